#!/bin/sh
# oracle/make_ref.sh -- TEST INFRASTRUCTURE.  Stages the UNMODIFIED reference package so that the
# GPU box (which has no /root/reference) can run the real `metran.Metran` class through the HIP
# boundary (tests/test_reference_dropin_gpu.py).  The reference is pure Python: there is nothing to
# compile, so "building" oracle/_ref is a verbatim copy of
#     /root/reference/metran/*.py                       (the package; imported through tests/golden/_refshim.py)
#     /root/reference/examples/data/B21B02140*_res.csv  (the input of the reference's own tests/conftest.py:13-24)
# into the git-ignored oracle/_ref/ (never committed, never imported by metran_amd; it travels to the
# GPU box with the snapshot exactly like the built .so files do).  No-op when the reference is absent.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${METRAN_REFERENCE_SRC:-/root/reference}"
if [ ! -d "$REF/metran" ]; then
    echo "make_ref: $REF/metran not present (GPU box?) -- keeping whatever oracle/_ref already holds"
    exit 0
fi
rm -rf "$HERE/_ref"
mkdir -p "$HERE/_ref/metran" "$HERE/_ref/examples/data"
cp "$REF"/metran/*.py "$HERE/_ref/metran/"
cp "$REF"/examples/data/B21B02140*_res.csv "$HERE/_ref/examples/data/"
( cd "$REF" && sha256sum metran/*.py examples/data/B21B02140*_res.csv ) > "$HERE/_ref/SHA256SUMS"
echo "make_ref: staged $(ls "$HERE/_ref/metran" | wc -l) modules + $(ls "$HERE/_ref/examples/data" | wc -l) data files into oracle/_ref"
