cd "$GRAFT_REPO_ROOT"; bash scripts/ab_c4f.sh ab/lib_side.so ab/lib_rt32768.so ab/lib_rt65536.so ab/lib_rt98304.so 2>&1 | head -9
