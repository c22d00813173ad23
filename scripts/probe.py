"""Dev probe: time filter / smoother kernels with different output subsets (GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

B = int(os.environ.get("B", 4096)); N, K, T = 8, 2, 1000
dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(B, N, K, T, seed=2000, device=dev)
kf = BatchedKalman(0, layout=os.environ.get("LAYOUT", "model_major"))
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
kf.enable_timing(True)
def run(tag, outs, smooth=False, reps=5):
    bufs = kf._alloc_outputs(B, list(outs))
    fs, ss = [], []
    for i in range(reps + 2):
        if smooth: kf.filter_smooth(d["phi"], d["q"], buffers=bufs)
        else: kf.filter(d["phi"], d["q"], buffers=bufs)
        f, s = kf.last_kernel_ms()
        if i >= 2: fs.append(f); ss.append(s)
    print("%-28s filter %.3f ms  smoother %.3f ms" % (tag, sum(fs)/len(fs), sum(ss)/len(ss) if smooth else 0))
    del bufs
run("loglik only", ())
run("F only", ("F",))
run("F,Pf", ("F", "Pf"))
run("F,Pf,Xp,Pp", ("F", "Pf", "Xp", "Pp"))
run("Pf,Pp", ("Pf", "Pp"))
run("full + smoother S,Ps", ("F", "Pf", "Xp", "Pp", "S", "Ps"), smooth=True)
run("F,Pf + smoother S only", ("F", "Pf", "S"), smooth=True)
run("F,Pf + smoother (no out)", ("F", "Pf"), smooth=True)
# fused projection path (row f2): filtered records only + projecting smoother
fs, ss = [], []
for i in range(6):
    r = kf.simulate_smoothed(d["phi"], d["q"])
    f, s = kf.last_kernel_ms()
    if i >= 2: fs.append(f); ss.append(s)
print("%-28s filter %.3f ms  smoother %.3f ms" % ("simulate_smoothed (f2)", sum(fs)/len(fs), sum(ss)/len(ss)))
