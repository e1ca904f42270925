"""Why is the first K-step call after a short warm-up slower than its repeats?  warm-up variants, then the same K-step call three times."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
K, W = 20, 5
def timed(r, k0, k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run(r.T - 1 - k0, k0, k, invariant=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for variant in ("W=5 once", "W=5 four times", "W=5 then 20 ms idle", "W=20 once"):
    r = Runner("bf16x3", 64, 13, W + K, 1, 0, dev)
    if variant == "W=5 once": r.run(r.T - 1, 0, W, invariant=True)
    elif variant == "W=5 four times":
        for _ in range(4): r.run(r.T - 1, 0, W, invariant=True)
    elif variant == "W=5 then 20 ms idle":
        r.run(r.T - 1, 0, W, invariant=True); torch.cuda.synchronize(); time.sleep(0.02)
    else: r.run(r.T - 1, 0, K, invariant=True)
    x0 = r.x.clone()
    res = []
    for rep in range(3):
        r.x.copy_(x0); res.append(timed(r, W, K))
    print("%-22s: %s ms/step" % (variant, " ".join("%.3f" % v for v in res)), flush=True)
    del r; torch.cuda.empty_cache()
