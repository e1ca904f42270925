"""Do two independent chains of dense kernels cover each other's ramps and tails?  Two contexts (two models), the same K-step call:
back to back on one stream vs. side by side on two streams.  The per-step dense chains of a group are independent of each other
(nothing of the encoder reads x), so a gain here is what running alternate steps of a group on two streams could buy."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
K = int(os.environ.get("K", 21)); B = int(os.environ.get("B", 64))
r1 = Runner("bf16x3", B, 13, K, 1, 0, dev); r2 = Runner("bf16x3", B, 13, K, 2, 0, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
T = r1.T
def go(two):
    r1.stream = s1.cuda_stream; r2.stream = (s2 if two else s1).cuda_stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r1.run(T - 1, 0, K, invariant=True); r2.run(T - 1, 0, K, invariant=True)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(2): go(False); go(True)
for rep in range(3):
    a = go(False); b = go(True)
    print("one stream %.3f ms/step | two streams %.3f ms/step  (per chain-step, 2 x %d steps)" % (a / (2 * K) * 1e3, b / (2 * K) * 1e3, K), flush=True)
