import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from oracle import cpu_ref
torch.set_grad_enabled(False)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
sd = make_synthetic_state_dict(seed=0); bt = make_synthetic_batch(2, seed=1, num_objects=8)
smp = cpu_ref.Sampler(sd); t = torch.tensor([999] * 2)
for n in [int(v) for v in sys.argv[1:]]:
    torch.set_num_threads(n)
    x = torch.randn(2, 1024, 3)
    t0 = time.time(); smp.p_sample(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"]); 
    print("threads %d: %.2f s per B=2 step" % (n, time.time() - t0), flush=True)
