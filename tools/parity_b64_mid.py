"""The round-6 parity cases of tests/test_hip_configs.py at full length on two seeds (GPU box; ~15 min of host time for the oracle):
one mixed-timestep p_sample step and five mid-schedule steps (t = 524 .. 520) at B = 64, f32 and bf16x3, raw and with the kernel's
equally-near 3-NN choice injected into the oracle.  Output kept as profiles/r06_parity_extra.txt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_configs as tc

torch.set_grad_enabled(False)
for seed in (1, 2):
    tc.test_config3_b64_mixed_timesteps_one_step(seed)
    tc.test_config3_b64_mid_schedule_segment(seed, 5)
print("all cases inside their bars")
