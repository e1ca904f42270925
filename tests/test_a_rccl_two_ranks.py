"""The N > 1 path on real devices: two ranks on the "nccl" (= RCCL) backend, one all-gather of the final samples.  Lives in
its own module that sorts FIRST, so that a multi-GPU driver run reaches it before anything else can stop the session (pytest -x);
skipped on a one-GPU lease (the same logic runs on CPU over gloo in tests/test_parallel_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion

DEV = "cuda:0"

_RANK_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from lsdm_amd.parallel import gather_results, shard_seed
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
torch.set_grad_enabled(False)
B, K, base = 2, 3, 40
m, diff = create_model_and_diffusion("proxd", device=lr, precision="bf16x3")
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.to("cuda:%%d" %% lr) for k, v in make_synthetic_batch(B, seed=shard_seed(base, rank)).items()}
torch.manual_seed(shard_seed(base, rank))
local = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False,
                           skip_timesteps=1000 - K)
out, cat, guid = gather_results(local, m.saved_cat, m.saved_guiding_points, dist)
if rank == 0:
    torch.save({"sample": out.cpu(), "saved_cat": cat.cpu(), "guiding": guid.cpu()}, %(out)r)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_rccl_allgather_matches_single_process_shards(tmp_path):
    """world size 2 on the "nccl" (= RCCL) backend: rank r's shard of the all-gathered result is bit-equal to a
    single-process run with seed base + r."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices (the driver's 8-GPU node); covered on CPU by tests/test_parallel_gloo.py")
    script, out = tmp_path / "rank.py", tmp_path / "gathered.pt"
    script.write_text(_RANK_SCRIPT % dict(root=ROOT, out=str(out)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), str(script)], check=True, env=env, timeout=900)
    res = torch.load(out)
    got = res["sample"]
    assert got.shape == (4, 1024, 3) and res["saved_cat"].shape == (4, 1, 13) and res["guiding"].shape == (4, 1024, 3)
    m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
    m.load_state_dict(make_synthetic_state_dict(seed=0))
    for r in range(2):
        bt = {k: v.to(DEV) for k, v in make_synthetic_batch(2, seed=40 + r).items()}
        torch.manual_seed(40 + r)
        ref = diff.p_sample_loop(m, [2, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False,
                                 skip_timesteps=997)
        assert torch.equal(got[2 * r:2 * r + 2], ref.cpu()), r
        assert torch.equal(res["saved_cat"][2 * r:2 * r + 2], m.saved_cat.cpu()), r
        assert torch.equal(res["guiding"][2 * r:2 * r + 2], m.saved_guiding_points.cpu()), r


def test_bench_two_gpus_reports_both_ranks_and_the_collective():
    """`python bench.py --gpus 2` as the driver launches it for N > 1 (it re-executes itself under torch.distributed.run): the JSON line
    carries both ranks' step times, the RCCL all-gather's duration and a whole-job value = 2 x 64 chains over the slowest rank."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices (the driver's 8-GPU node)")
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-hoisted",
                        "--no-others", "--no-cpu-baseline", "--no-parity"], check=True, env=env, timeout=1200, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["scaling"] == "weak"
    assert len(r["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in r["per_rank_ms_per_step"])
    assert r["allgather_us"] > 0 and r["allgather_bytes_per_rank"] == 64 * (2 * 3072 + 13) * 4
    assert r["config"]["global_batch"] == 128 and r["per_rank_seeds"] == [1, 2]
    assert abs(r["value"] - 2e3 / max(r["per_rank_ms_per_step"])) <= 1e-3 * r["value"]      # whole job: 2 ranks x steps / slowest rank
