"""Headline benchmark: denoising-steps/sec of the LSDM sampling loop on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one p_sample iteration (SceneDiffusionModel.forward + posterior update) of the whole per-GPU batch:
BASELINE.json configs[2] -- B = 64 chains, 9 slots x 1024 points (8 objects = the 8192-pt scene cloud, + the human
branch), C = 13, cosine T = 1000 -- every step with freshly drawn FPS starts and its own conditioning, as the reference
recomputes them.  Inputs, weights and the RNG draws of the timed steps (host generator, reference order) are resident
in HBM before the timed region; the K steps run inside ONE C-ABI call: lsdm_sample_steps_invariant (default -- what
p_sample_loop calls: the work that does not depend on the FPS draws is evaluated once per call, inside the timed call;
bit-identical to the strict entry point, checked every run) with lsdm_sample_steps (every stage every step) reported
beside it as `strict_recompute`.  Ranks are independent chain batches (seed = base + rank); the only
collective is one all-gather of the final samples.

dtype (--precision): "bf16x3" (default) = fp32 operands split into bf16 hi+lo, three bf16 MFMAs per product, fp32
accumulation -- fp32-class results that hold the same <= 1e-3 parity bar as the exact path (tests/test_hip_parity.py;
the live check below reports max-abs vs the CPU oracle); "f32" = exact v_mfma_f32_32x32x2_f32; "bf16" = plain bf16
products (reduced precision, configs[2]'s wording).  Geometry and sampler arithmetic are fp32 in every mode.  The other
two precisions are timed briefly and reported under "other_precisions"; they never enter `value`.

The JSON line also carries
  roofline     : the dominant kernel's algorithmic FLOP/s (HIP events on the launch stream, live in the timed region)
                 against the dense MFMA peak of the dtype (bf16x3 issues 3 MFMAs per algorithmic product, so its
                 attainable fraction of the bf16 peak is 1/3), plus whole-step figures and HBM traffic from the PMC pass;
  hoisted_conditioning : separate figure for the pinned-FPS / conditioning-evaluated-once mode;
  parity       : max-abs of one B=2 step against the CPU oracle, same seeds;
  cpu_baseline : the CPU oracle (oracle/cpu_ref.py, the checker -- never the product) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (before the HIP runtime loads): RCCL needs it on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: dense peaks.  f32-input MFMA 157.3 TF; bf16 MFMA 2.5 PF (never the 2:1-sparse figure)
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}
GFLOP_PER_SAMPLE_STEP = 14.58      # strict per-step recompute
# A profiled STAGE (the reference function it replaces) can be two launches in the bf16x3 mode; the rocprofv3 summary lists
# them separately (profiles/r01_*_kernel_stats.md), the HIP-event interval of the stage covers both.
STAGE_KERNELS_BF16X3 = {
    "sa_kernel<3,32,32,64>": "sa_r_kernel<3,32,32,64> over the item list of sa_pair_kernel (once per call in the loop-invariant schedule)",
    "sa_kernel<64,64,64,128>": "lin_u_kernel<64,64> (U, V per level-1 point; once per call in the loop-invariant schedule) + sa_ur_kernel<64,64,128>; "
                               "in the loop-invariant schedule: once per call the r = 0.2 ball of every source point (ball_query_kernel<1024>, ordered "
                               "scan) + memo_src_kernel + sa_ur_kernel on the item list of the uncrowded points (the memo F2), then per GROUP of steps (up to "
                               "LSDM_GROUP, default 28 / 32) memo_cent_kernel (copies F2 rows, lists the crowded centroids) + ONE sa_ur_kernel launch "
                               "over that list; avg_launch_ms is per step covered",
    "sa_kernel<128,128,128,256>": "lin_u_kernel<128,128> (U, V per level-2 point) + sa_p_kernel<128,128,256,12> (epilogues software-pipelined against the next block's MFMAs) over the item list of sa_pair_kernel (two ball-query groups per 32-row block where their distinct neighbours fit)",
    "sa_kernel<256,256,256,512>": "lin_u_kernel<256,256> (U, V per level-3 point) + sa_w_kernel over the item list of sa_pair_kernel",
    "fp_kernel<fp4>": "lin_u_kernel<512,256,false> (Y per level-4 point) + fp_pre_kernel<256,256,256,FUSE> (also writes fp3's Y); one launch per "
                      "group of steps (up to LSDM_GROUP, default 28 / 32) in the loop-invariant schedule",
    "fp_kernel<fp3>": "fp_pre_kernel<128,256,256,FUSE> (also writes fp2's Y)",
    "fp_kernel<fp2>": "fp_d_kernel<1024,256,64,256,128> (two-stage persistent kernel over Morton-ordered 32-point tiles, each distinct Y row landed once by "
                      "LDS-DMA) + its tile plans (fp_plan_kernel: one launch per group of steps in the loop-invariant schedule, its time inside this stage)",
    "fp_kernel<fp1+head>": "fp_c_kernel<1024,1024> (the five layers as a pipeline of stages on different waves, weights register-resident, the points walked along "
                           "the source cloud's Z curve)",
    "denoise_kernel": "denoise_r_kernel (one launch per group of steps and the guiding-point pass on the call's last forward only in the "
                      "loop-invariant schedule)",
}
GFLOP_PER_SAMPLE_STEP_HOISTED = 0.740   # the x/t-dependent part only


def cpu_baseline(batch, steps, C):
    """Times the CPU oracle: one p_sample step at B = 16 (BASELINE configs[1]'s batch, the largest batch the reference itself was measured
    at) -> `value`; and `steps` consecutive steps at the small batch `batch` beside it.  Both scaled to batches of 64 chains."""
    from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
    from oracle import cpu_ref
    # BASELINE.md 3 says all host cores; on the 256-core GPU-box host that is minutes per step for the oracle's many small ATen ops.
    # 16 threads is the fastest setting (tools/cpu_threads_probe.py, kept as profiles/r04_cpu_threads.txt: 8: 2.3 s, 16: 2.0 s, 32: 2.2 s,
    # 64: 3.6 s per B = 2 step, 128+: minutes), so the baseline is the oracle at ITS best thread count, stated in `cores`.
    threads = min(16, os.cpu_count())
    torch.set_num_threads(threads)
    sd = make_synthetic_state_dict(seed=0, max_cats=C)
    smp = cpu_ref.Sampler(sd)

    def timed(b, k):
        bt = make_synthetic_batch(b, seed=1, max_cats=C, num_objects=8)
        torch.manual_seed(1)
        x = torch.randn(b, 1024, 3)
        t = torch.tensor([999] * b)
        t0 = time.time()
        for j in range(k):
            x = smp.p_sample(x, bt["mask"], t - j, bt["given_objs"], bt["given_cats"], bt["text_features"])["sample"]
        return time.time() - t0

    timed(batch, 1)                     # warm-up (thread pool, allocator)
    dt_small = timed(batch, steps)
    dt16 = timed(16, 1)
    ss16, ss_small = 16 / dt16, batch * steps / dt_small
    return {"value": ss16 / 64.0, "unit": "denoising-steps/sec (batch of 64 chains)", "cores": threads,
            "kind": "port", "sample_steps_per_s": ss16,
            "sample": "1 p_sample step at B=16 (8 objects, C=%d; BASELINE configs[1]'s batch) with the torch-CPU oracle on %d of %d host cores, "
                      "%.1f s, after a warm-up step at B=%d; scaled to B=64 by /4 (cost per chain is flat in B: see small_batch)"
                      % (C, threads, os.cpu_count(), dt16, batch),
            "small_batch": {"value": ss_small / 64.0, "sample_steps_per_s": ss_small,
                            "sample": "%d p_sample steps at B=%d, %.1f s" % (steps, batch, dt_small)},
            "thread_sweep": "profiles/r04_cpu_threads.txt"}


def parity_check(precision, dev):
    """One p_sample step at B=2 through the product path vs the CPU oracle on the same seeds (the checker)."""
    from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
    from lsdm_amd.util.model_util import create_model_and_diffusion
    from oracle import cpu_ref
    sd = make_synthetic_state_dict(seed=0)
    model, diffusion = create_model_and_diffusion("proxd", device=dev.index, precision=precision)
    model.load_state_dict(sd)
    bt = make_synthetic_batch(2, seed=3)
    t = torch.tensor([900, 900])
    torch.manual_seed(1)
    x0 = torch.randn(2, 1024, 3)
    torch.manual_seed(2)
    ref = cpu_ref.Sampler(sd).p_sample(x0.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"])
    d = {k: v.to(dev) for k, v in bt.items()}
    torch.manual_seed(2)
    out = diffusion.p_sample(model, x0.clone().to(dev), d["mask"], t.to(dev), d["given_objs"], d["given_cats"], d["text_features"],
                             clip_denoised=False)
    torch.cuda.synchronize()
    return {"max_abs_sample": float((out["sample"].cpu() - ref["sample"]).abs().max()),
            "max_abs_pred_xstart": float((out["pred_xstart"].cpu() - ref["pred_xstart"]).abs().max()),
            "at_the_benched_batch": "tests/test_hip_configs.py::test_config3_b64_* (pytest -m gpu): B=64, 8 objects, C=13, steps from t=999, one step with a "
                                    "different timestep per chain and a mid-schedule segment vs the oracle with stage taps -- FPS / ball indices equal on all "
                                    "576 clouds x 4 levels, 3-NN distances bit-identical; what the caller reads (pred_xstart, guiding points, saved_cat) <= 1e-3 "
                                    "raw (measured <= 1.2e-4); a per-step sample <= 5e-5 (f32) / 1e-4 (bf16x3) of its magnitude with the kernel's equally-near "
                                    "neighbour choice injected into the oracle (raw: up to 1.1e-3 through the reference's unspecified order of exactly tied "
                                    "3-NN distances, profiles/r06_parity_extra.txt)",
            "against": "CPU oracle (oracle/cpu_ref.py, pinned to goldens recorded from the reference), one p_sample step, B=2, t=900"}


class Runner:
    """Model + resident inputs + RNG streams of one precision; run(t_first, k0, k) = k steps in one C-ABI call."""

    def __init__(self, precision, B, C, n_steps, seed, local_rank, dev, datatype="proxd", respacing="", ddim=False, surface=False, prompts_per_scene=1,
                 num_objects=8):
        from lsdm_amd import _lib
        from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
        from lsdm_amd.util.model_util import create_gaussian_diffusion, create_model_and_diffusion, get_default_diffusion
        self._lib = _lib
        self.B, self.C = B, C
        self.mode = _lib.MODE_DDIM if ddim else _lib.MODE_P_SAMPLE
        self.model, self.diffusion = create_model_and_diffusion(datatype, device=local_rank, precision=precision, max_cats=C)
        if respacing:
            self.diffusion = create_gaussian_diffusion(get_default_diffusion(), timestep_respacing=respacing)
        self.model.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C))
        self.ctx = self.model._context()
        self.model._reserve(B)
        self.model._set_schedule(self.diffusion._schedule_key(), self.diffusion._device_tables())
        bt = make_synthetic_batch(B // prompts_per_scene, seed=seed, max_cats=C, num_objects=num_objects, surface=surface)
        if prompts_per_scene > 1:      # HUMANISE (BASELINE configs[4]): chains 2 s, 2 s + 1 share scene s and differ in the prompt
            text = make_synthetic_batch(B, seed=seed + 1000, max_cats=C, num_objects=1)["text_features"]
            bt = {k: v.repeat_interleave(prompts_per_scene, dim=0) for k, v in bt.items()}
            bt["text_features"] = text
        self.bt = {k: v.to(dev).contiguous() for k, v in bt.items()}
        # RNG stream of the reference: x_T, then per step 4 FPS randints + randn, host generator
        torch.manual_seed(seed)
        x = torch.randn(B, 1024, 3)
        starts, noise = [], []
        for _ in range(n_steps):
            starts.append(torch.stack([torch.randint(0, n, (B * 9,), dtype=torch.long) for n in (1024, 1024, 256, 64)]))
            noise.append(torch.randn(B, 1024, 3))
        self.x = x.to(dev).contiguous()
        self.starts = torch.stack(starts).to(dev).contiguous()
        self.noise = torch.stack(noise).to(dev).contiguous()
        self.out_cat = torch.empty(B, 1, C, device=dev)
        self.out_x0, self.out_g = torch.empty_like(self.x), torch.empty_like(self.x)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.T = self.diffusion.num_timesteps

    def run(self, t_first, k0, k, hoisted=False, x=None, noise=None, invariant=False):
        p = lambda v: v.data_ptr()
        bt = self.bt
        x = self.x if x is None else x
        entry = "lsdm_sample_steps_hoisted" if hoisted else ("lsdm_sample_steps_invariant" if invariant else "lsdm_sample_steps")
        self.ctx.call(entry, self.mode, 0.0, self.B, t_first, k,
                      p(x), p(bt["mask"]), p(bt["given_objs"]), p(bt["given_cats"]), p(bt["text_features"]), p(self.starts[k0]),
                      p(self.noise[k0] if noise is None else noise), p(self.out_cat), p(self.out_x0), p(self.out_g), self.stream)


class StubRunner:
    """--stub-runner (tests/test_parallel_gloo.py: world-8 plumbing test on CPU, no GPU in the loop): the Runner interface with NO work behind
    it -- run() sleeps 2 ms per step (rank r: + 10 % per rank, so the slowest rank is known) and stamps the sample with the rank's seed.
    Everything around it (rank environment, seeds, barriers, the packed all-gather, max-over-ranks timing, the JSON contract fields) is
    bench.py's real code on the gloo backend.  Its `value` measures nothing and the line says so ("data": "stub")."""

    class _Ctx:
        def profile_begin(self): pass
        def profile_end(self): return []

    def __init__(self, precision, B, C, n_steps, seed, local_rank, dev, **_):
        self.B, self.C, self.T, self.seed = B, C, 1000, seed
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(B, 1024, 3, generator=g)
        self.out_cat = torch.zeros(B, 1, C)
        self.out_x0, self.out_g = torch.zeros_like(self.x), torch.zeros_like(self.x)
        self.ctx = StubRunner._Ctx()
        self.rank = int(os.environ.get("RANK", 0))

    def run(self, t_first, k0, k, hoisted=False, x=None, noise=None, invariant=False):
        time.sleep(0.002 * k * (1.0 + 0.1 * self.rank))
        (self.x if x is None else x).fill_(float(self.seed))
        self.out_cat.fill_(float(self.seed))
        self.out_g.fill_(-float(self.seed))


def contract_fields(a, world, K, W, B, C, dt, per_rank_dt, allgather_us, T, inv, data="synthetic"):
    """The bench contract's fields of the JSON line (driver: value = whole-job steps/s over the slowest rank's time)."""
    from lsdm_amd.parallel import shard_seed
    step_ms = dt / K * 1e3
    return {
        "metric": "denoising-steps/sec (B=64, 1024-pt clouds x 9 slots incl. the 655-vertex human branch, T=1000)",
        "value": world * K / dt, "unit": "denoising-steps/sec (one step = p_sample of a 64-chain batch per GPU; value sums the GPUs)",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.precision, "data": data,
        "sample_steps_per_s": world * B * K / dt,
        "config": {"workload": "BASELINE configs[2]: B=%d chains/GPU, 8 objects + human x 1024 pts (8192-pt scene cloud), C=13, random CLIP-text "
                               "features, cosine T=1000, p_sample steps t=%d..%d in ONE %s call on one caller stream, host-replayed RNG; %s"
                               % (B, T - 1 - W, T - W - K, "lsdm_sample_steps_invariant" if inv else "lsdm_sample_steps",
                                  "per-step recompute of the conditioning MLPs and of PointNet++ (FPS with fresh starts, levels 2-4, all FP "
                                  "stages); what does NOT depend on the FPS order is evaluated once per call INSIDE the timed call and "
                                  "re-ordered / copied per step: the level-1 work (balls, sa1 features, 3-NN candidates) and sa2's output of "
                                  "the points with fewer than 32 points within r = 0.2 (their ball is the same set every step; the others -- "
                                  "45 % of the centroids on this data -- go through sa2 every step) -- results bit-identical to the strict "
                                  "schedule, whose figure is under strict_recompute" if inv else
                                  "strict per-step recompute of everything the reference recomputes"),
                   "entry_point": "lsdm_sample_steps_invariant" if inv else "lsdm_sample_steps",
                   "global_batch": B * world, "chains_per_gpu": B,
                   "parallelism": "chains sharded over %d rank(s); one RCCL all-gather of final samples + saved_cat + guiding points (one packed buffer)" % world},
        "schedule": "loop-invariant (level-1 encoder work and the order-free part of sa2 once per %d-step call, inside the timed call; bit-identical to strict)" % K if inv
                    else "strict (every stage every step)",
        "rccl_ranks": world, "per_rank_seeds": [shard_seed(a.seed, q) for q in range(world)], "allgather_us": allgather_us,
        "allgather_bytes_per_rank": int(B * (2 * 3072 + C) * 4) if world > 1 else 0,
        "per_rank_ms_per_step": [round(v / K * 1e3, 4) for v in per_rank_dt],
    }


def config_legs(a, dev, local_rank, inv):
    """Short legs (never `value`) for the BASELINE configs that are not the headline but fit one GPU, through the same entry point (plus `ragged_k`:
    SURVEY.md 8d's k ~ U{1..8} variant of the headline workload):
    configs[1] (B = 16, C = 43: f32 -- its stated precision -- and bf16x3), configs[4]'s shapes on one GPU (HUMANISE defaults C = 11, 32 scenes x
    2 prompts, DDIM on the 'ddim250' schedule, reference util/model_util.py:56-85 + diffusion/gaussian_diffusion.py:790-811) and the headline
    workload on SURFACE-sampled objects (points on the box faces: PROX-like scans, posa/dataset.py:424-470) with sa2's memo hit rate."""
    import ctypes
    out = {}

    def leg(name, precision, B, C, K, note, **kw):
        r = Runner(precision, B, C, 2 + K, a.seed, local_rank, dev, **kw)
        T = r.T
        r.run(T - 1, 0, 2, invariant=inv)
        x0 = r.x.clone()
        ds = []
        for _ in range(3):                       # median of three repeats of the same K-step call (a fresh context's first long call can hiccup)
            r.x.copy_(x0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r.run(T - 3, 2, K, invariant=inv)
            torch.cuda.synchronize()
            ds.append(time.perf_counter() - t1)
        d = sorted(ds)[1]
        e = {"value": K / d, "value_is": "median of 3 repeats", "value_min": K / max(ds), "value_max": K / min(ds), "unit": "denoising-steps/sec (batch of %d chains)" % B, "ms_per_step": d / K * 1e3, "steps": K, "batch": B, "max_cats": C,
             "dtype": precision, "sample_steps_per_s": B * K / d, "schedule_T": T, "sampler": "ddim" if kw.get("ddim") else "p_sample",
             "entry_point": "lsdm_sample_steps_invariant" if inv else "lsdm_sample_steps", "finite": bool(torch.isfinite(r.x).all()), "workload": note}
        if name == "ragged_k":
            zero = int((r.bt["given_objs"].abs().amax(dim=(2, 3)) == 0).sum())
            e["clouds"] = {"per_step": B * 9, "all_zero_slots": zero, "live": B * 9 - zero,
                           "note": "k ~ U{1..8} objects per scene, the absent slots ZERO clouds (SURVEY.md 8d, posa/dataset.py:456): a cloud of 1024 identical "
                                   "points has an encoder output independent of the FPS draws -- one representative per call, the per-step encoder on the "
                                   "live clouds only (DESIGN.md 10; LSDM_DEG_FAST=0 = every cloud every step); bit-identical to it"}
        if kw.get("surface") or name == "headline_volume_reference":
            # sa2's memo: memo_cnt = {uncrowded source points of the call, crowded centroids of the call's last group}; the K steps are one group
            try:
                ptr, n = r.ctx.workspace_tensor("memo_cnt")
                cnt = (ctypes.c_int * 2)()
                hip = ctypes.CDLL("libamdhip64.so")
                if n >= 2 and hip.hipMemcpy(cnt, ctypes.c_void_p(ptr), ctypes.c_size_t(8), 2) == 0:
                    e["sa2_memo"] = {"uncrowded_source_points_frac": cnt[0] / float(B * 9 * 1024),
                                     "centroid_hit_rate": 1.0 - cnt[1] / float(B * 9 * 256 * K),
                                     "note": "fraction of the steps' sa2 centroids served by the per-call memo (fewer than 32 points within r = 0.2)"}
            except Exception as ex:     # (the memo is off: LSDM_SA2_MEMO=0 or a non-default schedule)
                e["sa2_memo"] = {"error": str(ex)}
            # sa3 / sa4's item lists: the 32-row blocks each step of the call's last group ran, per ball-query group
            try:
                hip = ctypes.CDLL("libamdhip64.so")
                it = {}
                for nm, S in (("sa3_items", 64), ("sa4_items", 16)):
                    ptr, n = r.ctx.workspace_tensor(nm)
                    k = min(K, int(n))
                    cnt = (ctypes.c_int * k)()
                    if hip.hipMemcpy(cnt, ctypes.c_void_p(ptr), ctypes.c_size_t(4 * k), 2) == 0:
                        it[nm.replace("_items", "_blocks_per_group")] = sum(cnt) / float(k * B * 9 * S)
                it["note"] = "two ball-query groups whose distinct neighbours fit into 32 rows share one MFMA block (1.0 = one block per group)"
                e["sa_item_lists"] = it
            except Exception as ex:     # (LSDM_SA_PAIR=0 or a non-default schedule)
                e["sa_item_lists"] = {"error": str(ex)}
        out[name] = e
        del r
        torch.cuda.empty_cache()

    leg("configs[1]_f32", "f32", 16, 43, 6, "BASELINE configs[1]: B=16, C=43 (655-vertex human branch), cosine T=1000, p_sample, exact f32 MFMA")
    leg("configs[1]_bf16x3", "bf16x3", 16, 43, 10, "BASELINE configs[1]'s shapes in the headline precision")
    leg("configs[4]_one_gpu", "bf16x3", 64, 11, 10, "BASELINE configs[4]'s shapes on ONE GPU: HUMANISE defaults (C=11), 32 scenes x 2 prompts (chains 2s, 2s+1 share "
        "given_objs), DDIM (eta = 0) on the 'ddim250' respaced schedule, steps t'=247..238", datatype="humanise", respacing="ddim250", ddim=True, prompts_per_scene=2)
    leg("headline_surface_sampled", "bf16x3", 64, 13, 10, "the headline workload with the 8 objects' points ON their box faces (surface scans) instead of in the "
        "volume: sparser balls, higher sa2-memo hit rate", surface=True)
    leg("headline_volume_reference", "bf16x3", 64, 13, 10, "the headline workload (volume-filled boxes) through the same 10-step leg, for the memo hit rate beside it")
    leg("ragged_k", "bf16x3", 64, 13, 10, "the headline workload with k ~ U{1..8} objects per scene instead of 8 (SURVEY.md 8d's default variant; mask = slots 1..k), "
        "through the same 10-step leg as headline_volume_reference", num_objects=None)
    out["ragged_k"]["vs_all_8_objects"] = out["ragged_k"]["value"] / out["headline_volume_reference"]["value"]
    return out


def respawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-executes itself under torch.distributed.run with one rank per GPU
    (the same command line the driver uses) and returns that job's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=7)
    ap.add_argument("--mode", default="invariant", choices=["strict", "invariant"],
                    help="headline entry point: lsdm_sample_steps_invariant (default; what p_sample_loop calls -- bit-identical to "
                         "strict, the FPS-order-independent level-1 encoder work evaluated once per call inside the timed call) or "
                         "lsdm_sample_steps (every stage recomputed every step); the other one is reported beside it")
    ap.add_argument("--batch", type=int, default=64, help="chains per GPU")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--precision", default=os.environ.get("LSDM_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the dense stacks (geometry and sampler arithmetic are always fp32)")
    ap.add_argument("--no-hoisted", action="store_true", help="skip the extra hoisted-conditioning measurement")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs in the other two precisions")
    ap.add_argument("--no-parity", action="store_true", help="skip the live parity check against the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configs (configs[1], configs[4] shapes, surface-sampled clouds)")
    ap.add_argument("--stub-runner", action="store_true",
                    help="plumbing test on CPU (gloo): no GPU work behind the Runner interface; the line's value measures nothing")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps at --cpu-batch after 1 warm-up (BASELINE.md §3); one more step is timed at B=16")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch one rank per GPU (python -m torch.distributed.run "
                 "--nproc-per-node %d bench.py --gpus %d ...) or run `python bench.py --gpus %d` directly" % (a.gpus, world, a.gpus, a.gpus, a.gpus))
    stub = a.stub_runner
    if torch.cuda.device_count() < a.gpus and not stub:
        sys.exit("bench.py: --gpus %d but only %d visible device(s) -- refusing to run ranks on shared GPUs" % (a.gpus, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not stub:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    sync = (lambda: None) if stub else torch.cuda.synchronize
    torch.set_grad_enabled(False)

    from lsdm_amd.parallel import gather_results

    B, K, W, C = a.batch, a.steps, a.warmup, 13

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    from lsdm_amd.parallel import shard_seed
    inv = a.mode == "invariant"
    r = (StubRunner if stub else Runner)(a.precision, B, C, W + K, shard_seed(a.seed, rank), local_rank, dev)
    T = r.T
    log("inputs resident; warm-up %d steps" % W)
    if W:
        r.run(T - 1, 0, W, invariant=inv)
    if dist:
        gather_results(r.x, r.out_cat, r.out_g, dist)   # untimed: RCCL communicator set-up + shard-size agreement belong to the warm-up
    sync()
    if dist:
        dist.barrier()
    sync()
    # Per-stage HIP events (lsdm_profile_*: one event between consecutive stages, recorded on the launch stream) of the SAME call --
    # same inputs, same RNG draws, from the same x -- in a pass of its own between the warm-up and the timed region, on every rank:
    # per-kernel durations need the kernels one at a time, so under instrumentation the library keeps every stage on the caller's
    # stream (its default runs the dense chains of alternate steps of a group on a side stream, where two kernels share the chip and
    # a per-kernel duration is not defined), and ~25 event records per step are not part of the product's work.  The roofline leg
    # comes from this pass; its wall time is reported beside `value`.  (It runs BEFORE the timed region, not after: a 5-step warm-up
    # is 11 ms of work and leaves the chip's clocks below their steady state -- tools/first_call_probe.py: first 20-step call after
    # it 2.22 ms/step, after 20 warm steps 2.16, every later repeat 2.15.)
    x_start = r.x.clone()
    xi = x_start.clone()
    sync()
    r.ctx.profile_begin()
    t1 = time.perf_counter()
    r.run(T - 1 - W, W, K, x=xi, invariant=inv)
    sync()
    dt_instr = time.perf_counter() - t1
    prof = r.ctx.profile_end()
    x_instr = xi
    if dist:
        dist.barrier()
    sync()
    log("timing %d steps (%s, %s)" % (K, a.precision, a.mode))
    ev = [None, None] if stub else [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    r.run(T - 1 - W, W, K, invariant=inv)
    tg0 = time.perf_counter()
    if not stub:
        ev[0].record()
    # the path's only collective (RCCL all-gather over xGMI; sizes agreed in the warm-up): what run/test_sdm.py:184-202 reads after the
    # loop -- final samples, saved_cat and the guiding points of every chain -- packed into one buffer (lsdm_amd/parallel.py)
    gathered, gathered_cat, gathered_guid = gather_results(r.x, r.out_cat, r.out_g, dist, agree=False)
    tg1 = time.perf_counter()
    if not stub:
        ev[1].record()
    sync()
    if dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    allgather_us = ((tg1 - tg0) * 1e6 if stub else ev[0].elapsed_time(ev[1]) * 1e3) if dist else 0.0
    log("timed region %.3f s" % dt)
    per_rank_dt = [dt]
    if dist:
        assert dist.get_world_size() == a.gpus
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)      # (every rank's own wall time of the timed region)
        every = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        per_rank_dt = [float(v) for v in every.cpu()]
        dt = max(per_rank_dt)                     # the job's time is its slowest rank's
    assert gathered.shape == (B * world, 1024, 3) and bool(torch.isfinite(gathered).all())
    assert gathered_cat.shape == (B * world, 1, C) and gathered_guid.shape == (B * world, 1024, 3) and bool(torch.isfinite(gathered_guid).all())
    x_final = r.x.clone()
    if stub:
        # the contract fields from the real bookkeeping; the gathered rows must be every rank's stamp in rank order
        stamps = [float(gathered[q * B, 0, 0]) for q in range(world)]
        assert stamps == [float(shard_seed(a.seed, q)) for q in range(world)], stamps
        if rank == 0:
            res = contract_fields(a, world, K, W, B, C, dt, per_rank_dt, allgather_us, T, inv, data="stub (no GPU work: plumbing test)")
            res["stub"] = True
            print(json.dumps(res), flush=True)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank != 0:
        dist.barrier()                 # rank 0 finishes its untimed reporting legs, then everyone tears down together
        dist.destroy_process_group()
        return

    # ---- everything below: rank 0 only, outside the timed region ----------------------------------------------------
    # Separate, clearly labelled figure (never `value`): hoisted-conditioning mode -- FPS starts pinned,
    # scene conditioning evaluated once per call, then KH steps of embedding MLPs + fused denoiser/sampler update.
    hoisted = None
    if not a.no_hoisted:
        KH = max(K, T)                                            # the complete T = 1000 chain in ONE call (conditioning once per call)
        nh = r.noise[:1].expand(KH, -1, -1, -1).contiguous()      # resident noise (0.8 GB at B = 64); values do not affect timing
        xh = r.x.clone()
        r.run(T - 1, 0, KH, hoisted=True, x=xh, noise=nh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r.run(T - 1, 0, KH, hoisted=True, x=xh, noise=nh)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        # executed work of the same schedule from a profiled (shorter) call: per-step stages vs the once-per-call conditioning
        KP = min(KH, 112)
        r.ctx.profile_begin()
        r.run(T - 1, 0, KP, hoisted=True, x=xh, noise=nh)
        torch.cuda.synchronize()
        ph = r.ctx.profile_end()
        per_step_names = ("denoise_kernel", "upsample12_kernel", "cond_kernel", "scene_attn_kernel")
        ex_h_step = sum(2.0 * q["executed_macs"] for q in ph if q["name"] in per_step_names) / KP / 1e9
        ex_h_call = sum(2.0 * q["executed_macs"] for q in ph if q["name"] not in per_step_names) / 1e9
        del nh
        hoisted = {"value": KH / dth, "unit": "denoising-steps/sec per GPU (B=%d)" % B, "steps": KH, "ms_per_step": dth / KH * 1e3,
                   "dtype": a.precision,
                   "roofline": {"bound": "mfma", "achieved": GFLOP_PER_SAMPLE_STEP_HOISTED * B / (dth / KH * 1e3), "peak": PEAK_TFLOPS[a.precision],
                                "unit": "TFLOP/s", "frac": GFLOP_PER_SAMPLE_STEP_HOISTED * B / (dth / KH * 1e3) / PEAK_TFLOPS[a.precision],
                                "note": "0.740 GFLOP per sample-step (SURVEY.md 8d: the x/t-dependent part) x B / ms_per_step; the once-per-call "
                                        "conditioning (13.8 GFLOP per sample) is inside the timed call and not counted as work",
                                "executed_gflop": ex_h_step + ex_h_call / KH, "executed_gflop_per_step_stages": ex_h_step,
                                "executed_gflop_once_per_call": ex_h_call,
                                "frac_executed": (ex_h_step + ex_h_call / KH) / (dth / KH * 1e3) / PEAK_TFLOPS[a.precision],
                                "mfma_issue_frac": (ex_h_step + ex_h_call / KH) / (dth / KH * 1e3) / PEAK_TFLOPS[a.precision]
                                                   * (3.0 if a.precision == "bf16x3" else 1.0),
                                "executed_note": "products issued per step of the batch (denoiser first pass + its shared S block, the timestep "
                                                 "half of the up-sampling / combine_extraction ONCE per step for the whole batch, the guiding-point "
                                                 "pass once per call) + the once-per-call conditioning / KH; the reference's 0.740 GFLOP counts the "
                                                 "up-sampling per chain and the guiding pass per step"},
                   "note": "NOT the headline metric: conditioning (human branch, PointNet++ geometry + dense stages, scene attention) "
                           "evaluated once per call with pinned FPS starts, included in the timed call; equals the reference only with "
                           "its per-step FPS re-draw pinned (tests: g6_hoisted golden)"}

    instrumented = {"ms_per_step": dt_instr / K * 1e3, "value": K / dt_instr, "steps": K,
                    "final_sample_bit_identical_to_timed_region": bool(torch.equal(x_instr, x_final)),
                    "order": "warm-up (%d steps) -> this pass -> barrier + synchronize -> timed region" % W,
                    "note": "the timed call with lsdm_profile_begin: per-stage HIP events, every stage on one stream (the source of `roofline`)"}
    del x_instr

    # The same K steps again: five more times through the headline entry point (run-to-run spread), once through the OTHER entry
    # point; all final samples must be bit-identical (the invariant mode only moves FPS-order-independent work out of the loop).
    def rerun(invariant, repeats=5):
        xi = x_start.clone()
        r.run(T - 1 - W, W, K, x=xi, invariant=invariant)      # warm
        ds = []
        for _ in range(repeats):
            xi.copy_(x_start)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r.run(T - 1 - W, W, K, x=xi, invariant=invariant)
            torch.cuda.synchronize()
            ds.append(time.perf_counter() - t1)
        ds.sort()
        d = ds[len(ds) // 2]
        return {"value": K / d, "ms_per_step": d / K * 1e3, "steps": K, "dtype": a.precision, "repeats": repeats,
                "value_is": "median of %d repeats of the same %d-step call" % (repeats, K), "value_min": K / ds[-1], "value_max": K / ds[0],
                "final_sample_bit_identical_to_timed_region": bool(torch.equal(xi, x_final))}
    # INTEGRATION.md path B: one model.forward (lsdm_step, mode 0) per call -- conditioning, POSA branch and PointNet++ evaluated
    # every call, nothing amortised over steps
    def forward_only(n=10):
        xf = x_start.clone()
        tt = torch.full((B,), T - 1, device=dev, dtype=torch.long)
        bt = r.bt
        r.model(xf, bt["mask"], tt, bt["given_objs"], bt["given_cats"], bt["text_features"])   # warm
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            r.model(xf, bt["mask"], tt, bt["given_objs"], bt["given_cats"], bt["text_features"])
        torch.cuda.synchronize()
        d = (time.perf_counter() - t1) / n
        return {"ms_per_forward": d * 1e3, "forwards_per_s": 1.0 / d, "calls": n, "dtype": a.precision,
                "entry_point": "SceneDiffusionModel.forward -> lsdm_step(mode 0), B=%d, host-drawn FPS starts uploaded per call" % B}
    fwd = forward_only()
    uninstrumented = rerun(inv)
    uninstrumented["note"] = ("the timed region repeated 5x (like it, without lsdm_profile_begin): median with min / max -- the run-to-run "
                              "spread to hold `value` against")
    res_spread = {"value_median": uninstrumented["value"], "value_min": uninstrumented["value_min"], "value_max": uninstrumented["value_max"]}
    other = rerun(not inv)
    other["entry_point"] = "lsdm_sample_steps" if inv else "lsdm_sample_steps_invariant"
    other["note"] = ("every stage recomputed every step (the reference's literal schedule of work)" if inv else
                     "level-1 ball query, sa1 features and 3-NN candidates evaluated once per call, re-ordered per step")

    per_launch_units = lambda name: B * 9 if name[:4] in ("ball", "sa_k", "thre", "fp_k", "fps_") else B
    rows = []
    for pr in prof:
        if pr["launches"] == 0:
            continue
        avg_ms = pr["total_ms"] / pr["launches"]
        flops = 2.0 * pr["macs_per_unit"] * per_launch_units(pr["name"])
        # executed work (include/lsdm_hip.h lsdm_profile_executed): products the stage's launches issued to the matrix pipe over the
        # window; a stage without an explicit figure (no matrix work, or a non-default kernel) reports its algorithmic count
        ex_flop = 2.0 * pr["executed_macs"] if pr["executed_macs"] > 0 else flops * pr["launches"]
        rows.append(dict(name=pr["name"], avg_ms=avg_ms, launches=pr["launches"], total_ms=pr["total_ms"],
                         tflops=flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0, gflop=flops / 1e9,
                         ex_gflop_step=ex_flop / K / 1e9, ex_tflops=ex_flop / (pr["total_ms"] * 1e-3) / 1e12 if pr["total_ms"] > 0 else 0.0))
    rows.sort(key=lambda q: -q["total_ms"])
    # the dominant MFMA kernel = the matrix stage with the largest share of the timed region (FPS etc. carry no matrix work;
    # once-per-call launches such as sa1 in the loop-invariant schedule have long launches but a small share)
    mfma_rows = [q for q in rows if q["gflop"] > 1.0]
    dom = mfma_rows[0]
    step_ms = dt / K * 1e3
    peak = PEAK_TFLOPS[a.precision]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(a.precision, {}).get(dom["name"])
    whole = GFLOP_PER_SAMPLE_STEP * B / step_ms
    issue = 3.0 if a.precision == "bf16x3" else 1.0          # matrix instructions issued per executed product
    ex_step_gflop = sum(q["ex_gflop_step"] for q in mfma_rows)
    whole_ex = ex_step_gflop / step_ms
    res = contract_fields(a, world, K, W, B, C, dt, per_rank_dt, allgather_us, T, inv)
    res.update({
        "roofline": {"bound": "mfma", "kernel": dom["name"],
                     "launches_of_the_stage": STAGE_KERNELS_BF16X3.get(dom["name"], dom["name"]) if a.precision == "bf16x3" else dom["name"],
                     "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                     "frac": dom["tflops"] / peak, "frac_of_attainable": dom["tflops"] / peak * (3.0 if a.precision == "bf16x3" else 1.0),
                     "traffic": traffic, "avg_launch_ms": dom["avg_ms"],
                     "avg_launch_ms_is": "HIP-event time of the stage per STEP covered (a stage batched over the steps of a group counts as that many "
                                         "launches; the dominant stage runs one launch per step), from the instrumented pass of the same call (see "
                                         "`instrumented`: stages in sequence on one stream; the timed region runs the dense chains of alternate steps on "
                                         "a side stream of the library, LSDM_STREAMS=%s); profiles/ are recorded with LSDM_STREAMS=1 for the same reason"
                                         % os.environ.get("LSDM_STREAMS", "2"),
                     "algorithmic_gflop_per_launch": dom["gflop"],
                     "executed_gflop": dom["ex_gflop_step"], "frac_executed": dom["ex_tflops"] / peak, "mfma_issue_frac": dom["ex_tflops"] * issue / peak,
                     "executed_is": "products the stage's launches ISSUE to the matrix pipe per step (lsdm_profile_executed: K padded to the k-step, "
                                    "split first layers, work the schedule evaluates once per call counted once) x 2 FLOP; frac_executed = that / "
                                    "stage time / peak; mfma_issue_frac = x%d matrix instructions per product (the figure PMC's MFMA-pipe "
                                    "utilisation measures against the actual clock); `frac` above stays reference FLOPs / time / peak" % int(issue),
                     "peak_note": {"f32": "dense f32-input MFMA (v_mfma_f32_32x32x2_f32)",
                                   "bf16x3": "dense bf16 MFMA; 3 MFMAs per algorithmic product -> attainable frac 0.333",
                                   "bf16": "dense bf16 MFMA"}[a.precision],
                     "whole_step": {"achieved": whole, "frac": whole / peak, "frac_of_f32_mfma_peak": whole / PEAK_TFLOPS["f32"],
                                    "note": "14.58 GFLOP per sample-step (SURVEY.md 8d) x B / ms_per_step",
                                    "executed_gflop": ex_step_gflop, "executed_gflop_per_sample_step": ex_step_gflop / B,
                                    "frac_executed": whole_ex / peak, "mfma_issue_frac": whole_ex * issue / peak,
                                    "executed_note": "sum of the matrix stages' executed products per step of the instrumented pass (once-per-call "
                                                     "work divided by the call's %d steps) x 2 FLOP / ms_per_step of the timed region" % K},
                     # `tflops` = the stage's REFERENCE work / its time, listed only where the launches execute (nearly) that work; where the
                     # schedule evaluates part of it once per call or not at all (the sa2 memo, sa1, the denoiser's shared rows, the split first
                     # layers) the figure would exceed what the hardware can do and is left out: `tflops_executed` is the honest column there
                     "kernels": [{"name": q["name"], "launches": q["launches"], "avg_ms": round(q["avg_ms"], 4),
                                  "ms_per_step": round(q["total_ms"] / K, 4),
                                  "tflops": round(q["tflops"], 2) if q["ex_tflops"] >= 0.8 * q["tflops"] else None,
                                  "executed_gflop": round(q["ex_gflop_step"], 3), "tflops_executed": round(q["ex_tflops"], 2),
                                  "frac_executed": round(q["ex_tflops"] / peak, 4), "mfma_issue_frac": round(q["ex_tflops"] * issue / peak, 4)} for q in rows]},
    })
    # no stage may claim more than the matrix pipe can issue (executed products x instructions per product <= peak), none more executed than
    # reference work beyond the k-step padding of the narrow layers
    # (recorded, not raised: the line must not be lost to a bookkeeping slip -- tests/test_hip_determinism.py asserts the list is empty)
    res["roofline"]["guard_violations"] = [q["name"] for q in rows if q["ex_tflops"] * issue > peak or (q["gflop"] > 1.0 and q["ex_tflops"] > 1.15 * q["tflops"])]
    if a.precision == "bf16x3":
        # The other roof these stages touch: bytes a launch pulls through the CUs' fill path (gathered rows +
        # weight fragments / panels streamed per tile or round; per cloud, x 9 B clouds) / stage time, against what tools/ubench/gather2_bw.hip
        # measures for random 1 KB rows with four gathering waves per CU and precomputed indices (profiles/r05_gather2_bw.txt: 25 TB/s L2-resident,
        # 8.7 TB/s out of the Infinity Cache; round 4's gather_bw.hip measured its own address arithmetic at four waves: 11 TB/s).  fp2 (round 5,
        # fp_d_kernel) lands every DISTINCT row of a Morton-ordered 32-point tile once (~26 of the 96 requested on this data; the figure below
        # assumes 26) -- it is no longer a fill-path kernel and is listed for the comparison with round 4's 2.56 GB per launch only.
        fill_mb_per_cloud = {"fp_kernel<fp2>": 32 * 26 * 1024 / 1e6 + 1024 * 256 / 1e6,                       # distinct Y rows per tile (estimate) + p1 rows
                             "sa_kernel<256,256,256,512>": 16 * 32 * 1024 / 1e6 + 4 * 786432 / 1e6,         # U rows, twelve 64 KB panels per four centroids
                             "fp_kernel<fp1+head>": 1024 * 3 * 512 / 1e6,                                   # rows (fp_c_kernel: the weights are register-resident)
                             "sa_kernel<128,128,128,256>": 64 * 32 * 512 / 1e6 + (64 / 12.0) * 131072 / 1e6}   # U rows, layer-3 panels per twelve centroids
        by_name = {q["name"]: q for q in rows}
        res["fill_path"] = {"unit": "TB/s", "ceiling_l2_resident_4_waves_per_cu": 25.0, "ceiling_infinity_cache": 8.7,
                            "source": "tools/ubench/gather2_bw.hip (profiles/r05_gather2_bw.txt); bytes = gathered rows + streamed weight fragments / panels per launch",
                            "stages": {n: {"gb_per_launch": round(mb * 9 * B / 1e3, 3), "tb_per_s": round(mb * 9 * B / 1e3 / by_name[n]["avg_ms"], 2),
                                           "frac_of_l2_ceiling": round(mb * 9 * B / 1e3 / by_name[n]["avg_ms"] / 25.0, 3)}
                                       for n, mb in fill_mb_per_cloud.items() if n in by_name and by_name[n]["avg_ms"] > 0}}
    if hoisted:
        res["hoisted_conditioning"] = hoisted
    res["forward_only"] = fwd
    res["instrumented"] = instrumented
    res["repeats"] = uninstrumented
    res["repeat_spread"] = res_spread
    res["strict_recompute" if inv else "invariant_motion"] = other
    del r
    torch.cuda.empty_cache()
    if not a.no_others and world == 1:
        others = {}
        for prec in ("f32", "bf16x3", "bf16"):
            if prec == a.precision:
                continue
            log("short run in %s" % prec)
            ko = 6 if prec == "f32" else 10
            ro = Runner(prec, B, C, 2 + ko, a.seed, local_rank, dev)
            ro.run(ro.T - 1, 0, 2, invariant=inv)
            torch.cuda.synchronize()
            x0o, dso = ro.x.clone(), []
            for _ in range(3):       # median of three repeats of the same call (a single shot after the set-up's idle time occasionally runs at a fraction of the rate)
                ro.x.copy_(x0o)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ro.run(ro.T - 3, 2, ko, invariant=inv)
                torch.cuda.synchronize()
                dso.append(time.perf_counter() - t1)
            d1 = sorted(dso)[1]
            others[prec] = {"value": ko / d1, "value_is": "median of 3 repeats", "value_min": ko / max(dso), "value_max": ko / min(dso), "ms_per_step": d1 / ko * 1e3,
                            "steps": ko, "n_gpus": 1, "whole_step_tflops": GFLOP_PER_SAMPLE_STEP * B / (d1 / ko * 1e3)}
            del ro
            torch.cuda.empty_cache()
        res["other_precisions"] = others
    if not a.no_configs and world == 1:
        log("short legs of the other BASELINE configs")
        res["configs"] = config_legs(a, dev, local_rank, inv)
    if not a.no_parity:
        log("parity check vs the CPU oracle")
        res["parity"] = parity_check(a.precision, dev)
    if not a.no_cpu_baseline and world == 1:      # the CPU leg is reported at N = 1 only
        log("cpu baseline (oracle) ...")
        res["cpu_baseline"] = cpu_baseline(a.cpu_batch, a.cpu_steps, C)
    print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
