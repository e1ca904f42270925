"""Run-to-run determinism of the whole forward under load (B = 16, 144 clouds: every persistent kernel runs several tiles per
work-group).  The reference is deterministic (models/sdm.py, pointnet2_utils.py have no atomics); so is every kernel here by
construction -- this test is the net for intra-kernel races and hardware hazards, which only show with the chip full
(tools/determinism_stress.py is the long form; round 4 found one in fp_ws_kernel with it: DESIGN.md 5, docs/HISTORY.md)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_repeated_forward_is_bit_identical_at_b16(precision):
    from determinism_stress import stress
    counts = stress(B=16, N=60 if precision == "bf16x3" else 12, precision=precision)
    assert not any(counts.values()), counts


@pytest.mark.parametrize("entry", ["invariant", "strict", "hoisted"])
def test_repeated_sampler_calls_are_bit_identical_at_b64(entry):
    """The benched configuration (B = 64: 576 clouds, 36 tiles per persistent work-group), six steps per call through each bulk entry point."""
    from determinism_stress import stress_loop
    assert stress_loop(B=64, K=6, N=12, precision="bf16x3", entry=entry) == 0


@pytest.mark.parametrize("compact", ["1", "0"])
def test_repeated_training_steps_agree(compact):
    """The training step runs FPS / ball queries / the compaction kernels on a stream of its own NEXT TO the conditioning GEMMs (train.hip): a
    lost update there would change indices -- discrete decisions -- and with them the loss at the 1e-3 level.  30 evaluations of the same
    batch (B = 6, run/train_sdm.py:247): loss terms equal to the first within the order of the atomic accumulations (2e-6 relative), the
    max-pool picks equal up to near-ties (<= 2e-5 of them), gradients within 1e-4 of the tensor's largest entry."""
    import torch
    from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
    from lsdm_amd.util.model_util import create_model_and_diffusion
    dev = torch.device("cuda", 0)
    B = 6
    old = os.environ.get("LSDM_TRAIN_COMPACT")
    os.environ["LSDM_TRAIN_COMPACT"] = compact          # (read when the training session is created)
    try:
        with torch.enable_grad():
            model, diffusion = create_model_and_diffusion("proxd", device=0)
            model.load_state_dict(make_synthetic_state_dict(seed=0))
            model.requires_grad_(True)
            model.train(False)
            bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=77, num_objects=8).items()}
            target = torch.zeros(B, 13, device=dev)
            target[:, 4] = 1
            torch.manual_seed(5)
            t = torch.randint(0, 1000, (B,), device=dev)
            noise = torch.randn(B, 1024, 3, device=dev)
            starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
            x_start = bt["given_objs"][:, 1].clone()
            first = None
            for it in range(30):
                model.zero_grad()
                terms = diffusion.training_losses(model, x_start, bt["mask"], t, bt["given_objs"], bt["given_cats"], target, y=bt["text_features"],
                                                  noise=noise, fps_starts=starts)
                terms["loss"].backward()
                cur = ({k: float(terms[k]) for k in ("loss", "mse", "cat_loss")}, [a.clone() for a in model._train_session.pool_choices(B)],
                       {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
                if first is None:
                    first = cur
                    continue
                for k in cur[0]:
                    assert abs(cur[0][k] - first[0][k]) <= 2e-6 * max(1.0, abs(first[0][k])), (it, k, cur[0][k], first[0][k])
                differing = sum(int((a != b).sum()) for a, b in zip(cur[1], first[1]))
                assert differing <= 2e-5 * sum(a.numel() for a in first[1]), (it, differing)
                for n in first[2]:
                    assert float((cur[2][n] - first[2][n]).abs().max()) <= 1e-4 * (float(first[2][n].abs().max()) + 1e-30), (it, n)
    finally:
        if old is None:
            os.environ.pop("LSDM_TRAIN_COMPACT", None)
        else:
            os.environ["LSDM_TRAIN_COMPACT"] = old


def test_bench_json_line_carries_the_contract_fields():
    """`python bench.py` as the driver runs it (N = 1; short): ONE JSON line with the contract's fields, the roofline of the dominant kernel with
    its three fractions, the fill-path figures of the gather-bound stages, value = 1000 / ms_per_step, and a short leg for every BASELINE config
    that fits one GPU beside the headline (configs[1] in f32 and bf16x3, configs[4]'s HUMANISE / DDIM-250 shapes, surface-sampled clouds)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-others", "--no-hoisted", "--no-cpu-baseline",
                        "--no-parity"], capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["dtype"] == "bf16x3" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) <= 1e-6 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # executed <= reference FLOPs (split first layers, the memo, the item lists); x3 matrix instructions per executed product
    assert 0 < rf["frac_executed"] <= rf["frac"] and abs(rf["mfma_issue_frac"] - 3 * rf["frac_executed"]) < 1e-9 and rf["mfma_issue_frac"] < 1.0
    # every stage of the list: executed work within the matrix pipe's issue rate and not above the stage's reference work; the reference-FLOP
    # column only where the launches execute that work
    assert rf["guard_violations"] == [] and all(k["mfma_issue_frac"] < 1.0 for k in rf["kernels"])
    assert all(k["tflops"] is None or k["tflops"] <= 2500.0 / 3 * 1.02 for k in rf["kernels"]), [(k["name"], k["tflops"]) for k in rf["kernels"]]
    assert set(r["fill_path"]["stages"]) >= {"fp_kernel<fp2>", "fp_kernel<fp1+head>"} and 0 < r["fill_path"]["stages"]["fp_kernel<fp1+head>"]["frac_of_l2_ceiling"] < 1.0
    assert r["repeats"]["final_sample_bit_identical_to_timed_region"] is True
    cf = r["configs"]
    assert set(cf) >= {"configs[1]_f32", "configs[1]_bf16x3", "configs[4]_one_gpu", "headline_surface_sampled", "ragged_k"}
    assert cf["ragged_k"]["clouds"]["live"] < 0.75 * cf["ragged_k"]["clouds"]["per_step"] and cf["ragged_k"]["vs_all_8_objects"] > 1.2
    for e in cf.values():
        assert e["value"] > 0 and e["finite"] is True and abs(e["value"] - 1e3 / e["ms_per_step"]) <= 1e-6 * e["value"]
    assert cf["configs[1]_f32"]["batch"] == 16 and cf["configs[1]_f32"]["max_cats"] == 43 and cf["configs[1]_f32"]["dtype"] == "f32"
    assert cf["configs[4]_one_gpu"]["sampler"] == "ddim" and cf["configs[4]_one_gpu"]["schedule_T"] == 250 and cf["configs[4]_one_gpu"]["max_cats"] == 11
    assert 0.0 < cf["headline_surface_sampled"]["sa2_memo"]["centroid_hit_rate"] <= 1.0
    for leg in ("headline_surface_sampled", "headline_volume_reference"):       # sa3 / sa4 over item lists: between one block per two groups and one per group
        il = cf[leg]["sa_item_lists"]
        assert 0.5 <= il["sa3_blocks_per_group"] < 1.0 and 0.5 <= il["sa4_blocks_per_group"] < 1.0, il
