"""GPU parity at the BASELINE configurations themselves, through the C ABI:

  configs[2]  B = 64, 8 objects, C = 13 (the bench workload): three p_sample steps against the oracle with every discrete
              decision compared on all 576 clouds x 4 levels, and the effect of the reference's unspecified order of exactly
              tied 3-NN distances bounded by construction (the kernel's equally-near choice injected into the oracle);
  configs[0]  run.test_sdm on the HIP path (default backend) against the same CLI driven by the oracle;
  configs[4]  HUMANISE defaults, two prompts per scene, ddim250 -- golden recorded from the real reference (G7);
  the argument variants of p_sample_loop (G9), INTEGRATION.md path B executed, state changes on a live model
  (re-loaded parameters, precision switches, batch-size changes, two schedules of equal length), input validation,
  the metric kernels, and the two-rank RCCL path when two devices are visible."""
import importlib.util
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT
from lsdm_amd import _lib
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_gaussian_diffusion, create_model_and_diffusion, get_default_diffusion
from oracle import cpu_ref
from test_hip_parity import T, maxabs, ws

torch.set_grad_enabled(False)
DEV = "cuda:0"
S_LEVEL = (1024, 256, 64, 16)
# workspace name of the 3-NN of each FP stage: (fine points, coarse points)
NN_WS = {"fp4": ("nn_idx1", 64, 16), "fp3": ("nn_idx2", 256, 64), "fp2": ("nn_idx3", 1024, 256), "fp1": ("nn_idx4", 1024, 1024)}


# ======================================================================================= configs[2]: B = 64
class B64:
    """Everything the B = 64 tests compare, computed once per session (three oracle steps at B = 64 are ~3 minutes)."""
    B, K, C = 64, 3, 13


@pytest.fixture(scope="module")
def b64():
    r = B64()
    B, K, C = r.B, r.K, r.C
    nc = B * 9
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = make_synthetic_state_dict(seed=0, max_cats=C)
    bt = make_synthetic_batch(B, seed=1, max_cats=C, num_objects=8)          # bench.py's workload
    d = {k: v.to(DEV).contiguous() for k, v in bt.items()}
    # the reference's RNG stream: x_T, then per step 4 FPS randints + randn_like
    torch.manual_seed(1)
    x_T = torch.randn(B, 1024, 3)
    starts, noise = [], []
    for _ in range(K):
        starts.append(cpu_ref.draw_fps_starts(nc))
        noise.append(torch.randn(B, 1024, 3))
    r.starts, r.noise, r.x_T, r.bt, r.d = starts, noise, x_T, bt, d

    # ---- HIP: K single-step calls per precision, geometry taps after every step
    r.hip, r.geo = {}, []
    for prec in ("f32", "bf16x3"):
        m, diff = create_model_and_diffusion("proxd", precision=prec)
        m.load_state_dict(sd)
        m._set_schedule(diff._schedule_key(), diff._device_tables())
        x = x_T.clone().to(DEV)
        steps = []
        for k in range(K):
            t = torch.full((B,), 999 - k, dtype=torch.long, device=DEV)
            nk = noise[k].to(DEV).contiguous()
            x = x.contiguous()
            _, x0, smp = m._run(_lib.MODE_P_SAMPLE, 0.0, x, d["mask"], t, d["given_objs"], d["given_cats"], d["text_features"],
                                fps_starts=starts[k], noise_fn=lambda: nk)
            geo = {}
            for l in range(4):
                geo["fps%d" % (l + 1)] = ws(m, "fps_idx%d" % (l + 1), (nc, S_LEVEL[l]), torch.int32).long()
                geo["ball%d" % (l + 1)] = ws(m, "ball%d" % (l + 1), (nc, S_LEVEL[l], 32), torch.int32).long()
            for name, (wsn, n_fine, _) in NN_WS.items():
                geo[name] = ws(m, wsn, (nc, n_fine, 3), torch.int32).long()
            if prec == "f32":
                r.geo.append(geo)
            else:       # geometry is fp32 in every precision mode: identical decisions
                assert all(torch.equal(geo[key], r.geo[k][key]) for key in geo), "geometry differs between precisions at step %d" % k
            steps.append(dict(sample=smp.cpu(), x0=x0.cpu(), x_mut=x.cpu(), cat=m.saved_cat.cpu(), guid=m.saved_guiding_points.cpu()))
            x = smp
        r.hip[prec] = steps
        # the bench path: the same K steps inside ONE library call (strict and invariant schedules)
        r.hip[prec + "_bulk"] = {}
        for entry in ("lsdm_sample_steps", "lsdm_sample_steps_invariant"):
            xb = x_T.clone().to(DEV)
            m._run_steps(_lib.MODE_P_SAMPLE, 0.0, xb, d["mask"], 999, K, d["given_objs"], d["given_cats"], d["text_features"],
                         torch.stack([torch.stack(s) for s in starts]), torch.stack(noise), invariant=entry.endswith("invariant"))
            torch.cuda.synchronize()
            r.hip[prec + "_bulk"][entry] = (xb.cpu(), m.saved_guiding_points.cpu(), m.saved_cat.cpu())
        del m
        torch.cuda.empty_cache()

    # ---- oracle: the scene encoder once per step (x-independent), then two chains -- raw, and with the kernel's 3-NN
    # choice injected for the clouds where it differs from the oracle's sort order
    smp = cpu_ref.Sampler(sd)
    clouds = bt["given_objs"].reshape(nc, 1024, 3)
    r.raw, r.inj, r.ties, r.taps = [], [], [], []
    x_raw, x_inj = x_T.clone(), x_T.clone()
    for k in range(K):
        taps = {}
        pcd = cpu_ref.scene_encoder(sd, clouds, starts[k], taps)
        tie_rows, bad = {}, torch.zeros(nc, dtype=torch.bool)
        for name in NN_WS:
            differs = (r.geo[k][name] != taps[name + ".nn_idx"]).any(dim=-1)          # (nc, N)
            tie_rows[name] = int(differs.sum())
            bad |= differs.any(dim=-1)
        sel = torch.nonzero(bad).flatten()
        pcd_inj = pcd.clone()
        if len(sel):
            inj = {name: r.geo[k][name][sel] for name in NN_WS}
            pcd_inj[sel] = cpu_ref.scene_encoder(sd, clouds[sel], [s[sel] for s in starts[k]], None, inj)
        t = torch.full((B,), 999 - k, dtype=torch.long)
        o_raw = smp.p_sample(x_raw, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"], fps_starts=starts[k],
                             noise=noise[k], pcd_override=pcd)
        # (clones: the next forward mutates its input in place, Q5)
        r.raw.append(dict(sample=o_raw["sample"].clone(), x0=o_raw["pred_xstart"], cat=smp.saved_cat, guid=smp.saved_guiding_points))
        o_inj = smp.p_sample(x_inj, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"], fps_starts=starts[k],
                             noise=noise[k], pcd_override=pcd_inj)
        r.inj.append(dict(sample=o_inj["sample"].clone(), x0=o_inj["pred_xstart"], cat=smp.saved_cat, guid=smp.saved_guiding_points))
        x_raw, x_inj = o_raw["sample"], o_inj["sample"]
        r.ties.append(dict(rows=tie_rows, clouds=int(len(sel))))
        r.taps.append({key: v for key, v in taps.items() if key.endswith(("fps_idx", "ball_idx", "nn_idx", "new_xyz"))})
    torch.set_num_threads(threads)
    return r


def test_config3_b64_fps_and_ball_indices_equal_on_all_clouds(b64):
    for k in range(b64.K):
        for l in range(4):
            assert torch.equal(b64.geo[k]["fps%d" % (l + 1)], b64.taps[k]["sa%d.fps_idx" % (l + 1)]), (k, l)
            assert torch.equal(b64.geo[k]["ball%d" % (l + 1)], b64.taps[k]["sa%d.ball_idx" % (l + 1)]), (k, l)


def test_config3_b64_three_nn_selected_distances_bit_identical_and_ties_counted(b64):
    """Wherever the kernel's three neighbours differ from the oracle's, the computed fp32 distances of the two choices are
    bit-identical (an exact tie, ordered by torch.sort's unspecified rule in the reference and by the lowest index here)."""
    clouds = b64.bt["given_objs"].reshape(-1, 1024, 3)
    total = 0
    for k in range(b64.K):
        xyz = [clouds] + [b64.taps[k]["sa%d.new_xyz" % (l + 1)] for l in range(4)]
        for name, fine, coarse in (("fp4", 3, 4), ("fp3", 2, 3), ("fp2", 1, 2), ("fp1", 0, 1)):
            got, ref = b64.geo[k][name], b64.taps[k][name + ".nn_idx"]
            rows = torch.nonzero((got != ref).any(dim=-1))                      # (n, 2): cloud, point
            total += len(rows)
            assert len(rows) == b64.ties[k]["rows"][name]
            if len(rows) == 0:
                continue
            cl, pt = rows[:, 0], rows[:, 1]
            dist = cpu_ref.square_distance(xyz[fine][cl, pt][:, None, :], xyz[coarse][cl])[:, 0]       # (n, S)
            dg, dr = torch.gather(dist, 1, got[cl, pt]), torch.gather(dist, 1, ref[cl, pt])
            assert torch.equal(dg.sort(dim=1)[0], dr.sort(dim=1)[0]), (k, name)
            # and they ARE the three smallest
            assert torch.equal(dg.sort(dim=1)[0], dist.sort(dim=1)[0][:, :3]), (k, name)
    points = b64.K * 576 * (64 + 256 + 1024 + 1024)
    print("3-NN rows with a different equally-near choice: %d of %d; per step %s" % (total, points, b64.ties))
    assert total <= points * 1e-4


@pytest.mark.parametrize("precision,tol_inj", [("f32", 5e-5), ("bf16x3", 1e-4)])
def test_config3_b64_chain_vs_oracle_with_tie_attribution(b64, precision, tol_inj):
    """Raw comparison within the 1e-3 bar; with the kernel's equally-near neighbour choice injected into the oracle the
    sample agrees to rounding (f32: 5e-5; bf16x3: 1e-4, its dropped 2^-18 term) -- so whatever exceeds that in the raw
    comparison is the tie order and nothing else."""
    for k in range(b64.K):
        h, raw, inj = b64.hip[precision][k], b64.raw[k], b64.inj[k]
        errs = {key: (maxabs(h[key], raw[key]), maxabs(h[key], inj[key])) for key in ("sample", "x0", "guid", "cat")}
        print("B=64 %s step %d: max-abs (raw, tie-injected) %s" % (precision, k, errs))
        assert errs["cat"][0] <= 1e-6
        # contractual RAW: what the caller of the loop reads (pred_xstart = the final sample at t = 0, saved_cat, the guiding points)
        for key in ("x0", "guid"):
            assert errs[key][0] <= 1e-3, (k, key, errs)
        # a per-step `sample` carries coef2 x the in-place guided x (Q5), i.e. the reference's unspecified order of exactly tied 3-NN distances
        # at full weight: held to rounding with the kernel's equally-near choice injected; the raw figure is printed with the tie count above
        # (8.8e-4 at another seed, profiles/r06_parity_extra.txt) and bounded loosely -- a kernel fault shows in the injected comparison
        for key in ("sample", "x0", "guid"):
            assert errs[key][1] <= tol_inj, (k, key, errs)
        assert errs["sample"][0] <= 5e-3, (k, errs, b64.ties[k])


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_config3_b64_bulk_calls_bit_equal_to_single_steps(b64, precision):
    last = b64.hip[precision][-1]
    for entry, (x, guid, cat) in b64.hip[precision + "_bulk"].items():
        assert torch.equal(x, last["sample"]), entry
        assert torch.equal(guid, last["guid"]) and torch.equal(cat, last["cat"]), entry


# ---- the same bar away from t = 999: one step with a different timestep per chain, and a mid-schedule segment (round 6) ----------------
def _b64_case(seed, ts, x_init):
    """len(ts) consecutive single-step calls at B = 64 (bench workload, batch seed `seed`) from x_init with per-chain timesteps ts[k] (B,),
    in f32 and bf16x3, against the oracle run twice -- raw, and with the kernel's equally-near 3-NN choice injected.  Returns
    {precision: [per step {key: (raw, injected)}]}, the tie rows per step."""
    B, C, nc, K = 64, 13, 576, len(ts)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = make_synthetic_state_dict(seed=0, max_cats=C)
    bt = make_synthetic_batch(B, seed=seed, max_cats=C, num_objects=8)
    d = {k: v.to(DEV).contiguous() for k, v in bt.items()}
    torch.manual_seed(100 + seed)
    starts = [cpu_ref.draw_fps_starts(nc) for _ in range(K)]
    noise = [torch.randn(B, 1024, 3) for _ in range(K)]
    hip, geo = {}, []
    for prec in ("f32", "bf16x3"):
        m, diff = create_model_and_diffusion("proxd", precision=prec)
        m.load_state_dict(sd)
        m._set_schedule(diff._schedule_key(), diff._device_tables())
        x = x_init.clone().to(DEV)
        steps = []
        for k in range(K):
            nk = noise[k].to(DEV).contiguous()
            x = x.contiguous()
            _, x0, smp = m._run(_lib.MODE_P_SAMPLE, 0.0, x, d["mask"], ts[k].to(DEV), d["given_objs"], d["given_cats"], d["text_features"],
                                fps_starts=starts[k], noise_fn=lambda: nk)
            if prec == "f32":
                geo.append({name: ws(m, wsn, (nc, n_fine, 3), torch.int32).long() for name, (wsn, n_fine, _) in NN_WS.items()})
            steps.append(dict(sample=smp.cpu(), x0=x0.cpu(), cat=m.saved_cat.cpu(), guid=m.saved_guiding_points.cpu()))
            x = smp
        hip[prec] = steps
        del m
        torch.cuda.empty_cache()
    smp = cpu_ref.Sampler(sd)
    clouds = bt["given_objs"].reshape(nc, 1024, 3)
    x_raw, x_inj = x_init.clone(), x_init.clone()
    errs, ties = {"f32": [], "bf16x3": []}, []
    for k in range(K):
        taps = {}
        pcd = cpu_ref.scene_encoder(sd, clouds, starts[k], taps)
        bad, rows = torch.zeros(nc, dtype=torch.bool), 0
        for name in NN_WS:
            differs = (geo[k][name] != taps[name + ".nn_idx"]).any(dim=-1)
            rows += int(differs.sum())
            bad |= differs.any(dim=-1)
        sel = torch.nonzero(bad).flatten()
        pcd_inj = pcd.clone()
        if len(sel):
            pcd_inj[sel] = cpu_ref.scene_encoder(sd, clouds[sel], [s_[sel] for s_ in starts[k]], None, {name: geo[k][name][sel] for name in NN_WS})
        outs = []
        for xk, pk in ((x_raw, pcd), (x_inj, pcd_inj)):
            o = smp.p_sample(xk, bt["mask"], ts[k], bt["given_objs"], bt["given_cats"], bt["text_features"], fps_starts=starts[k], noise=noise[k],
                             pcd_override=pk)
            outs.append(dict(sample=o["sample"].clone(), x0=o["pred_xstart"], cat=smp.saved_cat, guid=smp.saved_guiding_points))
        x_raw, x_inj = outs[0]["sample"], outs[1]["sample"]
        ties.append(dict(rows=rows, clouds=int(len(sel))))
        for prec in errs:
            h = hip[prec][k]
            e = {key: (maxabs(h[key], outs[0][key]), maxabs(h[key], outs[1][key])) for key in ("sample", "x0", "guid", "cat")}
            e["scale"] = float(outs[0]["sample"].abs().max())
            errs[prec].append(e)
    torch.set_num_threads(threads)
    return errs, ties


def _b64_check(tag, errs, ties):
    for prec, tol in (("f32", 5e-5), ("bf16x3", 1e-4)):
        for k, e in enumerate(errs[prec]):
            print("B=64 %s %s step %d: max-abs (raw, tie-injected) sample %.3e %.3e  pred_xstart %.3e %.3e  guiding %.3e %.3e  saved_cat %.1e; |sample|max %.1f; "
                  "3-NN rows with another equally-near choice: %d in %d clouds" % (tag, prec, k, e["sample"][0], e["sample"][1], e["x0"][0], e["x0"][1], e["guid"][0],
                                                                                  e["guid"][1], e["cat"][0], e["scale"], ties[k]["rows"], ties[k]["clouds"]))
            assert e["cat"][0] <= 1e-6, (tag, prec, k, e)
            for key in ("x0", "guid"):            # contractual, RAW: what the caller reads after the loop
                assert e[key][0] <= 1e-3, (tag, prec, k, key, e)
            # per-step sample: to rounding once the reference's unspecified tie order is taken out (relative to the sample's magnitude mid-schedule)
            for key in ("sample", "x0", "guid"):
                assert e[key][1] <= tol * max(1.0, e["scale"] / 4.0 if key == "sample" else 1.0), (tag, prec, k, key, e, ties[k])


@pytest.mark.parametrize("seed", [1])
def test_config3_b64_mixed_timesteps_one_step(seed):
    """One p_sample step at the benched batch with a DIFFERENT timestep per chain (t_b = (991 - 7 b) mod 1000: the whole schedule in one batch --
    tools/parity_b64.py's case, where the raw per-step sample sits at 8.5e-4 .. 1.0e-3 through the 3-NN tie order alone).  (An oracle step at
    B = 64 is a minute of host time: the suite runs this case on seed 1 and the segment below on seed 2; tools/parity_b64_mid.py runs both on both
    seeds with five-step segments -- profiles/r06_parity_extra.txt.)"""
    B = 64
    torch.manual_seed(seed)
    x = torch.randn(B, 1024, 3)
    ts = [torch.tensor([(991 - 7 * b) % 1000 for b in range(B)])]
    _b64_check("mixed-t seed %d" % seed, *_b64_case(seed, ts, x))


@pytest.mark.parametrize("seed,K", [(2, 3)])
def test_config3_b64_mid_schedule_segment(seed, K):
    """Steps t = 524 .. 524 - K + 1 at the benched batch from x_524 = q_sample(0, 524, noise) (p_sample_loop(skip_timesteps=475)'s start): the
    guidance term and coef2 are at full weight here, unlike at t = 999 .. 997 -- on the seed whose raw per-step sample exceeds 1e-3 in bf16x3
    through the tie order alone (1.03e-3 .. 1.11e-3; 7e-5 .. 3.4e-4 of a sample that has grown to 37 with the choice injected)."""
    B = 64
    _, diff = create_model_and_diffusion("proxd")
    torch.manual_seed(10 + seed)
    t0 = torch.full((B,), 524, dtype=torch.long)
    x = diff.q_sample(torch.zeros(B, 1024, 3), t0, torch.randn(B, 1024, 3))
    ts = [t0 - k for k in range(K)]
    _b64_check("t=524.. seed %d" % seed, *_b64_case(seed, ts, x))


# ======================================================================================= configs[0]: the CLI on the HIP path
def test_config1_cli_on_hip_backend_vs_oracle_backend(tmp_path, golden):
    """python -m run.test_sdm --synthetic 4 --use_ddim --timestep_respacing ddim50 with the DEFAULT backend (HipBackend),
    against the same CLI driven by the oracle (recorded by oracle/make_cli_fixture.py into g10_cli_ddim50: 200 oracle
    forwards, ten minutes of CPU time on the GPU box otherwise; LSDM_FULL_CONFIG0=1 re-runs the oracle CLI live instead):
    predictions / guiding points <= 1e-3, results.txt in the reference's format (run/test_sdm.py:211,227-231) with matching figures."""
    import run.test_sdm as cli
    argv = lambda root, out, dev: [str(root), "--synthetic", "4", "--device", dev, "--use_ddim", "--timestep_respacing", "ddim50",
                                   "--output_dir", str(out), "--fix_ori", "--test_on_valid_set", "--model_name", "sdm"]
    assert cli.main(argv(tmp_path / "d_hip", tmp_path / "hip", "cuda:0")) == 0
    if os.environ.get("LSDM_FULL_CONFIG0") == "1":
        from test_host_logic import OracleBackend
        assert cli.main(argv(tmp_path / "d_ref", tmp_path / "ref", "cpu"), backend_factory=OracleBackend) == 0
        ref = {sub: np.stack([np.load(tmp_path / "ref" / sub / f) for f in sorted(os.listdir(tmp_path / "ref" / sub))])
               for sub in ("predictions", "guiding_points")}
        ref_files = {sub: sorted(os.listdir(tmp_path / "ref" / sub)) for sub in ref}
        lr = open(tmp_path / "ref" / "results.txt").read().splitlines()
    else:
        g = golden("g10_cli_ddim50")
        ref = {sub: g[sub] for sub in ("predictions", "guiding_points")}
        ref_files = {sub: [str(f) for f in g[sub + "_files"]] for sub in ref}
        lr = str(g["results_txt"]).splitlines()
    for sub in ("predictions", "guiding_points"):
        files = sorted(os.listdir(tmp_path / "hip" / sub))
        assert files == ref_files[sub] and len(files) == 4
        for i, f in enumerate(files):
            a, b = np.load(tmp_path / "hip" / sub / f), ref[sub][i]
            assert a.shape == (1024, 3) and a.dtype == np.float32
            assert np.abs(a - b).max() <= 1e-3, (sub, f, np.abs(a - b).max())
    lh = open(tmp_path / "hip" / "results.txt").read().splitlines()
    assert len(lh) == len(lr) == 4 + 5
    assert all(re.fullmatch(r"Chamfer distance for seq Scene\d\d_\d{5}_01: \d+\.\d{4}", l) for l in lh[:4])
    for a, b, prefix in zip(lh[4:], lr[4:], ("Final Chamfer distance: ", "Final EMD: ", "Final F1 score: ", "Category accuracy: ",
                                              "Top 3 accuracy: ")):
        assert a.startswith(prefix) and b.startswith(prefix)
        assert abs(float(a[len(prefix):]) - float(b[len(prefix):])) <= 2e-3, (a, b)
    for a, b in zip(lh[:4], lr[:4]):
        assert a.split(":")[0] == b.split(":")[0] and abs(float(a.split(": ")[1]) - float(b.split(": ")[1])) <= 2e-3


def test_cli_batch_size_4_on_hip_backend_vs_oracle_backend(tmp_path, golden):
    """--batch_size 4 (an addition: the reference's loop is B = 1): the four synthetic samples as ONE batch of chains, ddim5, against
    the same CLI driven by the oracle (g10b_cli_ddim5_b4, oracle/make_cli_fixture.py --b4) -- files, per-sample lines, finals."""
    import run.test_sdm as cli
    out = tmp_path / "out"
    assert cli.main([str(tmp_path / "data"), "--synthetic", "4", "--device", "cuda:0", "--use_ddim", "--timestep_respacing", "ddim5",
                     "--batch_size", "4", "--output_dir", str(out), "--fix_ori", "--test_on_valid_set", "--model_name", "sdm"]) == 0
    g = golden("g10b_cli_ddim5_b4")
    for sub in ("predictions", "guiding_points"):
        files = sorted(os.listdir(out / sub))
        assert files == [str(f) for f in g[sub + "_files"]] and len(files) == 4
        for i, f in enumerate(files):
            a = np.load(out / sub / f)
            assert a.shape == (1024, 3) and a.dtype == np.float32 and np.abs(a - g[sub][i]).max() <= 1e-3, (sub, f)
    lh, lr = open(out / "results.txt").read().splitlines(), str(g["results_txt"]).splitlines()
    assert len(lh) == len(lr) == 9
    for a, b in zip(lh, lr):
        assert a.split(": ")[0] == b.split(": ")[0] and abs(float(a.split(": ")[1]) - float(b.split(": ")[1])) <= 2e-3, (a, b)


def test_cli_humanise_datatype_on_hip_backend(tmp_path):
    """--datatype humanise: the HUMANISE loader (own category table, C = 11, scene = seq[:9] + '_00') feeding the HIP path."""
    import run.test_sdm as cli
    out = tmp_path / "out"
    assert cli.main([str(tmp_path / "data"), "--synthetic", "4", "--datatype", "humanise", "--use_ddim", "--timestep_respacing", "ddim5",
                     "--output_dir", str(out), "--device", "cuda:0"]) == 0
    lines = open(out / "results.txt").read().splitlines()
    assert len(lines) == 4 + 5 and all(l.startswith("Chamfer distance for seq scene000") for l in lines[:4])
    assert len(os.listdir(out / "predictions")) == 4


# ======================================================================================= configs[4]: HUMANISE / ddim250 golden
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_config5_humanise_two_prompts_ddim250_vs_reference_golden(golden, precision):
    g = golden("g7_ddim250_humanise")
    B, C_, K, seed, bt_seed, txt_seed = [int(v) for v in g["meta"]]
    m, _ = create_model_and_diffusion("humanise", precision=precision)
    assert m.max_cats == C_ == 11
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C_))
    diff = create_gaussian_diffusion(get_default_diffusion(), timestep_respacing="ddim250")
    one = make_synthetic_batch(1, seed=bt_seed, max_cats=C_)
    bt = {k: v.repeat(2, *([1] * (v.dim() - 1))).contiguous() for k, v in one.items()}
    bt["text_features"] = make_synthetic_batch(2, seed=txt_seed, max_cats=C_)["text_features"]
    d = {k: v.to(DEV) for k, v in bt.items()}
    torch.manual_seed(seed)
    img = torch.randn(B, 1024, 3).to(DEV)
    for k, i in enumerate(list(range(250))[::-1][:K]):
        out = diff.ddim_sample(m, img, d["mask"], torch.tensor([i] * B, device=DEV), d["given_objs"], d["given_cats"],
                               d["text_features"], clip_denoised=False)
        img = out["sample"]
        assert maxabs(img, T(g["samples"][k])) <= 1e-3, k
    assert maxabs(m.saved_cat, T(g["saved_cat"])) <= 1e-6
    assert maxabs(m.saved_guiding_points, T(g["guiding"])) <= 1e-3
    # the bulk DDIM loop (one library call per chunk) against the per-step loop over the same range: skip_timesteps drops the
    # NOISIEST steps (gaussian_diffusion.py:726-733: x = q_sample(zeros, t_first, noise), then indices t_first .. 0), and both
    # loops draw the same stream (x_T, then per step 4 randints + randn_like)
    kw = dict(clip_denoised=False, skip_timesteps=250 - K)
    torch.manual_seed(seed)
    bulk = diff.ddim_sample_loop(m, [B, 1024, 3], d["mask"], d["given_objs"], d["given_cats"], d["text_features"], **kw)
    torch.manual_seed(seed)
    last = None
    for last in diff.ddim_sample_loop_progressive(m, [B, 1024, 3], d["mask"], d["given_objs"], d["given_cats"], d["text_features"], **kw):
        pass
    assert maxabs(bulk, last["sample"]) <= (1e-6 if precision == "f32" else 1e-5)


# ======================================================================================= p_sample_loop argument variants (G9)
def _g9_setup(golden):
    g = golden("g9_loop_variants")
    B, C_, K, bt_seed, _ = [int(v) for v in g["meta"]]
    m, diff = create_model_and_diffusion("proxd", max_cats=C_)
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C_))
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=bt_seed, max_cats=C_).items()}
    args = (m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"])
    return g, diff, args, B, K


@pytest.mark.parametrize("fused", [True, False])
def test_p_sample_loop_noise_init_image_dump_steps_vs_reference_golden(golden, fused):
    """noise= (no x_T draw), init_image + skip_timesteps (q_sample of the image at t = K-1), dump_steps -- recorded from the
    reference's p_sample_loop (gaussian_diffusion.py:611-759).  fused=False takes the unfused API path
    (p_mean_variance + tensor ops) through an identity denoised_fn."""
    g, diff, args, B, K = _g9_setup(golden)
    kw = dict(clip_denoised=False, skip_timesteps=1000 - K)
    if not fused:
        kw["denoised_fn"] = lambda v: v
    torch.manual_seed(93)
    out = diff.p_sample_loop(*args, noise=T(g["noise0"]).clone().to(DEV), **kw)
    assert maxabs(out, T(g["noise_arg"])) <= 1e-3
    torch.manual_seed(94)
    out = diff.p_sample_loop(*args, init_image=T(g["init"]).clone().to(DEV), **kw)
    assert maxabs(out, T(g["init_image"])) <= 1e-3
    torch.manual_seed(95)
    dump = diff.p_sample_loop(*args, dump_steps=[0, 2], **kw)
    assert isinstance(dump, list) and len(dump) == 2
    assert maxabs(dump[0], T(g["dump0"])) <= 1e-3 and maxabs(dump[1], T(g["dump2"])) <= 1e-3


def test_const_noise_uses_one_noise_row_for_every_chain(golden):
    """const_noise=True in the reference repeats noise[[0]] with FOUR repeat factors on a 3-D sample (an MDM leftover,
    gaussian_diffusion.py:546-548): one step returns a (B,B,1024,3) tensor and a second step cannot run (recorded in G9).
    This implementation gives the intended semantics -- chain j is noised with row 0 -- which is the diagonal [j, j] of the
    reference's single-step result."""
    g, diff, args, B, K = _g9_setup(golden)
    assert list(g["const_noise_shape"]) == [B, B, 1024, 3] and int(g["const_noise_two_steps_ok"]) == 0
    torch.manual_seed(96)
    out = diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=999, const_noise=True)
    assert out.shape == (B, 1024, 3)
    assert maxabs(out, T(g["const_noise_diag"])) <= 1e-3
    torch.manual_seed(96)     # and the per-step API path agrees with the bulk path
    last = [o for o in diff.p_sample_loop_progressive(*args, clip_denoised=False, skip_timesteps=999, const_noise=True)][-1]
    assert torch.equal(last["sample"], out)


def test_ddim_sample_loop_dump_steps(golden):
    g, _, args, B, K = _g9_setup(golden)
    diff = create_gaussian_diffusion(get_default_diffusion(), timestep_respacing="ddim50")
    torch.manual_seed(5)
    dump = diff.ddim_sample_loop(*args, clip_denoised=False, skip_timesteps=50 - K, dump_steps=[0, K - 1])
    torch.manual_seed(5)
    final = diff.ddim_sample_loop(*args, clip_denoised=False, skip_timesteps=50 - K)
    assert len(dump) == 2 and torch.equal(dump[1], final) and not torch.equal(dump[0], final)


# ======================================================================================= INTEGRATION.md path B, executed
def test_integration_path_b_hipforward():
    """integration/sdm_hip.py (shown verbatim in INTEGRATION.md) bound under a stand-in for the reference's module object:
    state_dict(), max_cats and _encode_text_clip are all it touches."""
    os.environ["LSDM_HIP_LIB"] = _lib.LIB_PATH
    spec = importlib.util.spec_from_file_location("sdm_hip", os.path.join(ROOT, "integration", "sdm_hip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sd = make_synthetic_state_dict(seed=0, max_cats=13)
    B = 3
    bt = make_synthetic_batch(B, seed=5)

    class RefModelStandIn:
        max_cats = 13

        def state_dict(self):
            full = dict(sd)
            full["clip_model.positional_embedding"] = torch.zeros(77, 512)     # real checkpoints carry the CLIP tower
            return full

        def _encode_text_clip(self, y):
            assert list(y) == ["p"] * B
            return bt["text_features"].to(DEV)

    fwd = mod.HipForward(RefModelStandIn(), device=0, max_batch=4)
    torch.manual_seed(3)
    x0 = torch.randn(B, 1024, 3)
    t = torch.tensor([987, 500, 3])
    torch.manual_seed(4)
    starts = cpu_ref.draw_fps_starts(B * 9)
    x_ref = x0.clone()
    cat_o, out_o, guid_o = cpu_ref.forward(sd, x_ref, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"], fps_starts=starts)
    x = x0.clone().to(DEV)
    d = {k: v.to(DEV) for k, v in bt.items()}
    torch.manual_seed(4)                     # HipForward draws the FPS starts where the reference does
    out_cat, out = fwd(x, d["mask"], t.to(DEV), d["given_objs"], d["given_cats"], y=["p"] * B)
    torch.cuda.synchronize()
    assert maxabs(out_cat, cat_o) <= 1e-6 and maxabs(out, out_o) <= 5e-4 and maxabs(x, x_ref) <= 5e-4
    assert maxabs(fwd.saved_guiding_points, guid_o) <= 5e-4
    fwd.close()


# ======================================================================================= state changes on a live model
def _chain(m, diff, bt, B, K, seed, **kw):
    torch.manual_seed(seed)
    out = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"][:B], bt["given_objs"][:B], bt["given_cats"][:B], bt["text_features"][:B],
                             clip_denoised=False, skip_timesteps=diff.num_timesteps - K, chunk_steps=K, **kw)
    return out.clone(), m.saved_guiding_points.clone(), m.saved_cat.clone()


def _fresh(precision="f32", C_=13):
    m, diff = create_model_and_diffusion("proxd", precision=precision, max_cats=C_)
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C_))
    return m, diff


def test_reload_and_precision_switch_keep_the_schedule_valid():
    """sample -> load_state_dict again -> set_precision("bf16x3") and back -> sample again == a fresh model, bit for bit.
    (lsdm_finalize_params frees the parameter arena; the schedule tables live in their own allocations.)"""
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(2, seed=12).items()}
    m, diff = _fresh()
    first = _chain(m, diff, bt, 2, 3, 8)
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=13))
    again = _chain(m, diff, bt, 2, 3, 8)
    m.set_precision("bf16x3")
    x3 = _chain(m, diff, bt, 2, 3, 8)
    m.set_precision("f32")
    back = _chain(m, diff, bt, 2, 3, 8)
    ref3 = _chain(*_fresh("bf16x3"), bt, 2, 3, 8)
    for a, b in ((first, again), (first, back), (x3, ref3)):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert not torch.equal(first[0], x3[0])
    # other weights through the same context
    m.load_state_dict(make_synthetic_state_dict(seed=1, max_cats=13))
    other = _chain(m, diff, bt, 2, 3, 8)
    m2, _ = create_model_and_diffusion("proxd")
    m2.load_state_dict(make_synthetic_state_dict(seed=1, max_cats=13))
    assert all(torch.equal(u, v) for u, v in zip(other, _chain(m2, diff, bt, 2, 3, 8)))


def test_two_schedules_of_equal_length_do_not_alias():
    """Two diffusion objects with the same number of steps and different betas on one model: each call uses its own tables
    (the upload is keyed on a digest of the tables, never on id())."""
    from lsdm_amd.diffusion import gaussian_diffusion as gd
    from lsdm_amd.diffusion.respace import SpacedDiffusion
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(2, seed=12).items()}
    mk = lambda name: SpacedDiffusion(use_timesteps=range(1000), betas=gd.get_named_beta_schedule(name, 1000),
                                      model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                                      loss_type=gd.LossType.MSE)
    cos, lin = mk("cosine"), mk("linear")
    assert cos._schedule_key() != lin._schedule_key() and cos._schedule_key() == mk("cosine")._schedule_key()
    m, _ = _fresh()
    a_cos = _chain(m, cos, bt, 2, 3, 8)
    a_lin = _chain(m, lin, bt, 2, 3, 8)
    b_cos = _chain(m, cos, bt, 2, 3, 8)
    assert all(torch.equal(u, v) for u, v in zip(a_cos, b_cos)) and not torch.equal(a_cos[0], a_lin[0])
    assert all(torch.equal(u, v) for u, v in zip(a_lin, _chain(_fresh()[0], lin, bt, 2, 3, 8)))


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16"])
def test_smaller_batch_after_larger_batch_on_one_model(precision):
    """A chain at B = 4 and then at B = 2 on the same model (the last partial batch of a loader): the multi-step entry points
    lay the per-step sets of a group out for the CURRENT batch, not for the reserved capacity -- equal to a fresh model,
    in every mode (strict, invariant, hoisted)."""
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(4, seed=21).items()}
    m, diff = _fresh(precision)
    _chain(m, diff, bt, 4, 5, 3)
    for kw in (dict(hoist_invariants=False), dict(hoist_invariants=True), dict(hoist_conditioning=True)):
        got = _chain(m, diff, bt, 2, 5, 4, **kw)
        want = _chain(_fresh(precision)[0], diff, bt, 2, 5, 4, **kw)
        assert all(torch.equal(u, v) for u, v in zip(got, want)), kw
    if precision == "f32":        # and against the oracle: the B = 2 chain is right, not merely reproducible
        smp = cpu_ref.Sampler(make_synthetic_state_dict(seed=0))
        c = {k: v[:2].cpu() for k, v in bt.items()}
        torch.manual_seed(4)
        ref = smp.sample_loop([2, 1024, 3], c["mask"], c["given_objs"], c["given_cats"], c["text_features"], skip_timesteps=995)
        assert maxabs(_chain(m, diff, bt, 2, 5, 4)[0], ref) <= 1e-3


def test_plain_bf16_multi_step_group_equals_single_steps():
    """bf16 packs the up-sampled conditioning of a chain in half the bytes of the other modes; steps 2.. of a group must find
    their rows (one call of K steps == K single-step calls, bit for bit)."""
    B, K = 2, 3
    m, diff = _fresh("bf16")
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=12).items()}
    torch.manual_seed(8)
    loop = [o["sample"] for o in diff.p_sample_loop_progressive(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"],
                                                                bt["text_features"], clip_denoised=False, skip_timesteps=1000 - K)][-1].clone()
    for inv in (False, True):
        assert torch.equal(_chain(m, diff, bt, B, K, 8, hoist_invariants=inv)[0], loop)


def test_chain_on_a_caller_stream_with_work_queued_behind_it():
    """The loop-invariant schedule forks a side stream off the caller's stream and joins it back (api.hip::stream_fork / stream_join): a
    chain launched on a non-default torch stream, with a consumer of its result enqueued right behind it on the same stream and no
    host synchronisation in between, gives the default-stream result bit for bit -- twice in a row (the events are re-used)."""
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(4, seed=41).items()}
    m, diff = _fresh("bf16x3")
    want = _chain(m, diff, bt, 4, 9, 23)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            got = _chain(m, diff, bt, 4, 9, 23)
            checksum = got[0].double().sum() + got[1].double().sum()      # queued behind the chain on the same stream
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
            assert float(checksum) == float(want[0].double().sum() + want[1].double().sum())
    torch.cuda.current_stream().wait_stream(side)


# ======================================================================================= retained A/B switches of the library
_SWITCHES = [{"LSDM_SA_SPLIT": "0"}, {"LSDM_FP_PRE": "0"}, {"LSDM_FP1_REG": "0"}, {"LSDM_DEN_REG": "0"}, {"LSDM_GROUP": "3"}, {"LSDM_BATCH_ENC": "0"},
             {"LSDM_BATCH_DEN": "0"}, {"LSDM_STREAMS": "1"}, {"LSDM_SA2_MEMO": "0"}, {"LSDM_FP2_WS": "0"}, {"LSDM_FP1_ORDER": "0"}, {"LSDM_SA_PAIR": "0"},
             {"LSDM_SA_PAIR": "2"}, {"LSDM_DEG_FAST": "0"}]
_SAME_ARITHMETIC = ("LSDM_GROUP", "LSDM_BATCH_ENC", "LSDM_BATCH_DEN", "LSDM_STREAMS", "LSDM_SA2_MEMO", "LSDM_FP2_WS", "LSDM_FP1_ORDER",
                    "LSDM_SA_PAIR", "LSDM_DEG_FAST")      # launch shapes / kernel forms with the same arithmetic in the same order: bit-identical results


@pytest.mark.parametrize("env", _SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_every_retained_switch_setting_holds_the_parity_bar(golden, env):
    """The LSDM_* switches the shipped library still reads select between IMPLEMENTATIONS of a stage (the fall-back kernels an
    optimisation replaced); each setting must hold the same bar as the default: one forward against the reference's golden G2 in
    bf16x3, and a 5-step chain within 1e-3 of the default build's (the switches are read when a context is created)."""
    g = golden("g2_forward_c13")
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(3, seed=5).items()}
    starts = [s.long() for s in T(g["starts"])]
    bt4 = {k: v.to(DEV) for k, v in make_synthetic_batch(4, seed=33).items()}

    def run():
        m, diff = _fresh("bf16x3")
        x = T(g["x_in"]).clone().to(DEV)
        out_cat, out = m(x, bt["mask"], T(g["t"]).long().to(DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=starts)
        fwd = (out_cat.clone(), out.clone(), x, m.saved_guiding_points.clone())
        return fwd + (_chain(m, diff, bt4, 4, 5, 17),)
    base = run()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        got = run()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    assert maxabs(got[0], T(g["out_cat"])) <= 1e-5 and maxabs(got[1], T(g["out"])) <= 1e-3
    assert maxabs(got[2], T(g["x_mut"])) <= 1e-3 and maxabs(got[3], T(g["guiding"])) <= 1e-3
    for a, b in zip(got[4], base[4]):
        assert maxabs(a, b) <= 1e-3
        if all(k in _SAME_ARITHMETIC for k in env):
            assert torch.equal(a, b), env


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_degenerate_clouds_fast_path_is_bit_identical(precision):
    """A cloud whose 1024 points are bit-identical (the all-zero padding slots of a scene with fewer than 8 objects, posa/dataset.py:456) has an
    encoder output that does not depend on the FPS draws (pointnet2_utils.py:60-104): lsdm_sample_steps_invariant evaluates one representative
    per distinct such cloud once per call and runs the per-step encoder on the live clouds only (api.hip: deg_fast).  Bit-equality of a 6-step
    chain's sample, guiding points and saved_cat against every cloud through the encoder every step (LSDM_DEG_FAST=0) and against the strict
    entry point, on: G2's ragged batch (4 / 2 / 7 objects), a batch with a non-zero constant cloud in two chains, a +0 / -0 mix (NOT degenerate
    by the bit rule), a cloud with one point moved (live), and a chain with NO objects; and the classification the call made."""
    def scenes():
        bt = make_synthetic_batch(3, seed=5)                          # G2's batch: 4 / 2 / 7 objects, the rest zero slots
        yield "g2 ragged", bt, 3, int((bt["given_objs"].abs().amax(dim=(2, 3)) == 0).sum()), 1
        bt = make_synthetic_batch(5, seed=21)
        objs = bt["given_objs"].clone()
        objs[0, 2] = torch.tensor([0.3, -1.2, 0.7])                   # a constant cloud ...
        objs[3, 5] = torch.tensor([0.3, -1.2, 0.7])                   # ... twice: one representative
        objs[1, 3] = 0.0
        objs[1, 3, 0::2, 1] = -0.0                                    # +0 / -0 mix: live
        objs[2, 4] = 0.0
        objs[2, 4, 777, 2] = 1e-3                                     # one point moved: live
        objs[4, 1:] = 0.0                                             # a chain without objects
        bt["given_objs"] = objs
        zero = objs.abs().amax(dim=(2, 3)) == 0
        bitzero = zero & ~torch.signbit(objs).any(dim=3).any(dim=2)
        yield "constructed", bt, 5, int(bitzero.sum()) + 2, 2
    for name, bt, B, n_deg, n_rep in scenes():
        d = {k: v.to(DEV) for k, v in bt.items()}

        def run(**kw):
            m, diff = _fresh(precision)
            out = _chain(m, diff, d, B, 6, 31, **kw)
            exp = ws(m, "exp_src", (B * 9,), torch.int32) if not kw and os.environ.get("LSDM_DEG_FAST") != "0" else None
            return out, exp
        fast, exp = run()
        slow, _ = _with_env({"LSDM_DEG_FAST": "0"}, run)
        strict, _ = run(hoist_invariants=False)
        assert int((exp < 0).sum()) == n_deg and int((-exp[exp < 0]).max()) == n_rep, (name, exp)
        live = exp[exp >= 0]
        assert torch.equal(live, torch.arange(len(live), dtype=torch.int32)), (name, exp)
        for a, b, c_ in zip(fast, slow, strict):
            assert torch.equal(a, b) and torch.equal(a, c_), name


def test_degenerate_clouds_fast_path_at_the_benched_batch():
    """B = 64 with k ~ U{1..8} objects per scene (SURVEY.md 8d's ragged variant: the bench's `ragged_k` leg): three steps, fast path on / off."""
    bt = make_synthetic_batch(64, seed=1, num_objects=None)
    d = {k: v.to(DEV) for k, v in bt.items()}
    n_zero = int((bt["given_objs"].abs().amax(dim=(2, 3)) == 0).sum())
    assert 150 < n_zero < 330

    def run():
        m, diff = _fresh("bf16x3")
        out = _chain(m, diff, d, 64, 3, 4)
        return out, (ws(m, "exp_src", (576,), torch.int32) if os.environ.get("LSDM_DEG_FAST") != "0" else None)
    fast, exp = run()
    slow, _ = _with_env({"LSDM_DEG_FAST": "0"}, run)
    assert int((exp < 0).sum()) == n_zero and int(exp.min()) == -1
    assert all(torch.equal(a, b) for a, b in zip(fast, slow))


def test_sa3_sa4_item_lists_agree_with_one_block_per_group():
    """sa3 / sa4 over item lists (two ball-query groups whose distinct neighbours fit into 32 rows together in one block, geometry.hip: sa_pair_kernel)
    against every group in a block of its own (LSDM_SA_PAIR=0), bit for bit over a 4-step chain, on scenes whose objects cover the
    extremes: a cloud scattered over 100 m (every group is its centroid alone: all of them pair up), a single point and an all-zero slot
    (every group has 32 distinct entries: nothing pairs), a line, a 2 cm blob, two points, ordinary objects; and the item counts the
    planner reports: blocks = groups - pairs, within what the groups' sizes allow."""
    B = 3
    bt = make_synthetic_batch(B, seed=41)
    g = torch.Generator().manual_seed(78)
    objs = bt["given_objs"].clone()
    objs[0, 1] = (torch.rand(1024, 3, generator=g) - 0.5) * 100.0
    objs[0, 2] = torch.tensor([0.3, -1.2, 0.7]).expand(1024, 3)
    objs[0, 3] = 0.0
    objs[1, 0, :, 0] = torch.linspace(-1.0, 1.0, 1024); objs[1, 0, :, 1:] = 0.0
    objs[1, 4] = torch.randn(1024, 3, generator=g) * 0.02
    objs[2, 5, ::2] = torch.tensor([0.0, 0.0, 0.0]); objs[2, 5, 1::2] = torch.tensor([0.05, 0.0, 0.1])
    objs[2, 6] = (torch.rand(1024, 3, generator=g) - 0.5) * 6.0
    bt["given_objs"] = objs
    bt = {k: v.to(DEV) for k, v in bt.items()}
    outs, items = {}, {}
    keep = os.environ.get("LSDM_SA_PAIR")
    try:
        for mode in ("1", "0"):
            os.environ["LSDM_SA_PAIR"] = mode
            m, diff = _fresh("bf16x3")
            outs[mode] = _chain(m, diff, bt, B, 4, 23)
            if mode == "1":
                items = {lvl: ws(m, name, (4,), torch.int32).clone() for lvl, name in ((3, "sa3_items"), (4, "sa4_items"))}
                live = int((ws(m, "exp_src", (B * 9,), torch.int32) >= 0).sum())      # (the constant clouds and zero slots run once per call)
            else:
                with pytest.raises(Exception):
                    m._ctx.workspace_tensor("sa3_items")
            del m, diff
    finally:
        if keep is None:
            del os.environ["LSDM_SA_PAIR"]
        else:
            os.environ["LSDM_SA_PAIR"] = keep
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)
    assert live < B * 9
    for lvl, S in ((3, 64), (4, 16)):
        groups = live * S
        n = items[lvl]
        # every step: at least half the groups (everything paired), fewer than all of them (the scattered cloud alone pairs S / 2 blocks away)
        assert bool((n >= groups // 2).all()) and bool((n <= groups - S // 2).all()), (lvl, n, groups)


def _most_pairs(classes):
    """Two-pointer matching on size classes (1..8 quartets of rows; a block holds 8): the largest class that still has a partner takes the smallest
    class left -- the number of pairs it forms is the maximum."""
    s = sorted(int(c) for c in classes)
    i, j, pairs = 0, len(s) - 1, 0
    while i < j:
        if s[i] + s[j] <= 8:
            pairs += 1; i += 1; j -= 1
        else:
            j -= 1
    return pairs


def test_sa_item_lists_are_an_exact_cover_with_the_most_pairs():
    """The planner's output itself (geometry.hip: sa_pair_kernel), read back after one scene-encoder call over 18 clouds (degenerate ones among
    them; 18432 / 4608 / 1152 / 288 centroids at the four SA levels: whole chunks of 256 and, at the last two, a partial one): every centroid is in exactly one block; a block's rows are
    its groups' first rows (A: 4k, B: 32 - 4k) shifted to the cloud's U rows and hold ALL their distinct entries; V rows and output rows are the
    groups' own; and the number of blocks is what the pair-maximising matching gives chunk by chunk."""
    nc = 18
    g = torch.Generator().manual_seed(79)
    objs = make_synthetic_batch(2, seed=63)["given_objs"].reshape(nc, 1024, 3).clone()
    objs[1] = 0.0
    objs[2] = torch.tensor([0.3, -1.2, 0.7]).expand(1024, 3)
    objs[4] = (torch.rand(1024, 3, generator=g) - 0.5) * 100.0
    objs[5, :, 0] = torch.linspace(-1.0, 1.0, 1024); objs[5, :, 1:] = 0.0
    objs[7] = torch.randn(1024, 3, generator=g) * 0.02
    objs[11] = (torch.rand(1024, 3, generator=g) - 0.5) * 6.0
    clouds = objs.to(DEV).contiguous()
    torch.manual_seed(6)
    starts = torch.stack([torch.randint(0, n, (nc,)) for n in (1024, 1024, 256, 64)]).to(DEV)
    keep = os.environ.get("LSDM_SA_PAIR")
    os.environ["LSDM_SA_PAIR"] = "2"        # (single steps take the lists from 6 chains on by default; 2 = at any size)
    try:
        m, _ = create_model_and_diffusion("proxd", precision="bf16x3")
        m.load_state_dict(make_synthetic_state_dict(seed=0))
        ctx = m._context()
    finally:
        if keep is None:
            del os.environ["LSDM_SA_PAIR"]
        else:
            os.environ["LSDM_SA_PAIR"] = keep
    m._reserve(2)
    out = torch.empty(nc, 1024, 3, device=DEV)
    ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), nc, out.data_ptr(), None)
    torch.cuda.synchronize()
    items = ws(m, "sa_step_items", (4,), torch.int32)
    for lvl, (S, NSRC, name) in enumerate(((1024, 1024, "sa1"), (256, 1024, "sa2"), (64, 256, "sa3"), (16, 64, "sa4"))):
        T1 = nc * S
        ball = ws(m, "ball%d" % (lvl + 1), (T1, 32), torch.int32).long()
        cidx = ws(m, "fps_idx%d" % (lvl + 1), (T1,), torch.int32).long()
        rows = ws(m, name + "_rows", (T1, 32), torch.int32).long()
        meta = ws(m, name + "_meta", (T1, 4), torch.int32).long()
        n = int(items[lvl])
        d = 1 + (ball[:, 1:] != ball[:, :1]).sum(1)                       # distinct entries of every group (the padding repeats the first)
        assert bool((ball[:, 1:] != ball[:, :1]).long().diff(dim=1).le(0).all())      # ... which come first
        cls = (d + 3) // 4
        want = sum(len(c) - _most_pairs(c) for c in cls.split(256))
        assert n == want, (name, n, want)
        r0 = (torch.arange(T1) // S) * NSRC
        k = meta[:n, 2] >> 24
        oa, ob, va, vb = meta[:n, 2] & 0xFFFFFF, meta[:n, 3], meta[:n, 0], meta[:n, 1]
        single = k == 8
        assert bool((k >= 1).all()) and bool((k <= 8).all()) and bool((oa[single] == ob[single]).all())
        seen = torch.cat([oa, ob[~single]])
        assert seen.numel() == T1 and bool((seen.sort().values == torch.arange(T1)).all()), name
        assert bool((va == r0[oa] + cidx[oa]).all()) and bool((vb == r0[ob] + cidx[ob]).all())
        assert bool((d[oa] <= 4 * k).all()) and bool((d[ob[~single]] <= 32 - 4 * k[~single]).all())
        col = torch.arange(32).expand(n, 32)
        from_a = col < (4 * k)[:, None]
        src = torch.where(from_a, ball[oa].gather(1, col), ball[ob].gather(1, (col - 4 * k[:, None]).clamp(min=0)))
        base = torch.where(from_a, r0[oa][:, None], r0[ob][:, None])
        assert torch.equal(rows[:n], src + base), name


def test_fp2_fp1_kernel_forms_agree_on_degenerate_and_scattered_clouds():
    """(Also sa3 / sa4 over item lists against one block per group, on the same clouds.)  fp2's planned kernel (fp_d_kernel: Morton-ordered tiles, rows deduplicated through LDS) and fp1's stage pipeline (fp_c_kernel) against the
    tile-per-work-group / register-chained forms they replaced, bit for bit, on the clouds that leave their fast paths: all-zero and single-point
    clouds (the level-1 order is not a permutation: the plan walks them in index order), a two-point cloud, a cloud scattered over 100 m (no two
    points of a tile share a coarse neighbour: up to 96 distinct rows per tile, beyond the 36 LDS slots -- the rows without a slot are read from
    global memory), a line, and ordinary clouds beside them."""
    nc = 9
    g = torch.Generator().manual_seed(77)
    clouds = torch.zeros(nc, 1024, 3)
    clouds[0] = make_synthetic_batch(1, seed=61)["given_objs"][0, 1]
    # clouds[1]: zeros (a padded slot)
    clouds[2] = torch.tensor([0.3, -1.2, 0.7]).expand(1024, 3)
    clouds[3, ::2] = torch.tensor([0.0, 0.0, 0.0]); clouds[3, 1::2] = torch.tensor([0.05, 0.0, 0.1])
    clouds[4] = (torch.rand(1024, 3, generator=g) - 0.5) * 100.0
    clouds[5, :, 0] = torch.linspace(-1.0, 1.0, 1024)
    clouds[6] = make_synthetic_batch(1, seed=62)["given_objs"][0, 0]
    clouds[7] = torch.randn(1024, 3, generator=g) * 0.02
    clouds[8] = (torch.rand(1024, 3, generator=g) - 0.5) * 6.0
    clouds = clouds.to(DEV).contiguous()
    torch.manual_seed(5)
    starts = torch.stack([torch.randint(0, n, (nc,)) for n in (1024, 1024, 256, 64)]).to(DEV)
    sd = make_synthetic_state_dict(seed=0)
    outs = {}
    keep = {k: os.environ.get(k) for k in ("LSDM_FP2_WS", "LSDM_SA_PAIR")}
    try:
        for name, env in (("new", {"LSDM_FP2_WS": "4", "LSDM_SA_PAIR": "2"}), ("old", {"LSDM_FP2_WS": "0", "LSDM_SA_PAIR": "0"})):
            os.environ.update(env)
            m, _ = create_model_and_diffusion("proxd", precision="bf16x3")
            m.load_state_dict(sd)
            ctx = m._context()
            m._reserve(1)
            out = torch.empty(nc, 1024, 3, device=DEV)
            ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), nc, out.data_ptr(), None)
            torch.cuda.synchronize()
            outs[name] = (out.clone(), ws(m, "fp2", (nc, 1024, 128)).clone(), ws(m, "feat3", (nc, 64, 256)).clone(), ws(m, "feat4", (nc, 16, 512)).clone(),
                          ws(m, "feat1", (nc, 1024, 64)).clone(), ws(m, "feat2", (nc, 256, 128)).clone())
            if name == "new":
                items = ws(m, "sa_step_items", (4,), torch.int32).clone()[2:]       # (by SA level: sa3, sa4)
            del m, ctx
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert bool(torch.isfinite(outs["new"][0]).all())
    # sa3 / sa4 over the step's item lists (sa_pair_kernel) against one block per ball-query group; the scattered cloud and the line pair up
    # completely, the single-point, zero and two-point clouds (every group = the first 32 points) not at all
    assert torch.equal(outs["new"][2], outs["old"][2]) and torch.equal(outs["new"][3], outs["old"][3]), "sa3 / sa4 differ between item lists and groups"
    assert torch.equal(outs["new"][4], outs["old"][4]) and torch.equal(outs["new"][5], outs["old"][5]), "sa1 / sa2 differ between item lists and groups"
    assert nc * 64 // 2 <= int(items[0]) <= nc * 64 - 32 and nc * 16 // 2 <= int(items[1]) <= nc * 16 - 8, items
    assert torch.equal(outs["new"][1], outs["old"][1]), "fp2 output differs between fp_d_kernel and fp_pre_kernel"
    assert torch.equal(outs["new"][0], outs["old"][0]), "encoder output differs between the new and the old kernel forms"


# ======================================================================================= clip_dim = 768 (BERT pooler width)
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_forward_with_768_wide_text_features_vs_reference_golden(golden, precision):
    """text_encoder_type="BERT" feeds embed_text with pooler_output (B,768) (model/sdm.py:146-149, 261-264): clip_dim = 768, embed_text
    768 -> 384 -> 256 -> 128.  The library reads the width off embed_text.0.weight; golden G13 is the reference built that way."""
    g = golden("g13_forward_clipdim768")
    B, C, D, bt_seed, _ = [int(v) for v in g["meta"]]
    m, _ = create_model_and_diffusion("proxd", precision=precision, clip_dim=D, text_encoder_type="BERT")
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C, clip_dim=D))
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=bt_seed, max_cats=C, text_dim=D).items()}
    x = T(g["x_in"]).to(DEV).clone()
    out_cat, out = m(x, bt["mask"], T(g["t"]).long().to(DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"],
                     fps_starts=[s.long() for s in T(g["starts"])])
    tol = 1e-4 if precision == "f32" else 1e-3
    assert maxabs(out_cat, T(g["out_cat"])) <= 1e-5 and maxabs(out, T(g["out"])) <= tol
    assert maxabs(m.saved_guiding_points, T(g["guiding"])) <= tol and maxabs(x, T(g["x_mutated"])) <= tol
    with pytest.raises(ValueError, match="clip_dim"):
        m(x, bt["mask"], T(g["t"]).long().to(DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"][:, :512].contiguous())


# ======================================================================================= device-RNG throughput mode
def test_device_rng_mode_is_deterministic_in_range_and_distribution():
    """host_rng=False ("device Philox for throughput mode"): x_T, the per-step FPS starts and the per-step noise
    all come from the device generator -- nothing is drawn or uploaded by the host inside the loop.  Checked: same seed -> the same
    sample bit for bit (bulk loop, strict and hoisted schedules), another seed -> another sample; the draws the loop hands to the
    library are in range per level and have the moments of their distributions; the per-step progressive loop consumes the device
    generator too (no host generator state is touched)."""
    B, K = 4, 6
    m, _ = _fresh("bf16x3")
    diff = create_gaussian_diffusion(get_default_diffusion(), host_rng=False)
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=21).items()}
    args = (m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"])
    seen = []
    orig = m._run_steps

    def spy(mode, eta, x, mask, t_first, k, given_objs, given_cats, text, starts, noise, **kw):
        seen.append((starts.clone(), noise.clone()))
        return orig(mode, eta, x, mask, t_first, k, given_objs, given_cats, text, starts, noise, **kw)
    m._run_steps = spy
    host_state = torch.get_rng_state()
    outs = {}
    for kw_name, kw in (("invariant", {}), ("strict", {"hoist_invariants": False}), ("hoisted", {"hoist_conditioning": True})):
        res = []
        for seed in (7, 7, 8):
            torch.manual_seed(seed)
            host_after_seed = torch.get_rng_state()
            res.append(diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=1000 - K, chunk_steps=4, **kw).clone())
            assert torch.equal(torch.get_rng_state(), host_after_seed)          # the host generator was not consumed
        assert torch.equal(res[0], res[1]) and not torch.equal(res[0], res[2]) and bool(torch.isfinite(res[0]).all()), kw_name
        outs[kw_name] = res[0]
    assert torch.equal(outs["invariant"], outs["strict"])                       # same draws, bit-identical schedules
    m._run_steps = orig
    torch.set_rng_state(host_state)
    starts = torch.cat([s for s, _ in seen if s.dim() == 3]).cpu()               # (steps, 4, B*9)
    noise = torch.cat([n for _, n in seen]).cpu()
    assert starts.is_cuda is False and all(s.is_cuda and n.is_cuda for s, n in seen)
    for l, n in enumerate((1024, 1024, 256, 64)):
        v = starts[:, l].reshape(-1).float()
        assert int(v.min()) >= 0 and int(v.max()) < n
        assert abs(float(v.mean()) - (n - 1) / 2) <= 4 * n / (12 * v.numel()) ** 0.5 + 1       # uniform: mean within 4 sigma
    assert abs(float(noise.mean())) <= 5 / noise.numel() ** 0.5 and abs(float(noise.var()) - 1) <= 0.01
    # per-step path (progressive loop): device draws as well
    torch.manual_seed(7)
    host_after_seed = torch.get_rng_state()
    last = [o["sample"] for o in diff.p_sample_loop_progressive(*args, clip_denoised=False, skip_timesteps=1000 - 2)][-1]
    assert torch.equal(torch.get_rng_state(), host_after_seed) and bool(torch.isfinite(last).all())


def test_input_validation_is_loud():
    m, diff = _fresh()
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(2, seed=3).items()}
    x = torch.zeros(2, 1024, 3, device=DEV)
    t = torch.tensor([5, 5], device=DEV)
    good = cpu_ref.draw_fps_starts(18)
    bad = [s.clone() for s in good]
    bad[2][0] = 256                                     # level 3 draws from [0, 256)
    with pytest.raises(ValueError, match="level 3"):
        m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=bad)
    # timesteps: host tensors are range-checked; device tensors are not read back (no per-step stall) unless opted in, and an
    # out-of-range device index is clamped into the table instead of reading past it
    with pytest.raises(ValueError, match="timesteps"):
        m(x, bt["mask"], torch.tensor([5, 5000]), bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=good)
    with pytest.raises(ValueError, match="timesteps"):   # sampler modes index the schedule tables (1000 rows)
        diff.p_sample(m, x, bt["mask"], torch.tensor([5, 1000]), bt["given_objs"], bt["given_cats"], bt["text_features"],
                      clip_denoised=False)
    os.environ["LSDM_CHECK_TIMESTEPS"] = "1"
    try:
        with pytest.raises(ValueError, match="timesteps"):
            m(x, bt["mask"], torch.tensor([5, 5000], device=DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=good)
    finally:
        del os.environ["LSDM_CHECK_TIMESTEPS"]
    # (not opted in: the call itself does not stall on a read-back -- its kernels see the index clamped -- and the error is raised
    # one call late, by the next call or by check_timesteps(), which the sampling loops call when they finish)
    xa, xb = x.clone(), x.clone()
    oa = m(xa, bt["mask"], torch.tensor([5, 5000], device=DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=good)
    with pytest.raises(ValueError, match="earlier call"):
        m.check_timesteps()
    ob = m(xb, bt["mask"], torch.tensor([5, 4999], device=DEV), bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=good)
    m.check_timesteps()
    assert torch.equal(oa[1], ob[1])
    with pytest.raises(ValueError):
        m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"][:, :, :5], y=bt["text_features"], fps_starts=good)
    m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=good)     # still usable
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == 0


# ======================================================================================= metric kernels
def test_nn_distance_kernel_vs_definition():
    from lsdm_amd.metrics import _nn_sq_dist_host, calculate_fscore, chamfer_distance, nn_sq_dist
    g = torch.Generator().manual_seed(3)
    a = torch.randn(3, 777, 3, generator=g)
    b = torch.randn(3, 2500, 3, generator=g) * 1.2          # three LDS tiles, ragged
    d2, idx = nn_sq_dist(a.to(DEV), b.to(DEV), return_index=True)
    want = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
    wd, wi = want.min(dim=2)
    assert maxabs(d2, wd) <= 1e-6
    sel = torch.gather(want, 2, idx.cpu().long()[..., None])[..., 0]
    assert maxabs(sel, wd) <= 1e-6                            # the reported index attains the minimum
    # the CLI's metrics on 1024-point clouds: device (HIP) vs the definitions evaluated on the host
    x = torch.randn(1, 1024, 3, generator=g)
    y = x + 0.08 * torch.randn(1, 1024, 3, generator=g)
    assert abs(float(chamfer_distance(x.to(DEV), y.to(DEV))) - float(chamfer_distance(x, y))) <= 1e-6
    fh, fd = calculate_fscore(x[0], y[0]), calculate_fscore(x[0].to(DEV), y[0].to(DEV))
    assert fh == fd and 0.0 < fd[0] < 1.0
    with pytest.raises(_lib.LsdmError):
        nn_sq_dist(x, y)
