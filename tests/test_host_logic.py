"""CPU tests of the product's host side: C-ABI surface, schedule tables, respacing, state_dict contract, the
fail-loud rule, data format + CLI plumbing (BASELINE configs[0]) with the oracle injected as the sampler."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from lsdm_amd import _lib
from lsdm_amd.diffusion.respace import space_timesteps
from lsdm_amd.synthetic import make_synthetic_state_dict
from lsdm_amd.util.model_util import create_gaussian_diffusion, create_model_and_diffusion, get_default_diffusion

torch.set_grad_enabled(False)


# ------------------------------------------------------------------------------------------------ C ABI
def test_c_abi_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lsdm_hip.h")).read()
    declared = set(re.findall(r"\b(lsdm_[a-z0-9_]+)\s*\(", header))
    declared -= {"lsdm_ctx", "lsdm_status", "lsdm_mode"}
    assert len(declared) >= 18
    lib = ctypes.CDLL(_lib.LIB_PATH)       # no compute calls: there is no GPU here
    for name in sorted(declared):
        assert hasattr(lib, name), "liblsdm_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    # ... and nothing else: the library is built with -fvisibility=hidden and linked through csrc/exports.map, so its dynamic symbol
    # table is exactly the C ABI (no mangled launch helpers, kernel handles or libstdc++ instantiations)
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    dynamic = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert dynamic == declared, sorted(dynamic ^ declared)[:10]
    for name in sorted(declared):
        assert re.search(r"^LSDM_API [^;]*\b%s\(" % name, header, flags=re.M), name


def test_no_cpu_fallback():
    """The product fails loudly without a HIP device instead of computing anything on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m, d = create_model_and_diffusion("proxd", use_cuda=False)
    args = (torch.zeros(1, 1024, 3), torch.zeros(1, 9), torch.zeros(1).long(), torch.zeros(1, 9, 1024, 3), torch.zeros(1, 9, 13))
    with pytest.raises(_lib.LsdmError):
        m(*args, y=torch.zeros(1, 512))
    with pytest.raises(_lib.LsdmError):
        d.p_sample_loop(m, [1, 1024, 3], args[1], args[3], args[4], torch.zeros(1, 512), clip_denoised=False, device="cpu")
    with pytest.raises(_lib.LsdmError, match="no HIP device"):
        _lib.Context(0)


def test_product_never_imports_the_oracle():
    import ast
    roots = [os.path.join(ROOT, d) for d in ("lsdm_amd", "model", "diffusion", "util", "run")]
    for root in roots:
        for base, _, files in os.walk(root):
            for f in files:
                if not f.endswith(".py"):
                    continue
                tree = ast.parse(open(os.path.join(base, f)).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), os.path.join(base, f)
                    if isinstance(node, ast.Call) and getattr(node.func, "id", "") in ("__import__", "exec", "eval"):
                        raise AssertionError("dynamic import/exec in product file %s" % os.path.join(base, f))


# ------------------------------------------------------------------------------------------------ schedules
@pytest.mark.parametrize("tag,resp", [("t1000", ""), ("ddim50", "ddim50"), ("ddim250", "ddim250")])
def test_schedule_tables_bit_exact_vs_reference(golden, tag, resp):
    g = golden("g1_schedule")
    d = create_gaussian_diffusion(get_default_diffusion(), timestep_respacing=resp)
    assert d.timestep_map == list(g["%s.timestep_map" % tag])
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_variance", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        np.testing.assert_array_equal(getattr(d, k), g["%s.%s" % (tag, k)], err_msg=k)
    assert d.num_timesteps == len(g["%s.betas" % tag])


def test_space_timesteps_sections():
    assert space_timesteps(1000, [1000]) == set(range(1000))
    assert space_timesteps(1000, "ddim50") == set(range(0, 1000, 20))
    assert space_timesteps(1000, "ddim250") == set(range(0, 1000, 4))
    assert space_timesteps(300, "10,15,20") == space_timesteps(300, [10, 15, 20])
    assert len(space_timesteps(300, [10, 15, 20])) == 45 and max(space_timesteps(300, [10, 15, 20])) == 299
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        space_timesteps(10, [20])


# ------------------------------------------------------------------------------------------------ state_dict
@pytest.mark.parametrize("datatype,C", [("proxd", 13), ("humanise", 11)])
def test_state_dict_contract(datatype, C):
    want = json.load(open(os.path.join(GOLDEN, "g0_state_dict_c%d.json" % C)))
    m, _ = create_model_and_diffusion(datatype, use_cuda=False)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == want                                  # names, shapes, dtypes and order of the reference
    res = m.load_state_dict(make_synthetic_state_dict(max_cats=C), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    sd = make_synthetic_state_dict(max_cats=C)
    sd["clip_model.positional_embedding"] = torch.zeros(77, 512)   # real checkpoints also carry the CLIP tower
    m.load_state_dict(sd, strict=True)


def test_constructor_signature_matches_reference():
    import inspect
    from lsdm_amd.model.sdm import SceneDiffusionModel
    sig = inspect.signature(SceneDiffusionModel.__init__)
    names = list(sig.parameters)[1:]
    assert names == ["seg_len", "modality", "clip_version", "clip_dim", "dropout", "n_layer", "n_head", "f_vert", "dim_ff",
                     "cat_emb", "mesh_ds_dir", "posa_path", "latent_dim", "cond_mask_prob", "device", "vert_dims", "obj_cat",
                     "data_rep", "njoints", "use_cuda", "pcd_points", "pcd_dim", "xyz_dim", "max_cats", "translation_params",
                     "pcd_backbone_type", "human_backbone_type", "text_encoder_type", "kwargs"]
    assert sig.parameters["max_cats"].default == 13 and sig.parameters["clip_dim"].default == 768
    fwd = list(inspect.signature(SceneDiffusionModel.forward).parameters)
    assert fwd[:8] == ["self", "x", "mask", "timesteps", "given_objs", "given_cats", "y", "force_mask"]


# ------------------------------------------------------------------------------------------------ CLI / config[0]
class OracleBackend:
    """Test-only sampler: the CPU oracle behind the CLI's backend protocol (the product default is HipBackend)."""

    def __init__(self, args, text_encoder):
        from oracle import cpu_ref
        C = 13 if args.datatype == "proxd" else 11
        self.smp = cpu_ref.Sampler(make_synthetic_state_dict(seed=0, max_cats=C), timestep_respacing=args.timestep_respacing or None)
        self.text_encoder, self.use_ddim = text_encoder, args.use_ddim

    def sample(self, shape, mask, given_objs, given_cats, y):
        out = self.smp.sample_loop(shape, mask, given_objs, given_cats, self.text_encoder(list(y)), use_ddim=self.use_ddim)
        return out, self.smp.saved_cat, self.smp.saved_guiding_points


def test_cli_plumbing_synthetic_proxd_test(tmp_path):
    """BASELINE configs[0]: run.test_sdm on 4 synthetic proxd_test samples, DDIM, CPU (plumbing + file outputs).
    ddim50 when LSDM_FULL_CONFIG0=1 (about 4 minutes of oracle time), ddim5 otherwise."""
    import run.test_sdm as cli
    resp = "ddim50" if os.environ.get("LSDM_FULL_CONFIG0") == "1" else "ddim5"
    out = tmp_path / "out"
    rc = cli.main([str(tmp_path / "data"), "--synthetic", "4", "--device", "cpu", "--use_ddim", "--timestep_respacing", resp,
                   "--output_dir", str(out), "--fix_ori", "--test_on_valid_set", "--model_name", "sdm"], backend_factory=OracleBackend)
    assert rc == 0
    lines = open(out / "results.txt").read().splitlines()
    assert len(lines) == 4 + 5
    assert all(re.fullmatch(r"Chamfer distance for seq Scene\d\d_\d{5}_01: \d+\.\d{4}", l) for l in lines[:4])
    for l, prefix in zip(lines[4:], ("Final Chamfer distance: ", "Final EMD: ", "Final F1 score: ", "Category accuracy: ", "Top 3 accuracy: ")):
        assert l.startswith(prefix) and re.fullmatch(r"-?\d+\.\d{4}", l[len(prefix):])
    for sub in ("predictions", "guiding_points"):
        files = sorted(os.listdir(out / sub))
        assert len(files) == 4
        a = np.load(out / sub / files[0])
        assert a.shape == (1024, 3) and a.dtype == np.float32 and np.isfinite(a).all()
    # dataset format: third context line unstripped / no trailing newline (Q14), human slot mask 0
    from lsdm_amd.data import ProxTextDataset
    ds = ProxTextDataset(str(tmp_path / "data" / "proxd_test"), objs_data_dir=str(tmp_path / "data" / "objs"))
    mask, verts, cats, tgt, tcat, prompt = ds[0]
    assert mask[0] == 0 and mask.sum() >= 1 and verts.shape == (9, 1024, 3) and cats.shape == (9, 13) and cats[0, 0] == 1
    assert tgt.shape == (1024, 3) and tcat.sum() == 1 and prompt.startswith("synthetic prompt")


def test_integration_md_shows_the_executed_binding_verbatim():
    """INTEGRATION.md path B is integration/sdm_hip.py, the file tests/test_hip_configs.py executes on the GPU."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = open(os.path.join(ROOT, "integration", "sdm_hip.py")).read()
    assert "```python\n" + code + "```" in doc


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`--gpus N` is honoured or refused, never ignored: with a launcher's WORLD_SIZE that disagrees the run stops before
    touching a device (without one, bench.py re-executes itself under torch.distributed.run with N ranks)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    import bench
    src = open(bench.__file__).read()
    assert "respawn_ranks(a.gpus)" in src and "torch.distributed.run" in src


def test_traffic_json_matches_the_latest_kernel_record():
    """profiles/traffic.json (bench.py's roofline.traffic) must come from the newest round's PMC pass: its source tag is the newest
    `rNN_x_kernel_stats.md` under profiles/ that has a PMC file beside it, and every kernel it sums over is a kernel of that record."""
    import glob
    import json
    import re
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["bf16x3"]
    tag = re.match(r"pmc_(r\d\d_[a-z]+)\.json", tj["_source"])
    assert tag, tj["_source"]
    tag = tag.group(1)
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_kernel_stats.md")))
    with_pmc = [s for s in stats if os.path.exists(s.replace("_kernel_stats.md", "_pmc.json")) and "train" not in s]
    assert with_pmc and os.path.basename(with_pmc[-1]).startswith(tag), (tag, os.path.basename(with_pmc[-1]))
    record = open(with_pmc[-1].replace(".md", ".csv")).read()      # (the .md keeps the 40 largest kernels, the .csv all of them)
    for stage, e in tj.items():
        if isinstance(e, dict):
            for k in e["launches"]:
                assert k.split("<")[0].replace("lsdm::", "") in record, (stage, k)
    # ... and that record is of ONE workload (round 5's averaged B = 16 and B = 64 launches: sa_w min 70 / avg 252 / max 342 us): from round 6 on the
    # dominant dense kernel's fastest launch must lie within 20 % of its average
    if int(tag[1:3]) >= 6:
        import csv
        rows = list(csv.DictReader(open(with_pmc[-1].replace(".md", ".csv"))))
        dom = max((r for r in rows if "sa_w_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
        assert float(dom["MinNs"]) >= 0.8 * float(dom["AverageNs"]), (dom["Name"][:40], dom["MinNs"], dom["AverageNs"])


def test_design_md_keeps_its_own_size_rule():
    """DESIGN.md describes the code as it is and stays within 40 KiB; measurements and negative results go to docs/HISTORY.md."""
    n = os.path.getsize(os.path.join(ROOT, "DESIGN.md"))
    assert n <= 40 * 1024, "DESIGN.md is %d bytes (> 40 KiB): move history to docs/HISTORY.md" % n


def test_bench_rank_bookkeeping_world8_gloo_stub():
    """The driver's 8-GPU command line -- `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W` -- with the GPU
    work stubbed at the Runner boundary (--stub-runner: gloo, CPU, no library call): everything AROUND the kernels is bench.py's real code.  Eight
    ranks read RANK / LOCAL_RANK / WORLD_SIZE, seed = base + rank, pass the barriers, run the packed all-gather of samples + saved_cat + guiding
    points (rank order checked inside bench.py), take the max over the ranks' times, and rank 0 prints ONE line: value = 8 x steps / slowest rank."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    K, W = 4, 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(K), "--warmup", str(W), "--stub-runner"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["stub"] is True and r["data"].startswith("stub")
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["steps"] == K and r["warmup"] == W and r["scaling"] == "weak"
    assert r["per_rank_seeds"] == [1, 2, 3, 4, 5, 6, 7, 8]
    assert r["config"]["global_batch"] == 512 and r["config"]["chains_per_gpu"] == 64
    assert len(r["per_rank_ms_per_step"]) == 8 and r["allgather_us"] > 0 and r["allgather_bytes_per_rank"] == 64 * (2 * 3072 + 13) * 4
    slowest = max(r["per_rank_ms_per_step"])
    assert abs(r["ms_per_step"] - slowest) <= 1e-3 * slowest                       # the job's time is its slowest rank's
    assert abs(r["value"] - 8 * 1e3 / r["ms_per_step"]) <= 1e-6 * r["value"]      # whole-job steps/s: the sum over the GPUs
    # (each rank's time ends after the collective and the barrier, so the ranks' times agree to within the barrier's skew -- the stub's
    # slowest rank, 1.7x the fastest in run(), sets all of them)
    assert min(r["per_rank_ms_per_step"]) >= 0.002 * 1.7 * 1e3 * 0.9
    assert abs(r["sample_steps_per_s"] - 64 * r["value"]) <= 1e-6 * r["sample_steps_per_s"]


def test_metrics_known_answers():
    from lsdm_amd.metrics import accuracy, calculate_fscore, chamfer_distance, emd
    a = torch.tensor([[[0.0, 0, 0], [1, 0, 0]]])
    b = torch.tensor([[[0.0, 0, 0], [1, 0, 0.5]]])
    assert abs(float(chamfer_distance(a, b)) - 0.25) < 1e-6          # (0 + .25)/2 + (0 + .25)/2
    assert abs(emd(a, b) - 0.25) < 1e-6
    f, p, r = calculate_fscore(a[0], b[0], th=0.1)
    assert (f, p, r) == (0.5, 0.5, 0.5)
    out = torch.tensor([[0.1, 0.5, 0.2, 0.2], [0.7, 0.1, 0.1, 0.1]])
    assert [float(v) for v in accuracy(out, torch.tensor([2, 0]), topk=(1, 3))] == [50.0, 100.0]


def test_generated_code_has_no_barrier_without_the_lds_wait_and_no_packed_fp32_next_to_mfma(tmp_path):
    """hipcc (ROCm 7.2, gfx950) emitted a loop-header s_barrier WITHOUT `s_waitcnt lgkmcnt(0)` after the loop latch's LDS stores in round 4
    (fp_ws_kernel: another wave read the tile before the stores had landed; common.hpp::wg_sync now waits explicitly).
    tools/barrier_wait_check.py walks the control-flow graph of every compiled kernel backwards from each barrier; the negative control
    is that loop shape as a listing and must be flagged."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or shutil.which("make") is None:
        pytest.skip("no hipcc / make")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "lsdm_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j8", "check-barriers", "ASMDIR=%s" % (tmp_path / "asm")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "0 unguarded barrier(s)" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # negative control: the shape hipcc produced for round 4's fp_ws_kernel (removed in round 6) -- a loop-header barrier whose only predecessor
    # LDS store sits in the loop latch, no wait between them on the back edge -- as a hand-written listing; the guarded form beside it passes
    bare = tmp_path / "bare"
    bare.mkdir()
    body = ("\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n.LBB0_1:\n\ts_barrier\n\tds_read_b128 v[0:3], v4\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 v0, v0, v1\n"
            "\tds_write_b64 v5, v[0:1]\n%s\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    for name, guard in (("unguarded", ""), ("guarded", "\ts_waitcnt lgkmcnt(0)\n")):
        (bare / (name + "-gfx950.s")).write_text("_Z6kernelPf:\n" + body % guard + ".Lfunc_end0:\n")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "barrier_wait_check.py"), str(bare / "guarded-gfx950.s")], capture_output=True, text=True)
    assert r.returncode == 0 and "0 unguarded barrier(s)" in r.stdout, r.stdout[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "barrier_wait_check.py"), str(bare / "unguarded-gfx950.s")], capture_output=True, text=True)
    assert r.returncode == 1 and "_Z6kernelPf" in r.stdout and "1 unguarded barrier(s)" in r.stdout, r.stdout[-2000:]
    # The same generated code holds NO packed fp32 arithmetic in any translation unit that issues matrix instructions, and NO operand-selected
    # packed fp32 instruction in ANY unit (round 4's lost updates came from v_pk_fma_f32 with operand selection in a wave sharing its SIMD with
    # an MFMA wave; the Makefile removes the compiler's own packing, geometry.hip keeps explicit plain-pair forms).  Negative controls: a
    # unit compiled WITHOUT the flag mixes the two; geometry.hip compiled WITH the SLP vectoriser on gets operand-selected forms -- both flagged.
    r = subprocess.run(["make", "-C", csrc, "check-packed", "ASMDIR=%s" % (tmp_path / "asm")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "0 violation(s)" in r.stdout and "all on plain register pairs" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    pk = tmp_path / "pk"
    pk.mkdir()
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-inline-asm",
                    "--save-temps=obj", "-c", os.path.join(csrc, "denoise_reg.hip"), "-o", str(pk / "denoise_reg.o")], capture_output=True, timeout=900, check=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "packed_fp32_check.py")] + [str(p) for p in pk.glob("*-gfx950.s")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "denoise_r_kernel" in r.stdout, r.stdout[-2000:]
    geo = tmp_path / "geo"
    geo.mkdir()
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-inline-asm", "-ffp-contract=off",
                    "--save-temps=obj", "-c", os.path.join(csrc, "geometry.hip"), "-o", str(geo / "geometry.o")], capture_output=True, timeout=900, check=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "packed_fp32_check.py")] + [str(p) for p in geo.glob("*-gfx950.s")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "op_sel" in r.stdout, r.stdout[-2000:]
