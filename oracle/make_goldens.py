"""Generates tests/golden/*.npz by running the REAL reference (read-only tree at /root/reference) on
CPU through oracle/ref_shims.py.  Build-container only; the outputs are small data fixtures (inputs
are regenerated from seeds by lsdm_amd/synthetic.py; files hold expected outputs, taps and RNG draws).

    python -m oracle.make_goldens [--only g0,g1,g2,g3,g4,g5,long] [--long-steps 1000]

Golden sets:
  G0  state_dict names/shapes/dtypes of the reference model (proxd C=13, humanise C=11).
  G1  schedule tables: cosine-1000 and the ddim50 / ddim250 respaced processes (float64).
  G2  one SceneDiffusionModel.forward, B=3 ragged masks, C in {13, 43}: outputs + module taps.
  G3  PointNet++ known answers on 3 clouds (one all-zero): FPS / ball-query / 3-NN indices, features.
  G4  short p_sample_loop chains (skip_timesteps = 1000-K) with per-step samples; B in {1,3,16}.
  G6  hoisted conditioning: the reference with its FPS re-draw pinned to the first forward's starts, K = 4 steps.
  G5  DDIM-50: reference p_mean_variance + the arithmetic of gaussian_diffusion.py:790-811.
  G7  BASELINE configs[4]: humanise defaults (C=11), two prompts sharing one scene, 'ddim250' composed as G5, K = 3 steps.
  G8  CLI metrics: emd / accuracy of the reference's util/evaluation.py on seeded clouds and logits.
  G9  p_sample_loop argument variants (gaussian_diffusion.py:611-759): noise=, init_image + skip_timesteps, dump_steps;
      and what const_noise=True does in the reference (a 4-D repeat inherited from MDM: a (B,B,1024,3) sample).
  G11 dataset loaders: every tuple the reference's posa.dataset.ProxDataset_txt / HUMANISE yield on the trees written by
      lsdm_amd.data.write_synthetic_dataset / write_synthetic_humanise (seeded), keyed by prompt, plus the reference's own
      iteration order (os.listdir of the context directory, posa/dataset.py:374,503) as recorded in THIS container.
  G12 the training objective (gaussian_diffusion.py:1256-1342) on the real reference, B = 2, in model.eval() and in model.train()
      (batch-statistics BatchNorm, Dropout(0.5) of the encoder head with the mask the global generator gives): the three loss terms,
      every parameter's gradient (small tensors in full; large ones as norm / sum / a strided sample), the BatchNorm running
      statistics after the forward.  pytorch3d's chamfer_distance is absent: its published definition (oracle/train_ref.py) is
      patched in -- the chamfer term's parity is unpinned.
  G13 the other text tower's width: the reference built with clip_dim = 768 (the BERT pooler_output width, model/sdm.py:261-264, with
      embed_text 768 -> 384 -> 256 -> 128): one forward at B = 2 with injected (B,768) features -- outputs, saved_cat, guiding points,
      the embedding tap.  (The encoder itself -- transformers' bert-base-uncased -- is third party and not reachable offline: the
      boundary is its (B,768) feature, like CLIP's (B,512).)
  G14 BASELINE configs[1] at depth: B = 16, C = 43, first K = 5 steps (t = 999..995) and last K = 5 steps (init_image + skip_timesteps
      = 995, t = 4..0) of p_sample_loop, per-step samples + saved_cat + guiding points.
  G15 error accumulation at BASELINE configs[1]'s shape beyond ten steps (SURVEY.md 8(d) "Config 2" (ii)): B = 16, C = 43, K = 25 steps in the
      middle of the schedule (init_image + skip_timesteps = 475: q_sample to t = 524, then t = 524..500), every fifth sample.
  long  full T=1000 chain at B=1 (every 100th step + final); ~20 min of CPU.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shims  # noqa: E402

SDM, mu, clip = ref_shims.import_reference()
import torch  # noqa: E402
from diffusion import gaussian_diffusion as gd  # noqa: E402  (reference module)
from diffusion.respace import SpacedDiffusion, space_timesteps  # noqa: E402
import model.pcd_backbone.pointnet2_utils as pnu  # noqa: E402
from lsdm_amd.synthetic import make_synthetic_state_dict, make_synthetic_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def ref_model(max_cats, seed=0):
    kw = mu.get_default_model_proxd()
    kw["max_cats"] = max_cats
    m = SDM(**kw, use_cuda=False).eval()
    m.load_state_dict(make_synthetic_state_dict(seed=seed, max_cats=max_cats))
    return m


def ref_diffusion(respacing=None):
    """util/model_util.py:127-163 with an explicit timestep_respacing."""
    betas = gd.get_named_beta_schedule("cosine", 1000, 1.0)
    return SpacedDiffusion(
        use_timesteps=space_timesteps(1000, respacing if respacing else [1000]), betas=betas,
        model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
        loss_type=gd.LossType.MSE, rescale_timesteps=False, lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0,
        lambda_cat=0.1)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype == np.int64 and v.size > 16 and v.max(initial=0) < 32767 and v.min(initial=0) >= 0:
            v = v.astype(np.int16)
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def g0():
    """state_dict contract of the reference model (names, shapes, dtypes, registration order)."""
    import json
    for C in (13, 11):
        kw = mu.get_default_model_proxd() if C == 13 else mu.get_default_model_humanise()
        m = SDM(**kw, use_cuda=False)
        spec = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        with open(os.path.join(OUT, "g0_state_dict_c%d.json" % C), "w") as fh:
            json.dump(spec, fh)
        print("wrote g0_state_dict_c%d.json (%d entries)" % (C, len(spec)))


def g1():
    out = {}
    for tag, resp in (("t1000", None), ("ddim50", "ddim50"), ("ddim250", "ddim250")):
        d = ref_diffusion(resp)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_variance",
                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            out["%s.%s" % (tag, k)] = getattr(d, k)
        out["%s.timestep_map" % tag] = np.asarray(d.timestep_map, dtype=np.int64)
    save("g1_schedule", **out)


def g2():
    for C in (13, 43):
        m = ref_model(C)
        taps = {}
        hooks = [
            m.pcd_backbone.register_forward_hook(lambda mod, i, o: taps.__setitem__("pcd", o)),
            m.human_backbone.register_forward_hook(lambda mod, i, o: taps.__setitem__("hm", o)),
            m.pcd_attention.register_forward_hook(lambda mod, i, o: taps.__setitem__("o_rows", o[0])),
            m.attn_layer.register_forward_hook(lambda mod, i, o: taps.__setitem__("w", o[1])),
            m.translation_layer.register_forward_hook(lambda mod, i, o: taps.__setitem__("tr", o)),
            m.embed_text.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc", o)),
            m.embed_cat.register_forward_hook(lambda mod, i, o: taps.__setitem__("ecat", o)),
            m.embed_timestep.register_forward_hook(lambda mod, i, o: taps.__setitem__("ets", o)),
            m.upsampling_layer.register_forward_hook(lambda mod, i, o: taps.__setitem__("e", o)),
            m.combine_extraction.register_forward_hook(lambda mod, i, o: taps.__setitem__("emb", o)),
        ]
        B = 3
        bt = make_synthetic_batch(B, seed=5, max_cats=C)
        torch.manual_seed(11)
        x = torch.randn(B, 1024, 3)
        x_in = x.clone()
        t = torch.tensor([987, 500, 3])
        clip._model.features = bt["text_features"]
        torch.manual_seed(12)
        starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
        torch.manual_seed(12)
        out_cat, out = m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=["p"] * B)
        for h in hooks:
            h.remove()
        o_rows = taps["o_rows"].view(B, 9, 1024, 12)
        assert (o_rows - o_rows[:, :, :1]).abs().max().item() == 0.0  # Q7
        save("g2_forward_c%d" % C, x_in=x_in, t=t, starts=torch.stack([s for s in starts]),
             out_cat=out_cat, out=out, guiding=m.saved_guiding_points, x_mut=x,
             pcd=taps["pcd"].reshape(B, 9, -1), hm=taps["hm"], o=o_rows[:, :, 0], w=taps["w"][:, 0],
             tr=taps["tr"], enc=taps["enc"], ecat=taps["ecat"], ets=taps["ets"][0],
             e_sub=taps["e"].permute(0, 2, 1)[:, ::16], emb_sub=taps["emb"][:, ::16])


def g3():
    """Geometry known answers recorded by wrapping the reference's own helper functions."""
    m = ref_model(13)
    bb = m.pcd_backbone
    bt = make_synthetic_batch(2, seed=21, num_objects=3)
    clouds = torch.stack([bt["given_objs"][0, 0], bt["given_objs"][0, 2], torch.zeros(1024, 3),
                          bt["given_objs"][1, 1] * 0.3])
    rec = {"fps": [], "ball": [], "idx3": [], "sqd": []}
    o_fps, o_ball, o_index, o_sq = pnu.farthest_point_sample, pnu.query_ball_point, pnu.index_points, pnu.square_distance

    def w_fps(xyz, npoint):
        r = o_fps(xyz, npoint)
        rec["fps"].append(r.clone())
        return r

    def w_ball(radius, nsample, xyz, new_xyz):
        r = o_ball(radius, nsample, xyz, new_xyz)
        rec["ball"].append(r.clone())
        return r

    def w_index(points, idx):
        if idx.dim() == 3 and idx.shape[-1] == 3:
            rec["idx3"].append(idx.clone())
        return o_index(points, idx)

    def w_sq(src, dst):
        r = o_sq(src, dst)
        rec["sqd"].append(r.clone())
        return r

    pnu.farthest_point_sample, pnu.query_ball_point, pnu.index_points, pnu.square_distance = w_fps, w_ball, w_index, w_sq
    feats = {}
    hooks = []
    for name in ("sa1", "sa2", "sa3", "sa4"):
        hooks.append(getattr(bb, name).register_forward_hook(
            lambda mod, i, o, name=name: feats.__setitem__(name, (o[0].permute(0, 2, 1), o[1].permute(0, 2, 1)))))
    for name in ("fp4", "fp3", "fp2", "fp1"):
        hooks.append(getattr(bb, name).register_forward_hook(
            lambda mod, i, o, name=name: feats.__setitem__(name, o.permute(0, 2, 1))))
    try:
        torch.manual_seed(31)
        starts = [torch.randint(0, n, (clouds.shape[0],)) for n in (1024, 1024, 256, 64)]
        torch.manual_seed(31)
        out = bb(clouds)
    finally:
        pnu.farthest_point_sample, pnu.query_ball_point, pnu.index_points, pnu.square_distance = o_fps, o_ball, o_index, o_sq
        for h in hooks:
            h.remove()
    arrs = dict(clouds=clouds, starts=torch.stack(starts), out=out)
    SUB, FSUB = (8, 2, 1, 1), (1, 2, 8, 8)  # point-axis sub-sampling of the stored feature taps
    for i in range(4):
        arrs["fps%d" % (i + 1)] = rec["fps"][i]
        arrs["ball%d" % (i + 1)] = rec["ball"][i]
        arrs["sa%d_xyz" % (i + 1)] = feats["sa%d" % (i + 1)][0]
        arrs["sa%d_points" % (i + 1)] = feats["sa%d" % (i + 1)][1][:, ::SUB[i]]
    for i, name in enumerate(("fp4", "fp3", "fp2", "fp1")):
        arrs["%s_idx" % name] = rec["idx3"][i]
        arrs["%s_out" % name] = feats[name][:, ::FSUB[i]]
    # raw squared distances of the sa3 ball query (64x256) and fp3 (256x64): pins the fp32 rounding sequence
    arrs["sqd_sa3"] = rec["sqd"][2]
    arrs["sqd_fp3"] = rec["sqd"][5]
    save("g3_pointnet", **arrs)


def _chain(m, diff, B, C, K, seed, bt_seed, record_every=1):
    bt = make_synthetic_batch(B, seed=bt_seed, max_cats=C)
    clip._model.features = bt["text_features"]
    samples = []
    torch.manual_seed(seed)
    steps = 0
    for i, out in enumerate(diff.p_sample_loop_progressive(
            m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], ["p"] * B,
            clip_denoised=False, skip_timesteps=diff.num_timesteps - K)):
        if (i % record_every) == record_every - 1 or i == K - 1:
            samples.append(out["sample"].clone())
        steps += 1
    return torch.stack(samples), m.saved_cat.clone(), m.saved_guiding_points.clone()


def g4():
    diff = ref_diffusion()
    for tag, B, C, K in (("b1", 1, 13, 4), ("b3", 3, 13, 4), ("b16c43", 16, 43, 2)):
        m = ref_model(C)
        t0 = time.time()
        samples, cat, guid = _chain(m, diff, B, C, K, seed=100 + B, bt_seed=40 + B)
        print("chain %s: %.1f s" % (tag, time.time() - t0))
        if B == 16:
            samples = samples[-1:]
        save("g4_chain_%s" % tag, samples=samples, saved_cat=cat, guiding=guid, meta=np.array([B, C, K, 100 + B, 40 + B]))


def g6():
    """Hoisted conditioning: the REAL reference with its per-forward FPS re-draw pinned -- torch.randint
    (only used by farthest_point_sample on this path, pointnet2_utils.py:72) returns the first forward's four draws on
    every later forward; everything else (noise stream, sampler) is the stock reference."""
    diff = ref_diffusion()
    B, C, K = 2, 13, 4
    m = ref_model(C)
    real, rec, n = torch.randint, [], [0]

    def pinned(*a, **k):
        i = n[0]
        n[0] += 1
        if i < 4:
            rec.append(real(*a, **k))
            return rec[i]
        return rec[i % 4].clone()

    torch.randint = pinned
    try:
        samples, cat, guid = _chain(m, diff, B, C, K, seed=77, bt_seed=78)
    finally:
        torch.randint = real
    assert n[0] == 4 * K
    save("g6_hoisted", samples=samples, saved_cat=cat, guiding=guid, starts=torch.stack(rec), meta=np.array([B, C, K, 77, 78]))


def g5():
    """DDIM composed exactly as SURVEY.md A9: the reference's p_mean_variance on the respaced process,
    followed by the reference's ddim_sample arithmetic (gaussian_diffusion.py:790-811) re-evaluated
    with the reference's own helpers (eta = 0)."""
    diff = ref_diffusion("ddim50")
    B, C, K = 2, 13, 3
    m = ref_model(C)
    bt = make_synthetic_batch(B, seed=52, max_cats=C)
    clip._model.features = bt["text_features"]
    torch.manual_seed(61)
    img = torch.randn(B, 1024, 3)
    samples = []
    for i in list(range(diff.num_timesteps))[::-1][:K]:
        t = torch.tensor([i] * B)
        x = img
        out = diff.p_mean_variance(m, x, bt["mask"], t, bt["given_objs"], bt["given_cats"], ["p"] * B, clip_denoised=False)
        eps = diff._predict_eps_from_xstart(x, t, out["pred_xstart"])
        ab = gd._extract_into_tensor(diff.alphas_cumprod, t, x.shape)
        abp = gd._extract_into_tensor(diff.alphas_cumprod_prev, t, x.shape)
        sigma = 0.0 * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        noise = torch.randn_like(x)
        mean_pred = out["pred_xstart"] * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
        nz = (t != 0).float().view(-1, 1, 1)
        img = mean_pred + nz * sigma * noise
        samples.append(img.clone())
    save("g5_ddim50", samples=torch.stack(samples), meta=np.array([B, C, K, 61, 52]))


def g7():
    """configs[4] shapes on the real reference: get_default_model_humanise (C = 11), 2 prompts x 1 scene, ddim250 (stride 4,
    Q8: the model sees the respaced index), composed exactly like G5."""
    diff = ref_diffusion("ddim250")
    assert diff.num_timesteps == 250
    B, C, K = 2, 11, 3
    kw = mu.get_default_model_humanise()
    m = SDM(**kw, use_cuda=False).eval()
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C))
    one = make_synthetic_batch(1, seed=31, max_cats=C)
    bt = {k: v.repeat(2, *([1] * (v.dim() - 1))).contiguous() for k, v in one.items()}
    bt["text_features"] = make_synthetic_batch(2, seed=32, max_cats=C)["text_features"]
    clip._model.features = bt["text_features"]
    torch.manual_seed(9)
    img = torch.randn(B, 1024, 3)
    samples = []
    for i in list(range(diff.num_timesteps))[::-1][:K]:
        t = torch.tensor([i] * B)
        x = img
        out = diff.p_mean_variance(m, x, bt["mask"], t, bt["given_objs"], bt["given_cats"], ["p"] * B, clip_denoised=False)
        eps = diff._predict_eps_from_xstart(x, t, out["pred_xstart"])
        ab = gd._extract_into_tensor(diff.alphas_cumprod, t, x.shape)
        abp = gd._extract_into_tensor(diff.alphas_cumprod_prev, t, x.shape)
        sigma = 0.0 * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        noise = torch.randn_like(x)
        mean_pred = out["pred_xstart"] * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
        nz = (t != 0).float().view(-1, 1, 1)
        img = mean_pred + nz * sigma * noise
        samples.append(img.clone())
    save("g7_ddim250_humanise", samples=torch.stack(samples), saved_cat=m.saved_cat, guiding=m.saved_guiding_points,
         meta=np.array([B, C, K, 9, 31, 32]))


def g8():
    """emd / accuracy of the reference's util/evaluation.py (open3d is an import-only placeholder there; calculate_fscore
    needs the real package and is pinned against its definition instead, tests/test_metrics.py)."""
    import util.evaluation as ev
    g = torch.Generator().manual_seed(5)
    a = torch.randn(3, 1024, 3, generator=g)
    b = a + 0.3 * torch.randn(3, 1024, 3, generator=g)
    emds = [float(ev.emd(a[i:i + 1], b[i:i + 1])) for i in range(3)]
    emds.append(float(ev.emd(a[0, :200], b[1, :333])))              # unequal sizes, 2-D inputs
    logits = torch.randn(16, 13, generator=g)
    target = torch.randint(0, 13, (16,), generator=g)
    acc = [float(v) for v in ev.accuracy(logits, target, topk=(1, 3, 5))]
    save("g8_metrics", a=a, b=b, emd=np.array(emds), logits=logits, target=target, acc=np.array(acc))


def g9():
    """Argument variants of p_sample_loop on the real reference, B = 2, C = 13, K = 3 steps."""
    diff = ref_diffusion()
    B, C, K = 2, 13, 3
    m = ref_model(C)
    bt = make_synthetic_batch(B, seed=91, max_cats=C)
    clip._model.features = bt["text_features"]
    args = (m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], ["p"] * B)
    g = torch.Generator().manual_seed(92)
    noise0 = torch.randn(B, 1024, 3, generator=g)
    init = 0.5 * torch.randn(B, 1024, 3, generator=g)
    out = {}
    torch.manual_seed(93)      # noise= given: no x_T draw; skip_timesteps with init_image None -> q_sample(zeros)
    out["noise_arg"] = diff.p_sample_loop(*args, noise=noise0.clone(), clip_denoised=False, skip_timesteps=1000 - K)
    torch.manual_seed(94)      # init_image: x_T drawn, then q_sample(init_image, t = K-1)
    out["init_image"] = diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=1000 - K, init_image=init.clone())
    torch.manual_seed(95)
    dump = diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=1000 - K, dump_steps=[0, 2])
    assert isinstance(dump, list) and len(dump) == 2
    out["dump0"], out["dump2"] = dump
    # const_noise=True: `noise[[0]].repeat(B, 1, 1, 1)` on a 3-D sample gives (B,1,1024,3); one step returns a (B,B,1024,3)
    # tensor whose [i, j] = mean_j + sigma * noise_0 ... recorded as evidence; a second step cannot run on it
    torch.manual_seed(96)
    cn = diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=999, const_noise=True)
    out["const_noise_shape"] = np.array(cn.shape)
    out["const_noise_diag"] = torch.stack([cn[j, j] for j in range(B)])     # chain j with noise row 0
    try:
        torch.manual_seed(96)
        diff.p_sample_loop(*args, clip_denoised=False, skip_timesteps=998, const_noise=True)
        out["const_noise_two_steps_ok"] = np.array(1)
    except Exception as e:
        print("const_noise over two steps fails in the reference:", type(e).__name__, str(e)[:100])
        out["const_noise_two_steps_ok"] = np.array(0)
    save("g9_loop_variants", noise0=noise0, init=init, meta=np.array([B, C, K, 91, 92]), **out)


def g11():
    """The reference's two dataset classes on the synthetic trees (the generator is seeded, so the test re-writes the same trees and
    compares its own loaders with these tuples).  The path must contain exactly the underscores / components the reference parses
    (`data_dir.split('_')[1]`, posa/dataset.py:369; `data_dir.split('/')[-1]`, :497): a fresh directory under /tmp without underscores."""
    import shutil
    import tempfile
    import posa.dataset as pds
    from lsdm_amd.data import write_synthetic_dataset, write_synthetic_humanise
    root = tempfile.mkdtemp(prefix="lsdmg11", dir="/tmp")
    assert "_" not in root
    try:
        out = {}
        for tag, writer, cls, kw in (("prox", lambda r: write_synthetic_dataset(r, num_samples=4, seed=11), pds.ProxDataset_txt, dict(max_cats=13)),
                                     ("hum", lambda r: write_synthetic_humanise(r, num_scenes=2, prompts_per_scene=2, seed=12), pds.HUMANISE,
                                      dict(max_cats=11))):
            data_dir, objs_dir = writer(root)
            ds = cls(data_dir, objs_data_dir=objs_dir, **kw)
            items = [ds[i] for i in range(len(ds))]
            order = [it[5] for it in items]                      # prompts in the reference's os.listdir order (this container)
            by_prompt = sorted(range(len(items)), key=lambda i: items[i][5])
            out[tag + "_prompts_listdir_order"] = np.array(order)
            out[tag + "_seq_names_listdir_order"] = np.array(list(ds.seq_names))
            out[tag + "_prompts"] = np.array([items[i][5] for i in by_prompt])
            for j, name in enumerate(("mask", "given_objs", "given_cats", "target_obj", "target_cat")):
                out[tag + "_" + name] = torch.stack([torch.as_tensor(items[i][j]).float() for i in by_prompt])
            out[tag + "_len"] = np.array(len(ds))
        save("g11_loaders", meta=np.array([4, 11, 2, 2, 12]), **out)
    finally:
        shutil.rmtree(root)


GRAD_FULL = 4096      # gradients of tensors up to this size are stored in full; larger ones as a strided sample of this many entries


def grad_digest(g):
    """What G12 keeps of one gradient tensor: (l2 norm, sum, numel) in float64 and a strided sample (everything when small)."""
    flat = g.detach().double().reshape(-1)
    stride = max(1, flat.numel() // GRAD_FULL)
    return np.array([float(flat.norm()), float(flat.sum()), flat.numel(), stride]), flat[::stride].float().numpy()


def g12():
    from oracle import train_ref
    gd.chamfer_distance = lambda x, y: (train_ref.chamfer_distance(x, y), None)      # pytorch3d stand-in (its published definition)
    B, C = 2, 13
    diff = ref_diffusion()
    bt = make_synthetic_batch(B, seed=121, max_cats=C)
    g = torch.Generator().manual_seed(122)
    x_start = bt["given_objs"][:, 1] + 0.05 * torch.randn(B, 1024, 3, generator=g)      # a target cloud near the first object
    noise = torch.randn(B, 1024, 3, generator=g)
    t = torch.tensor([37, 641])
    target_cat = torch.zeros(B, C)
    target_cat[0, 4] = 1
    target_cat[1, 9] = 1
    out = dict(x_start=x_start, noise=noise, t=t, target_cat=target_cat, meta=np.array([B, C, 121, 122]))
    for mode in ("eval", "train"):
        m = ref_model(C)
        m.train(mode == "train")
        m.requires_grad_(True)
        for p_ in m.clip_model.parameters() if hasattr(m, "clip_model") else ():
            p_.requires_grad_(False)
        clip._model.features = bt["text_features"]
        torch.manual_seed(123)
        state = torch.get_rng_state()
        starts = [torch.randint(0, n, (B * 9,), dtype=torch.long) for n in (1024, 1024, 256, 64)]
        drop = torch.empty(B * 9, 128, 1024).bernoulli_(0.5)          # what F.dropout draws next on the same generator (train mode)
        torch.set_rng_state(state)
        with torch.enable_grad():
            terms = diff.training_losses(m, x_start.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], target_cat, y=["p"] * B,
                                         noise=noise.clone())
            loss = (terms["loss"] * torch.ones(B)).mean()              # run/train_sdm.py:76 with the uniform sampler's weights
            loss.backward()
        out[mode + "_fps_starts"] = torch.stack([s_.clone() for s_ in starts])
        if mode == "train":
            out["train_dropout_mask"] = np.packbits(drop.numpy().astype(np.uint8).reshape(-1))
        for k in ("loss", "mse", "cat_loss"):
            out["%s_%s" % (mode, k)] = np.array(float(terms[k]))
        sd_after = m.state_dict()
        names, none = [], []
        for name, p_ in m.named_parameters():
            if name.startswith("clip_model."):
                continue
            if p_.grad is None:
                none.append(name)
                continue
            names.append(name)
            out["%s_gstat/%s" % (mode, name)], out["%s_grad/%s" % (mode, name)] = grad_digest(p_.grad)
        out[mode + "_grad_names"], out[mode + "_grad_none"] = np.array(names), np.array(none)
        if mode == "train":
            for k, v in sd_after.items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    out["train_running/" + k] = v.detach().clone()
        print(mode, {k: float(terms[k]) for k in ("loss", "mse", "cat_loss")}, "params with grad:", len(names), "without:", none)
    save("g12_training", **out)


def g13():
    B, C, D = 2, 13, 768
    kw = mu.get_default_model_proxd()
    kw["max_cats"], kw["clip_dim"] = C, D
    m = SDM(**kw, use_cuda=False).eval()          # text_encoder_type stays "CLIP": the stub returns whatever width is injected
    m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C, clip_dim=D))
    bt = make_synthetic_batch(B, seed=131, max_cats=C, text_dim=D)
    clip._model.features = bt["text_features"]
    torch.manual_seed(132)
    x = torch.randn(B, 1024, 3)
    t = torch.tensor([11, 777])
    state = torch.get_rng_state()
    starts = torch.stack([torch.randint(0, n, (B * 9,), dtype=torch.long) for n in (1024, 1024, 256, 64)])
    torch.set_rng_state(state)
    taps = {}
    hook = m.embed_text.register_forward_hook(lambda mod, i, o: taps.__setitem__("enc", o.detach().clone()))
    x_in = x.clone()
    out_cat, out = m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], ["p"] * B)
    hook.remove()
    save("g13_forward_clipdim768", x_in=x_in, t=t, starts=starts, out_cat=out_cat, out=out, guiding=m.saved_guiding_points, x_mutated=x,
         enc=taps["enc"], meta=np.array([B, C, D, 131, 132]))


def g14():
    """BASELINE configs[1] at the depth SURVEY.md 8(d) "Config 2" asks for: B = 16, C = 43, the FIRST K = 5 steps of the stock loop
    (t = 999..995, from x_T) and the LAST K = 5 steps (init_image + skip_timesteps = 995: q_sample to t = 4, then t = 4..0 with the
    coef1[0] = 1 / no-noise end).  Per step: the sample on every 8th point (the last step in full); after the last forward saved_cat
    and every 2nd guiding point.  Two files of ~0.4 MB."""
    diff = ref_diffusion()
    B, C, K = 16, 43, 5
    m = ref_model(C)
    bt = make_synthetic_batch(B, seed=141, max_cats=C)
    clip._model.features = bt["text_features"]
    args = (m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], ["p"] * B)
    for tag, seed, kw in (("first", 142, {}), ("last", 143, {"skip_timesteps": 1000 - K})):
        extra = {}
        if tag == "last":
            g = torch.Generator().manual_seed(144)
            kw["init_image"] = 0.7 * torch.randn(B, 1024, 3, generator=g)
            extra["init_seed"] = np.array(144)
        torch.manual_seed(seed)
        t0, got = time.time(), []
        for i, out in enumerate(diff.p_sample_loop_progressive(*args, clip_denoised=False, **kw)):
            got.append(out["sample"].clone())
            if len(got) == K:
                break
        print("g14 %s: %.1f s" % (tag, time.time() - t0))
        save("g14_config2_%s" % tag, steps_sub=torch.stack(got)[:, :, ::8], final=got[-1], saved_cat=m.saved_cat,
             guiding_sub=m.saved_guiding_points[:, ::2], meta=np.array([B, C, K, seed, 141]), **extra)


def g15():
    """Mid-schedule accumulation at configs[1]'s shape: B = 16, C = 43, K = 25 consecutive steps from t = 524 (gaussian_diffusion.py:684-759:
    init_image + skip_timesteps = 475 draws q_sample's noise first, then the stock loop).  Every fifth sample on every 4th point, the 25th in
    full; saved_cat and every 2nd guiding point after the last forward.  ~20 min of CPU; one file < 1 MB."""
    diff = ref_diffusion()
    B, C, K, SKIP = 16, 43, 25, 475
    m = ref_model(C)
    bt = make_synthetic_batch(B, seed=151, max_cats=C)
    clip._model.features = bt["text_features"]
    g = torch.Generator().manual_seed(154)
    init = 0.7 * torch.randn(B, 1024, 3, generator=g)
    torch.manual_seed(152)
    t0, got = time.time(), []
    for i, out in enumerate(diff.p_sample_loop_progressive(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], ["p"] * B,
                                                           clip_denoised=False, skip_timesteps=SKIP, init_image=init)):
        if i % 5 == 4:
            got.append(out["sample"].clone())
            print("g15 step %d: %.1f s" % (i + 1, time.time() - t0), flush=True)
        if i + 1 == K:
            break
    save("g15_config2_mid", steps_sub=torch.stack(got)[:, :, ::4], final=got[-1], saved_cat=m.saved_cat,
         guiding_sub=m.saved_guiding_points[:, ::2], meta=np.array([B, C, K, 152, 151, SKIP, 154]))


def long_chain(steps):
    diff = ref_diffusion()
    m = ref_model(13)
    t0 = time.time()
    samples, cat, guid = _chain(m, diff, 1, 13, steps, seed=777, bt_seed=70, record_every=100)
    print("long chain: %.1f s" % (time.time() - t0))
    save("g4_chain_long_b1", samples=samples, saved_cat=cat, guiding=guid, meta=np.array([1, 13, steps, 777, 70]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g0,g1,g2,g3,g4,g5,g6")
    ap.add_argument("--long-steps", type=int, default=1000)
    a = ap.parse_args()
    todo = a.only.split(",")
    for name in todo:
        if name == "long":
            long_chain(a.long_steps)
        else:
            globals()[name]()
