"""GPU tests of the training objective: diffusion.training_losses(...).backward() through lsdm_train_step
(lsdm_amd/csrc/train.hip) against G12 -- loss terms and d loss / d theta of every parameter recorded from the REAL reference in
model.eval() and model.train() -- and against torch.autograd of the CPU oracle; the optimizer loop of run.train_sdm."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
from test_oracle_golden import T, g12_check_grads, g12_inputs

DEV = torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _grad_enabled():
    """(the parity modules switch autograd off process-wide at import; training needs it)"""
    with torch.enable_grad():
        yield


def _model(C=13, train=False):
    model, diffusion = create_model_and_diffusion("proxd", device=0)
    model.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C))
    model.requires_grad_(True)
    model.train(train)
    return model, diffusion


def _g12_step(g, mode, model, diffusion):
    B, C, bt, x_start, noise, t, target_cat = g12_inputs(g)
    starts = [s.long() for s in T(g[mode + "_fps_starts"])]
    keep = None
    if mode == "train":
        keep = torch.from_numpy(np.unpackbits(g["train_dropout_mask"])[:B * 9 * 128 * 1024].reshape(B * 9, 128, 1024).copy())
    d = {k: v.to(DEV) for k, v in bt.items()}
    terms = diffusion.training_losses(model, x_start.to(DEV), d["mask"], t.to(DEV), d["given_objs"], d["given_cats"], target_cat.to(DEV),
                                      y=d["text_features"], noise=noise.to(DEV), fps_starts=starts, dropout_keep=keep)
    return B, terms


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_losses_and_gradients_vs_reference_golden(golden, mode):
    """Against the record of the REAL reference (G12): loss terms within 1e-4 relative; every gradient tensor within 3e-3 in relative
    l2 error (and its norm within 3e-3), its entries within 1e-2 of the tensor's largest (the next test holds the tight bar).  The entry-wise slack is the two
    discontinuities of the backward pass: where two neighbours of a PointNet++ group tie to the last bits, WHICH one the max-pool
    routes the gradient to -- and whether a ReLU at a pre-activation of +-1e-7 passes it -- depends on the summation order of the
    layer before (a handful of the 2.2 M arg-maxes / 150 M activations; each moves one row's contribution in one channel; the next
    test pins the first kind).  The three dead attention tensors have no gradient; in training mode the BatchNorm running
    statistics after the step are checked."""
    g = golden("g12_training")
    model, diffusion = _model(train=mode == "train")
    before = {k: v.clone() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    B, terms = _g12_step(g, mode, model, diffusion)
    loss = (terms["loss"] * torch.ones(B, device=DEV)).mean()            # run/train_sdm.py:76 with the uniform sampler's weights
    loss.backward()
    torch.cuda.synchronize()
    for k in ("loss", "mse", "cat_loss"):
        ref = float(g["%s_%s" % (mode, k)])
        assert abs(float(terms[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(terms[k]), ref)
    grads = {n: p.grad for n, p in model.named_parameters()}
    worst = g12_check_grads(g, mode, grads, rel_l2=3e-3, rel_max=1e-2)
    print("worst sampled gradient deviation:", worst)
    after = model.state_dict()
    if mode == "train":
        for k in before:
            ref = T(g["train_running/" + k])
            assert float((after[k].cpu() - ref).abs().max()) <= 1e-4 * (1.0 + float(ref.abs().max())), k
        assert int(after["pcd_backbone.bn1.num_batches_tracked"]) == 1
    else:
        assert all(torch.equal(after[k], before[k]) for k in before)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_gradients_strict_with_pool_choices_attributed(golden, mode):
    """The strict bar, against the EXACT gradient of the reference's objective: torch.autograd of the oracle (pinned to the reference
    by G12, tests/test_oracle_golden.py) evaluated in float64 on the G12 batch -- discrete geometry decisions in fp32 as the
    reference takes them, and the max-pool routing attributed: the oracle sends each pooled gradient to the neighbour the HIP path
    picked, and every pick that differs from the fp32 oracle's own arg-max must be a near-tie (the picked activation within 1e-5
    relative of the maximum).  Then d loss / d theta of EVERY parameter: relative l2 error <= 1e-3 per tensor; entry-wise <= 1e-3
    of the tensor's largest entry in eval mode, <= 1e-2 in training mode (batch-statistics BatchNorm shifts every pre-activation by
    a few ulp, so a ReLU sitting on zero may flip: one row of one channel).  Why float64: in training mode torch's OWN fp32
    evaluation is 6e-4 .. 1e-3 away from this gradient in every PointNet++ tensor (printed below), so an fp32 record cannot hold
    a 1e-3 bar against anything but itself; in eval mode the two agree to 2e-6."""
    from oracle import cpu_ref, train_ref
    g = golden("g12_training")
    model, diffusion = _model(train=mode == "train")
    B, terms = _g12_step(g, mode, model, diffusion)
    terms["loss"].backward()
    picks = model._train_session.pool_choices(B)
    _, C, bt, x_start, noise, t, target_cat = g12_inputs(g)
    sd = make_synthetic_state_dict(seed=0, max_cats=C)
    starts = [s.long() for s in T(g[mode + "_fps_starts"])]
    mk_train = lambda: None
    if mode == "train":
        keep = torch.from_numpy(np.unpackbits(g["train_dropout_mask"])[:B * 9 * 128 * 1024].reshape(B * 9, 128, 1024).astype(np.float32))
        mk_train = lambda: cpu_ref.TrainMode(dropout_mask=keep)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tab = cpu_ref.Sampler(sd).tab
    args = (sd, tab, x_start.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], target_cat, bt["text_features"], noise, starts)
    # the fp32 oracle's own pooling inputs: how many of the HIP picks differ, and by how much
    taps = {"want_pool_in": True}
    with torch.no_grad():
        train_ref.training_losses(*[a.clone() if torch.is_tensor(a) else a for a in args], lambda_cat=0.1, train=mk_train(), taps=taps)
    differing = 0
    for l, pick in enumerate(picks):
        h = taps["sa%d.pool_in" % (l + 1)]                                    # (BO,C,32,S)
        top, arg = h.max(dim=2)
        mine = torch.gather(h, 2, pick.long().permute(0, 2, 1).unsqueeze(2)).squeeze(2)
        gap = ((top - mine).abs() / (top.abs() + 1e-30)).permute(0, 2, 1)
        assert float(gap.max()) <= 1e-5, ("sa%d" % (l + 1), float(gap.max()))
        differing += int(((arg.permute(0, 2, 1) != pick.long()) & (gap > 0)).sum())
    print("max-pool picks that differ from the oracle's arg-max (all within 1e-5 of the maximum):", differing)
    assert differing <= 200
    del taps
    inject = {"sa%d" % (l + 1): p for l, p in enumerate(picks)}
    losses, ref = train_ref.losses_and_grads(*args, lambda_cat=0.1, train=mk_train(), pool_inject=inject, dtype=torch.float64)
    _, ref32 = train_ref.losses_and_grads(*args, lambda_cat=0.1, train=mk_train(), pool_inject=inject)
    for k in ("loss", "mse", "cat_loss"):
        assert abs(float(terms[k]) - losses[k]) <= 1e-5 * max(1.0, abs(losses[k])), (k, float(terms[k]), losses[k])
    worst, worst32, bad = (0.0, 0.0, ""), (0.0, 0.0, ""), []
    for n, p in model.named_parameters():
        if ref[n] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        if mode == "train" and float(ref[n].norm()) <= 1e-6 and n.startswith("pcd_backbone.") and n.endswith(".bias"):
            assert float(p.grad.norm()) <= 1e-3, n          # conv bias in front of a batch-statistics BatchNorm: exact gradient 0
            continue
        scale, nrm = float(ref[n].abs().max()) + 1e-30, float(ref[n].norm()) + 1e-30
        d = p.grad.cpu().double() - ref[n]
        err, l2 = float(d.abs().max()) / scale, float(d.norm()) / nrm
        d32 = ref32[n].double() - ref[n]
        worst = max(worst, (err, l2, n))
        worst32 = max(worst32, (float(d32.abs().max()) / scale, float(d32.norm()) / nrm, n))
        if l2 > (1e-3 if mode == "eval" else 3e-3) or err > (1e-3 if mode == "eval" else 1e-2):
            bad.append((n, err, l2))
    print("worst deviation from the float64 gradient (entry-wise, l2, tensor): HIP", worst, " torch fp32 oracle", worst32)
    assert not bad, bad


def test_training_objective_without_grad_and_determinism(golden):
    """torch.no_grad(): values only, no graph; two evaluations of the same batch give the same loss and the same gradients up to the
    order of the atomic accumulations (1e-5 relative)."""
    g = golden("g12_training")
    model, diffusion = _model()
    with torch.no_grad():
        _, terms = _g12_step(g, "eval", model, diffusion)
    assert not terms["loss"].requires_grad and abs(float(terms["loss"]) - float(g["eval_loss"])) <= 2e-3
    grads = []
    for _ in range(2):
        model.zero_grad()
        _, terms = _g12_step(g, "eval", model, diffusion)
        terms["loss"].backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    for n in grads[0]:
        scale = float(grads[0][n].abs().max()) + 1e-30
        assert float((grads[0][n] - grads[1][n]).abs().max()) <= 1e-4 * scale, n
    # gradient accumulation across backward calls is torch's (p.grad += ...): a second backward doubles
    _, terms = _g12_step(g, "eval", model, diffusion)
    terms["loss"].backward()
    n = "output_process.pose_final.2.weight"
    assert float((dict(model.named_parameters())[n].grad - 2 * grads[1][n]).abs().max()) <= 1e-3 * float(grads[1][n].abs().max())
    # an evaluation under no_grad between a forward and its backward runs the objective only (flags & 2): the session's gradient buffers
    # keep the first batch's gradients, the backward is the same as without it, and the model is not marked for a parameter re-upload
    model.zero_grad()
    _, terms = _g12_step(g, "eval", model, diffusion)
    model._dirty = False
    with torch.no_grad():
        _, ev = _g12_step(g, "eval", model, diffusion)
    assert model._dirty is False and abs(float(ev["loss"]) - float(terms["loss"])) <= 1e-6 * abs(float(terms["loss"]))
    terms["loss"].backward()
    assert float((dict(model.named_parameters())[n].grad - grads[1][n]).abs().max()) <= 1e-4 * float(grads[1][n].abs().max())
    # ... but a second objective WITH gradients rewrites those buffers: the stale backward is refused instead of returning the other batch's
    _, first = _g12_step(g, "eval", model, diffusion)
    _, second = _g12_step(g, "eval", model, diffusion)
    with pytest.raises(RuntimeError, match="rewritten by a later"):
        first["loss"].backward()
    second["loss"].backward()


def test_training_gradients_vs_oracle_autograd_humanise_defaults():
    """A second configuration against torch.autograd of the CPU oracle run live (C = 11, B = 3, ragged masks, other timesteps)."""
    from oracle import cpu_ref, train_ref
    B, C = 3, 11
    model, diffusion = create_model_and_diffusion("humanise", device=0)
    sd = make_synthetic_state_dict(seed=0, max_cats=C)
    model.load_state_dict(sd)
    model.requires_grad_(True)
    model.eval()
    bt = make_synthetic_batch(B, seed=77, max_cats=C)
    gen = torch.Generator().manual_seed(78)
    x_start = bt["given_objs"][:, 1] + 0.1 * torch.randn(B, 1024, 3, generator=gen)
    noise = torch.randn(B, 1024, 3, generator=gen)
    t = torch.tensor([3, 500, 998])
    target_cat = torch.zeros(B, C)
    target_cat[torch.arange(B), torch.tensor([1, 5, 10])] = 1
    torch.manual_seed(79)
    starts = cpu_ref.draw_fps_starts(B * 9)
    tab = cpu_ref.Sampler(sd).tab
    losses, ref = train_ref.losses_and_grads(sd, tab, x_start.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], target_cat,
                                             bt["text_features"], noise, starts, lambda_cat=diffusion.lambda_cat)
    d = {k: v.to(DEV) for k, v in bt.items()}
    terms = diffusion.training_losses(model, x_start.to(DEV), d["mask"], t.to(DEV), d["given_objs"], d["given_cats"], target_cat.to(DEV),
                                      y=d["text_features"], noise=noise.to(DEV), fps_starts=starts)
    terms["loss"].backward()
    for k in ("loss", "mse", "cat_loss"):
        assert abs(float(terms[k]) - losses[k]) <= 1e-4 * max(1.0, abs(losses[k])), (k, float(terms[k]), losses[k])
    worst = (0.0, "")
    for n, p in model.named_parameters():
        if ref[n] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        scale = float(ref[n].abs().max()) + 1e-30
        err = float((p.grad.cpu() - ref[n]).abs().max()) / scale
        worst = max(worst, (err, n))
        assert err <= 1e-3, (n, err)
    print("worst:", worst)


def test_train_cli_steps_the_optimizer_and_the_loss_goes_down(tmp_path, capsys):
    """python -m run.train_sdm --synthetic 8 --epochs 3: AdamW steps on the HIP gradients; the epoch objective decreases, the
    parameters move, a checkpoint in the reference's layout is written and loads back into the sampling path."""
    import run.train_sdm as cli
    torch.manual_seed(5)
    np.random.seed(5)
    rc = cli.main(["--train_data_dir", str(tmp_path / "tr"), "--valid_data_dir", str(tmp_path / "va"), "--synthetic", "8", "--epochs", "3",
                   "--batch_size", "4", "--lr", "1e-3", "--save_interval", "3", "--valid_skip_timesteps", "997", "--out_dir", str(tmp_path / "out"),
                   "--experiment", "t", "--device", "cuda:0"])
    assert rc == 0
    out = capsys.readouterr().out
    totals = [float(l.split(": ")[1]) for l in out.splitlines() if l.startswith("====> Total_train_loss")]
    assert len(totals) == 3 and totals[-1] < totals[0], totals
    assert "====>Recon_loss_semantics" in out
    ck = torch.load(tmp_path / "out" / "t" / "model_ckpt" / "epoch_0002.pt", map_location="cpu")
    assert ck["epoch"] == 2 and "total_valid_loss" in ck
    sd0 = make_synthetic_state_dict(seed=0, max_cats=13)
    moved = [k for k in sd0 if k in ck["model_state_dict"] and sd0[k].dtype == torch.float32 and not torch.equal(sd0[k], ck["model_state_dict"][k].cpu())]
    assert len(moved) > 150
    model, diffusion = create_model_and_diffusion("proxd", device=0)
    model.load_state_dict(ck["model_state_dict"])
    best = torch.load(tmp_path / "out" / "t" / "model_ckpt" / "best_model_cfd.pt", map_location="cpu")
    assert "total_cfd" in best and best["epoch"] == 2                 # run/train_sdm.py:326-335
    # the scalars of the reference's SummaryWriter (run/train_sdm.py:104, 176-178), with its tags, under <out_dir>/<experiment>/tb_log
    logd = tmp_path / "out" / "t" / "tb_log"
    if (logd / "scalars.jsonl").exists():         # tensorboard absent: the JSON-lines fallback
        import json
        recs = [json.loads(l) for l in (logd / "scalars.jsonl").read_text().splitlines()]
        assert [r["step"] for r in recs if r["tag"] == "total/train_total_loss"] == [0, 1, 2]
        assert [abs(r["value"] - t) < 1e-3 for r, t in zip([r for r in recs if r["tag"] == "total/train_total_loss"], totals)] == [True] * 3
        assert {r["tag"] for r in recs if r["step"] == 2} == {"total/train_total_loss", "recon_loss_semantics/validate", "total_cfd/validate", "total_acc/validate"}
    else:
        assert any(f.name.startswith("events.out.tfevents") for f in logd.iterdir())
    # --load_ckpt resumes at checkpoint['epoch'] + 1 (run/train_sdm.py:284-289): epochs 3..4 run, epoch_0002.pt is not rewritten
    before = os.path.getmtime(tmp_path / "out" / "t" / "model_ckpt" / "epoch_0002.pt")
    rc = cli.main(["--train_data_dir", str(tmp_path / "tr"), "--valid_data_dir", str(tmp_path / "va"), "--synthetic", "8", "--epochs", "5",
                   "--batch_size", "4", "--lr", "1e-3", "--save_interval", "5", "--valid_skip_timesteps", "997", "--out_dir", str(tmp_path / "out"),
                   "--experiment", "t", "--device", "cuda:0", "--load_ckpt", str(tmp_path / "out" / "t" / "model_ckpt" / "epoch_0002.pt")])
    assert rc == 0
    out = capsys.readouterr().out
    assert "loading stats of epoch 3" in out and "Training epoch 3" in out and "Training epoch 4" in out and "Training epoch 0" not in out
    assert os.path.getmtime(tmp_path / "out" / "t" / "model_ckpt" / "epoch_0002.pt") == before
    assert torch.load(tmp_path / "out" / "t" / "model_ckpt" / "epoch_0004.pt", map_location="cpu")["epoch"] == 4


def test_train_cli_forward_only_reports_the_eval_objective(tmp_path, capsys):
    """--forward_only: model.eval(), no optimizer, no graph -- the per-batch objective and the validation report; parameters unchanged."""
    import run.train_sdm as cli
    torch.manual_seed(6)
    np.random.seed(6)
    rc = cli.main(["--train_data_dir", str(tmp_path / "tr"), "--valid_data_dir", str(tmp_path / "va"), "--synthetic", "4", "--epochs", "1",
                   "--batch_size", "2", "--forward_only", "--valid_skip_timesteps", "998", "--device", "cuda:0"])
    assert rc == 0
    out = capsys.readouterr().out
    assert out.count("====> Total_train_loss") == 1 and "====>Recon_loss_semantics" in out
    assert not (tmp_path / "training").exists()


def test_training_step_at_the_reference_batch_size_and_beyond():
    """B = 6 (run/train_sdm.py:247) and B = 24 (1.77 M grouped rows in sa1: row tiles beyond the 65 535 limit of a grid's y dimension):
    finite losses and gradients, every live parameter receives one."""
    for B in (6, 24):
        model, diffusion = _model(train=True)
        bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=50 + B, num_objects=8).items()}
        target_cat = torch.zeros(B, 13, device=DEV)
        target_cat[:, 2] = 1
        torch.manual_seed(B)
        terms = diffusion.training_losses(model, bt["given_objs"][:, 1].clone(), bt["mask"], torch.randint(0, 1000, (B,), device=DEV), bt["given_objs"],
                                          bt["given_cats"], target_cat, y=bt["text_features"])
        terms["loss"].backward()
        assert bool(torch.isfinite(terms["loss"])) and float(terms["mse"]) > 0
        missing = [n for n, p in model.named_parameters() if p.grad is None and not n.startswith("attn_layer.v_proj") and not n.startswith("attn_layer.out_proj")]
        assert not missing, missing
        assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
        assert float(dict(model.named_parameters())["pcd_backbone.sa1.mlp_convs.0.weight"].grad.abs().max()) > 0
        del model._train_session


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_compacted_groups_agree_with_every_group_on_32_rows(mode):
    """The compacted, weighted evaluation of the PointNet++ stacks (train.hip: cg_* kernels, LSDM_TRAIN_COMPACT=1, the default: a padded
    ball-query group runs on its distinct rows plus one weighted stand-in for the repeats) against every group on its 32 rows, on scenes
    whose objects cover the extremes -- a cloud scattered over 100 m (every group is its centroid alone: 2 rows instead of 32), a single
    point and an all-zero slot (32 distinct entries everywhere: nothing to compact), a line, a 2 cm blob -- in eval and in training mode
    (batch statistics with the weights; running statistics after the step).  Same max-pool picks; loss terms within 1e-5; every gradient
    within 2e-3 relative l2 (the sums run over fewer, weighted terms: the backward pass's ReLU / arg-max discontinuities move with the
    last bits, as between this build and the oracle)."""
    B = 3
    bt = make_synthetic_batch(B, seed=43)
    g = torch.Generator().manual_seed(80)
    objs = bt["given_objs"].clone()
    objs[0, 2] = (torch.rand(1024, 3, generator=g) - 0.5) * 100.0
    objs[0, 3] = torch.tensor([0.3, -1.2, 0.7]).expand(1024, 3)
    objs[1, 4] = 0.0
    objs[1, 5, :, 0] = torch.linspace(-1.0, 1.0, 1024); objs[1, 5, :, 1:] = 0.0
    objs[2, 6] = torch.randn(1024, 3, generator=g) * 0.02
    bt["given_objs"] = objs
    d = {k: v.to(DEV) for k, v in bt.items()}
    x_start = objs[:, 1].clone().to(DEV)
    target_cat = torch.zeros(B, 13, device=DEV)
    target_cat[torch.arange(B), torch.tensor([2, 7, 11])] = 1
    torch.manual_seed(9)
    t = torch.randint(0, 1000, (B,))
    noise = torch.randn(B, 1024, 3)
    starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
    keep = torch.empty(B * 9, 128, 1024).bernoulli_(0.5).to(torch.uint8) if mode == "train" else None
    res = {}
    old = os.environ.get("LSDM_TRAIN_COMPACT")
    try:
        for c in ("1", "0"):
            os.environ["LSDM_TRAIN_COMPACT"] = c                      # (read when the training session is created)
            model, diffusion = _model(train=mode == "train")
            terms = diffusion.training_losses(model, x_start, d["mask"], t.to(DEV), d["given_objs"], d["given_cats"], target_cat, y=d["text_features"],
                                              noise=noise.to(DEV), fps_starts=starts, dropout_keep=keep)
            terms["loss"].backward()
            res[c] = ({k: float(terms[k]) for k in ("loss", "mse", "cat_loss")}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                      [a.clone() for a in model._train_session.pool_choices(B)],
                      {k: v.clone() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")})
            del model, diffusion
    finally:
        if old is None:
            os.environ.pop("LSDM_TRAIN_COMPACT", None)
        else:
            os.environ["LSDM_TRAIN_COMPACT"] = old
    a, b = res["1"], res["0"]
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-5 * max(1.0, abs(b[0][k])), (k, a[0][k], b[0][k])
    differing = sum(int((x != y).sum()) for x, y in zip(a[2], b[2]))
    total = sum(x.numel() for x in a[2])
    assert differing <= 2e-5 * total, (differing, total)             # (near-ties of two distinct rows may flip with the statistics' last bits)
    assert set(a[1]) == set(b[1])
    for n in a[1]:
        ref = b[1][n]
        err = float((a[1][n] - ref).norm()) / max(float(ref.norm()), 1e-12)
        if mode == "train" and n.endswith(".bias") and (".mlp_convs." in n or n == "pcd_backbone.conv1.bias"):
            # a conv bias in front of a batch-statistics BatchNorm has a ZERO gradient: what both builds hold there is rounding residue
            wmax = float(b[1][n[:-4] + "weight"].abs().max())
            assert float(a[1][n].abs().max()) <= 1e-4 * wmax and float(ref.abs().max()) <= 1e-4 * wmax, (n, float(a[1][n].abs().max()), wmax)
            continue
        assert err <= 2e-3 or float((a[1][n] - ref).abs().max()) <= 1e-7, (n, err, float((a[1][n] - ref).abs().max()))
    for k in a[3]:
        assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-5 * max(1.0, float(b[3][k].abs().max())), k
