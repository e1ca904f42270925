"""``python -m run.train_sdm`` -- the forward side of the reference's training script (run/train_sdm.py).

What exists: the flags of run/train_sdm.py:187-213, the per-batch objective of ``train()`` (:30-107: uniform timestep sampler,
``diffusion.training_losses``, ``(losses["loss"] * weights).mean()``) and the whole of ``validate()`` (:110-183: p_sample_loop,
squared-error / chamfer / category accuracy, the reference's report line) -- all evaluated on the HIP path.
What does not: parameter updates.  The HIP kernels have no backward pass, so ``--forward_only`` (report the objective of
every batch without stepping) is the only training mode; without it the script stops with a clear error BEFORE touching
an optimizer.  Additions: ``--synthetic N``, ``--device``, ``--forward_only``, ``--valid_skip_timesteps`` (shorter chains for
smoke runs; 0 = the reference's full T = 1000)."""
import argparse
import sys

import torch
from torch.utils.data import DataLoader

from ..data import (HumaniseDataset, ProxTextDataset, synthetic_text_features, write_synthetic_dataset, write_synthetic_humanise)
from ..diffusion.resample import create_named_schedule_sampler
from ..metrics import chamfer_distance


def build_parser():
    p = argparse.ArgumentParser(description="")
    p.add_argument("--train_data_dir", type=str, default="data/proxd_train")
    p.add_argument("--valid_data_dir", type=str, default="data/proxd_valid")
    p.add_argument("--load_ckpt", type=str, default=None)
    p.add_argument("--posa_path", type=str, default="training/posa/model_ckpt/best_model_recon_acc.pt")
    p.add_argument("--out_dir", type=str, default="training/")
    p.add_argument("--experiment", type=str, default="default_experiment")
    p.add_argument("--save_interval", type=int, default=50)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--epochs", type=int, default=1000)
    p.add_argument('--fix_ori', dest='fix_ori', action='store_const', const=True, default=False)
    p.add_argument("--encoder_mode", type=int, default=1)
    p.add_argument("--decoder_mode", type=int, default=1)
    p.add_argument("--n_layer", type=int, default=3)
    p.add_argument("--n_head", type=int, default=4)
    p.add_argument("--dim_ff", type=int, default=512)
    p.add_argument("--f_vert", type=int, default=64)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--jump_step", type=int, default=8)
    p.add_argument("--max_frame", type=int, default=256)
    p.add_argument("--eval_epochs", type=int, default=10)
    p.add_argument("--datatype", type=str, default="proxd")
    # additions
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--forward_only", action="store_true")
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--objs_dir", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=6, help="run/train_sdm.py:247 hard-codes 6")
    p.add_argument("--valid_skip_timesteps", type=int, default=0)
    return p


def training_objective(model, diffusion, loader, device, sampler=None):
    """One pass of run/train_sdm.py:52-82 without the optimizer: -> (mean objective, per-batch list of term dicts)."""
    sampler = sampler or create_named_schedule_sampler('uniform', diffusion)
    total, batches = 0.0, []
    for mask, given_objs, given_cats, target_obj, target_cat, y in loader:
        mask, given_objs, given_cats = mask.to(device), given_objs.to(device), given_cats.to(device)
        target_obj, target_cat = target_obj.float().to(device), target_cat.to(device)
        t, weights = sampler.sample(target_obj.shape[0], device)
        losses = diffusion.training_losses(model, target_obj, mask, t, given_objs, given_cats, target_cat, y=y)
        loss = (losses["loss"] * weights).mean()
        total += float(loss)
        batches.append({k: float(v) for k, v in losses.items()})
    return total / max(len(batches), 1), batches


def validate(model, diffusion, loader, device, skip_timesteps=0):
    """run/train_sdm.py:110-183: -> (recon_loss_semantics, chamfer distance, category accuracy), each a batch mean."""
    tot_sq, tot_cfd, tot_acc, n = 0.0, 0.0, 0.0, 0
    with torch.no_grad():
        for mask, given_objs, given_cats, target_obj, target_cat, y in loader:
            mask, given_objs, given_cats = mask.to(device), given_objs.to(device), given_cats.to(device)
            target_obj, target_cat = target_obj.float().to(device), target_cat.to(device)
            sample = diffusion.p_sample_loop(model, list(target_obj.shape), mask, given_objs, given_cats, y=y, clip_denoised=False,
                                             model_kwargs=None, skip_timesteps=skip_timesteps, init_image=None, progress=False,
                                             dump_steps=None, noise=None, const_noise=False)
            tot_sq += float(((sample - target_obj) ** 2).mean())
            tot_cfd += float(chamfer_distance(sample, target_obj))
            pred = torch.argmax(model.saved_cat.squeeze(1), dim=1)
            tot_acc += int((pred == torch.argmax(target_cat, dim=1)).sum())
            n += 1
    out = (tot_sq / n, tot_cfd / n, tot_acc / n)
    print('====>Recon_loss_semantics = {:.4f} , Chamfer distance = {:.4f}, Category acc = {:.4f}'.format(*out))
    return out


def main(argv=None):
    from ..synthetic import make_synthetic_state_dict
    from ..util.model_util import create_model_and_diffusion
    args = build_parser().parse_args(argv)
    device = torch.device(args.device)
    text_encoder = None
    if args.synthetic:
        mk = write_synthetic_dataset if args.datatype == "proxd" else (lambda root, n, seed: write_synthetic_humanise(root, (n + 1) // 2, 2, seed=seed))
        train_dir, objs = mk(args.train_data_dir, args.synthetic, seed=1)
        valid_dir, objs_v = mk(args.valid_data_dir, args.synthetic, seed=2)
        text_encoder = synthetic_text_features
    else:
        train_dir, valid_dir, objs, objs_v = args.train_data_dir, args.valid_data_dir, args.objs_dir, args.objs_dir
    if args.datatype == "proxd":
        mkds = lambda d, o: ProxTextDataset(d, objs_data_dir=o or 'data/protext/objs', max_cats=13)
    else:
        mkds = lambda d, o: HumaniseDataset(d, objs_data_dir=o or 'data/humanise/objs', max_cats=11)
    train_loader = DataLoader(mkds(train_dir, objs), batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
    valid_loader = DataLoader(mkds(valid_dir, objs_v), batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
    model, diffusion = create_model_and_diffusion(args.datatype, device=device.index or 0)
    if args.load_ckpt is not None:
        model.load_state_dict(torch.load(args.load_ckpt, map_location="cpu")['model_state_dict'])
    elif args.synthetic:
        model.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=model.max_cats))
    if text_encoder is not None:
        model.text_encoder = text_encoder
    if not args.forward_only:
        raise NotImplementedError("run.train_sdm: parameter updates need the backward pass of the HIP kernels, which is not built "
                                  "(SURVEY.md 8 f-4); use --forward_only to evaluate the training objective and the validation pass")
    for e in range(args.epochs):
        print('Training epoch {}'.format(e))
        total, _ = training_objective(model, diffusion, train_loader, device)
        print('====> Total_train_loss: {:.4f}'.format(total))
        if e % args.save_interval == args.save_interval - 1 or e == args.epochs - 1:
            validate(model, diffusion, valid_loader, device, args.valid_skip_timesteps)
    return 0


if __name__ == '__main__':
    sys.exit(main())
