"""``python -m run.train_sdm`` -- drop-in for the reference's training script (run/train_sdm.py) on the HIP path.

The flags of run/train_sdm.py:187-213; ``train()`` (:30-107: ``model.train()``, a fresh ``AdamW(lr)`` per epoch, uniform timestep
sampler, ``diffusion.training_losses`` -> ``(losses["loss"] * weights).mean()`` -> ``backward`` -> ``optimizer.step()``); ``validate()``
(:110-183: ``model.eval()``, p_sample_loop, squared error / chamfer / category accuracy, the reference's report line); checkpoints
``<out_dir>/<experiment>/model_ckpt/epoch_%04d.pt`` + ``best_model_valid_loss.pt`` / ``best_model_cfd.pt`` every ``save_interval``
epochs (:296-330).  The objective and its gradients come from ``lsdm_train_step`` (lsdm_amd/training.py); the optimizer is torch's
AdamW on the device tensors the library reads in place.  The scalars the reference hands to its SummaryWriter (run/train_sdm.py:104,176-178,
292) go to ``<out_dir>/<experiment>/tb_log``: through ``torch.utils.tensorboard`` when the package is importable, as ``scalars.jsonl`` (one
``{"tag", "value", "step"}`` object per line) otherwise -- and are printed like the reference prints them.
Additions: ``--synthetic N``, ``--device``, ``--forward_only`` (report the eval-mode objective of every batch without stepping),
``--valid_skip_timesteps`` (shorter validation chains for smoke runs; 0 = the reference's full T = 1000), ``--batch_size``."""
import argparse
import os
import sys
import time

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
    p.add_argument('--host_rng', action='store_true',
                   help='draw the noise, the FPS starts and the dropout mask on the host in the order a CPU run of the reference draws them (parity with '
                        'the oracle; the 7 M-element dropout mask of a B = 6 batch costs ~65 ms per step).  Default: the device generator, as the '
                        'reference does when it trains on a GPU')
    return p


def training_objective(model, diffusion, loader, device, sampler=None, optimizer=None):
    """One pass of run/train_sdm.py:52-82: -> (mean objective, per-batch list of term dicts).  With ``optimizer`` every batch is a
    training step (zero_grad, backward, step); without, the objective is only evaluated (no graph, no update)."""
    sampler = sampler or create_named_schedule_sampler('uniform', diffusion)
    total, batches = 0.0, []
    for mask, given_objs, given_cats, target_obj, target_cat, y in loader:
        mask, given_objs, given_cats = mask.to(device), given_objs.to(device), given_cats.to(device)
        target_obj, target_cat = target_obj.float().to(device), target_cat.to(device)
        t, weights = sampler.sample(target_obj.shape[0], device)
        if optimizer is not None:
            optimizer.zero_grad()
        with torch.set_grad_enabled(optimizer is not None):
            losses = diffusion.training_losses(model, target_obj, mask, t, given_objs, given_cats, target_cat, y=y)
            loss = (losses["loss"] * weights).mean()
        if optimizer is not None:
            loss.backward()
            optimizer.step()
        total += float(loss)
        batches.append({k: float(v) for k, v in losses.items() if v.dim() == 0})
    return total / max(len(batches), 1), batches


def train(model, diffusion, loader, device, lr):
    """run/train_sdm.py:30-107: one epoch in ``model.train()`` with a fresh AdamW (the reference builds its optimizer inside train())."""
    model.train()
    model.requires_grad_(True)
    params = [p for p in model.parameters() if p.requires_grad]
    # (fused: one launch over all parameters instead of ~25 multi-tensor launches per step -- the same update rule)
    optimizer = torch.optim.AdamW(params, lr=lr, fused=bool(params) and all(p.is_cuda for p in params))
    total, _ = training_objective(model, diffusion, loader, device, optimizer=optimizer)
    print('====> Total_train_loss: {:.4f}'.format(total))
    return total


def validate(model, diffusion, loader, device, skip_timesteps=0):
    """run/train_sdm.py:110-183: -> (recon_loss_semantics, chamfer distance, category accuracy), each a batch mean."""
    tot_sq, tot_cfd, tot_acc, n = 0.0, 0.0, 0.0, 0
    model.eval()
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


class ScalarLog:
    """writer.add_scalar of run/train_sdm.py:292 -- tensorboard's SummaryWriter when importable, a JSON-lines file with the same tags otherwise."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer, self.file = SummaryWriter(log_dir), None
        except Exception:
            self.writer, self.file = None, open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
        else:
            import json
            self.file.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self.file.flush()

    def close(self):
        (self.writer or self.file).close()


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
    diffusion.host_rng = bool(args.host_rng)
    starting_epoch = 0
    if args.load_ckpt is not None:      # resume as run/train_sdm.py:284-289: weights + the epoch counter
        checkpoint = torch.load(args.load_ckpt, map_location="cpu")
        model.load_state_dict(checkpoint['model_state_dict'])
        starting_epoch = int(checkpoint.get('epoch', -1)) + 1
        print('loading stats of epoch {}'.format(starting_epoch))
    elif args.synthetic:
        model.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=model.max_cats))
    if text_encoder is not None:
        model.text_encoder = text_encoder
    save_dir = os.path.join(args.out_dir, args.experiment, "model_ckpt")
    best_valid, best_cfd = float('inf'), float('inf')
    writer = ScalarLog(os.path.join(args.out_dir, args.experiment, "tb_log"))
    for e in range(starting_epoch, args.epochs):
        print('Training epoch {}'.format(e))
        t0 = time.time()
        if args.forward_only:
            model.eval()
            total, _ = training_objective(model, diffusion, train_loader, device)
            print('====> Total_train_loss: {:.4f}'.format(total))
        else:
            total = train(model, diffusion, train_loader, device, args.lr)
        print('training_time = {:.4f}'.format(time.time() - t0))
        writer.add_scalar('total/train_total_loss', total, e)
        if e % args.save_interval == args.save_interval - 1 or (args.forward_only and e == args.epochs - 1):
            valid_loss, cfd, acc = validate(model, diffusion, valid_loader, device, args.valid_skip_timesteps)
            writer.add_scalar('recon_loss_semantics/validate', valid_loss, e)
            writer.add_scalar('total_cfd/validate', cfd, e)
            writer.add_scalar('total_acc/validate', acc, e)
            if args.forward_only:
                continue
            os.makedirs(save_dir, exist_ok=True)
            data = {'epoch': e, 'model_state_dict': model.state_dict(), 'total_train_loss': total, 'total_valid_loss': valid_loss}
            torch.save(data, os.path.join(save_dir, 'epoch_{:04d}.pt'.format(e)))
            if valid_loss < best_valid:
                print("Updated best model due to new lowest valid_loss. Current epoch: {}".format(e))
                best_valid = valid_loss
                torch.save(data, os.path.join(save_dir, 'best_model_valid_loss.pt'))
            if cfd < best_cfd:
                print("Updated best model due to new highest total_cfd. Current epoch: {}".format(e))
                best_cfd = cfd
                torch.save(dict(data, total_cfd=cfd), os.path.join(save_dir, 'best_model_cfd.pt'))
    writer.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
