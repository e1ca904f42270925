"""``python -m run.test_sdm`` -- drop-in for the reference's evaluation CLI (run/test_sdm.py).

Keeps the reference's flags (run/test_sdm.py:34-66; the architecture flags are parsed and ignored exactly as
there) and its outputs: ``<output_dir>/results.txt`` ("Chamfer distance for seq ..." per sample, then the five
"Final ..." lines, :211,227-231), ``predictions/<stem>.npy`` and ``guiding_points/<stem>.npy`` (float32 (1024,3)).
Additions (defaults preserve reference behaviour): ``--device``, ``--use_ddim``, ``--timestep_respacing``,
``--synthetic N`` (write and evaluate an N-sample synthetic proxd_test tree with synthetic weights -- no dataset,
checkpoint or CLIP weights are reachable offline), ``--objs_dir``, ``--batch_size B`` (the reference evaluates one sample per
sampling loop, run/test_sdm.py:111 -- 9 clouds per launch; B > 1 runs B samples as one batch of chains, metrics and files still
per sample; results then differ from B = 1 because the RNG stream is consumed per batch and the reference's mask quirks Q2/Q3
depend on B) and ``--order sorted|listdir`` (sample enumeration; ``listdir`` replays the reference's os.listdir order on this
file system, lsdm_amd/data.py).

The sampling backend is the HIP path; ``main(argv, backend_factory=...)`` lets tests inject another one.
"""
import argparse
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

from ..data import (HumaniseDataset, ProxTextDataset, synthetic_text_features, write_synthetic_dataset,
                    write_synthetic_humanise)
from ..metrics import accuracy, calculate_fscore, chamfer_distance, emd


def build_parser():
    p = argparse.ArgumentParser(description="")
    p.add_argument("data_dir", type=str, help="path to POSA_temp dataset dir")
    p.add_argument("--load_model", type=str, default="training/contactformer/model_ckpt/best_model_recon_acc.pt")
    p.add_argument("--posa_path", type=str, default="training/posa/model_ckpt/epoch_0349.pt")
    p.add_argument("--visualize", dest="visualize", action='store_const', const=True, default=False)
    p.add_argument("--scene_dir", type=str, default="data/scenes")
    p.add_argument("--tpose_mesh_dir", type=str, default="data/mesh_ds")
    p.add_argument("--save_video", dest='save_video', action='store_const', const=True, default=False)
    p.add_argument("--output_dir", type=str, default="../test_output")
    p.add_argument("--cam_setting_path", type=str, default="posa/support_files/ScreenCamera_0.json")
    p.add_argument("--single_seq_name", type=str, default="BasementSittingBooth_00142_01")
    p.add_argument("--test_on_train_set", dest='do_train', action='store_const', const=True, default=False)
    p.add_argument("--test_on_valid_set", dest='do_valid', action='store_const', const=True, default=False)
    p.add_argument("--model_name", type=str, default="default_model")
    p.add_argument("--fix_ori", dest='fix_ori', action='store_const', const=True, default=False)
    p.add_argument("--encoder_mode", type=int, default=1)
    p.add_argument("--decoder_mode", type=int, default=1)
    p.add_argument("--n_layer", type=int, default=3)
    p.add_argument("--n_head", type=int, default=4)
    p.add_argument("--jump_step", type=int, default=8)
    p.add_argument("--dim_ff", type=int, default=512)
    p.add_argument("--f_vert", type=int, default=64)
    p.add_argument("--max_frame", type=int, default=256)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--datatype", type=str, default="proxd")
    # additions
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--use_ddim", action="store_true")
    p.add_argument("--timestep_respacing", type=str, default="")
    p.add_argument("--synthetic", type=int, default=0, help="write and evaluate N synthetic samples under data_dir")
    p.add_argument("--objs_dir", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=1, help="samples per sampling loop (reference: 1)")
    p.add_argument("--order", type=str, default="sorted", choices=["sorted", "listdir"])
    return p


class HipBackend:
    """create_model_and_diffusion + p_sample_loop / ddim_sample_loop on the HIP device."""

    def __init__(self, args, text_encoder=None):
        from ..synthetic import make_synthetic_state_dict
        from ..util.model_util import create_gaussian_diffusion, create_model_and_diffusion, get_default_diffusion
        self.device = torch.device(args.device)
        idx = self.device.index or 0
        self.model, _ = create_model_and_diffusion(args.datatype, device=idx)
        self.diffusion = create_gaussian_diffusion(get_default_diffusion(), timestep_respacing=args.timestep_respacing)
        self.model.eval()
        if args.synthetic and not os.path.exists(args.load_model):
            self.model.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=self.model.max_cats))
        else:
            self.model.load_state_dict(torch.load(args.load_model, map_location="cpu")['model_state_dict'])
        if text_encoder is not None:
            self.model.text_encoder = text_encoder
        self.use_ddim = args.use_ddim

    def sample(self, shape, mask, given_objs, given_cats, y):
        fn = self.diffusion.ddim_sample_loop if self.use_ddim else self.diffusion.p_sample_loop
        mv = lambda v: v.to(self.device)
        out = fn(self.model, shape, mv(mask), mv(given_objs), mv(given_cats), y=y, clip_denoised=False, model_kwargs=None,
                 skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        # the sample stays on the device: chamfer / F-score run there (run/test_sdm.py:185-186: pred = sample.float().to(device))
        return out.float(), self.model.saved_cat, self.model.saved_guiding_points


def list_mean(values):
    return sum(values) / len(values)


def main(argv=None, backend_factory=None):
    args = build_parser().parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    torch.manual_seed(args.seed)
    data_dir, objs_dir = args.data_dir, args.objs_dir
    text_encoder = None
    if args.synthetic:
        if args.datatype == 'proxd':
            data_dir, objs_dir = write_synthetic_dataset(args.data_dir, args.synthetic, seed=args.seed)
        else:   # HUMANISE layout: two prompts per scene (BASELINE configs[4])
            data_dir, objs_dir = write_synthetic_humanise(args.data_dir, (args.synthetic + 1) // 2, 2, seed=args.seed)
        text_encoder = synthetic_text_features
    # the reference picks the loader by --datatype (run/test_sdm.py:110-117); each has its own category table
    if args.datatype == 'proxd':
        dataset = ProxTextDataset(data_dir, objs_data_dir=objs_dir or 'data/protext/objs', max_cats=13, order=args.order)
    else:
        dataset = HumaniseDataset(data_dir, objs_data_dir=objs_dir or 'data/humanise/objs', max_cats=11, order=args.order)
    if args.batch_size < 1:
        raise ValueError("--batch_size must be >= 1")
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=0)
    backend = (backend_factory or HipBackend)(args, text_encoder)
    lookup = {}
    for fn in os.listdir(os.path.join(data_dir, 'context')):
        with open(os.path.join(data_dir, 'context', fn)) as fh:
            lookup[fh.readlines()[0].strip()] = fn.split('.')[0]
    chamfer_list, emd_list, f1_list, total_acc, total_topk = [], [], [], [], []
    os.makedirs(os.path.join(args.output_dir, 'predictions'), exist_ok=True)
    os.makedirs(os.path.join(args.output_dir, 'guiding_points'), exist_ok=True)
    with open(os.path.join(args.output_dir, "results.txt"), "w+") as f:
        for mask, given_objs, given_cats, target_obj, target_cat, y in loader:
            with torch.no_grad():
                pred, pred_cat, guiding = backend.sample(list(target_obj.shape), mask, given_objs, given_cats, y)
            target_obj = target_obj.float().to(pred.device)
            pred_cat = pred_cat.squeeze(1).cpu()
            tgt = torch.argmax(target_cat, dim=1)
            for i in range(pred.shape[0]):          # (one iteration in the reference's B = 1 loop)
                pi, ti = pred[i:i + 1], target_obj[i:i + 1]
                chamfer_s = float(chamfer_distance(pi, ti))
                chamfer_list.append(chamfer_s)
                emd_list.append(emd(pi, ti))
                f1_list.append(calculate_fscore(pi.squeeze(0), ti.squeeze(0))[0])
                total_topk.append(accuracy(pred_cat[i:i + 1], tgt[i:i + 1], topk=(3,))[0].item())
                total_acc.append((torch.argmax(pred_cat[i:i + 1], dim=1) == tgt[i:i + 1]).sum().item())
                stem = lookup[y[i]]
                f.write("Chamfer distance for seq {}: {:.4f}".format(stem, chamfer_s) + '\n')
                np.save(os.path.join(args.output_dir, 'predictions', stem + '.npy'), pred[i].cpu().numpy())
                np.save(os.path.join(args.output_dir, 'guiding_points', stem + '.npy'), guiding[i].cpu().numpy())
        f.write("Final Chamfer distance: {:.4f}".format(list_mean(chamfer_list)) + '\n')
        f.write("Final EMD: {:.4f}".format(list_mean(emd_list)) + '\n')
        f.write("Final F1 score: {:.4f}".format(list_mean(f1_list)) + '\n')
        f.write("Category accuracy: {:.4f}".format(list_mean(total_acc)) + '\n')
        f.write("Top 3 accuracy: {:.4f}".format(list_mean(total_topk)) + '\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())
