"""Factory with the reference's entry points (util/model_util.py:16-163)."""
from ..diffusion import gaussian_diffusion as gd
from ..diffusion.respace import SpacedDiffusion, space_timesteps
from ..model.sdm import SceneDiffusionModel


def load_model_wo_clip(model, state_dict):
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected) == 0
    assert all(k.startswith('clip_model.') for k in missing)


def get_default_model_proxd():
    """util/model_util.py:26-48 (max_cats falls back to the constructor default 13)."""
    return {
        'seq_len': 256, 'modality': 'text', 'clip_version': 'ViT-B/32', 'clip_dim': 512, 'dropout': 0.1, 'n_layer': 6,
        'n_head': 8, 'f_vert': 64, 'dim_ff': 512, 'd_hid': 256, 'mesh_ds_dir': "data/mesh_ds", 'posa_path': None,
        'latent_dim': 128, 'pcd_dim': 3, 'cond_mask_prob': 1.0, 'device': 0, 'vert_dims': 655, 'obj_cat': 8,
        'data_rep': 'rot6d', 'njoints': 251,
    }


def get_default_model_humanise():
    """util/model_util.py:50-73."""
    args = get_default_model_proxd()
    args['max_cats'] = 11
    return args


def get_default_diffusion():
    """util/model_util.py:76-85."""
    return {"lambda_fc": 0.0, "lambda_rcxyz": 0.0, "lambda_vel": 0.0, "lambda_cat": 0.1, "noise_schedule": "cosine",
            "sigma_small": True}


def create_gaussian_diffusion(args, timestep_respacing='', host_rng=True):
    """util/model_util.py:127-163: 1000 cosine steps, START_X, FIXED_SMALL, MSE, no timestep rescaling.
    ``timestep_respacing`` ('' = all steps, 'ddim50', 'ddim250', ...) is hard-wired to '' in the reference."""
    steps = 1000
    betas = gd.get_named_beta_schedule(args['noise_schedule'], steps, 1.)
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing if timestep_respacing else [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_SMALL if args['sigma_small'] else gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.MSE,
        rescale_timesteps=False,
        lambda_vel=args['lambda_vel'], lambda_rcxyz=args['lambda_rcxyz'], lambda_fc=args['lambda_fc'],
        lambda_cat=args['lambda_cat'], host_rng=host_rng)


def create_model_and_diffusion(datatype, **model_overrides):
    kw = get_default_model_proxd() if datatype == "proxd" else get_default_model_humanise()
    kw.update(model_overrides)
    return SceneDiffusionModel(**kw), create_gaussian_diffusion(get_default_diffusion())
