from lsdm_amd.diffusion.gaussian_diffusion import *  # noqa: F401,F403
from lsdm_amd.diffusion.gaussian_diffusion import GaussianDiffusion, ModelMeanType, ModelVarType, LossType, get_named_beta_schedule, _extract_into_tensor  # noqa: F401
