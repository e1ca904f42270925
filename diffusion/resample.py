"""Drop-in alias of the reference package layout (diffusion/resample.py)."""
from lsdm_amd.diffusion.resample import UniformSampler, create_named_schedule_sampler  # noqa: F401
