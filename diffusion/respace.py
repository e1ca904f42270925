from lsdm_amd.diffusion.respace import SpacedDiffusion, space_timesteps, _WrappedModel  # noqa: F401
