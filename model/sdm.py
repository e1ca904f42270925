from lsdm_amd.model.sdm import *  # noqa: F401,F403
from lsdm_amd.model.sdm import SceneDiffusionModel  # noqa: F401
