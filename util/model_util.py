from lsdm_amd.util.model_util import *  # noqa: F401,F403
