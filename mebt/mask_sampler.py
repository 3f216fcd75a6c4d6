from mebt_amd.mask_sampler import *  # noqa: F401,F403
