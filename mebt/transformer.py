from mebt_amd.transformer import *  # noqa: F401,F403
