from mebt_amd.modules.encoders import *  # noqa: F401,F403
