from mebt_amd.modules.gpt import *  # noqa: F401,F403
