"""import-name alias: `mebt.modules.codebook.Codebook` (reference mebt/modules/codebook.py)"""
from mebt_amd.vqgan import Codebook  # noqa: F401
