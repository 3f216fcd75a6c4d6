"""import-name alias of the checkpoint loaders of reference mebt/download.py:50-61 (the Google-Drive download helpers and
the I3D loader of the FVD metric are outside the hot path)"""
from mebt_amd.vqgan import load_vqgan  # noqa: F401
from mebt import load_transformer  # noqa: F401
