"""import-name alias: `mebt.vqgan.VQGAN` (reference mebt/vqgan.py) -> the MI355X-native first stage"""
from mebt_amd.vqgan import *  # noqa: F401,F403
from mebt_amd.vqgan import VQGAN, Encoder, Decoder, ResBlock, SamePadConv3d, SamePadConvTranspose3d, Normalize, load_vqgan  # noqa: F401
