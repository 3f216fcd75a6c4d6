"""mebt_amd — MI355X-native (gfx950) implementation of the MeBT transformer hot path.

Host side mirrors the reference's interface (mebt.transformer.Net2NetTransformer,
mebt.mask_sampler.MaskGen, mebt.modules.gpt.GPT); all device work goes through the C ABI of
libmebt_hip.so (include/mebt_hip.h).  There is no CPU or eager-PyTorch fallback."""
__version__ = "0.1.0"
