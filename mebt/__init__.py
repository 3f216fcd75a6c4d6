"""Import-name compatibility layer: the reference's YAML configs and scripts address
`mebt.transformer.Net2NetTransformer`, `mebt.mask_sampler.MaskGen`, `mebt.modules.gpt.GPT`
(configs/*/*.yaml:2,49) — here those names resolve to the MI355X-native implementation in
`mebt_amd`.  Nothing is implemented in this package."""
from mebt_amd.transformer import Net2NetTransformer  # noqa: F401
from mebt_amd.mask_sampler import MaskGen  # noqa: F401
from mebt_amd.vqgan import VQGAN, load_vqgan  # noqa: F401
from mebt_amd.data import TokenData as VideoData  # noqa: F401  (the `vtokens` token-grid contract of reference data.py:236-305)


def load_transformer(ckpt_path, device=None):
    """counterpart of reference mebt/download.py:56-61: load a Lightning-format checkpoint, eval mode"""
    model = Net2NetTransformer.load_from_checkpoint(ckpt_path).eval()
    return model.to(device) if device is not None else model
