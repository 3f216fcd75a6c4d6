"""Import-name compatibility layer: the reference's YAML configs and scripts address
`mebt.transformer.Net2NetTransformer`, `mebt.mask_sampler.MaskGen`, `mebt.modules.gpt.GPT`
(configs/*/*.yaml:2,49) — here those names resolve to the MI355X-native implementation in
`mebt_amd`.  Nothing is implemented in this package."""
from mebt_amd.transformer import Net2NetTransformer  # noqa: F401
from mebt_amd.mask_sampler import MaskGen  # noqa: F401
from mebt_amd.vqgan import VQGAN, load_vqgan  # noqa: F401
from mebt_amd.data import TokenData as VideoData  # noqa: F401  (the `vtokens` token-grid contract of reference data.py:236-305)


def load_transformer(gpt_ckpt, vqgan_ckpt=None, device=None):
    """reference mebt/download.py:56-61, same signature (`load_transformer(args.gpt_ckpt, vqgan_ckpt=None)` at
    draft_and_revise_videos.py:138 / sample_vqgan_transformer_videos.py:218): load a Lightning-format checkpoint, eval mode.
    The reference ignores `vqgan_ckpt` (the first stage comes from the checkpoint's own config) and leaves the model on the CPU;
    `device`, when given, is honoured here."""
    model = Net2NetTransformer.load_from_checkpoint(gpt_ckpt).eval()
    return model.to(device) if device is not None else model
