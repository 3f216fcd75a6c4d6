"""GPT — parameter tree and entry point of the latent-bottleneck transformer core.

Mirrors the constructor, attribute and state-dict names of reference mebt/modules/gpt.py:198-253
(`blocks.{i}.ln1|ln2|attn.{key,query,value,proj}|mlp.{0,2}`, `ln_f`, `head`) so reference
checkpoints load.  The nn.Modules below only *hold* parameters: no torch op of theirs ever runs.
All compute goes through the C ABI (embed + blocks + head are one `mebt_forward` call, driven by
`mebt_amd.transformer.Net2NetTransformer`); `GPT.forward` on pre-embedded tensors is served by the
same library through the owning model.
"""
import torch
import torch.nn as nn

MODES = ("latent_enc", "latent_self", "latent_dec", "lt2l")


class GPTConfig:
    """hyper-parameter holder (reference gpt.py:79-89)"""

    def __init__(self, vocab_size, block_size, **kwargs):
        self.vocab_size = vocab_size
        self.block_size = block_size
        for k, v in kwargs.items():
            setattr(self, k, v)


class CrossAttention(nn.Module):
    """parameters of reference gpt.py:98-117 (4 x Linear(d,d)); compute lives in the HIP library"""

    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0                      # gpt.py:107
        self.key = nn.Linear(config.n_embd, config.n_embd)
        self.query = nn.Linear(config.n_embd, config.n_embd)
        self.value = nn.Linear(config.n_embd, config.n_embd)
        self.proj = nn.Linear(config.n_embd, config.n_embd)
        self.n_head = config.n_head


class Block(nn.Module):
    """parameters of reference gpt.py:143-157"""

    def __init__(self, config, mode):
        super().__init__()
        self.ln1 = nn.LayerNorm(config.n_embd)
        self.ln2 = nn.LayerNorm(config.n_embd)
        self.attn = CrossAttention(config)
        self.mlp = nn.Sequential(
            nn.Linear(config.n_embd, 4 * config.n_embd),
            nn.GELU(),
            nn.Linear(4 * config.n_embd, config.n_embd),
            nn.Dropout(config.resid_pdrop),
        )
        self.mode = mode


class GPT(nn.Module):
    def __init__(self, vocab_size, block_size, n_layer=12, n_head=8, n_embd=256, embd_pdrop=0., resid_pdrop=0.,
                 attn_pdrop=0., n_unmasked=0, vtokens_pos=False, mode=[]):
        super().__init__()
        config = GPTConfig(vocab_size=vocab_size, block_size=block_size, embd_pdrop=embd_pdrop,
                           resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop, n_layer=n_layer, n_head=n_head,
                           n_embd=n_embd, n_unmasked=n_unmasked, mode=mode)
        if len(config.mode) < n_layer:                                  # gpt.py:208-209 (mutates the list in place)
            config.mode += ['maskgit' for _ in range(n_layer - len(config.mode))]
        assert config.n_layer == len(config.mode)                       # gpt.py:213
        self.blocks = nn.Sequential(*[Block(config, m) for m in config.mode])
        self.ln_f = nn.LayerNorm(config.n_embd)
        self.head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.block_size = config.block_size
        self.apply(self._init_weights)
        self.config = config
        self._owner = None        # set by Net2NetTransformer: the object that owns the native engine

    def get_block_size(self):
        return self.block_size

    @staticmethod
    def _init_weights(module):
        """N(0, 0.02) weights, zero biases, unit LayerNorm (reference gpt.py:225-232)"""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def forward(self, sos_emb, contexts, targets, mask_emb, attn_bias=None, debug=False):
        """GPT.forward(sos_emb[B,NS,d], contexts[B,NC,d], targets[B,NT,d], mask_emb, attn_bias=0.)
        -> (logits[B,NT,V], None)   (reference gpt.py:234-253).  Only attn_bias == 0 exists in the
        reference (transformer.py:281,321)."""
        if self._owner is None:
            raise RuntimeError("GPT must be owned by a Net2NetTransformer: its compute runs in libmebt_hip.so")
        if attn_bias is not None and not (isinstance(attn_bias, (int, float)) and attn_bias == 0):
            raise NotImplementedError("attn_bias other than 0 is dead code in the reference and is not built")
        return self._owner()._gpt_forward_embedded(sos_emb, contexts, targets), None
