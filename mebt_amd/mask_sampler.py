"""MaskGen — host-side mirror of reference mebt/mask_sampler.py (same constructor, attributes and
method names), restated from scratch.  The index bookkeeping is tiny integer work and stays in
torch; the one device-heavy step (ordering the targets by confidence, `gumbel_top_k` + gathers,
mask_sampler.py:178-246) runs as one HIP kernel (mebt_op_next_mask)."""
import numpy as np
import torch
from torch import nn


def _cosine(t):
    return torch.cos(0.5 * np.pi * t)


def _cosine_plus(t):
    return 0.5 * (1 + torch.cos(np.pi * t))


SCHEDULES = {
    "cosine": _cosine,                                   # mask_sampler.py:35-36
    "cosine_plus": _cosine_plus,                         # :39-40
    "linear": lambda t: 1.0 - t,                         # :44-45
    "quadratic": lambda t: (1.0 - t) ** 2.0,             # :48-49
    "square": lambda t: 1.0 - t ** 2.0,                  # :52-53
    "cube": lambda t: 1.0 - t ** 3.0,                    # :56-57
    "sqrt": lambda t: 1.0 - t ** 0.5,                    # :60-61
    "convex": lambda t: (1.0 - t) ** 3.0,                # :64-65
}
_METHODS = ['iid', 'mlm', 'partial_mlm', 'block', 'ar', 'phase', 'grid', 'frame', 'interpolate',
            'stochastic_phase', 'stochastic_grid', 'fdm', 'softfdm', 'clip']   # :68


class MaskGen(nn.Module):
    def __init__(self, iid=False, schedule='cosine', max_token=256, method=None, shape=(4, 16, 16),
                 t_range=(0., 1.), budget=1024):
        super().__init__()
        if schedule not in SCHEDULES and schedule != 'ar':
            raise ValueError(f'Unsupported schedule: {schedule}')                # :14-15
        self.method = method if method is not None else ('iid' if iid else 'mlm')   # :17-20
        if self.method not in _METHODS:
            raise ValueError(f'Unsupported method: {self.method}')              # :22-23
        self.schedule = schedule
        self.device = None
        self.shape = shape
        self.seq_len = np.prod(shape)
        self.max_token = max_token
        self.dense = True
        self.range = t_range if t_range is not None else (0., 1.)
        self.budget = budget
        self.noise_hook = None      # tests inject Exp(1)/Normal draws: fn(kind, shape) -> tensor

    @property
    def schedule_fn(self):                                                        # :71-73
        return SCHEDULES[self.schedule]

    # ------------------------------------------------------------------------------------------
    def divide_indices(self, indices, t, vid_t, prior_t, debug=False):
        """mask_sampler.py:75-115: split each row's permutation into (context, target)."""
        mask_ratio = self.schedule_fn(torch.as_tensor(t, dtype=torch.float32))
        if self.training or debug:
            max_T = self.shape[0]
            num_pos = int(np.prod(self.shape[1:]))
            prior_t = prior_t / prior_t.sum()
            T = np.random.choice(vid_t, p=prior_t)                               # :88
            if max_T != T:
                start_t = np.random.randint(0, max_T - T + 1)                    # :90
                lo, hi = start_t * num_pos, (start_t + T) * num_pos
                keep = (indices >= lo) & (indices < hi)
                # every row is a permutation, so each keeps exactly T*num_pos entries, in order (:97-98)
                indices = indices[keep].view(indices.shape[0], T * num_pos)
        seq_len = int(np.prod(indices.shape[1:]))                                # :101
        n_masked = int(torch.ceil(mask_ratio * seq_len).to(torch.long))          # :102
        n_ctx = seq_len - n_masked
        budget = self.budget if (self.training or debug) else seq_len            # :105-108
        n_tgt = min(budget, seq_len - n_ctx)                                     # :111
        return indices[:, :n_ctx], indices[:, -n_tgt:], seq_len                  # :113-115

    # ------------------------------------------------------------------------------------------
    def _draw(self, kind, like):
        if self.noise_hook is not None:
            return self.noise_hook(kind, tuple(like.shape)).to(like.device, torch.float32)
        if kind == "randn":
            return torch.randn_like(like)
        return torch.empty_like(like).exponential_()

    def generate_next_mask(self, context_indices, target_indices, score, t, strategy='maskgit',
                           context_temperature=4.5, n_masked_toks=None, debug=False):
        """mask_sampler.py:189-246.  The descending order of (score/sum)/q^ctemp, q~Exp(1), decides
        which targets become context; computed on the GPU by mebt_op_next_mask."""
        from . import _lib
        if score is not None and strategy != 'ar':
            assert target_indices.shape == score.shape                          # :193-194
        B, NC = context_indices.shape
        NT = target_indices.shape[1]
        if strategy == 'ar':                                                      # :239-246
            nc = torch.cat([context_indices, target_indices[:, :1]], 1)
            nt = target_indices[:, 1:]
            return (nc, nt, torch.zeros_like(target_indices[:, :1])) if debug else (nc, nt)
        if strategy not in ('maskgit', 'random', 'mlm', 'bootstrap'):
            raise ValueError(strategy)
        if strategy in ('random', 'bootstrap'):                                   # :206-208
            score = self._draw("randn", score)
            context_temperature = 0.0
        seq_len = NC + NT
        if n_masked_toks is None:                                                 # :212-216
            tt = torch.full((B,), fill_value=float(t)) if isinstance(t, float) else t
            n_masked = int(torch.ceil(self.schedule_fn(tt)[0] * seq_len).to(torch.long))
        else:
            n_masked = int(n_masked_toks[0].long())
        if strategy == 'bootstrap':
            n_masked = NT - 1                                                     # :218-219
        n_ctx = seq_len - n_masked
        if n_ctx <= NC:                                                           # :222-225
            return (context_indices, target_indices, None) if debug else (context_indices, target_indices)
        n_new = n_ctx - NC
        noise = self._draw("exp", score)                                          # :182
        dev = score.device
        ci = context_indices.contiguous()
        ti = target_indices.contiguous()
        sc = score.to(torch.float32).contiguous()
        new_ci = torch.empty(B, NC + n_new, dtype=torch.long, device=dev)
        new_ti = torch.empty(B, NT - n_new, dtype=torch.long, device=dev)
        lib = _lib.load()
        _lib.check(lib.mebt_op_next_mask(_lib.ptr(ci) if NC > 0 else None, _lib.ptr(ti), _lib.ptr(sc), _lib.ptr(noise),
                                         float(context_temperature), n_new, B, NC, NT, _lib.ptr(new_ci),
                                         _lib.ptr(new_ti) if NT - n_new > 0 else None, _lib.cur_stream()))
        if debug:
            return new_ci, new_ti, None
        return new_ci, new_ti

    # ------------------------------------------------------------------------------------------
    def _perms(self, B, N, device):
        if self.noise_hook is not None:
            return torch.stack([self.noise_hook("perm", (N,)) for _ in range(B)]).to(device)
        return torch.stack([torch.randperm(N) for _ in range(B)]).to(device)     # :331 / :351 (CPU draw)

    def create_gibbs_revise_mask(self, context_indices, target_indices, num_unit_gibbs_steps, device):
        """mask_sampler.py:317-336: step i re-predicts chunk i given everything else."""
        B = target_indices.shape[0]
        N = int(np.prod(target_indices.shape[1:]))
        n = num_unit_gibbs_steps
        assert N % n == 0                                                         # :328
        w = N // n
        tgt = torch.gather(target_indices, 1, self._perms(B, N, device))
        ctxs = torch.stack([torch.cat([context_indices, tgt[:, (i + 1) * w:], tgt[:, :i * w]], 1) for i in range(n)])
        tgts = torch.stack([tgt[:, i * w:(i + 1) * w] for i in range(n)])
        return ctxs, tgts

    def create_gibbs_draft_mask(self, context_indices, target_indices, num_unit_gibbs_steps, device):
        """mask_sampler.py:338-356: step i predicts ALL not-yet-context tokens given the first i chunks."""
        B = target_indices.shape[0]
        N = int(np.prod(target_indices.shape[1:]))
        n = num_unit_gibbs_steps
        assert N % n == 0                                                         # :348
        w = N // n
        tgt = torch.gather(target_indices, 1, self._perms(B, N, device))
        ctxs = [torch.cat([context_indices, tgt[:, :i * w]], 1) for i in range(n)]
        tgts = [tgt[:, i * w:] for i in range(n)]
        return ctxs, tgts
