"""Tensor helpers of the reference's `mebt/utils.py` that sit on the hot path's boundary: `shift_dim` (:30-53; the VQGAN's
channel-last round trips and the sampling scripts), `view_range` (:61-78) and `accuracy` (:80-94; `shared_step`'s top-1 /
top-5 — the training path itself uses the fused loss kernel, this is the standalone function callers import)."""
import torch


def shift_dim(x, src_dim=-1, dest_dim=-1, make_contiguous=True):
    """move dimension `src_dim` to position `dest_dim`, the others keep their order: shift_dim(x, 1, -1) is
    (b, c, t, h, w) -> (b, t, h, w, c)"""
    n = x.dim()
    src = src_dim + n if src_dim < 0 else src_dim
    dst = dest_dim + n if dest_dim < 0 else dest_dim
    assert 0 <= src < n and 0 <= dst < n
    y = torch.movedim(x, src, dst)
    return y.contiguous() if make_contiguous else y


def view_range(x, i, j, shape):
    """reshape dimensions [i, j) of x to `shape`: view_range(x[b, thw, c], 1, 2, (t, h, w)) -> [b, t, h, w, c]"""
    n = x.dim()
    i = i + n if i < 0 else i
    j = n if j is None else (j + n if j < 0 else j)
    assert 0 <= i < j <= n
    return x.view(tuple(x.shape[:i]) + tuple(shape) + tuple(x.shape[j:]))


def accuracy(output, target, topk=(1,)):
    """[top-k accuracy in percent for k in topk], each a 1-element tensor: `output` [rows, classes] scores, `target` [rows]"""
    with torch.no_grad():
        rows = target.size(0)
        ranked = output.topk(max(topk), dim=1, largest=True, sorted=True).indices          # [rows, maxk]
        hit = ranked == target.reshape(-1, 1)
        return [hit[:, :k].any(dim=1).float().sum(0, keepdim=True) * (100.0 / rows) for k in topk]
