"""Long-video sampling drivers — counterparts of the two functions of reference
sample_vqgan_transformer_videos.py that every shipped pipeline uses to produce the *draft* code map
(scripts/valid_dnr_*.sh): `bidirect_sample` (:22-94, bootstrap + MaskGIT sampling + sliding-window
continuation) and `extrapolate` (:96-157, continuation of given tokens with `edit=True`).

They drive `Net2NetTransformer.sample` exactly like the reference (same argument order, same index
sets); `log['samples']` (the pixel decode through the 3D-VQGAN first stage, mebt_amd/vqgan.py) is
filled when the model carries a `first_stage_model`.
"""
import numpy as np
import torch


def _decode(model, code_map, total_length):
    fs = getattr(model, "first_stage_model", None)
    if fs is None:
        return None
    try:
        img = fs.decode(code_map)
    except RuntimeError:                                   # per-sample fallback on OOM, reference :77-82
        img = torch.cat([fs.decode(code_map[i:i + 1]) for i in range(code_map.shape[0])], 0)
    return (torch.clamp(img, -0.5, 0.5) + 0.5)[:, :, :total_length, :, :]


@torch.no_grad()
def bidirect_sample(model, batch_size, total_length, step_size, context_size, temperature=1.0, top_k=None, top_p=None,
                    frame_n_steps=8, vid_n_steps=8, frame_c_temp=4.5, vid_c_temp=4.5, no_phase=False,
                    ctemp_schedule='linear', strategy='maskgit', bootstrap=0):
    T, H, W = model.mask_sampler.shape[-3:]
    ratio = 0.25                                            # 4 video frames per latent frame, reference :29
    step = int(step_size * ratio)
    ctx = int(context_size * ratio)
    shape = (batch_size, step, H, W)
    dev = model.device
    log = {"class_label": torch.zeros(batch_size, 1, dtype=torch.long, device=dev)}
    x = torch.zeros(shape, dtype=torch.long, device=dev)
    ci = ti = None
    boot_probs = None
    # The reference keeps the [B, N, V] probability maps of debug=True (:41-46) only to read them at the final codes (:86-92).  A
    # position's final code was drawn in the last step that had it as a target, which is also the step that last wrote its map row:
    # the value read is that draw's own score.  So the drivers track the [B, N] chosen-probability map instead (2.1 GB less written
    # per step at block 8192); MEBT_BIDIRECT_FULL_MAP=1 keeps the full maps.
    import os
    chosen = os.environ.get("MEBT_BIDIRECT_FULL_MAP", "0") != "1"
    if bootstrap > 0:                                       # one token revealed per step, reference :41-42
        x, ci, ti, _, _, boot_probs = model.sample(x, None, 1., None, None, bootstrap, ci, ti, context_temperature=vid_c_temp,
                                                   skips=False, ctemp_schedule=ctemp_schedule, strategy='bootstrap', debug=True,
                                                   _chosen_probs=chosen)
    x, ci, _, _, _, final_probs = model.sample(x, None, temperature, top_k, top_p, vid_n_steps, ci, ti,
                                               context_temperature=vid_c_temp, skips=False, ctemp_schedule=ctemp_schedule,
                                               strategy=strategy, debug=True, _chosen_probs=chosen)
    vq = x.reshape(shape)
    code_map = [vq]
    curr_t = step
    while curr_t < total_length * ratio:                    # sliding window: keep the last `ctx` latent frames, reference :55-71
        new_x = torch.zeros(shape, dtype=torch.long, device=dev)
        new_x[:, :ctx] = vq[:, -ctx:]
        ci = torch.arange(H * W * ctx, device=dev).repeat(batch_size, 1)
        ti = torch.arange((step - ctx) * H * W, device=dev).repeat(batch_size, 1) + H * W * ctx
        x = model.sample(new_x, None, temperature, top_k, top_p, vid_n_steps, ci, ti, context_temperature=vid_c_temp,
                         skips=False, ctemp_schedule=ctemp_schedule, strategy=strategy)[0]
        vq = x.reshape(shape)
        code_map.append(vq[:, ctx:])
        curr_t += step - ctx
    code_map = torch.cat(code_map, 1)
    if code_map.shape[1] == 1:
        code_map = code_map.expand(-1, 4, H, W)
    log["code_maps"] = code_map
    samples = _decode(model, code_map, total_length)
    if samples is not None:
        log["samples"] = samples
    prob_map = final_probs if boot_probs is None else torch.where(final_probs < 0., boot_probs, final_probs)
    # log-probability of the chosen codes of the first window (the reference gathers with the whole code
    # map, which only type-checks when there was no continuation — the case of every shipped script)
    if prob_map.dim() == 2:                                 # chosen-probability maps: already the value at the chosen code
        log["score"] = prob_map.log().sum(-1)
    else:
        first = code_map.reshape(batch_size, -1)[:, :prob_map.shape[1]]
        log["score"] = torch.gather(prob_map, -1, first.unsqueeze(-1)).squeeze(-1).log().sum(-1)
    return log


@torch.no_grad()
def extrapolate(model, vq_input, total_length, step_size, context_size, temperature=1.0, top_k=None, top_p=None,
                frame_n_steps=8, vid_n_steps=8, frame_c_temp=4.5, vid_c_temp=4.5, no_phase=False,
                ctemp_schedule='linear', strategy='maskgit', bootstrap=0):
    B, T, H, W = vq_input.shape
    ratio = 0.25
    step = int(step_size * ratio)
    ctx = int(context_size * ratio)
    assert T == step                                        # reference :106
    total = int(total_length * ratio)
    jump = step - ctx
    n_jumps = int(np.ceil((total - step) / jump))
    dev = model.device
    log = {"class_label": torch.zeros(B, 1, dtype=torch.long, device=dev)}
    code_map = [vq_input.clone()]
    idx = torch.arange(H * W * step, device=dev).repeat(B, 1).view(B, step, H, W)
    ci = idx[:, :ctx].reshape(B, -1)
    ti = idx[:, ctx:].reshape(B, -1)
    x = vq_input
    for _ in range(n_jumps):                                # reference :135-145
        nxt = torch.zeros_like(x)
        nxt[:, :ctx] = code_map[-1][:, -ctx:]
        x = model.sample(nxt.view(B, -1), None, temperature, top_k, top_p, vid_n_steps, ci, ti, context_temperature=vid_c_temp,
                         skips=False, edit=True)[0]
        x = x.view(B, step, H, W)
        code_map.append(x[:, ctx:].clone())
    code_map = torch.cat(code_map, 1)
    log["code_maps"] = code_map
    samples = _decode(model, code_map, total_length)
    if samples is not None:
        log["samples"] = samples
    return log


@torch.no_grad()
def draft_and_revise_sample(model, batch_size, total_length, step_size, context_size, n_draft, draft_t, draft_k, draft_p, n_revise,
                            revise_t, revise_k, revise_p, M, draft=None):
    """Counterpart of `sample` in reference draft_and_revise_videos.py:22-60: one draft-and-revise pass over a zero (or given
    draft) code map of `step_size / 4` latent frames, then the first stage's decode.  `draft` (array-like of token ids,
    e.g. the code map a `bidirect_sample` run saved) skips the draft phase, as the shipped scripts do (`--np_draft`)."""
    assert total_length == step_size                           # reference :27
    T, H, W = model.mask_sampler.shape[-3:]
    step = int(step_size * 0.25)
    shape = (batch_size, step, H, W)
    dev = model.device
    skip_draft = draft is not None
    x = torch.as_tensor(draft, dtype=torch.long, device=dev) if skip_draft else torch.zeros(shape, dtype=torch.long, device=dev)
    x = model.draft_and_revise(x, None, n_draft, draft_t, draft_k, draft_p, n_revise, revise_t, revise_k, revise_p, M, skip_draft)
    code_map = x.reshape(shape)
    if code_map.shape[1] == 1:
        code_map = code_map.expand(-1, 4, H, W)
    log = {"class_label": torch.zeros(batch_size, 1, dtype=torch.long, device=dev), "code_maps": code_map}
    samples = _decode(model, code_map, total_length)
    if samples is not None:
        log["samples"] = samples
    return log
