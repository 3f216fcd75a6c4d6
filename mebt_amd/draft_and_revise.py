#!/usr/bin/env python3
"""Draft-and-revise command line — counterpart of reference draft_and_revise_videos.py:64-198 (same flags, same output names).

  python -m mebt_amd.draft_and_revise --gpt_ckpt run.ckpt --exp_name ucf --batch_size 4 --n_sample 8 --total_length 128 --step_size 128 \\
      --np_draft results/ucf/numpy_files_128/ucf101/VID_n_steps32_..._codemap.npy --n_revise 32 --revise_t 1.0 --M 2 --save_codemap

`--np_draft <codemap.npy>` revises the code maps a `python -m mebt_amd.sample --save_codemap` run wrote (the shipped pipelines,
scripts/valid_dnr_*.sh): the draft phase is skipped, `n_draft` / the context temperature are parsed from the file name like the
reference does (:118-133)."""
import argparse
import os

import numpy as np
import torch

from .scripts_common import add_common_args, resolve_checkpoint, load_model, save_video_grid, write_outputs


def build_parser():
    parser = argparse.ArgumentParser()
    parser = add_common_args(parser)
    parser.add_argument('--n_draft', type=int, default=8)
    parser.add_argument('--draft_t', type=float, default=1.0)
    parser.add_argument('--draft_p', type=float, default=None)
    parser.add_argument('--draft_k', type=int, default=None)
    parser.add_argument('--n_revise', type=int, default=8)
    parser.add_argument('--revise_t', type=float, default=1.0)
    parser.add_argument('--revise_p', type=float, default=None)
    parser.add_argument('--revise_k', type=int, default=None)
    parser.add_argument('--M', type=int, default=2)
    parser.add_argument('--np_draft', type=str, default=None)
    parser.set_defaults(total_length=16)
    return parser


def apply_np_draft(args):
    """reference :118-133: a given draft fixes n_draft (parsed from the file name) and disables the draft phase's sampling knobs"""
    postfix = ''
    if args.np_draft is None:
        return None, postfix
    draft = np.load(args.np_draft)
    if 'n_steps' in args.np_draft:
        args.n_draft = int(args.np_draft.split('VID_n_steps')[-1].split('_')[0])
    else:
        args.n_draft = 0
    if 'maskgit_cosine' in args.np_draft:
        ctemp = float(args.np_draft.split('ctemp')[-1].split('_')[0][:3])
        postfix += f'_ctemp{ctemp}'
    args.draft_t, args.draft_p, args.draft_k = 0.0, None, None
    return draft, postfix


def output_names(args, postfix):
    """reference :140-158 (its `_dp{args.draft_p}` pieces are plain strings without the f prefix: reproduced literally)"""
    tag = f'VID_dnr_nd{args.n_draft}_dt{args.draft_t}_nr{args.n_revise}_rt{args.revise_t}_M{args.M}' + postfix
    if args.draft_p is not None:
        tag += '_dp{args.draft_p}'
    if args.draft_k is not None:
        tag += '_dk{args.draft_k}'
    if args.revise_p is not None:
        tag += '_rp{args.revise_p}'
    if args.revise_k is not None:
        tag += '_rk{args.revise_k}'
    tag += f'_run{args.run}'
    return f'{args.save}/videos_{args.total_length}/{args.dataset}/{tag}', f'{args.save}/numpy_files_{args.total_length}/{args.dataset}/{tag}'


def main(argv=None):
    from .config import load_config
    from .sampling import draft_and_revise_sample
    args, unknown = build_parser().parse_known_args(argv)
    config = load_config(args.base, [u for u in unknown if "=" in u])
    resolution = config.data.resolution if ("data" in config and config.data.get("image_folder", False)) else args.resolution
    resolve_checkpoint(args)
    print(args.gpt_ckpt)
    draft, postfix = apply_np_draft(args)
    os.makedirs(args.save, exist_ok=True)
    gpt = load_model(args)
    save_dir, save_np = output_names(args, postfix)
    print('generating and saving video to %s...' % save_dir)
    os.makedirs(save_dir, exist_ok=True)
    all_data, all_code = [], []
    n_row = int(np.sqrt(args.batch_size))
    n_batch = args.n_sample // args.batch_size + min(1, args.n_sample % args.batch_size)       # :165
    with torch.no_grad():
        for sample_id in range(n_batch):
            draft_batch = None if draft is None else draft[sample_id * args.batch_size:(sample_id + 1) * args.batch_size]
            bs = args.batch_size if draft_batch is None else len(draft_batch)
            if bs == 0:
                break
            logs = draft_and_revise_sample(gpt, bs, total_length=args.total_length, step_size=args.step_size, context_size=args.context_size,
                                           n_draft=args.n_draft, draft_t=args.draft_t, draft_k=args.draft_k, draft_p=args.draft_p,
                                           n_revise=args.n_revise, revise_t=args.revise_t, revise_k=args.revise_k, revise_p=args.revise_p,
                                           M=args.M, draft=draft_batch)
            if "samples" in logs:
                if args.save_videos and sample_id < args.save_n:
                    save_video_grid(logs['samples'], os.path.join(save_dir, 'generation_%d.%s' % (sample_id, args.format)), n_row)
                all_data.append(logs['samples'].cpu().numpy())
            all_code.append(logs['code_maps'].cpu().numpy())
            if args.verbose:
                print(f"batch {sample_id + 1}/{n_batch}: code map {tuple(logs['code_maps'].shape)}", flush=True)
    if args.np_draft is not None:                                              # :185-187
        os.makedirs(os.path.dirname(save_np), exist_ok=True)
        with open(save_np + '.txt', 'w') as f:
            f.write(args.np_draft)
    write_outputs(args, save_np, all_data, all_code, resolution)
    return save_np


if __name__ == "__main__":
    main()
