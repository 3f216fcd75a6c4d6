#!/usr/bin/env python3
"""Draft sampler command line — counterpart of reference sample_vqgan_transformer_videos.py:160-297 (same flags, same output
names): MaskGIT-style `bidirect_sample` of new clips, or `extrapolate` of the code maps in `--base_np`.

  python -m mebt_amd.sample --gpt_ckpt run.ckpt --exp_name ucf --batch_size 4 --n_sample 8 --total_length 128 --step_size 128 \\
      --vid_n_steps 32 --vid_c_temp 2.0 --bootstrap 64 --top_k 32 --no_phase --save_codemap --dataset ucf101

Loads a Lightning-format checkpoint through the restricted unpickler (mebt_amd/lightning_shim.py), samples on the MI355X engine,
writes `<save_np>_codemap.npy` / `<save_np>.npy` and (with --save_videos) animated grids.  Config YAMLs (`--base`) are only read
for `data.resolution`, like the reference."""
import argparse
import os

import numpy as np
import torch

from .scripts_common import add_common_args, resolve_checkpoint, load_model, save_video_grid, write_outputs


def build_parser():
    parser = argparse.ArgumentParser()
    parser = add_common_args(parser)
    parser.add_argument('--base_np', type=str, default='')
    parser.add_argument('--top_k', type=int, default=None)
    parser.add_argument('--temp', type=float, default=1.0)
    parser.add_argument('--frame_c_temp', type=float, default=4.5)
    parser.add_argument('--vid_c_temp', type=float, default=1.0)
    parser.add_argument('--frame_n_steps', type=int, default=16)
    parser.add_argument('--vid_n_steps', type=int, default=128)
    parser.add_argument('--bootstrap', type=int, default=0)
    parser.add_argument('--top_p', type=float, default=None)
    parser.add_argument('--no_phase', action='store_true')
    parser.add_argument('--schedule', type=str, default='cosine')
    parser.add_argument('--decoding_strategy', type=str, default='maskgit', choices=['maskgit', 'random', 'ar'])
    parser.add_argument('--ctemp_schedule', type=str, default='linear', choices=['linear', 'constant', 'cosine'])
    parser.set_defaults(total_length=32)
    return parser


def output_names(args):
    """reference :221-243"""
    tag = f'VID_n_steps{args.vid_n_steps}'
    if args.top_k is not None:
        tag += f'_k{args.top_k}'
    if args.top_p is not None:
        tag += f'_p{args.top_p}'
    tag += f'_temp{args.temp}_ctemp{args.vid_c_temp}{args.ctemp_schedule}_{args.decoding_strategy}_{args.schedule}'
    if not args.no_phase:
        raise AssertionError("the reference asserts --no_phase (sample_vqgan_transformer_videos.py:235)")
    tag += '_no_phase' + f'_run{args.run}'
    return f'{args.save}/videos_{args.total_length}/{args.dataset}/{tag}', f'{args.save}/numpy_files_{args.total_length}/{args.dataset}/{tag}'


def main(argv=None):
    from .config import load_config
    from .sampling import bidirect_sample, extrapolate
    args, unknown = build_parser().parse_known_args(argv)
    config = load_config(args.base, [u for u in unknown if "=" in u])
    resolution = config.data.resolution if ("data" in config and config.data.get("image_folder", False)) else args.resolution
    resolve_checkpoint(args)
    print(args.gpt_ckpt)
    os.makedirs(args.save, exist_ok=True)
    gpt = load_model(args)
    gpt.mask_sampler.schedule = args.schedule                                   # :219
    save_dir, save_np = output_names(args)
    print('generating and saving video to %s...' % save_dir)
    os.makedirs(save_dir, exist_ok=True)
    all_data, all_code = [], []
    n_row = min(int(np.sqrt(args.batch_size)), 4)
    n_batch = args.n_sample // args.batch_size + 1                              # :249
    kw = dict(total_length=args.total_length, step_size=args.step_size, context_size=args.context_size, temperature=args.temp,
              top_k=args.top_k, top_p=args.top_p, frame_n_steps=args.frame_n_steps, vid_n_steps=args.vid_n_steps,
              frame_c_temp=args.frame_c_temp, vid_c_temp=args.vid_c_temp, no_phase=args.no_phase, ctemp_schedule=args.ctemp_schedule,
              strategy=args.decoding_strategy, bootstrap=args.bootstrap)
    vq_np = np.load(args.base_np) if args.base_np else None
    with torch.no_grad():
        for sample_id in range(n_batch):
            if vq_np is None:
                logs = bidirect_sample(gpt, args.batch_size, **kw)
            else:
                vq_x = torch.as_tensor(vq_np[sample_id * args.batch_size:(sample_id + 1) * args.batch_size]).long().cuda()
                if vq_x.shape[0] == 0:
                    break
                logs = extrapolate(gpt, vq_x, **kw)
            if "samples" in logs:
                if args.save_videos and sample_id < args.save_n:
                    save_video_grid(logs['samples'], os.path.join(save_dir, 'generation_%d.%s' % (sample_id, args.format)), n_row,
                                    fps=10 if vq_np is None else 30)
                all_data.append(logs['samples'].cpu().numpy())
            all_code.append(logs['code_maps'].cpu().numpy())
            if args.verbose:
                print(f"batch {sample_id + 1}/{n_batch}: code map {tuple(logs['code_maps'].shape)}", flush=True)
    write_outputs(args, save_np, all_data, all_code, resolution, codemap_limit=args.n_sample)
    return save_np


if __name__ == "__main__":
    main()
