"""What the two sampling command lines share (reference sample_vqgan_transformer_videos.py:160-297 and
draft_and_revise_videos.py:64-198): the common flags, checkpoint resolution, the output naming scheme and the writers."""
import math
import os
from glob import glob

import numpy as np


def add_common_args(parser):
    """flags both reference scripts define with the same meaning (+ the data flags they inherit from VideoData and the two
    Lightning trainer flags their bodies read)"""
    from .data import TokenData
    parser.add_argument('--base', nargs='*', default=[], metavar="base_config.yaml")
    parser = TokenData.add_data_specific_args(parser)
    parser.add_argument('--default_root_dir', type=str, default=None)          # pl.Trainer.add_argparse_args (the only trainer flags the scripts read)
    parser.add_argument('--gpus', default=None)
    parser.add_argument('--gpt_ckpt', type=str, default='')
    parser.add_argument('--exp_name', type=str, default='')
    parser.add_argument('--save', type=str, default='./results/mebt')
    parser.add_argument('--total_length', type=int, default=None)
    parser.add_argument('--context_size', type=int, default=12)
    parser.add_argument('--step_size', type=int, default=16)
    parser.add_argument('--run', type=int, default=0)
    parser.add_argument('--n_sample', type=int, default=2048)
    parser.add_argument('--dataset', type=str, default='mshapes', choices=['ucf101', 'stl', 'taichi', 'mshapes'])
    parser.add_argument('--format', type=str, default='gif', choices=['webp', 'mp4', 'gif', 'avi'])
    parser.add_argument('--save_videos', action='store_true')
    parser.add_argument('--save_n', type=int, default=5)
    parser.add_argument('--save_codemap', action='store_true')
    parser.add_argument('--no_np', action='store_true')
    parser.add_argument('--latest', action='store_true')
    parser.add_argument('-v', '--verbose', action='store_true')
    parser.add_argument('--dtype', default=None, choices=['bf16', 'f32'], help="engine precision (default: MEBT_COMPUTE_DTYPE or bf16)")
    return parser


def resolve_checkpoint(args):
    """--gpt_ckpt, or the experiment's best / latest checkpoint under logs/<exp_name>/ (reference :201-210 / :107-116); sets
    args.save = results/<exp_name>[_latest] exactly like the scripts do"""
    ver = args.exp_name
    args.save = f'results/{ver}'
    if args.gpt_ckpt == '':
        if not args.latest:
            found = glob(f'logs/{ver}/lightning_logs/version_0/checkpoints/best_checkpoint.ckpt')
            if not found:
                raise FileNotFoundError(f"no --gpt_ckpt and no logs/{ver}/lightning_logs/version_0/checkpoints/best_checkpoint.ckpt")
            args.gpt_ckpt = found[0]
        else:
            ckpts = glob(f'logs/{ver}/lightning_logs/version_0/checkpoints/*/loss=*.ckpt')
            iters = [int(ckpt.split('step=')[-1].split('-train')[0]) for ckpt in ckpts]
            if not iters:
                raise FileNotFoundError(f"--latest: no logs/{ver}/lightning_logs/version_0/checkpoints/*/loss=*.ckpt")
            max_iter = max(iters)
            args.gpt_ckpt = glob(f'logs/{ver}/lightning_logs/version_0/checkpoints/*step={max_iter}-train/loss=*.ckpt')[0]
            args.save += '_latest'
    return args.gpt_ckpt


def load_model(args):
    """load_transformer(args.gpt_ckpt, vqgan_ckpt=None).cuda().eval() of the scripts (:218 / :138) through the restricted unpickler"""
    import torch
    from .transformer import Net2NetTransformer
    model = Net2NetTransformer.load_from_checkpoint(args.gpt_ckpt)
    if args.dtype:
        model.compute_dtype = args.dtype
    if not torch.cuda.is_available():
        raise RuntimeError("mebt_amd has no CPU path: the sampling scripts need the MI355X (cuda) device")
    return model.cuda().eval()


def save_video_grid(video, fname, nrow=None, fps=10):
    """reference mebt/utils.py:149-171: [B, C, T, H, W] in [0, 1] -> one animated grid file"""
    from PIL import Image
    b, c, t, h, w = video.shape
    video = (video.permute(0, 2, 3, 4, 1).cpu().numpy() * 255).astype('uint8')
    if nrow is None:
        nrow = math.ceil(math.sqrt(b))
    ncol = math.ceil(b / nrow)
    pad = 1
    grid = np.zeros((t, (pad + h) * nrow + pad, (pad + w) * ncol + pad, c), dtype='uint8')
    for i in range(b):
        r, cc = i // ncol, i % ncol
        grid[:, (pad + h) * r:(pad + h) * r + h, (pad + w) * cc:(pad + w) * cc + w] = video[i]
    frames = [Image.fromarray(f, 'RGB') for f in grid]
    frames[0].save(fname, quality=95, save_all=True, append_images=frames[1:], duration=1000 / fps, loop=0, optimize=False)
    print('saved videos to', fname)


def write_outputs(args, save_np, all_data, all_code, resolution, codemap_limit=None):
    """the tail of both scripts (:275-291 / :180-198): `<save_np>_codemap.npy` (token ids) and `<save_np>.npy` (uint8 videos
    [n, T, H, W, C], a random subset of n_sample) — the video file only when a first stage produced pixel samples"""
    os.makedirs(os.path.dirname(save_np), exist_ok=True)
    if args.save_codemap:
        print('saving code_map numpy file to %s...' % (save_np + '_codemap'))
        code = np.concatenate(all_code, 0)
        np.save(save_np + '_codemap', code if codemap_limit is None else code[:codemap_limit])
    if not args.no_np:
        if not all_data:
            print('no first stage attached (vtokens model): no pixel samples to save, token ids only (--save_codemap)')
            return
        print('saving numpy file to %s...' % save_np)
        data = np.concatenate(all_data, 0)               # (the reference stacks with np.array: the same for equal batches, an error for a short last one)
        data = np.transpose(data.reshape(-1, 3, args.total_length, resolution, resolution), (0, 2, 3, 4, 1))      # B T H W C
        n_total = data.shape[0]
        data = (data * 255).astype(np.uint8)[np.random.permutation(n_total)[:args.n_sample]]
        np.save(save_np, data)
