"""Person replacement from already-unscreened outputs (reference tools/replace/replace.py:44-86), on the GPU:
per frame  res = u8(fg' * m' + bg * (1 - m'))  with fg', m' = rescale_fg(shift_fg(., dx, dy), 1.2).

Same inputs as the reference (fg_%06d.jpg / alphamask_%06d.jpg of the target, frame_%06d.jpg / alphamask of the source,
one background image); paths come from --data_root instead of being hard-coded (replace.py:95-101).  Frames are
independent: under torchrun every rank takes a contiguous range; the mean centroid offset (replace.py:28-41) is one
all-reduce of (sum dx, sum dy, count).
"""
import argparse
import os
import os.path as osp
import sys
from glob import glob

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from video_unscreen_amd import _native as nv
from video_unscreen_amd.dist import shard_range
from video_unscreen_amd.utils.fileio import read_img, save_img


def centroid(mask_dev):
    """get_center (reference utils/imgprocess.py:7-14, as replace.py:32,35 call it on the 3-channel mask): COLOR_RGB2GRAY, integer
    moments on the device (usc_gray_moments_u8), int(m10 / (m00 + 1e-6)), int(m01 / (m00 + 1e-6))."""
    _, mom = nv.gray_moments(mask_dev, rgb_order=True)
    m00, m10, m01 = (float(v) for v in mom.tolist())
    return int(m10 / (m00 + 1e-6)), int(m01 / (m00 + 1e-6))


def run(args):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    numframes = len(sorted(glob(osp.join(args.tgt_data_dir, 'fg_*.jpg'))))
    assert numframes > 0
    os.makedirs(args.dst_data_dir, exist_ok=True)
    lo, hi = shard_range(numframes, rank, world)

    if args.dx is None or args.dy is None:
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        for fid in range(lo, hi):
            dst_mask = torch.from_numpy(read_img(osp.join(args.tgt_data_dir, "alphamask_{:06d}.jpg".format(fid)))).to(dev)
            src_mask = torch.from_numpy(read_img(osp.join(args.src_data_dir, "alphamask_{:06d}.jpg".format(fid)))).to(dev)
            src_mask = nv.resize(src_mask, dst_mask.shape[0], dst_mask.shape[1], 1)        # adaptive_resize (replace.py:34)
            dc, sc = centroid(dst_mask), centroid(src_mask)
            acc += torch.tensor([sc[0] - dc[0], sc[1] - dc[1], 1.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(acc)
        dx_mean, dy_mean = float(acc[0] / acc[2]), float(acc[1] / acc[2])
    else:
        dx_mean, dy_mean = args.dx, args.dy
    print("Correspondence mean: ", dx_mean, dy_mean)

    M = np.float32([[1, 0, dx_mean], [0, 1, dy_mean]]).astype(np.float64)
    bg = None
    for fid in range(lo, hi):
        dst_fg = torch.from_numpy(read_img(osp.join(args.tgt_data_dir, "fg_{:06d}.jpg".format(fid)))).to(dev)
        dst_mask = torch.from_numpy(read_img(osp.join(args.tgt_data_dir, "alphamask_{:06d}.jpg".format(fid)))).to(dev)
        h, w = dst_fg.shape[:2]
        src_image = nv.resize(torch.from_numpy(read_img(osp.join(args.src_data_dir, "frame_{:06d}.jpg".format(fid)))).to(dev), h, w, 1)
        if bg is None:
            bg = nv.resize(torch.from_numpy(read_img(args.src_bg_image)).to(dev), h, w, 1)
        fg_s = nv.rescale_fg(nv.warp_affine(dst_fg, M, h, w), 1.2)
        m_s = nv.rescale_fg(nv.warp_affine(dst_mask, M, h, w), 1.2)
        res = nv.blend(fg_s, m_s, bg)
        save_img(res.cpu().numpy(), osp.join(args.dst_data_dir, "res_{:06d}.jpg".format(fid)))
        save_img(torch.cat((src_image, res), 1).cpu().numpy(), osp.join(args.dst_data_dir, "compare_{:06d}.jpg".format(fid)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--src', type=str, default="test5")
    parser.add_argument('--tgt', type=str, default=None)
    parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_REPLACE_ROOT", "../data/replace/edn"))
    parser.add_argument('--dx', type=float, default=None)
    parser.add_argument('--dy', type=float, default=None)
    args = parser.parse_args()
    if args.tgt is None:
        args.tgt = "out" + args.src[-1]                    # the reference's convention (replace.py:95)
    args.src_data_dir = osp.join(args.data_root, "unscreen_img", args.src)
    args.src_bg_image = osp.join(args.src_data_dir, "../bg/test5_case.jpg")
    args.tgt_data_dir = osp.join(args.data_root, "unscreenbg_img", args.tgt)
    args.dst_data_dir = osp.join(args.data_root, "merge_test_img", "{}_{}".format(args.src, args.tgt))
    run(args)
