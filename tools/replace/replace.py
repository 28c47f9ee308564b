"""Person replacement from already-unscreened outputs (reference tools/replace/replace.py:44-86), on the GPU:
per frame  res = u8(fg' * m' + bg * (1 - m'))  with fg', m' = rescale_fg(shift_fg(., dx, dy), 1.2)  -  one launch per frame
(video_unscreen_amd/pipeline_replace.py, usc_replace_compose_u8).

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

import torch
import torch.distributed as dist

from video_unscreen_amd.dist import shard_range
from video_unscreen_amd.pipeline_replace import ReplacePipeline
from video_unscreen_amd.utils.fileio import read_img
from video_unscreen_amd.utils.frameio import AsyncLoader, AsyncSaver


def run(args):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (rehearsal knobs, as in bench.py: ranks may share a device and talk over gloo - tests/test_tools_gpu.py on a one-GPU box)
    backend = os.environ.get("USC_BENCH_BACKEND", "nccl")
    if os.environ.get("USC_BENCH_SHARE_GPU"):
        local = local % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ext = args.io                                            # jpg = the reference's files; npy = raw arrays (no codec: bit-exact runs)
    numframes = len(sorted(glob(osp.join(args.tgt_data_dir, 'fg_*.' + ext))))
    assert numframes > 0
    os.makedirs(args.dst_data_dir, exist_ok=True)
    lo, hi = shard_range(numframes, rank, world)
    tgt = lambda name, fid: osp.join(args.tgt_data_dir, "{}_{:06d}.{}".format(name, fid, ext))       # noqa: E731
    src = lambda name, fid: osp.join(args.src_data_dir, "{}_{:06d}.{}".format(name, fid, ext))       # noqa: E731

    # Sizes come from frame 0 on every rank (not from the rank's own first frame: a rank whose range is empty - more ranks than frames -
    # has none, and must still reach the collectives below).  Every frame of a clip has its first frame's size (fixed ring slots).
    h, w = read_img(tgt("fg", 0)).shape[:2]
    bg_npy = osp.splitext(args.src_bg_image)[0] + ".npy"
    bg_path = bg_npy if ext == "npy" and osp.exists(bg_npy) else args.src_bg_image
    pipe = ReplacePipeline(h, w, torch.from_numpy(read_img(bg_path)), device=dev, scale=1.2, max_frames=max(1, hi - lo))

    # Files are decoded by worker processes into a page-locked ring and go up asynchronously (AsyncLoader), the two output images of a frame
    # are encoded by worker processes (AsyncSaver): the loop's own thread only enqueues (the reference: cv2.imread x 5 and cv2.imwrite x 2
    # per frame on it, ~60 ms of codec work against 0.03 ms of kernel).
    sh, sw = read_img(src("frame", 0)).shape[:2]
    workers = args.io_workers
    if args.dx is None or args.dy is None:                   # comp_dx_dy (replace.py:28-41): moments on the device, one host read
        groups = [(tgt("alphamask", fid), src("alphamask", fid)) for fid in range(lo, hi)]
        if groups:                                           # (an empty range contributes (0, 0, 0) to the all-reduce)
            for i, (dst_mask, src_mask) in enumerate(AsyncLoader(groups, [(h, w, 3), (sh, sw, 3)], dev, workers=workers)):
                pipe.centroid_pair(i, dst_mask, src_mask)
        acc = torch.tensor(pipe.offset_sums(hi - lo), dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(acc)
        dx_mean, dy_mean = float(acc[0] / acc[2]), float(acc[1] / acc[2])
    else:
        dx_mean, dy_mean = args.dx, args.dy
    print("Correspondence mean: ", dx_mean, dy_mean)

    M = pipe.shift_matrix(dx_mean, dy_mean)
    groups = [(tgt("fg", fid), tgt("alphamask", fid), src("frame", fid)) for fid in range(lo, hi)]
    saver = AsyncSaver([(h, w, 3), (h, 2 * w, 3)], workers=workers if groups else 1)
    try:
        for i, (fg, mask, frame) in enumerate(AsyncLoader(groups, [(h, w, 3), (h, w, 3), (sh, sw, 3)], dev, workers=workers) if groups else ()):
            res, cmp_img = pipe.compose(fg, mask, frame, M)
            saver.save((res, cmp_img), (osp.join(args.dst_data_dir, "res_{:06d}.{}".format(lo + i, ext)),
                                        osp.join(args.dst_data_dir, "compare_{:06d}.{}".format(lo + i, ext))))
    finally:
        saver.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--src', type=str, default="test5")
    parser.add_argument('--tgt', type=str, default=None)
    parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_REPLACE_ROOT", "../data/replace/edn"))
    parser.add_argument('--io', type=str, default="jpg", choices=["jpg", "npy"], help='file format of inputs and outputs (npy: raw arrays, no codec)')
    parser.add_argument('--io_workers', type=int, default=8, help='decoder / encoder processes (not in the reference)')
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
