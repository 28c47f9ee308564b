"""bg online mode on MI355X: the frame loop of the reference's tools/unscreen/bg.py:22-104 (image / video segmentation ->
matting -> background inpainting -> background-difference refinement -> matting -> foreground), device-resident.

Same CLI as the reference (--cfg, -vid/--video_id, --range a-b) plus --data_root.  Reads <data_root>/src_img/<vid>/*.jpg,
writes segmask_%06d.jpg, bg_%06d.jpg, alphamask_%06d.jpg, fg_%06d.jpg to <data_root>/test_bg_img/<vid>/.  The loop is
sequential (tracking flag, alpha_pre, previous frame), so under torchrun every rank takes a contiguous frame range and
starts it like a new clip - what the reference's own --range does.
"""
import argparse
import json
import os
import os.path as osp
import sys
import time
from glob import glob

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from video_unscreen_amd.dist import shard_range
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
from video_unscreen_amd.utils.fileio import parallel_read_img
from video_unscreen_amd.utils.frameio import AsyncSaver


def main(cfg):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    print("Processing ", cfg['data'])
    src_img_dir, dst_img_dir = cfg["data"]["src_img_dir"], cfg["data"]["dst_img_dir"]
    os.makedirs(dst_img_dir, exist_ok=True)
    st = time.time()
    frame_paths = sorted(glob(osp.join(src_img_dir, cfg["data"]["src_img_tmpl"])))
    first = 0
    if cfg["data"]["range"]:
        a, b = cfg["data"]["range"]
        frame_paths, first = frame_paths[a:b], a
    lo, hi = shard_range(len(frame_paths), rank, world)
    frame_list = parallel_read_img(frame_paths[lo:hi])
    h, w, _ = frame_list[0].shape
    numframes = len(frame_list)
    print('{} frames. Reading Data Done! {:.2f}s'.format(numframes, time.time() - st))
    st = time.time()
    pipe = BgOnlinePipeline(cfg, h, w, device=dev)
    print('Agent built in {:.2f}s'.format(time.time() - st))
    st = time.time()
    tracked = 0
    # the JPEGs of a frame are encoded by worker processes while the loop goes on (utils/frameio.py: AsyncSaver)
    saver = AsyncSaver([(h, w), (h, w, 3), (h, w), (h, w, 3)], workers=int(cfg["data"].get("io_workers", 8)))
    try:
        for i in range(numframes):
            out = pipe.step(torch.from_numpy(frame_list[i]).to(dev))
            tracked += int(out["tracking"])
            fid = first + lo + i
            name = lambda kind: osp.join(dst_img_dir, '{}_{:06d}.jpg'.format(kind, fid))      # noqa: E731
            if "bgimg" in out:
                saver.save((out["segmask"], out["bgimg"], out["alpha"], out["fg"]), (name('segmask'), name('bg'), name('alphamask'), name('fg')))
            else:
                saver.save((out["segmask"], None, None, None), (name('segmask'), None, None, None))
        torch.cuda.synchronize()
    finally:
        saver.close()
    print('{} frames ({} tracked), files written, in {:.2f}s on rank {}'.format(numframes, tracked, time.time() - st, rank))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, default='./configs/bg.json')
    parser.add_argument('-vid', '--video_id', type=str, default="1")
    parser.add_argument('--range', type=str, default=None, help="eg. 400-700")
    parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_DATA_ROOT", "./data/greenmat"))
    parser.add_argument('--io_workers', type=int, default=8, help='JPEG encoder processes (not in the reference)')
    args = parser.parse_args()
    cfg = json.load(open(args.cfg))
    cfg["data"] = {"range": [int(i) for i in args.range.split('-')] if args.range else None,
                   "video_id": args.video_id,
                   "src_img_dir": osp.join(args.data_root, 'src_img', args.video_id),
                   "src_img_tmpl": '*.jpg',
                   "dst_img_dir": osp.join(args.data_root, 'test_bg_img', args.video_id),
                   "dst_vid_dir": osp.join(args.data_root, 'video'), "io_workers": args.io_workers}
    main(cfg)
