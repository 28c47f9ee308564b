"""green mode on MI355X: the frame loop of the reference's tools/unscreen/green.py:31-141 (segmentation -> colour filtering
-> object removal -> matting -> colour correction -> foreground), device-resident.

Same CLI as the reference (--cfg, -vid/--video_id, --range a-b) plus --data_root (the reference hard-codes its dataset
path).  Reads <data_root>/src_img/<vid>/*.jpg, writes alphamask_%06d.jpg, fg_%06d.jpg, bg_%06d.jpg to
<data_root>/test_img/<vid>/ and the fg / alphamask videos.  The loop is sequential in the reference (tracking flag,
alpha_pre, warm-started mixtures), so under torchrun every rank takes a contiguous frame range and starts it like a new
clip - exactly what the reference's own --range does.
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
from video_unscreen_amd.pipeline_green import GreenPipeline
from video_unscreen_amd.utils.fileio import parallel_read_img, save_video
from video_unscreen_amd.utils.frameio import AsyncSaver


def print_statistic(runtime, tracking_count, numframes):
    print(f'{tracking_count} / {numframes} use tracking')
    print('-' * 10 + 'runtime' + '-' * 10)
    for key, value in runtime.items():
        print(f'{key:>16s}: {value / max(numframes, 1):.4f}s')
    print('-' * 27 + '\n')


def main(cfg):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    src_img_dir, dst_img_dir = cfg['data']['src_img_dir'], cfg["data"]["dst_img_dir"]
    os.makedirs(dst_img_dir, exist_ok=True)
    os.makedirs(cfg['data']['dst_vid_dir'], exist_ok=True)

    st = time.time()
    frame_paths = sorted(glob(osp.join(src_img_dir, cfg["data"]["src_img_tmpl"])))
    first = 0
    if cfg["data"]["range"]:
        a, b = cfg["data"]["range"]
        frame_paths, first = frame_paths[a:b], a
    lo, hi = shard_range(len(frame_paths), rank, world)
    frame_list = parallel_read_img(frame_paths[lo:hi])
    numframes = len(frame_list)
    h, w, _ = frame_list[0].shape
    print('{} frames. Reading Data Done! {:.2f}s'.format(numframes, time.time() - st))

    st = time.time()
    pipe = GreenPipeline(cfg, h, w, device=dev, cf_random_state=cfg["data"].get("cf_seed"))
    print(f'Building Agents Done! {time.time() - st:.2f}s')

    runtime = {'frame': 0.0}
    t_loop = time.time()
    # the three JPEGs of a frame are encoded by worker processes while the loop goes on (utils/frameio.py: AsyncSaver)
    saver = AsyncSaver([(h, w, 3), (h, w), (h, w, 3)], workers=int(cfg["data"].get("io_workers", 8)))
    try:
        for i in range(numframes):
            st = time.time()
            out = pipe.step(torch.from_numpy(frame_list[i]).to(dev))
            torch.cuda.synchronize()
            runtime['frame'] += time.time() - st
            fid = first + lo + i
            saver.save((out["fg"], out["alpha"], out["bg"]),
                       (osp.join(dst_img_dir, 'fg_{:06d}.jpg'.format(fid)), osp.join(dst_img_dir, 'alphamask_{:06d}.jpg'.format(fid)),
                        osp.join(dst_img_dir, 'bg_{:06d}.jpg'.format(fid))))
    finally:
        saver.close()
    print('{} frames, files written, in {:.2f}s on rank {}'.format(numframes, time.time() - t_loop, rank))
    print_statistic(runtime, pipe.tracking_count, numframes)
    if world > 1:
        dist.barrier()
    if rank == 0:
        save_video(cfg, 'fg')
        save_video(cfg, 'alphamask')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, default='./configs/green.json')
    parser.add_argument('-vid', '--video_id', type=str, default="1")
    parser.add_argument('--range', type=str, default=None, help="eg. 400-700")
    parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_DATA_ROOT", "./data/greenmat"))
    parser.add_argument('--io_workers', type=int, default=8, help='JPEG encoder processes (not in the reference)')
    parser.add_argument('--cf_seed', type=int, default=None, help='seed of the colour filter\'s EM fits (not in the reference, whose fits are unseeded)')
    args = parser.parse_args()
    cfg = json.load(open(args.cfg))
    cfg["data"] = {"range": [int(i) for i in args.range.split('-')] if args.range else None,
                   "video_id": args.video_id,
                   "src_img_dir": osp.join(args.data_root, 'src_img', args.video_id),
                   "src_img_tmpl": '*.jpg',
                   "dst_img_dir": osp.join(args.data_root, 'test_img', args.video_id),
                   "dst_vid_dir": osp.join(args.data_root, 'video'), "io_workers": args.io_workers, "cf_seed": args.cf_seed}
    main(cfg)
