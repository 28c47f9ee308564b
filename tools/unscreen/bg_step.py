"""bg_step / bg_offline mode on MI355X: the stage-3 loop of the reference's tools/unscreen/bg_offline.py:133-177,
device-resident and optionally frame-sharded over GPUs.  The reference ships with stages 1-2 switched off (`if 0:`) and
reads their products from disk; `--stages 123` runs them here too (default "3" = the reference as shipped):
  1  per frame segmask (HumanSegAgent, or STM while tracking) + one matting pass + per-frame background   bg_offline.py:51-97
  2  overall background: masked mean over all frames, ONE all-reduce over the frame shards, hole filled  bg_offline.py:99-130
  3  bg-difference ensemble mask -> object filter -> trimap -> matting -> foreground                  bg_offline.py:133-177

Same CLI as the reference (--cfg, -vid/--video_id, --range a-b) plus --data_root (the reference hard-codes its
dataset path, bg_offline.py:187).  Inputs, as in the reference: <data_root>/src_img/<vid>/*.jpg frames and, in
<data_root>/test_img/<vid>/, segmask_*.jpg, bg_*.jpg and always_bg.jpg.  Outputs alphamask_%06d.jpg and
fg_%06d.jpg next to them.  Under torchrun every rank takes a contiguous frame range (equivalent to the
reference's --range) and writes its own files; rank 0 muxes the video.
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

import numpy as np
import torch
import torch.distributed as dist

from video_unscreen_amd.bgestimate import OverallBackground
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.dist import broadcast_state, load_and_broadcast, shard_range
from video_unscreen_amd.stm import STMAgent
from video_unscreen_amd.vmatting import VMattingAgent
from video_unscreen_amd.pipeline import BgStepPipeline
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
from video_unscreen_amd.hostio import stream_clip
from video_unscreen_amd.utils.fileio import parallel_read_img, read_img, save_img, save_video
from video_unscreen_amd.utils.frameio import AsyncSaver, FrameReader, FrameWriter


def main(cfg):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    print("Processing ", cfg['data'])
    src_img_dir, dst_img_dir = cfg["data"]["src_img_dir"], cfg["data"]["dst_img_dir"]
    os.makedirs(dst_img_dir, exist_ok=True)

    st = time.time()
    stages = cfg["data"].get("stages", "3")
    ext = cfg["data"].get("io", "jpg")
    frame_paths = sorted(glob(osp.join(src_img_dir, '*.' + ext))) or sorted(glob(osp.join(src_img_dir, cfg["data"]["src_img_tmpl"])))
    first = 0
    if cfg["data"]["range"]:
        a, b = cfg["data"]["range"]
        frame_paths, first = frame_paths[a:b], a
    lo, hi = shard_range(len(frame_paths), rank, world)          # this rank's contiguous range
    # stages 1 and 2 walk the clip in memory like the reference (bg_offline.py:36); stage 3 streams it (below)
    frames = parallel_read_img(frame_paths[lo:hi]) if ("1" in stages or "2" in stages) else None
    h, w = (frames[0] if frames else read_img(frame_paths[min(lo, len(frame_paths) - 1)])).shape[:2]   # (a rank may own no frame)
    print('{} frames. Reading Data Done! {:.2f}s'.format(hi - lo, time.time() - st))
    masks, bgs = [], []

    if "1" in stages:
        # every rank starts its range untracked, exactly like a --range run of the reference
        print("get segmask using video segmentation")
        st = time.time()
        # HumanSegAgent + STMAgent + VMattingAgent, as bg_offline.py:13,44-47; rank 0 reads the checkpoints, the others receive them
        w1 = load_and_broadcast({"schp": cfg["binseg"].get("model_path"), "stm": cfg["stm"].get("model_path"),
                                 "vmatting": cfg["vmatting"].get("model_path")}, dev, world)
        idx = dev.index or 0
        pipe1 = BgOnlinePipeline(cfg, h, w, device=dev,
                                 segagent=HumanSegAgent(**dict(cfg["binseg"], cuda_device=idx, state_dict=w1["schp"])),
                                 stmagent=STMAgent(**dict(cfg["stm"], cuda_device=idx, state_dict=w1["stm"])),
                                 vmatagent=VMattingAgent(**dict(cfg["vmatting"], cuda_device=idx, state_dict=w1["vmatting"])))
        saver = AsyncSaver([(h, w), (h, w, 3)], workers=int(cfg["data"].get("io_workers", 6)))     # the two files of a frame are encoded off this thread
        try:
            for i, f in enumerate(frames):
                fid = first + lo + i
                fd = torch.from_numpy(f).to(dev)
                out = pipe1.step(fd, passes=1)
                # without a foreground the reference appends no background and its stage 3 then refuses the unequal lists
                # (bg_offline.py:68-71,140); here such a frame is its own background
                bg_dev = out["bgimg"] if "bgimg" in out else fd
                saver.save((out["segmask"], bg_dev), (osp.join(dst_img_dir, 'segmask_{:06d}.{}'.format(fid, ext)),
                                                     osp.join(dst_img_dir, 'bg_{:06d}.{}'.format(fid, ext))))
                masks.append(out["segmask"].cpu().numpy()[..., None].repeat(3, 2))
                bgs.append(bg_dev.cpu().numpy())
        finally:
            saver.close()
        del pipe1
        print('stage 1: {} frames in {:.2f}s on rank {}'.format(len(frames), time.time() - st, rank))

    def shard_files(prefix):
        """This rank's intermediate files of one kind, selected BY FRAME ID ('<prefix>_<fid:06d>.<ext>' for the frame ids of its
        range - what stage 1 of this tool writes, whatever --range was), in the run's format or as the reference's JPEGs; a
        directory that holds exactly one file per frame of the (ranged) clip, or of the whole clip, under other numbers is paired
        positionally like the reference does (bg_offline.py:134-141)."""
        exts = list(dict.fromkeys((ext, 'jpg')))
        for e in exts:
            named = [osp.join(dst_img_dir, '{}_{:06d}.{}'.format(prefix, first + lo + i, e)) for i in range(hi - lo)]
            if all(osp.exists(q) for q in named):
                return named
        for e in exts:
            paths = sorted(glob(osp.join(dst_img_dir, '{}_*.{}'.format(prefix, e))))
            if len(paths) == len(frame_paths):
                return paths[lo:hi]
            if paths and len(paths) >= first + len(frame_paths):
                return paths[first:first + len(frame_paths)][lo:hi]
        raise FileNotFoundError("no {}_<frame id>.{{{}}} files for frames {}..{} in {} (run stage 1 first, or with the same --range)".format(
            prefix, ",".join(exts), first + lo, first + hi - 1, dst_img_dir))

    if world > 1:
        dist.barrier()
    if "2" in stages:
        print("calculate overall bg image according to video segmentation and temporal constraints")
        st = time.time()
        if not masks:
            masks = parallel_read_img(shard_files('segmask'))
        acc = OverallBackground(h, w, device=dev)
        for f, m in zip(frames, masks):
            acc.add(torch.from_numpy(f).to(dev), torch.from_numpy(np.ascontiguousarray(m[..., 0])).to(dev))
        acc.reduce()                                   # RCCL all-reduce of one (h*w*4) float32 bucket
        if rank == 0:
            save_img(acc.finalize()["bg"].cpu().numpy(), osp.join(dst_img_dir, 'always_bg.jpg'))
        del acc
        print('stage 2: {:.2f}s on rank {}'.format(time.time() - st, rank))
        if world > 1:
            dist.barrier()
    if "3" not in stages:
        if world > 1:
            dist.destroy_process_group()
        return
    st = time.time()
    # rank 0 owns the checkpoint and the overall background; RCCL broadcast to the others
    sd = None
    if rank == 0:
        sd = torch.load(cfg["vmatting"]["model_path"], map_location="cpu")["state_dict"]
    bg_always = torch.from_numpy(read_img(osp.join(dst_img_dir, 'always_bg.jpg'))) if rank == 0 else \
        torch.empty((h, w, 3), dtype=torch.uint8)
    sd, bg_always = broadcast_state(sd, bg_always, dev, world)
    pipe = BgStepPipeline(cfg, h, w, device=dev, state_dict=sd)
    print('Building Agents Done! {:.2f}s'.format(time.time() - st))

    # Stage 3 streams: decoder processes -> page-locked ring -> uploads -> the pipelined frame loop -> downloads -> encoder
    # processes (utils/frameio.py, hostio.stream_clip).  Nothing is held but the rings; the lossy JPEG round trip of the masks
    # and backgrounds (bg_offline.py:134-141) happens only when those files are JPEGs (--io jpg, the reference's layout).
    mask_paths = shard_files('segmask')
    bg_paths = shard_files('bg')
    groups = list(zip(frame_paths[lo:hi], bg_paths, mask_paths))        # (frame, bg, mask): the upload staging order (hostio.in_layout)
    workers = int(cfg["data"].get("io_workers", 6))
    print("calculate fg image according to bg image")
    st = time.time()                                   # the files -> files figure includes starting the codec processes
    reader = FrameReader(groups, [(h, w, 3), (h, w, 3), (h, w)], workers=workers, depth=16)
    writer = FrameWriter([(h, w), (h, w, 3)], workers=workers, depth=16)

    def out_paths(i):
        fid = first + lo + i
        return (osp.join(dst_img_dir, 'alphamask_{:06d}.{}'.format(fid, ext)), osp.join(dst_img_dir, 'fg_{:06d}.{}'.format(fid, ext)))

    n = stream_clip(pipe, reader, writer, out_paths, bg_always)
    torch.cuda.synchronize()
    writer.close()
    reader.close()
    dt = time.time() - st
    print('{} frames in {:.2f}s on rank {} ({:.1f} frames/s, files -> files, {} codec)'.format(n, dt, rank, n / max(dt, 1e-9), ext))
    if world > 1:
        dist.barrier()
    if rank == 0:
        save_video(cfg, 'fg')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, default='./configs/bg.json')
    parser.add_argument('-vid', '--video_id', type=str, default="1")
    parser.add_argument('--range', type=str, default=None, help="eg. 400-700")
    parser.add_argument('--stages', type=str, default="3", help='which of the stages 1, 2, 3 to run, e.g. "123"')
    parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_DATA_ROOT", "./data/greenmat"))
    parser.add_argument('--io', type=str, default="jpg", choices=["jpg", "npy"],
                        help='file format of the intermediate and output images: jpg = the reference layout (lossy mask round trip), npy = raw arrays, no codec')
    parser.add_argument('--io_workers', type=int, default=6, help='decoder and encoder processes of stage 3')
    args = parser.parse_args()

    cfg = json.load(open(args.cfg))
    cfg["data"] = {"range": [int(i) for i in args.range.split('-')] if args.range else None,
                   "video_id": args.video_id, "stages": args.stages, "io": args.io, "io_workers": args.io_workers,
                   "src_img_dir": osp.join(args.data_root, 'src_img', args.video_id),
                   "src_img_tmpl": '*.jpg',
                   "dst_img_dir": osp.join(args.data_root, 'test_img', args.video_id),
                   "dst_vid_dir": osp.join(args.data_root, 'video')}
    main(cfg)
