"""File I/O helpers with the reference's names (reference unscreen/utils/fileio.py), without cv2 / mmcv:
Pillow reads / writes the JPEGs (BGR order is restored so arrays look exactly like cv2.imread's), a thread pool
replaces multiprocessing.Pool(48).  Codec work is outside the measured hot path."""
import os
import os.path as osp
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime
from glob import glob

from .._imgcodec import read_img, save_img

__all__ = ["read_txt_list", "write_txt_list", "return_date", "return_time", "read_img", "parallel_read_img",
           "save_img", "save_video", "delete_mode"]


def read_txt_list(txt_fn):
    with open(txt_fn, "r") as f:
        return f.read().splitlines()


def write_txt_list(txt_fn, txt_list):
    with open(txt_fn, "w") as f:
        for item in txt_list:
            f.write("{}\n".format(item))


def return_date():
    return datetime.now().strftime('%m%d')


def return_time():
    return datetime.now().strftime('%m%d') + "_" + datetime.now().strftime('%H%M')


def parallel_read_img(framepaths, workers=16):
    with ThreadPoolExecutor(workers) as pool:
        return list(pool.map(read_img, framepaths))


def save_video(cfg, mode, fps=30):
    """frames -> mp4 with the system ffmpeg when there is one (the reference uses mmcv.frames2video)."""
    vid = cfg["data"]["video_id"]
    os.makedirs(cfg['data']['dst_vid_dir'], exist_ok=True)
    out = osp.join(cfg['data']['dst_vid_dir'], '{}_{}.mp4'.format(vid, mode))
    if shutil.which("ffmpeg") is None:
        print("ffmpeg not found: frames of mode '{}' stay in {}".format(mode, cfg['data']['dst_img_dir']))
        return None
    subprocess.check_call(["ffmpeg", "-y", "-loglevel", "error", "-framerate", str(fps), "-i",
                           osp.join(cfg['data']['dst_img_dir'], mode + "_%06d.jpg"), "-pix_fmt", "yuv420p", out])
    print("saving the {} of video {}".format(mode, vid))
    return out


def delete_mode(cfg, mode):
    for p in sorted(glob(osp.join(cfg["data"]["save_img_dir"], '{}_*.jpg'.format(mode)))):
        os.remove(p)
