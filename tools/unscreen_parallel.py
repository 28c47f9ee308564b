"""Loop videos of a list through tools/unscreen.sh (reference tools/unscreen_parallel.py)."""
import argparse
import os
import os.path as osp
import sys

sys.path.append(".")
from video_unscreen_amd.utils.fileio import read_txt_list  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--script', type=str, default="bg_step")
parser.add_argument('--gpuid', type=str, default="0")
parser.add_argument('--range', type=str, default=None, help="eg. 0-40")
parser.add_argument('--data_root', type=str, default=os.environ.get("UNSCREEN_DATA_ROOT", "./data/greenmat"))
args = parser.parse_args()

vid_list = read_txt_list(osp.join(args.data_root, 'meta/vid_list.txt'))
if args.range:
    start, end = args.range.split("-")
    vid_list = vid_list[int(start): int(end)]
for vid in vid_list:
    os.system("bash tools/unscreen.sh {} {} {} --data_root {}".format(args.script, vid, args.gpuid, args.data_root))
print('\nVideos Done: ', len(vid_list))
