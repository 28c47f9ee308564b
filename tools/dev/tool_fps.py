"""Files -> files rate of the drop-in entry point: writes a synthetic 1080p dataset under a scratch
directory twice - raw .npy (no codec) and .jpg (the reference's layout) - and runs tools/unscreen/bg_step.py --stages 3 on each.
The line the tool prints ("N frames in T s ... frames/s, files -> files") is the figure; it covers decoder processes -> pinned
ring -> uploads -> pipelined loop -> downloads -> encoder processes, not the start-up (weights, agents, first graph).

    python tools/dev/tool_fps.py [--frames 240] [--workers 6,16] [--size 1080x1920]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=480)
    ap.add_argument("--workers", type=str, default="6,12")
    ap.add_argument("--size", type=str, default="1080x1920")
    ap.add_argument("--scratch", type=str, default=None, help="defaults to /dev/shm when it has room (no disk in the figure), else a tmp dir")
    args = ap.parse_args()
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W = (int(v) for v in args.size.split("x"))
    n = args.frames
    base = args.scratch or ("/dev/shm" if shutil.disk_usage("/dev/shm").free > 4 * n * H * W * 8 else None)
    tmp = tempfile.mkdtemp(prefix="usc_tool_fps_", dir=base)
    try:
        clip = SyntheticClip(H, W, n, config_id=2, device="cuda" if torch.cuda.is_available() else "cpu")
        wpath = os.path.join(tmp, "unet.pth")
        torch.save({"state_dict": synthetic_state_dict(0)}, wpath)
        cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
        cfg["vmatting"]["model_path"] = wpath
        cpath = os.path.join(tmp, "bg.json")
        json.dump(cfg, open(cpath, "w"))
        for ext in ("npy", "jpg"):
            root = os.path.join(tmp, ext)
            src, dst = os.path.join(root, "src_img", "1"), os.path.join(root, "test_img", "1")
            os.makedirs(src)
            os.makedirs(dst)
            save_img(clip.bg_always.cpu().numpy(), os.path.join(dst, "always_bg.jpg"))
            t0 = time.time()
            for t in range(n):
                f = {k: v.cpu() for k, v in clip.frame(t).items() if torch.is_tensor(v)}
                if ext == "npy":
                    np.save(os.path.join(src, f"{t:06d}.npy"), f["frame"].numpy())
                    np.save(os.path.join(dst, f"segmask_{t:06d}.npy"), f["seg"].numpy())
                    np.save(os.path.join(dst, f"bg_{t:06d}.npy"), f["bg"].numpy())
                else:
                    save_img(f["frame"].numpy(), os.path.join(src, f"{t:06d}.jpg"))
                    save_img(np.stack([f["seg"].numpy()] * 3, -1), os.path.join(dst, f"segmask_{t:06d}.jpg"))
                    save_img(f["bg"].numpy(), os.path.join(dst, f"bg_{t:06d}.jpg"))
            print(f"[tool_fps] wrote {n} x {H}x{W} {ext} groups under {root} in {time.time() - t0:.1f} s", flush=True)
            for w in (int(v) for v in args.workers.split(",")):
                secs = {}
                for m in (n // 2, n):                     # two clip lengths: the difference is the steady-state rate without start-up
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "bg_step.py"), "--cfg", cpath, "--video_id", "1",
                                        "--data_root", root, "--io", ext, "--io_workers", str(w), "--range", "0-%d" % m],
                                       capture_output=True, text=True, cwd=ROOT)
                    line = [l for l in r.stdout.splitlines() if "files -> files" in l]
                    if not line:
                        print(f"[tool_fps] {ext} io_workers {w}: FAILED rc {r.returncode}\n{(r.stdout + r.stderr)[-2000:]}", flush=True)
                        break
                    secs[m] = float(line[0].split(" frames in ")[1].split("s")[0])
                    print(f"[tool_fps] {ext} io_workers {w}: " + line[0], flush=True)
                if len(secs) == 2 and secs[n] > secs[n // 2]:
                    print(f"[tool_fps] {ext} io_workers {w}: steady state {(n - n // 2) / (secs[n] - secs[n // 2]):.1f} frames/s "
                          f"(start-up of stage 3 = {secs[n // 2] - (n // 2) * (secs[n] - secs[n // 2]) / (n - n // 2):.2f} s)", flush=True)
            shutil.rmtree(root)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
