"""tools/replace/replace.py files -> files at 1080p (JPEGs in /dev/shm): wall time of a 120-frame run minus a 24-frame run, / 96.
python tools/dev/replace_tool_fps.py [script ...]"""
import os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.utils.fileio import save_img
H, W = 1080, 1920
base = tempfile.mkdtemp(prefix="usc_rep_", dir="/dev/shm")
try:
    clip, clip2 = SyntheticClip(H, W, 24, config_id=2, device="cpu"), SyntheticClip(H, W, 24, config_id=3, device="cpu")
    frames = []
    for t in range(24):
        f, g = clip.frame(t), clip2.frame(t)
        a = (f["alpha_true"].numpy() * 255).astype(np.uint8)
        frames.append(((f["frame"].numpy() * (a[..., None] / 255.)).astype(np.uint8), np.stack([a] * 3, -1), g["frame"].numpy(),
                       np.stack([(g["alpha_true"].numpy() * 255).astype(np.uint8)] * 3, -1)))
    for n in (24, 120):
        root = os.path.join(base, str(n))
        src, tgt, bgd = (os.path.join(root, *p) for p in (("unscreen_img", "test7"), ("unscreenbg_img", "out7"), ("unscreen_img", "bg")))
        for d in (src, tgt, bgd):
            os.makedirs(d)
        save_img(clip2.bg_always.numpy(), os.path.join(bgd, "test5_case.jpg"))
        for t in range(n):
            fg, m, fr, sm = frames[t % 24]
            save_img(fg, os.path.join(tgt, f"fg_{t:06d}.jpg")); save_img(m, os.path.join(tgt, f"alphamask_{t:06d}.jpg"))
            save_img(fr, os.path.join(src, f"frame_{t:06d}.jpg")); save_img(sm, os.path.join(src, f"alphamask_{t:06d}.jpg"))
    for script in sys.argv[1:] or ["tools/replace/replace.py"]:
        times = {}
        for n in (24, 120):
            t = time.time()
            r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--src", "test7", "--data_root", os.path.join(base, str(n))],
                               capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
            times[n] = time.time() - t
            if r.returncode:
                print(r.stdout[-1500:], r.stderr[-3000:])
                sys.exit(1)
        print("%-28s 24 frames %.2f s, 120 frames %.2f s -> %.1f frames/s between the two" % (script, times[24], times[120], 96 / (times[120] - times[24])), flush=True)
finally:
    shutil.rmtree(base, ignore_errors=True)
