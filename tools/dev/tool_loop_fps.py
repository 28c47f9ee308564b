"""Frame loop of tools/unscreen/green.py and bg.py, files -> files (1080p JPEGs in /dev/shm): the loop time the tool prints (frames decoded
before it; every output file written when it ends; the first frames build the networks) for 40 and 160 frames, and the rate between the two.
python tools/dev/tool_loop_fps.py [script ...]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.utils.fileio import save_img
H, W = 1080, 1920
base = tempfile.mkdtemp(prefix="usc_tool_", dir="/dev/shm")
try:
    for cid, name in ((1, "green"), (2, "bg")):
        clip = SyntheticClip(H, W, 40, config_id=cid, device="cpu")
        frames = [clip.frame(t)["frame"].numpy() for t in range(40)]
        for n in (40, 160):
            d = os.path.join(base, f"{name}{n}", "src_img", "1")
            os.makedirs(d)
            for t in range(n):
                save_img(frames[t % 40] if (t // 40) % 2 == 0 else frames[39 - t % 40], os.path.join(d, f"frame_{t:06d}.jpg"))      # (forwards, backwards, ...: continuous motion)
    for script in sys.argv[1:] or ["tools/unscreen/green.py", "tools/unscreen/bg.py"]:
        name = "green" if "green" in script else "bg"
        cfg = json.load(open(os.path.join(ROOT, "configs", name + ".json")))
        for k in ("binseg", "vmatting", "stm"):
            if k in cfg:
                cfg[k]["model_path"] = "synthetic"
        cfg_path = os.path.join(base, name + ".json")
        json.dump(cfg, open(cfg_path, "w"))
        times = {}
        for n in (40, 160):
            t = time.time()
            r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--cfg", cfg_path, "--data_root", os.path.join(base, f"{name}{n}")],
                               capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
            import re
            m = re.search(r"in ([0-9.]+)s on rank", r.stdout)
            times[n] = float(m.group(1)) if m else float('nan')
            if r.returncode:
                print(r.stdout[-1500:], r.stderr[-3000:])
                sys.exit(1)
        print("%-28s loop of 40 frames %.2f s, of 160 frames %.2f s -> %.1f frames/s between the two" % (script, times[40], times[160], 120 / (times[160] - times[40])), flush=True)
finally:
    shutil.rmtree(base, ignore_errors=True)
