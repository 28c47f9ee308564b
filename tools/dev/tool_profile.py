"""cProfile of tools/unscreen/bg.py (or green.py) on a 60-frame 1080p dataset in /dev/shm: python tools/dev/tool_profile.py bg|green"""
import cProfile, json, os, pstats, runpy, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.utils.fileio import save_img
name = sys.argv[1] if len(sys.argv) > 1 else "bg"
H, W, N = 1080, 1920, 60
base = tempfile.mkdtemp(prefix="usc_tool_", dir="/dev/shm")
try:
    clip = SyntheticClip(H, W, N, config_id=1 if name == "green" else 2, device="cpu")
    d = os.path.join(base, "src_img", "1")
    os.makedirs(d)
    for t in range(N):
        save_img(clip.frame(t)["frame"].numpy(), os.path.join(d, f"frame_{t:06d}.jpg"))
    cfg = json.load(open(os.path.join(ROOT, "configs", name + ".json")))
    for k in ("binseg", "vmatting", "stm"):
        if k in cfg:
            cfg[k]["model_path"] = "synthetic"
    cfg_path = os.path.join(base, name + ".json")
    json.dump(cfg, open(cfg_path, "w"))
    sys.argv = [os.path.join(ROOT, "tools", "unscreen", name + ".py"), "--cfg", cfg_path, "--data_root", base]
    pr = cProfile.Profile()
    pr.enable()
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_callees("pipeline_bg.py:.*step|pipeline_green.py:.*step"); st.print_stats("agent.py|convnet|net_hip", 25)
finally:
    shutil.rmtree(base, ignore_errors=True)
