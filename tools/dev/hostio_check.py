"""Byte comparison of the frame loop over device-resident inputs with the host -> host loop (hostio.HostFrameIO) at the bench's size, several
passes each: which frames / which output differ, and does the device loop repeat itself?   python tools/dev/hostio_check.py [frames] [H W]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd import hostio
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 27
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
dev = torch.device("cuda", 0)
seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
clip = SyntheticClip(H, W, K, config_id=2, device=dev)
clip.gen.manual_seed(20260002)
fr = [clip.frame(t) for t in range(K)]
keys = ("alpha", "fg", "trimap", "gate", "alphaor", "seg")


def dev_run():
    kept = []
    pipe.reset()
    pipe.run(iter([(f["frame"], None, f["bg"], clip.bg_always) for f in fr]), on_result=lambda n, o: kept.append({k: o[k].clone() for k in keys}))
    torch.cuda.synchronize()
    return kept


def diff(a, b, ks=keys):
    out = []
    for n in range(K):
        d = {k: int((a[n][k] != b[n][k]).sum()) for k in ks if k in a[n] and k in b[n]}
        if any(d.values()):
            out.append((n, {k: v for k, v in d.items() if v}))
    return out

def sync_run():
    kept = []
    pipe.reset()
    for f in fr:
        o = pipe.step(f["frame"], None, f["bg"], clip.bg_always)
        kept.append({k: o[k].clone() for k in keys})
    torch.cuda.synchronize()
    return kept

sync = sync_run()
print("sync loop pass 1 vs pass 0:", diff(sync_run(), sync) or "identical", flush=True)
ref = dev_run()
print("pipelined device loop pass 0 vs sync loop:", diff(ref, sync) or "identical", flush=True)
for rep in range(3):
    r = dev_run()
    print("pipelined device loop pass %d vs sync loop:" % (rep + 1), diff(r, sync) or "identical", flush=True)
ref = sync
for rep in range(0):
    print("device loop pass %d vs pass 0:" % (rep + 1), diff(dev_run(), ref) or "identical", flush=True)
groups = [hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), None) for f in fr]
a_h, f_h = hostio.packed_outputs(K, H, W)
for mode in ("hostwait", "own", "hostwait"):
    io = hostio.HostFrameIO(pipe, down_mode=mode)
    for rep in range(3):
        a_h.zero_(); f_h.zero_()
        side = []
        pipe.reset()
        col = io.collect(a_h, f_h)

        def on_result(n, o):
            side.append({k: o[k].clone() for k in keys})
            col(n, o)
        pipe.run(io.inputs(groups, clip.bg_always), on_result=on_result)
        io.drain()
        torch.cuda.synchronize()
        got = [{"alpha": a_h[n].cuda(), "fg": f_h[n].cuda()} for n in range(K)]
        print("host loop (%s) pass %d: downloaded vs device loop:" % (mode, rep), diff(got, ref, ("alpha", "fg")) or "identical",
              "| its device-side tensors vs device loop:", diff(side, ref) or "identical", flush=True)
    io.close()
