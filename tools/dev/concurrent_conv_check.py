"""Do the conv stacks give the same bits when two of them run at the same time on two streams as when each has the GPU to itself?
Matting net (544 x 960) on the main stream, human-parsing net (12 frames per forward) on a second stream; every output is compared on the
device with the one computed alone.   python tools/dev/concurrent_conv_check.py [default|nosplit] [rounds]
nosplit: plans without the K-halves-on-two-workgroups form (usc_conv_force_winograd_form(-1, -1, 1))."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd import _native as nv
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nv.load()
if mode == "nosplit":
    nv.load().usc_conv_force_winograd_form(-1, -1, 1)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
from video_unscreen_amd.synth import SyntheticClip
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
g = torch.Generator(device="cpu"); g.manual_seed(5)
x = torch.rand((544, 960, 8), generator=g); x[..., 7] = 0
net.x8.copy_(x.to(dev))
seg = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
seg.set_plan_batch(12)
clip = SyntheticClip(1080, 1920, 12, config_id=2, device=dev)
frames = [clip.frame(t)["frame"] for t in range(12)]
masks = [torch.empty((1080, 1920), dtype=torch.uint8, device=dev) for _ in range(12)]
ref = net().clone()
torch.cuda.synchronize()
seg.forward_device_batch(frames, outs=masks)
torch.cuda.synchronize()
mref = torch.stack(masks).clone()
main = torch.cuda.current_stream(dev)
s2 = torch.cuda.Stream(dev)


def run(concurrent):
    bad, mbad = [], []
    torch.cuda.synchronize()
    for r in range(rounds):
        if concurrent:
            with torch.cuda.stream(s2):
                seg.forward_device_batch(frames, outs=masks)
                mbad.append((torch.stack(masks) != mref).sum())
        for _ in range(10):
            p = net()
            bad.append((p != ref).sum())
        if not concurrent:
            seg.forward_device_batch(frames, outs=masks)
            mbad.append((torch.stack(masks) != mref).sum())
    torch.cuda.synchronize()
    bad = [int(b) for b in bad]; mbad = [int(b) for b in mbad]
    return sum(1 for b in bad if b), sum(bad), len(bad), sum(1 for b in mbad if b), sum(mbad), len(mbad)

for name, c in (("one stream, in turn", False), ("two streams at once", True), ("one stream, in turn", False), ("two streams at once", True)):
    nb, px, n, mb, mpx, mn = run(c)
    print("%-8s %-20s matting forwards with wrong values: %d of %d (%d values); parsing forwards with wrong mask pixels: %d of %d (%d pixels)" % (mode, name, nb, n, px, mb, mn, mpx), flush=True)
