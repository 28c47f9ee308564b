"""Which kernel variant every conv layer of the five networks runs (usc_conv_plan_query over the planned descriptors), at the sizes
the bench / the configs use.  Prints per network the count of layers per (kernel, Winograd form, K groups, K halves) and lists the
layers on a given variant:   python tools/dev/plan_dump.py [--list "winograd-lin,KW=1"]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from video_unscreen_amd import _native as nv  # noqa: E402


def arrays(net):
    if hasattr(net, "_segments"):
        return [arr for kind, _, arr in net._segments if kind == "conv"]
    return [net._arr]


def describe(net, name, want):
    cnt = collections.Counter()
    for arr in arrays(net):
        for d, p in zip(arr, nv.conv_plans(arr, len(arr))):
            key = p["plan"] if p["winograd"] else "direct (NT,RG,S,KW) " + p["plan"]
            cnt[key] += 1
            if want and want in key:
                print(f"    {name}: {key}: in {d.IH}x{d.IW}x{d.Cin} -> out {d.OH}x{d.OW}x{d.Cout}, taps {d.ntaps}, batch {d.batch} (plan {d.plan_batch}), workgroups {p['workgroups']}")
    print(f"{name}: " + ", ".join(f"{k}: {v}" for k, v in sorted(cnt.items())))


def main():
    want = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--list" else None
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import HipMattingNet
    for hw in ((544, 960), (256, 256), (1088, 1920)):
        describe(HipMattingNet(synthetic_state_dict(0), *hw), "matting %dx%d" % hw, want)
    from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
    sd = schp_synthetic_state_dict(0)
    for b in (1, 4, 12):
        describe(HipSchpNet(sd, 473, 473, batch=b, plan_batch=b), f"schp 473x473 batch {b}", want)
    from video_unscreen_amd.binseg.deeplab_net import HipDeepLabNet, deeplab_synthetic_state_dict
    sd = deeplab_synthetic_state_dict(0)
    for b in (1, 4):
        describe(HipDeepLabNet(sd, 513, 513, batch=b, plan_batch=b), f"deeplab 513x513 batch {b}", want)
    from video_unscreen_amd.stm.net import stm_synthetic_state_dict
    from video_unscreen_amd.stm.net_hip import HipStmNet
    sd = stm_synthetic_state_dict(0)
    for hw in ((544, 960), (272, 480)):
        net = HipStmNet(sd, *hw, capacity=3)
        for part in ("mem", "qry", "dec"):
            describe(getattr(net, part), "stm %dx%d %s" % (hw + (part,)), want)
    from video_unscreen_amd.iseg.net import iseg_synthetic_state_dict
    from video_unscreen_amd.iseg.net_hip import HipIsegNet
    describe(HipIsegNet(iseg_synthetic_state_dict(0), 544, 960), "iseg 544x960 (flip pair)", want)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
