"""Measure every admissible plan (tile width x grid K split x K groups) of every distinct conv layer shape of the path's
networks on the device (usc_conv_force_plan) and compare with the plan the library's cost model picks.

Run on the GPU box:  python tools/dev/tune_convs.py gpurun_out/tune   -> gpurun_out/tune/tune_log.json + a summary line.
A shape's key is (pixels per image x plan_batch, Cin, Cout, taps, input stride, phases): the quantities the planner sees.
Round-1 result (profiles/r01/conv_plan_tuning.md; first run, before the row groups): the model's plans cost 10.27 ms over 115 shapes, the best measured
plans 9.82 ms (-4.4 %), and wiring the measured plans in did not move bench.py (272.2 -> 272.1 fps), so no table is kept."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net_hip import ConvDesc

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tune"
os.makedirs(out_dir, exist_ok=True)
lib = nv.load()
st = torch.cuda.current_stream().cuda_stream


def nets():
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import HipMattingNet
    from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
    from video_unscreen_amd.binseg.deeplab_net import HipDeepLabNet, deeplab_synthetic_state_dict
    from video_unscreen_amd.stm.net import stm_synthetic_state_dict
    from video_unscreen_amd.stm.net_hip import HipStmNet
    which = os.environ.get("NETS", "unet,schp,deeplab,stm").split(",")
    if "unet" in which:
        n = HipMattingNet(synthetic_state_dict(0), 544, 960)
        yield "unet 544x960", n, list(n._descs)
    if "schp" in which:
        sb = int(os.environ.get("SCHP_B", "4"))
        n = HipSchpNet(schp_synthetic_state_dict(0), 473, 473, batch=sb, plan_batch=sb)
        yield "schp 473x473 b%d" % sb, n, [d for k, a, arr in n._segments if k == "conv" for d in arr]
    if "deeplab" in which:
        n = HipDeepLabNet(deeplab_synthetic_state_dict(0), 513, 513, batch=4, plan_batch=4)
        yield "deeplab 513x513 b4", n, [d for k, a, arr in n._segments if k == "conv" for d in arr]
    if "stm" in which:
        n = HipStmNet(stm_synthetic_state_dict(0), 272, 480, capacity=3)
        descs = []
        for part in (n.mem, n.qry, n._read_for(2), n.dec):
            descs += [d for k, a, arr in part._segments if k == "conv" for d in arr]
        if descs:
            yield "stm 272x480", n, descs


def time_layer(d, reps=12):
    arr = (ConvDesc * 1)(d)
    for _ in range(2):
        if lib.usc_conv_sequence_f32(arr, 1, st) != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.usc_conv_sequence_f32(arr, 1, st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


table, log, counts = {}, [], {}
for name, net, descs in nets():
    seen = 0
    for d in descs:
        if d.kind != 0:
            continue
        nb = max(d.batch, 1)
        nphase = max(d.nphase, 1)
        key = (d.OH * d.OW // nb * max(d.plan_batch, 1), d.Cin, d.Cout, d.ntaps, d.in_stride, nphase)
        counts[(name, key)] = counts.get((name, key), 0) + 1
        if key in table:
            continue
        seen += 1
        nch = (d.ntaps * d.Cin + 31) // 32
        lib.usc_conv_force_plan(0, 0, 0, 0)
        base = time_layer(d)
        best = (base, 0, 0, 0, 0)
        rows = []
        for nt, rg in (((1, 4), (2, 4), (1, 2), (1, 1)) if d.Cout % 64 == 0 else ((1, 4), (1, 2), (1, 1))):
            for s in (1, 2, 3, 4, 6, 8, 12, 16):
                for kw in (1, 2, 4, 8, 16):
                    if (nt == 2 and kw > 2) or (rg == 4 and kw > 4) or (rg < 4 and not 4 <= rg * kw <= 16):
                        continue
                    if s * kw > 1 and nch // (s * kw) < 2:
                        continue
                    lib.usc_conv_force_plan(nt, rg, s, kw)
                    t = time_layer(d)
                    if t is None:
                        continue
                    rows.append((nt, rg, s, kw, round(t, 2)))
                    if t < best[0]:
                        best = (t, nt, rg, s, kw)
        lib.usc_conv_force_plan(0, 0, 0, 0)
        # keep the planner's own choice unless a measured plan wins by > 3 % (timer noise)
        if best[1] and best[0] < 0.97 * base:
            table[key] = best[1:]
        else:
            table[key] = None
        plan = nv.conv_plans((ConvDesc * 1)(d), 1)[0]
        log.append({"net": name, "key": key, "planner_us": round(base, 2), "best_us": round(best[0], 2), "best": best[1:], "all": rows,
                    "planner_plan": plan["plan"], "two_outputs": bool(d.out2), "has_slab": bool(d.slab), "slab_bytes": int(d.slab_bytes),
                    "plane_bytes": int(d.OHf * d.OWf * d.Cout * 4)})
        print(f"{name:20s} {key} planner {base:7.1f} us  best {best[0]:7.1f} us  nt/RG/S/KW={best[1:]}", flush=True)
    print(f"{name}: {seen} new shapes", flush=True)
    del net
    torch.cuda.empty_cache()

for r in log:
    r["count"] = counts.get((r["net"], tuple(r["key"])), 1)
json.dump(log, open(os.path.join(out_dir, "tune_log.json"), "w"))
print("planner total %.1f us, tuned total %.1f us" % (sum(r["planner_us"] for r in log), sum(min(r["planner_us"], r["best_us"]) for r in log)))
