"""Per-layer table of the matting UNet's conv launches from a rocprofv3 kernel trace (the evidence behind bench.py's
`roofline`): shape, algorithmic GFLOP, average kernel duration, TFLOP/s, fraction of the f32-MFMA peak.

  1. rocprofv3 --kernel-trace -d <dir> -o lt -- python3 tools/dev/layer_table.py run [forwards] > <dir>/layers.json
       launches every layer of `forwards` forwards one after the other on ONE stream (usc_conv_sequence_f32: no graph, no
       branch overlap, nothing else on the device) and prints the layer list (shape, flops, launches per layer) as JSON;
  2. python3 tools/dev/layer_table.py table <dir>/layers.json <kernel_trace.csv> > profiles/rNN/unet_layer_table.md
       orders the trace's conv dispatches by start time, drops the warm-up forwards, folds them modulo the launches of one
       forward and averages per layer.

The sum of the table's averages is the per-forward kernel time that `roofline.ms_per_forward` of bench.py (HIP events
around the same serial launches) must reproduce."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PEAK = 157.3
WARM = 5


def run(forwards):
    import torch
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import HipMattingNet
    net = HipMattingNet(synthetic_state_dict(0), 544, 960)        # launch mode "stream": the product's
    net.x8.normal_()
    torch.cuda.synchronize()
    for _ in range(WARM + forwards):
        net()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(forwards):
        net()
    e1.record()
    e1.synchronize()
    layers = []
    plans = nv.conv_plans(net._arr, len(net._descs))
    for i, (d, p) in enumerate(zip(net._descs, plans)):
        M = d.OH * d.OW
        nph = max(d.nphase, 1)
        layers.append({"i": i, "M": M * nph, "Cin": d.Cin, "Cout": d.Cout, "taps": d.ntaps, "in_stride": d.in_stride,
                       "out_stride": d.out_stride, "kind": d.kind, "gflop": 2.0 * M * d.Cout * d.ntaps * d.Cin * nph / 1e9,
                       "launches": p["launches"], "plan": p["plan"], "workgroups": p["workgroups"]})
    json.dump({"forwards_traced": WARM + 2 * forwards, "warm": WARM, "layers": layers,
               "hip_event_ms_per_forward": e0.elapsed_time(e1) / forwards}, sys.stdout)


def table(layers_json, trace_csv):
    meta = json.load(open(layers_json))
    layers = meta["layers"]
    per_fwd = sum(l["launches"] for l in layers)
    rows = []
    for r in csv.DictReader(open(trace_csv)):
        name = r.get("Kernel_Name") or r.get("Name")
        if any(k in name for k in ("conv_mfma_kernel", "conv_wino_kernel", "splitk_reduce_kernel", "head_conv_kernel")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    assert len(rows) % per_fwd == 0, (len(rows), per_fwd)
    nf = len(rows) // per_fwd
    rows = rows[meta["warm"] * per_fwd:]
    nf -= meta["warm"]
    dur = [0.0] * per_fwd
    for j, (_, d) in enumerate(rows):
        dur[j % per_fwd] += d / nf
    print("# Matting UNet (1 x 8 x 544 x 960, f32): per-layer kernel time, serial single-stream launches\n")
    print(f"rocprofv3 --kernel-trace of `tools/dev/layer_table.py run`: {nf} forwards after {meta['warm']} warm-up forwards, "
          f"{per_fwd} launches per forward, one stream, nothing else on the device.  GFLOP = algorithmic 2·M·N·K of the layer "
          f"(padded channels of the 7-channel input included: +0.04 GFLOP over the reference module's 132.75); peak = {PEAK} TFLOP/s "
          "(f32 MFMA, MI355X_MICROARCH.md).\n")
    print("| # | M (pixels) | Cin | Cout | taps | plan (NT,RG,S,KW) | workgroups | GFLOP | avg us | TFLOP/s | frac of peak |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    k = 0
    tot_us = tot_gf = 0.0
    for l in layers:
        us = sum(dur[k:k + l["launches"]]) / 1e3
        k += l["launches"]
        tot_us += us
        tot_gf += l["gflop"]
        tf = l["gflop"] / us * 1e3 if us > 0 else 0.0
        what = "head (N = 1, HBM-bound)" if l["kind"] == 1 else l["plan"]
        print(f"| {l['i']} | {l['M']} | {l['Cin']} | {l['Cout']} | {l['taps']} | {what} | {l['workgroups']} | {l['gflop']:.3f} | {us:.1f} | {tf:.1f} | {tf / PEAK:.3f} |")
    print(f"\n**Sum over the {len(layers)} layers: {tot_us / 1e3:.3f} ms of kernel time per forward, {tot_gf:.2f} GFLOP executed** "
          f"-> {tot_gf / tot_us * 1e3:.1f} TFLOP/s executed; against the reference module's 132.75 GFLOP: "
          f"**{132.75 / tot_us * 1e3:.1f} TFLOP/s = {132.75 / tot_us * 1e3 / PEAK:.3f} of the f32-MFMA peak**.")
    print(f"\nHIP events around the same serial launches in the same (traced) process: {meta['hip_event_ms_per_forward']:.3f} ms per forward "
          "(kernel time + the gaps between consecutive launches on the stream).")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    else:
        table(sys.argv[2], sys.argv[3])
