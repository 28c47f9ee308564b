"""Turn rocprofv3 CSVs (kernel-trace --stats run + two --pmc runs) into the markdown tables committed under profiles/.

usage: summarize_profiles.py <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> <frames_in_pmc_run> <title>
FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes
of wide (16 B/lane) coalesced streaming reads; narrower access widths are uncalibrated, so both the raw and the x2 value
are listed for the read side and the one that applies is named per kernel."""
import collections
import csv
import re
import sys

stats_f, fetch_f, write_f, frames, title = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
OURS = ("conv_mfma_kernel", "conv_wino_kernel", "splitk_reduce_kernel", "head_conv_kernel", "bgdiff_gate_kernel", "seg_post_fused_kernel", "bgdiff_point_kernel", "gate_kernel", "trimap_small_kernel",
        "trimap_up_kernel", "vmat_pre_kernel", "vmat_post_kernel", "get_fg_kernel", "vmat_post_fg_kernel", "seg_pre_kernel", "seg_up_kernel",
        "seg_warp_argmax_kernel", "upsample_ac_nhwc_kernel", "count_ge_kernel", "morph_tile_kernel", "inrange_hsv_kernel")


def short(name):
    """conv_mfma_kernel<NT, KW, IN_RELU> -> conv_mfma_kernel<NT>: the 32- and the 64-column tile (K groups and the input ReLU
    do not change the traffic pattern)."""
    for o in OURS:
        if o in name:
            if o != "conv_mfma_kernel":
                return o
            m = re.search(r"conv_mfma_kernel<(\d)", name) or re.search(r"conv_mfma_kernelILi(\d)E", name)
            return o + "<%s>" % (m.group(1) if m else "?")
    return None


stats = {}
tot = 0.0
for r in csv.DictReader(open(stats_f)):
    tot += float(r["TotalDurationNs"])
    s = short(r["Name"])
    if s:
        e = stats.setdefault(s, [0, 0.0])
        e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])


def pmc(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        s = short(r["Kernel_Name"])
        if s:
            agg[s][0] += 1; agg[s][1] += float(r["Counter_Value"])
    return agg


fe, wr = pmc(fetch_f), pmc(write_f)
print(f"## {title}\n")
print("| kernel | launches (trace run) | avg duration us | share of GPU time | FETCH_SIZE KiB / launch | x2 (wide-read correction) MB | WRITE_SIZE MB / launch |")
print("|---|---|---|---|---|---|---|")
for k, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    f = fe.get(k, [0, 0.0]); w = wr.get(k, [0, 0.0])
    fk = f[1] / f[0] if f[0] else float("nan")
    wk = w[1] / w[0] if w[0] else float("nan")
    print(f"| {k} | {c} | {t/c/1e3:.2f} | {100*t/tot:.1f}% | {fk:.0f} | {2*fk*1024/1e6:.2f} | {wk*1024/1e6:.2f} |")
conv_f = sum(v[1] for k, v in fe.items() if k.startswith(("conv_mfma", "conv_wino", "splitk", "head_conv")))
conv_w = sum(v[1] for k, v in wr.items() if k.startswith(("conv_mfma", "conv_wino", "splitk", "head_conv")))
heads = fe.get("head_conv_kernel", [0, 0.0])[0]            # one head launch per matting forward
if frames <= 0:
    frames = heads
print(f"\nAll conv-stack launches (conv_mfma + conv_wino + splitk_reduce + head) divided by the {frames} forwards of the PMC run: "
      f"FETCH_SIZE {conv_f/frames*1024/1e6:.1f} MB raw / {2*conv_f/frames*1024/1e6:.1f} MB x2, WRITE_SIZE {conv_w/frames*1024/1e6:.1f} MB per forward.")
import json
json.dump({"forwards": frames, "fetch_raw_bytes_per_forward": conv_f / frames * 1024, "write_bytes_per_forward": conv_w / frames * 1024,
           "note": "conv_mfma + conv_wino + splitk_reduce + head_conv launches; FETCH_SIZE raw (x2 for 16-B/lane streaming reads per MI355X_MICROARCH.md)"},
          open(sys.argv[6], "w")) if len(sys.argv) > 6 else None
