"""MFMA pipe utilisation per kernel from a rocprofv3 pass with --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace.

usage: summarize_mfma.py <counter_collection.csv> <kernel_trace.csv> <forwards in the run> <unprofiled ms per forward>"""
import collections
import csv
import re
import sys

cc, kt, forwards, ms_unprofiled = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])


def short(name):
    if "conv_mfma_kernel" in name:
        m = re.search(r"conv_mfma_kernel<(\d+), (\d+), (\d+)", name)
        return "conv_mfma_kernel<NT=%s, RG=%s, KW=%s>" % m.groups() if m else "conv_mfma_kernel"
    if "conv_wino_kernel" in name:
        m = re.search(r"conv_wino_kernel<(\d+)", name)
        return "conv_wino_kernel<KW=%s>" % m.group(1) if m else "conv_wino_kernel"
    for o in ("splitk_reduce_kernel", "head_conv_kernel"):
        if o in name:
            return o
    return None


dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(kt)):
    s = short(r["Kernel_Name"])
    if s:
        dur[s][0] += 1
        dur[s][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
busy = collections.defaultdict(float)
for r in csv.DictReader(open(cc)):
    s = short(r["Kernel_Name"])
    if s and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[s] += float(r["Counter_Value"])
if forwards <= 0:                                 # one head launch per matting forward
    forwards = dur["head_conv_kernel"][0]
print("## MFMA utilisation of the conv stack (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, bench.py --no-seg --sync-loop, %d frames)\n" % forwards)
print("| kernel | launches | avg duration in this pass (us) | SQ_VALU_MFMA_BUSY_CYCLES / launch | MFMA pipe busy = busy / (duration x 2.4 GHz x 1024 SIMDs) |")
print("|---|---|---|---|---|")
tb = tt = 0.0
for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    b = busy.get(k, 0.0)
    tb += b
    tt += t
    print("| %s | %d | %.1f | %.3e | %.1f %% |" % (k, n, t / n / 1e3, b / n, 100 * b / (t * 2.4 * 1024) if t else 0))
nm = tb / 64
print("\nConv stack of one forward: %.3e MFMA-busy SIMD-cycles (= 64 cycles x %.3e v_mfma_f32_32x32x2_f32 = %.1f GFLOP issued incl. tile "
      "padding), %.2f ms of kernel time in this (serialised, counter-collecting) pass -> %.1f %% of the MFMA cycles at 2.4 GHz; against the "
      "%.2f ms of an unprofiled forward: %.1f %%." % (tb / forwards, nm / forwards, nm / forwards * 4096 / 1e9, tt / forwards / 1e6,
                                                     100 * tb / (tt * 2.4 * 1024), ms_unprofiled, 100 * tb / forwards / (ms_unprofiled * 1e6 * 2.4 * 1024)))
