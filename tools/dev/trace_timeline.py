"""Timeline statistics of a rocprofv3 kernel trace of bench.py: how many kernels are in flight over the steady-state third of the
run, kernel time per frame by kernel, and the longest idle gaps.   python tools/dev/trace_timeline.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
# steady state of the first frame loop: from the 50th to the 110th get_fg launch (each loop of bench.py runs warm-up + steps frames)
fg = [x for x in ev if 'get_fg_kernel' in x[2]]
lo, hi = (int(v) for v in (sys.argv[2:4] if len(sys.argv) > 3 else (50, 110)))
t0, t1 = fg[min(lo, len(fg) - 2)][0], fg[min(hi, len(fg) - 1)][0]
win = [x for x in ev if x[0] >= t0 and x[1] <= t1]
frames = max(1, sum(1 for x in win if 'get_fg_kernel' in x[2]))
pts = []
for s, e, _, _ in win:
    pts += [(s, 1), (e, -1)]
pts.sort()
conc, cur, last = collections.Counter(), 0, None
for t, d in pts:
    if last is not None:
        conc[cur] += t - last
    cur += d
    last = t
span = t1 - t0
print(f"window {span / 1e6:.1f} ms = {frames} frames, {span / 1e3 / frames:.1f} us/frame; kernel time {sum(e - s for s, e, _, _ in win) / 1e3 / frames:.1f} us/frame")
print("kernels in flight: " + ", ".join(f"{k}: {100 * conc[k] / span:.1f}%" for k in sorted(conc)))
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n, q in win:
    n = re.sub(r"\(.*", "", re.sub(r"<.*", "", n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")))[:44]
    tot[n] += e - s
    cnt[n] += 1
for n, t in tot.most_common(28):
    print(f"  {n:46s} {cnt[n] / frames:6.2f} calls/frame {t / 1e3 / frames:8.1f} us/frame  avg {t / 1e3 / cnt[n]:7.1f} us")
