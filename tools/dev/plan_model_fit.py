"""Offline check of the direct conv planner's cost model (usc_conv.hip plan_layer) against measured plans: tools/dev/tune_convs.py times
every admissible (NT, RG, S, KW) of every layer shape; this script replays the model's choice for a set of constants and adds up what
the CHOSEN plans measured (weighted by how often a shape occurs), so constants can be compared without a GPU.

  python tools/dev/plan_model_fit.py gpurun_out/r5_tune/b12/tune_log.json [gpurun_out/r5_tune/b4/tune_log.json ...]"""
import itertools, json, sys

WSLAB_TAIL = 65536


def exists(nt2, rg, kw):
    if nt2:
        return rg == 4 and kw <= 2
    if rg == 4:
        return kw <= 4
    return 4 <= rg * kw <= 16


def choose(r, c):
    m_plan, cin, cout, ntaps, stride, nphase = r["key"]
    nt2 = cout % 64 == 0 and ((m_plan + 127) // 128) * (cout // 64) * nphase >= c["nt2_min"]
    if r.get("two_outputs"):
        pass                                    # (split_c alignment is not in the log: the few two-output layers keep their measured planner plan)
    nblocks = cout // 64 if nt2 else cout // 32
    nch = (ntaps * cin + 31) // 32
    plane_mb = m_plan * cout * 4e-6 * nphase
    slots = c["slots2"] if nt2 else c["slots1"]
    best, pick = 1e30, None
    rg = 4
    while rg >= 1:
        blocks = (m_plan + 32 * rg - 1) // (32 * rg) * nblocks * nphase
        for s in range(1, 17):
            if s > 1 and (not r.get("has_slab", True) or r.get("two_outputs") or s * r.get("plane_bytes", 0) + WSLAB_TAIL > r.get("slab_bytes", 1 << 60)):
                break
            kw = 1
            while kw <= 16:
                if exists(nt2, rg, kw) and not ((s > 1 and nch // (s * kw) < c["min_steps"]) or (kw > 1 and nch // (s * kw) < c["kw_steps"])):
                    n = (nch + s * kw - 1) // (s * kw)
                    wps = rg * kw / 4.0
                    wgs = blocks * s
                    per_cu = (wgs + 255) // 256
                    fit = max(1, int(slots / wps))
                    t, left = 0.0, per_cu
                    while left > 0:
                        W = min(left, fit) * wps
                        t += n * max(c["t_step"] * W, c["t_lone"]) + c["t_fixed"]
                        left -= fit
                    if s > 1:
                        t += c["t_reduce"] + plane_mb * (s + 1.0) / 3.0
                    t += 0.3 * (kw - 1) / (2.0 if kw >= 8 else 1.0) + (0.2 if rg < 4 else 0.0)
                    if t < best - 1e-9:
                        best, pick = t, (2 if nt2 else 1, 4 if nt2 else rg, s, kw)
                kw *= 2
        rg >>= 1
    return pick


def measured(r, pick):
    for nt, rg, s, kw, t in r["all"]:
        if (nt, rg, s, kw) == tuple(pick):
            return t
    return None


def total(logs, c, verbose=False):
    tot = miss = 0
    for r in logs:
        if r["key"][3] == 9 and r["key"][4] == 1 and r["key"][5] == 1:
            continue                                        # 3x3 / stride 1: the Winograd planner's
        if r.get("two_outputs") or "winograd" in str(r.get("planner_plan", "")):
            continue
        pick = choose(r, c)
        t = measured(r, pick)
        if t is None:
            miss += 1
            t = r["planner_us"]
        tot += t * r.get("count", 1)
        if verbose:
            print(r["net"][:14], r["key"], "model", pick, t, "best", r["best"], r["best_us"], "x", r.get("count", 1))
    return tot, miss


if __name__ == "__main__":
    logs = []
    for p in sys.argv[1:]:
        logs += json.load(open(p))
    base = dict(nt2_min=512, min_steps=9, kw_steps=4, slots1=4.0, slots2=3.0, t_step=0.43, t_lone=1.5, t_fixed=3.0, t_reduce=8.0)
    best_possible = sum(min(min(t for *_, t in r["all"]), r["planner_us"]) * r.get("count", 1) for r in logs
                        if not (r["key"][3] == 9 and r["key"][4] == 1 and r["key"][5] == 1) and not r.get("two_outputs") and "winograd" not in str(r.get("planner_plan", "")))
    print("best measured plan per shape: %.1f us" % best_possible)
    print("shipped constants (slots 4 / 3): %.1f us (%d picks not in the table)" % total(logs, base))
    res = []
    for s1, s2, ts, tl, tf, n2, ms, ks in itertools.product((4.0, 5.0), (3.0, 4.0), (0.30, 0.36, 0.43, 0.5), (1.0, 1.5, 2.0), (2.0, 3.0, 4.5, 6.0), (256, 512, 1024, 2048), (6, 9), (2, 4)):
        c = dict(base, slots1=s1, slots2=s2, t_step=ts, t_lone=tl, t_fixed=tf, nt2_min=n2, min_steps=ms, kw_steps=ks)
        res.append((total(logs, c)[0], c))
    res.sort(key=lambda x: x[0])
    for t, c in res[:12]:
        print("%.1f us  " % t, {k: c[k] for k in ("slots1", "slots2", "t_step", "t_lone", "t_fixed", "nt2_min", "min_steps", "kw_steps")})
