#!/usr/bin/env python3
"""bench.py - frames/sec of the bg_step (stage 3) unscreen loop on synthetic 1080p, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--height 1080 --width 1920] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one frame through the whole per-frame path (segmentation -> difference mask -> object filter -> trimap ->
matting pre / network / post -> foreground).  `value` is SURVEY.md section 8d's end-to-end rate: every frame (and its
per-frame background) starts in pinned host memory and its alpha + foreground end in pinned host memory, uploads and
downloads on copy streams beside the compute (video_unscreen_amd/hostio.py); the same loop with inputs resident in HBM
and no download is reported beside it as `device_resident`.  With N ranks the clip is sharded into contiguous frame
ranges (no data-path collective: frames only couple through alpha_pre inside a shard, exactly like the reference's
--range runs); weights and the overall background are broadcast from rank 0 over RCCL before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

REPLACE_PINNED_BUDGET = 12 << 30  # --mode replace: page-locked bytes per rank the clip tensors may take
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
MFMA_F32_PEAK_TFLOPS = 157.3     # f32-input MFMA dense peak (no xf32 on gfx950)
UNET_GFLOP_544x960 = 132.75      # SURVEY.md section 6, torch flop counter on the reference module


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--device-resident", action="store_true",
                    help="time the loop with inputs already in HBM and outputs left there (round 1's figure) as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-single-thread", action="store_true", help="skip the one-frame single-thread leg of the CPU baseline")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--no-interleave", action="store_true", help="skip the opt-in `interleaved_shards` figure (two sub-ranges per matting forward)")
    ap.add_argument("--no-spin-up", action="store_true", help="skip the untimed clock spin-up in front of the warm-up frames (counter-collecting profile runs)")
    ap.add_argument("--sync-loop", action="store_true", help="use the synchronous step() instead of the pipelined run()")
    ap.add_argument("--cpu-frames", type=int, default=5)
    ap.add_argument("--seg-batch", type=int, default=0,
                    help="frames per human-parsing forward in the pipelined loop (bit-identical masks, fewer launches per frame); "
                         "0 = the largest of 12, 8, 6, 4 that divides --steps (a short last group would ride in a full forward), else 8")
    ap.add_argument("--object-filter", default="auto", choices=["auto", "device", "host"],
                    help="remove_invalid_objects: device (= auto) = label-based HIP kernels, host = the border-following C++ twin on "
                         "worker threads inside the pipelined run()")
    ap.add_argument("--no-seg", action="store_true",
                    help="stage-3 loop only (segmentation masks are inputs, as in the reference's bg_offline.py with its "
                         "stages 1-2 disabled); default: the human-parsing net runs on every frame (binseg + trimap + vmatting)")
    ap.add_argument("--mode", default="bg_step", choices=["bg_step", "green", "bg_online", "replace"],
                    help="bg_step (default, BASELINE.json configs[1] / [3]); green (configs[0]: SegAgent / colour filtering loop) and "
                         "bg_online (configs[2]: STM tracking + inpainting loop) time the other frame loops the same way; replace "
                         "(configs[4], the compositing half: tools/replace/replace.py's centroid pass + per-frame composite, host -> host)")
    ap.add_argument("--clip-frames", type=int, default=0,
                    help="strong-scaling form: ONE clip of that many frames, sharded into contiguous frame ranges over the ranks exactly as "
                         "the reference's --range runs cut it (tools/unscreen/bg_offline.py:34-35, 184-192); `steps` becomes the clip "
                         "length, `value` = clip frames / max-over-ranks time, \"scaling\": \"strong\".  BASELINE.json configs[3] = "
                         "--height 2160 --width 3840 --clip-frames 300")
    ap.add_argument("--net", default="hip", choices=["hip", "miopen"],
                    help="conv stack: hand-written f32-MFMA kernels (default) or PyTorch-ROCm / MIOpen")
    return ap.parse_args()


def time_kernel(fn, iters=30, warm=5):
    """Average launch duration in seconds, HIP events on torch's current stream (= the stream the C ABI launches on)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def kernel_table(pipe, sample, H, W):
    """Per-stage achieved GB/s against the algorithmic byte counts of SURVEY.md section 8d."""
    from video_unscreen_amd import _native as nv
    frame, seg, bg_i, bg_always = sample
    N = H * W
    n_lo = int(np.prod(nv.get_target_size(H, W, pipe.long_side, 1)))
    n_in = pipe.in_h * pipe.in_w
    seg_b = seg.numel()
    slot = pipe.slots[0]
    out = pipe.step(frame, seg, bg_i, bg_always)
    alpha, tri = out["alpha"].clone(), out["trimap"].clone()
    gate = slot.gate.clone()
    pred = out["pred"].clone().view(pipe.in_h, pipe.in_w)
    rows = []

    def add(name, fn, nbytes):
        t = time_kernel(fn)
        rows.append({"kernel": name, "us": round(t * 1e6, 2), "alg_bytes": int(nbytes),
                     "GBps": round(nbytes / t / 1e9, 1), "frac_hbm_peak": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)})

    add("bgdiff_mask (bgdiff_gate_kernel)", lambda: nv.bgdiff_mask(frame, bg_i, bg_always, seg, pipe.beta, pipe.thr,
                                                             out_bg=slot.bg, out_alpha=slot.gate, ws=pipe.ws_bgdiff),
        9 * N + seg_b + 3 * N + N)
    add("trimap (small + up)", lambda: nv.trimap(gate, pipe.long_side, pipe.ksize, pipe.iters, out=slot.trimap, ws=pipe.ws_trimap),
        N + n_lo)
    add("vmat_pre", lambda: nv.vmat_pre(frame, alpha, tri, pipe.in_h, pipe.in_w, layout=pipe._layout, out=pipe.x7),
        5 * N + (32 if pipe._layout == nv.NHWC8 else 28) * n_in)
    # the loop's back half runs postprocess + paste + get_fg as ONE launch: pred + trimap in, frame + bg in, alpha + fg out
    add("vmat_post_fg (postprocess + get_fg, one launch)",
        lambda: nv.vmat_post_fg(pred, tri, frame, slot.bg, paste_thr=1, out_u8=alpha, out_fg=out["fg"]), 11 * N + 4 * n_in)
    cnt = torch.empty(1, dtype=torch.int64, device=frame.device)
    add("count_ge", lambda: nv.count_ge(alpha, 128, out=cnt), N)
    t_net = time_kernel(lambda: pipe.net(pipe.x7), iters=20, warm=3)           # the product path: planned stream launches
    t_graph = t_net
    if hasattr(pipe.net, "set_launch_mode"):
        pipe.net.set_launch_mode("graph")
        t_graph = time_kernel(lambda: pipe.net(pipe.x7), iters=20, warm=3)     # the same layers as one hipGraph replay
        pipe.net.set_launch_mode("stream")
    return rows, t_net, t_graph


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


from video_unscreen_amd import usable_cpus  # noqa: E402  (affinity mask capped by the cgroup quota)


def cpu_baseline(cfg, pipe, clip_cpu_frames, bg_always, nframes, threads=None):
    """The CPU oracle (C restatement, 1 thread) + the same networks on torch-CPU with `threads` threads (None = torch's default = all
    cores the process may use), on a bounded sample; every call restarts the alpha_pre chain."""
    import __graft_entry__ as ge
    import oracle
    oracle.build()
    seg_cpu = None
    if pipe.seg_agent is not None:
        from video_unscreen_amd.binseg.schp_net import SchpNetTorch
        seg_cpu = SchpNetTorch(pipe.seg_agent.state_dict)
    pipe.cpu_net()
    saved = torch.get_num_threads()
    if threads is not None:
        torch.set_num_threads(int(threads))
    try:
        alpha_pre = None
        t0 = time.time()
        for f in clip_cpu_frames[:nframes]:
            o = ge.oracle_frame(oracle, cfg, pipe, f["frame"], f["seg"], f["bg"], bg_always, alpha_pre, seg_net_cpu=seg_cpu)
            alpha_pre = o["alpha"]
        dt = time.time() - t0
    finally:
        torch.set_num_threads(saved)
    return nframes / dt, dt


def pmc_traffic(H, W):
    """Fabric bytes of the conv stack per forward from the committed rocprofv3 PMC passes of this workload (newest round first:
    profiles/rNN/pmc_conv_traffic.json for 1920 x 1080, profiles/rNN/pmc_conv_traffic_<W>x<H>.json for any other size; produced by
    tools/dev/run_profiles.sh + summarize_profiles.py): FETCH_SIZE x 2 (gfx950 reports half the bytes of 16-B/lane streaming reads,
    MI355X_MICROARCH.md) + WRITE_SIZE.  Counters cannot be collected inside this process; null when no file matches the frame size."""
    root = os.path.dirname(os.path.abspath(__file__))
    name = "pmc_conv_traffic.json" if (H, W) == (1080, 1920) else f"pmc_conv_traffic_{W}x{H}.json"
    path = next((q for q in (os.path.join(root, "profiles", r, name) for r in ("r05", "r04", "r03", "r02", "r01")) if os.path.exists(q)), None)
    if path is None:
        return None
    z = json.load(open(path))
    return int(2 * z["fetch_raw_bytes_per_forward"] + z["write_bytes_per_forward"])


def other_mode(args, rank, world, dev):
    """green / bg_online: one step = one frame of that mode's (sequential) loop on this rank's own synthetic clip."""
    import copy
    from video_unscreen_amd.synth import SyntheticClip
    H, W, K, Wm = args.height, args.width, args.steps, args.warmup
    root = os.path.dirname(os.path.abspath(__file__))
    if args.mode == "green":
        from video_unscreen_amd.pipeline_green import GreenPipeline
        cfg = json.load(open(os.path.join(root, "configs", "green.json")))
        for k in ("binseg", "vmatting"):
            cfg[k]["model_path"] = "synthetic"
        pipe = GreenPipeline(cfg, H, W, device=dev, cf_random_state=0)
        clip = SyntheticClip(H, W, K + Wm, config_id=1, device=dev)
        what = "green: SegAgent (DeepLabv3+, untracked frames) / tracking + GMM colour filtering + object filter + trimap_withbg + matting UNet-18 + color_correct + get_fg"
    else:
        from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
        cfg = json.load(open(os.path.join(root, "configs", "bg.json")))
        for k in ("binseg", "vmatting", "stm"):
            cfg[k]["model_path"] = "synthetic"
        pipe = BgOnlinePipeline(cfg, H, W, device=dev)
        clip = SyntheticClip(H, W, K + Wm, config_id=2, device=dev)
        what = "bg_online: HumanSegAgent / STM tracking + object filter + trimap + matting UNet-18 + get_bg + Laplace inpainting + bg-difference gate + object filter + trimap + matting UNet-18 + get_fg"
    clip.gen.manual_seed(20260000 + 7919 * rank)
    frames = [clip.frame(t)["frame"] for t in range(K + Wm)]
    for f in frames[:Wm]:
        pipe.step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tracked = 0
    for f in frames[Wm:]:
        tracked += int(pipe.step(f)["tracking"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        roof, extra = mode_roofline(args, pipe, frames[-1])
        print(json.dumps({
            "metric": f"frames/sec end-to-end unscreen ({args.mode}, {W}x{H})", "value": round(world * K / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{what}, {W}x{H}, {K} frames/GPU, one independent clip per GPU", "tracked_frames_rank0": tracked,
                       "weights": "keyed-deterministic synthetic (no checkpoint offline)"},
            "roofline": roof, **extra,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else mode_cpu_baseline(args, cfg, H, W)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def replace_mode(args, rank, world, dev):
    """BASELINE.json configs[4], the loop of the reference's tools/replace/replace.py:28-82: one step = one frame through BOTH passes
    (centroid moments of the two alpha masks; shift_fg -> rescale_fg(1.2) -> float64 blend -> side-by-side image).  Inputs start in
    pinned host memory (target fg + alphamask, source frame + alphamask), the side-by-side image ends there (its right half is res).
    Frames are independent: every rank takes its own contiguous range, the only exchange is the all-reduce of (sum dx, sum dy, n)."""
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.pipeline_replace import ReplaceHostLoop, ReplacePipeline
    from video_unscreen_amd.synth import SyntheticClip
    H, W, Ktot, Wm = args.height, args.width, args.steps, args.warmup
    n = H * W
    # Host and HBM footprints are O(frames of a clip): 18 B/px page-locked (two input tensors, one output) + 6 B/px of HBM per frame.  The
    # clip is therefore capped by a page-locked budget per rank (12 GiB: 345 frames at 1920 x 1080, 86 at 3840 x 2160 - eight ranks then
    # pin 96 GiB, not 360); more steps than that run as several clips over the same buffers, each with its own centroid pass.
    K = max(1, min(Ktot, max(8, REPLACE_PINNED_BUDGET // (18 * n))))
    clips = [K] * (Ktot // K) + ([Ktot % K] if Ktot % K else [])
    tgt = SyntheticClip(H, W, K, config_id=2, device=dev)
    src = SyntheticClip(H, W, K, config_id=3, device=dev, n_objects=1)   # (another silhouette: the mean centroid offset is not zero)
    tgt.gen.manual_seed(20260002 + 7919 * rank)
    src.gen.manual_seed(20260003 + 7919 * rank)
    # one page-locked (K, 2, H, W, 3) tensor per pass: the loop moves eight frames per copy
    masks_h = torch.empty((K, 2, H, W, 3), dtype=torch.uint8).pin_memory()
    inputs_h = torch.empty((K, 2, H, W, 3), dtype=torch.uint8).pin_memory()
    for t in range(K):
        f, g = tgt.frame(t), src.frame(t)
        a = (f["alpha_true"] * 255).to(torch.uint8)
        m3 = a.unsqueeze(-1).expand(H, W, 3).contiguous()
        fg = (f["frame"].float() * (a.float() / 255.).unsqueeze(-1)).to(torch.uint8)
        sm3 = (g["alpha_true"] * 255).to(torch.uint8).unsqueeze(-1).expand(H, W, 3).contiguous()
        masks_h[t, 0].copy_(m3)
        masks_h[t, 1].copy_(sm3)
        inputs_h[t, 0].copy_(fg)
        inputs_h[t, 1].copy_(g["frame"])
    pipe = ReplacePipeline(H, W, src.bg_always, device=dev, scale=1.2, max_frames=K)
    loop = ReplaceHostLoop(pipe, K, ring=3, group=8)           # eight frames (100 MB) per copy: tools/dev/replace_group_probe.py - 2065 frames/s against 1042 at four
    compare_h = torch.empty((K, H, 2 * W, 3), dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()

    def job(k):                                              # both passes over the first k frames; the mean needs every rank's sums
        acc = torch.tensor(loop.pass1(masks_h[:k]), dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(acc)
        dx, dy = float(acc[0] / acc[2]), float(acc[1] / acc[2])
        loop.pass2(inputs_h[:k], pipe.shift_matrix(dx, dy), compare_h)
        return dx, dy

    for _ in range(2):
        job(max(1, min(Wm, K)))                              # W warm-up frames (first launches, page-locked buffers touched), twice
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in reversed(clips):                               # (the full clip last: its offsets and composites are what self_check looks at)
        dx, dy = job(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        M = pipe.shift_matrix(dx, dy)
        # not timed: the loop's output against the stand-alone kernel chain on HBM-resident copies of the same inputs
        bad = []
        for i in sorted({0, 1, K // 2, K - 1}):
            res_c, cmp_c = pipe.compose_chain(inputs_h[i][0].to(dev), masks_h[i][0].to(dev), inputs_h[i][1].to(dev), M)
            if not torch.equal(compare_h[i], cmp_c.cpu()):
                bad.append(i)
        if bad:
            print("bench.py: the replace loop's output differs from the stand-alone kernel chain on frames %s" % bad, file=sys.stderr, flush=True)
            sys.exit(1)
        # the same work with everything resident in HBM (no copies): what the kernels alone sustain
        d_in = [(inputs_h[i][0].to(dev), masks_h[i][0].to(dev), inputs_h[i][1].to(dev), masks_h[i][1].to(dev)) for i in range(min(K, 32))]
        cmp_d = torch.empty((H, 2 * W, 3), dtype=torch.uint8, device=dev)

        def resident():
            for i, (fg, m, fr, sm) in enumerate(d_in):
                pipe.centroid_pair(i, m, sm)
                pipe.compose(fg, m, fr, M, compare=cmp_d, want_res=False)
        t_res = _net_time(resident, iters=5, warm=2) / len(d_in)
        fg, m, fr, sm = d_in[0]
        t_k = time_kernel(lambda: pipe.compose(fg, m, fr, M, compare=cmp_d, want_res=False), iters=50)
        t_m = time_kernel(lambda: pipe.centroid_pair(0, m, sm), iters=50)
        t_chain = time_kernel(lambda: pipe.compose_chain(fg, m, fr, M), iters=20)
        alg = 12 * n + 6 * n                                  # read fg, mask, bg, source frame (3 N each); write [source | res] (6 N)
        roof = {"bound": "hbm", "kernel": "replace_compose_kernel (shift_fg + rescale_fg x2, float64 blend, side-by-side image: one launch per frame)",
                "achieved": round(alg / t_k / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / t_k / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": None, "alg_bytes_per_launch": alg, "us_per_launch": round(t_k * 1e6, 2),
                "stand_alone_chain_us": round(t_chain * 1e6, 2)}
        rows = [{"kernel": "replace_compose_kernel", "us": round(t_k * 1e6, 2), "alg_bytes": alg, "GBps": round(alg / t_k / 1e9, 1)},
                {"kernel": "gray_moments_kernel x2 (+ 2 memsets; get_center of both masks)", "us": round(t_m * 1e6, 2), "alg_bytes": 6 * n,
                 "GBps": round(6 * n / t_m / 1e9, 1)}]
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = replace_cpu_baseline(inputs_h, masks_h, src.bg_always.cpu().numpy(), H, W)
        print(json.dumps({
            "metric": f"frames/sec end-to-end replace ({W}x{H})", "value": round(world * Ktot / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": Ktot, "warmup": Wm, "ms_per_step": round(dt / Ktot * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"replace (tools/replace/replace.py): centroid pass + shift_fg / rescale_fg(1.2) / blend / side-by-side, {W}x{H}, "
                                   f"{Ktot} frames/GPU" + (f" as {len(clips)} clips of <= {K} (page-locked budget per rank)" if len(clips) > 1 else "") +
                                   f", frame-sharded x{world}, one all-reduce of the centroid offsets per clip",
                       "boundary": "pinned host -> pinned host: two alpha masks (pass 1), fg and source frame (pass 2) uploaded (12 B/px), side-by-side image downloaded (6 B/px), eight frames per copy, "
                                   "per frame inside the timed region; the target masks stay in HBM between the two passes",
                       "correspondence_mean": [dx, dy]},
            "pcie_bytes_per_frame": {"up": 12 * n, "down": 6 * n},
            "device_resident": {"value": round(1.0 / t_res, 1), "unit": "frames/s", "note": "same kernels (moments of both masks + one composite launch per frame), inputs in HBM, nothing copied"},
            "self_check": {"frames": len({0, 1, K // 2, K - 1}), "loop_equals_stand_alone_kernel_chain": True},
            "roofline": roof, "kernels": rows, "cpu_baseline": cpu}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def replace_cpu_baseline(inputs_h, masks_h, bg, H, W, budget_s=12.0):
    """The reference's loop body on the CPU oracle (one thread, like the reference's own cv2 + NumPy loop): get_center of both masks,
    shift_fg, rescale_fg, the float64 blend and the concatenation, on as many frames of the same clip as fit the time budget."""
    import oracle
    oracle.build()
    M = None
    t0 = time.time()
    done = 0
    for (fg, fr), (dm, sm) in zip(inputs_h, masks_h):
        fg, fr, dm, sm = fg.numpy(), fr.numpy(), dm.numpy(), sm.numpy()
        dc, sc = oracle.get_center(dm), oracle.get_center(oracle.resize_linear(sm, H, W))
        if M is None:
            M = np.float32([[1, 0, sc[0] - dc[0]], [0, 1, sc[1] - dc[1]]]).astype(np.float64)
        res = oracle.blend(oracle.rescale_fg(oracle.warp_affine(fg, M, H, W), 1.2), oracle.rescale_fg(oracle.warp_affine(dm, M, H, W), 1.2), bg)
        np.concatenate((oracle.resize_linear(fr, H, W), res), axis=1)
        done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return {"value": round(done / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} frames {W}x{H}: both passes of replace.py on the C oracle, 1 thread, {dt:.1f} s",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}


def _net_time(run, iters=10, warm=2):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def mode_cpu_baseline(args, cfg, H, W):
    """green / bg_online: the loop body restated on the CPU oracle (oracle/chains.py) with the same networks on torch-CPU, on a
    bounded sample of the same synthetic clip: one untracked frame (the segmentation network runs) + the tracked frames after it."""
    import oracle
    from oracle import chains
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    oracle.build()
    nthr = min(torch.get_num_threads(), usable_cpus())
    saved = torch.get_num_threads()
    torch.set_num_threads(nthr)
    try:
        mat = MattingNet(synthetic_state_dict(0), device="cpu", dtype=torch.float32)
        st = {}
        if args.mode == "green":
            from video_unscreen_amd.binseg.deeplab_net import DeepLabNetTorch, deeplab_synthetic_state_dict
            n = 3
            clip = SyntheticClip(H, W, n, config_id=1, device="cpu")
            cc = cfg["colorfiltering"]
            cf = chains.OracleColorFilter(oracle, cc["input_long_side"], tuple(cc["bg_ncomp"]), tuple(cc["fg_ncomp"]), cc["max_num_samples"],
                                          cc["color_prior_winsize"], random_state=0)
            seg = DeepLabNetTorch(deeplab_synthetic_state_dict(0))
            frames = [clip.frame(t)["frame"].numpy() for t in range(n)]
            t0 = time.time()
            tracked = sum(int(chains.green_frame(oracle, cfg, st, f, cf, seg, mat)["tracking"]) for f in frames)
        else:
            from video_unscreen_amd.binseg.schp_net import SchpNetTorch, schp_synthetic_state_dict
            from video_unscreen_amd.stm.net import StmNetTorch, stm_synthetic_state_dict
            n = 2
            clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
            schp, stm = SchpNetTorch(schp_synthetic_state_dict(0)), StmNetTorch(stm_synthetic_state_dict(0))
            frames = [clip.frame(t)["frame"].numpy() for t in range(n)]
            t0 = time.time()
            tracked = sum(int(chains.bg_online_frame(oracle, cfg, st, f, schp, stm, mat)["tracking"]) for f in frames)
        dt = time.time() - t0
    finally:
        torch.set_num_threads(saved)
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": f"{n} frames {W}x{H} ({n - tracked} untracked + {tracked} tracked): C oracle (1 thread; sklearn EM / scipy sparse solve where the "
                      f"reference calls them) + the same networks on torch-CPU ({nthr} threads), {dt:.1f} s",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}


def mode_roofline(args, pipe, frame):
    """green / bg_online: the dominant kernels of those loops are the same conv kernels (usc_conv.hip) on other networks - DeepLabv3+
    over the crops of an untracked green frame, the space-time memory network's three graphs for a tracked bg-online frame - plus, in
    the bg-online loop, the float64 Laplace solver (HBM-bound).  Each is timed here with HIP events on the stream it runs on;
    flops = 2 M N K of the planned layers (ConvGraphNet.plan_gflop), bytes of the solver = its vector traffic per iteration (see
    DESIGN.md section 6)."""
    extra = {}
    if args.mode == "green":
        agent = pipe.segagent
        net = agent.model_for(agent.crop_batch)
        t = _net_time(lambda: net.forward(None))
        gf = net.plan_gflop()
        tf = gf / t / 1e3
        roof = {"bound": "mfma", "kernel": f"conv_mfma_kernel + conv_wino_kernel (DeepLabv3+/R50, {agent.crop_batch} crops of 513 x 513 per forward, {net.n_layers()} launches)",
                "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "alg_gflop_per_forward": round(gf, 1), "ms_per_forward": round(t * 1e3, 3)}
        mnet = next(iter(getattr(pipe.vmatagent, "_nets", {}).values()), None)
        if mnet is not None:                            # the matting net of the tracked frames (the headline mode's roofline kernel)
            tm = _net_time(lambda: mnet())
            gm = UNET_GFLOP_544x960 * (mnet.H * mnet.W) / (544 * 960)
            extra["roofline_matting"] = {"bound": "mfma", "achieved": round(gm / tm / 1e3, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": round(gm / tm / 1e3 / MFMA_F32_PEAK_TFLOPS, 4), "ms_per_forward": round(tm * 1e3, 3)}
        return roof, extra
    stm = next(iter(getattr(pipe.stmagent, "_nets", {}).values()), None)
    roof = None
    if stm is not None:
        def step():
            stm.mem.run()
            stm.qry.run()
            stm._read_for(1).run()
            stm.dec.run()
        t = _net_time(step)
        gf = stm.mem.plan_gflop() + stm.qry.plan_gflop() + stm._read_for(1).plan_gflop() + stm.dec.plan_gflop()
        tf = gf / t / 1e3
        roof = {"bound": "mfma", "kernel": "conv_mfma_kernel + conv_wino_kernel (space-time memory network: memorize + query trunk + memory read + decoder)",
                "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "alg_gflop_per_forward": round(gf, 1), "ms_per_forward": round(t * 1e3, 3)}
    # the Laplace fill of the frame's own hole: bytes = iterations x unknowns x 130 B - per unknown and preconditioned CG iteration the four
    # level-0 passes move 12 ("down": r f64 in, z f32 out) + 17 ("up": z, r, a quarter of z_c in, z out) + 28 (p = z + beta p, A p) + 52
    # (x, r update + the two dot products) bytes, the float32 coarser levels a third of 22 + 40 more (their two passes also read the level's
    # operator: a diagonal and four edge weights per cell) (round 3's float64 / seven-pass form: 195); time = HIP events around one converged solve
    from video_unscreen_amd import _native as nv
    out = pipe.step(frame)
    if "bg_holes" in out and "hole" in out:
        bg, hole = out["bg_holes"], out["hole"]
        _, iters = nv.regionfill(bg, hole, want_iters=True)
        t = _net_time(lambda: nv.regionfill(bg, hole, budget=iters + (iters & 1)), iters=3, warm=1)     # (the default budget of 48 would time 13 no-op launches per unused iteration too)
        unknowns = 3 * int((hole > 0).sum().item())
        nbytes = iters * unknowns * 130
        extra["roofline_fill"] = {"bound": "hbm", "kernel": "usc_regionfill.hip (CG in float64, tiled float32 multigrid V-cycle with Galerkin coarse operators as preconditioner; 13 dependent launches per iteration: latency-bound)", "achieved": round(nbytes / t / 1e9, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                  "iterations": iters, "unknowns": unknowns, "ms_per_solve": round(t * 1e3, 3)}
    return roof, extra


def gpu_numa_node(index, sysfs="/sys"):
    """NUMA node of GPU `index`'s PCIe root from sysfs, or None (no NUMA information, node -1, no such device)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        dom, bus, devid = getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", f"{dom:04x}:{bus:02x}:{devid:02x}.0", "numa_node")).read())
        return node if node >= 0 else None
    except Exception:           # noqa: BLE001
        return None


def bind_near_gpu(local, local_world=1, sysfs="/sys"):
    """Multi-rank runs: keep this rank's host threads (and so its pinned staging memory, first-touched by them) on the NUMA node its
    GPU's PCIe root hangs off - uploads and downloads then do not cross the socket interconnect.  Best effort: any missing piece
    (no NUMA information, a cpuset that excludes the node) leaves the affinity as it is.  Every thread the process already has is
    re-bound too (torch may have started its intra-op pool while importing: sched_setaffinity(0) only reaches the calling thread and
    threads created later).  Returns (description, ranks sharing the node) - the second for sizing this rank's CPU pool: the local
    ranks whose GPU reports the same node (NPS4 hosts have eight nodes, not two)."""
    try:
        node = gpu_numa_node(local, sysfs)
        if node is None:
            return "no NUMA node reported", max(1, local_world)
        cpus = set()
        for part in open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"node {node}: outside this process's cpuset", max(1, local_world)
        tids = [int(t) for t in os.listdir("/proc/self/task")] if os.path.isdir("/proc/self/task") else [0]
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
            except OSError:         # (a thread that has just exited)
                pass
        sharers = sum(1 for g in range(max(1, local_world)) if gpu_numa_node(g, sysfs) == node)
        return f"node {node}: {len(cpus)} cpus", max(1, sharers)
    except Exception as e:      # noqa: BLE001 - placement is an optimisation, never a reason to fail
        return f"unchanged ({type(e).__name__})", max(1, local_world)


def strong_seg_batch(clip_frames, world):
    """Frames per human-parsing forward of the strong-scaling form: every rank plans for the same batch - the largest range
    (ceil(F / world)) cut into the fewest groups of <= 12 frames, evenly: a 38-frame range runs as 10 + 10 + 10 + 8, not 12 + 12 + 12 + 2
    (a short last group rides in a full forward)."""
    kmax = -(-clip_frames // world)
    return -(-kmax // -(-kmax // 12))


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves - one child `torch.distributed.run`
    (the command line the driver uses), one process per GPU - and pass rank 0's JSON line through.  This process has made no GPU
    call (importing torch and counting devices do not initialise HIP), and it starts a CHILD and waits: it never replaces itself.
    The reference's analogue is tools/unscreen.sh:3-17, which takes the GPU list on its command line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("USC_BENCH_SHARE_GPU"):
        print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a line "
                  f"whose n_gpus would not be what was asked for", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # Rehearsal knobs (tests/test_tools_gpu.py exercises the N > 1 code path on a one-GPU box): ranks may share a device and
    # talk over gloo.  The driver's runs use neither: one rank per GPU, backend "nccl" (= RCCL over xGMI).
    backend = os.environ.get("USC_BENCH_BACKEND", "nccl")
    if os.environ.get("USC_BENCH_SHARE_GPU"):
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    numa, sharers = (bind_near_gpu(local, local_world) if world > 1 and not os.environ.get("USC_BENCH_SHARE_GPU")
                     else ("not bound (single rank)", local_world))
    if world > 1:
        # the ranks of a node share its cores: each keeps torch's CPU pool (network set-up: folding, weight packing) to its share
        # (bound to a NUMA node, a rank shares that node's cores with the ranks whose GPUs hang off the same node)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // sharers)))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_native()
    if world > 1:
        dist.barrier()
    if args.mode == "replace":
        return replace_mode(args, rank, world, dev)
    if args.mode != "bg_step":
        return other_mode(args, rank, world, dev)
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.dist import broadcast_state, shard_range
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import synthetic_state_dict

    H, W, K, Wm = args.height, args.width, args.steps, args.warmup
    cfg = default_config()
    strong = args.clip_frames > 0
    if strong:
        # ONE clip of F frames cut into contiguous ranges (the reference's --range, bg_offline.py:34-35): this rank times its own
        # range [lo, hi) from a fresh alpha_pre, after Wm untimed warm-up frames (the head of its range, run and discarded)
        F_clip = args.clip_frames
        if F_clip < world:
            print(f"bench.py: --clip-frames {F_clip} < {world} ranks", file=sys.stderr)
            sys.exit(2)
        lo, hi = shard_range(F_clip, rank, world)
        K = hi - lo                                          # this rank's timed frames
        Wm = min(Wm, K)

    # weights + overall background: rank 0 owns them, everyone else receives them over RCCL / xGMI
    clip = SyntheticClip(H, W, F_clip if strong else (K + Wm) * world, config_id=2, device=dev)
    sd = synthetic_state_dict(0) if rank == 0 else None
    extra = {}
    if not args.no_seg:
        from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
        extra["schp"] = schp_synthetic_state_dict(0) if rank == 0 else None          # 267 MB: rank 0's, one bucket with the UNet's
    sd, bg_always, extra = broadcast_state(sd, clip.bg_always if rank == 0 else torch.empty_like(clip.bg_always), dev, world, extra=extra)
    seg_agent = None
    if not args.no_seg:
        from video_unscreen_amd.binseg import HumanSegAgent
        seg_agent = HumanSegAgent(model_path=None, cuda_device=local, state_dict=extra["schp"])
    if args.seg_batch > 0:
        seg_batch = args.seg_batch
    elif strong:
        seg_batch = strong_seg_batch(F_clip, world)
    else:
        seg_batch = next((b for b in (12, 8, 6, 4) if K % b == 0), 8)
    pipe = BgStepPipeline(cfg, H, W, device=dev, state_dict=sd, net_backend=args.net, seg_agent=seg_agent,
                          seg_batch=1 if args.sync_loop else seg_batch, object_filter=args.object_filter)

    # this rank's contiguous frame range (not timed).  Host boundary (default): each frame + its per-frame background (+ its
    # mask with --no-seg) sit in pinned host memory, and pinned host buffers receive alpha + fg.
    from video_unscreen_amd.hostio import HostFrameIO, pack_host_group, packed_outputs
    if not strong:
        lo, hi = shard_range((K + Wm) * world, rank, world)
    clip.gen.manual_seed(20260002 + 7919 * rank)
    first = 0 if strong else Wm                              # index of the first timed frame in this rank's list
    host_io = not args.device_resident
    frames, frames_h = [], []
    for t in range(lo, hi):
        f = clip.frame(t)
        if host_io:      # one page-locked buffer per frame, laid out like the upload staging slot: the group goes up with ONE copy
            frames_h.append(pack_host_group(f["frame"].cpu(), f["bg"].cpu(), f["seg"].cpu() if seg_agent is None else None))
        if not host_io or (rank == 0 and t - lo < first + K):
            frames.append((f["frame"], f["seg"], f["bg"], bg_always))
    alpha_h = fg_h = io = None
    if host_io:
        alpha_h, fg_h = packed_outputs(K, H, W)               # frame k's alpha and fg are neighbours: ONE download per frame
        io = HostFrameIO(pipe)
    torch.cuda.synchronize()

    def run(inputs):                                         # inputs resident in HBM, outputs stay there
        if args.sync_loop:
            for i in inputs:
                pipe.step(*i)
        else:
            pipe.run(iter(inputs))

    def run_host(first, n):                                  # pinned host -> pinned host
        if args.sync_loop:
            for j in range(n):
                fh, sh, bh = frames_h[first + j]
                out = pipe.step(fh.to(dev, non_blocking=True), sh.to(dev, non_blocking=True) if sh is not None else None,
                                bh.to(dev, non_blocking=True), bg_always)
                alpha_h[j].copy_(out["alpha"], non_blocking=True)
                fg_h[j].copy_(out["fg"], non_blocking=True)
        else:
            t_a = time.perf_counter()
            pipe.run(io.inputs(frames_h[first:first + n], bg_always), on_result=io.collect(alpha_h, fg_h))
            t_b = time.perf_counter()
            io.drain()                                       # every download enqueued and finished (the copy thread may trail the loop)
            if os.environ.get("USC_BENCH_DEBUG"):            # where the host spent the pass (stderr)
                from video_unscreen_amd.hostio import PendingCopy
                print("run_host(%d, %d): enqueue %.1f ms, drain %.1f ms; blocked so far: %.1f ms for staging slots, %.1f ms for output ring entries" % (
                    first, n, (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3, io.slot_waited * 1e3, PendingCopy.waited * 1e3), file=sys.stderr)

    def spin_up(seconds=1.0):
        """The clip is generated and pinned on the CPU for seconds while the GPU idles in a low power state, and W warm-up frames
        (30 ms at W = 10) do not bring its clocks back: untimed matting forwards on a scratch input do (measured: the same 300
        timed frames run 3.5 % slower without, and 1 - 2 % slower after 0.25 s than after 1.5 s; nothing here touches the
        pipeline's state or the frames)."""
        if args.no_spin_up:
            return
        x = torch.zeros_like(pipe.x7)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                pipe.net(x)
            torch.cuda.synchronize()

    def rehearse(fn, seconds=0.6):
        """Untimed passes of the very loop that is timed next (whole path: uploads, parsing net, stage-3 loop, downloads - the matting
        forwards of spin_up() exercise one of its three streams only): the first pass of a process measured 3 % slower than any later
        one (profiles/r02).  The timed region itself stays W warm-up frames + exactly K timed frames."""
        if args.no_spin_up:
            return
        t0 = time.perf_counter()
        passes = 0
        while time.perf_counter() - t0 < seconds or passes < 2:           # (a 300-frame pass is longer than 0.6 s: at least two)
            pipe.reset()
            fn()
            torch.cuda.synchronize()
            passes += 1

    def timed(fn_warm, fn):
        pipe.reset()
        spin_up()
        rehearse(fn)
        pipe.reset()
        fn_warm()
        torch.cuda.synchronize()
        if strong:
            pipe.reset()                                     # the timed range starts like a reference --range run: alpha_pre = its first gate
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    if host_io:
        dt = timed(lambda: run_host(0, Wm), lambda: run_host(first, K))
    else:
        dt = timed(lambda: run(frames[:Wm]), lambda: run(frames[first:first + K]))
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total = F_clip if strong else world * K                 # frames all ranks processed inside the timed region

    result = None
    if rank == 0:
        fps = total / dt
        K_line = total if strong else K
        result = {
            "metric": "frames/sec end-to-end unscreen (bg_step, 1080p)" if (H, W) == (1080, 1920) else f"frames/sec end-to-end unscreen (bg_step, {W}x{H})",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K_line, "warmup": Wm,
            "ms_per_step": round(dt / K_line * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("bg_step: " + ("human-parsing segmentation (SCHP ResNet-101) + " if seg_agent is not None else "")
                                    + "stage-3 loop (bgdiff mask + object filter + trimap + matting UNet-18 + get_fg), "
                                    + (f"{W}x{H}, ONE clip of {F_clip} frames cut into {world} contiguous frame range(s) (rank 0: {K} frames)"
                                       if strong else f"{W}x{H}, {K} frames/GPU, frame-sharded x{world}")),
                       "boundary": ("pinned host -> pinned host: frame + per-frame background" + (" + mask" if seg_agent is None else "")
                                    + " uploaded, alpha + fg downloaded per frame inside the timed region (SURVEY.md 8d)") if host_io
                                   else "device resident: inputs in HBM when the timed region starts, outputs left in HBM",
                       "binseg": (f"HumanSegAgent on every frame ({pipe.seg_batch} frames per network forward)" if seg_agent is not None
                                  else "masks are inputs (--no-seg)"),
                       "net_input": [pipe.in_h, pipe.in_w], "loop": "sync" if args.sync_loop else "pipelined", "convs": args.net,
                       "parity": ("not bit-exact end to end: gate / object filter / trimap / foreground are bit-exact against the CPU chain given the same "
                                  "inputs, alpha is within 1e-3 (float) / 1 LSB, and the segmentation mask - an argmax over float logits from two f32 "
                                  "summation orders - differs from the CPU chain in 2 - 16 of the frame's pixels, every one an argmax tie below 1e-4 "
                                  "(tests/test_gpu_seg.py, profiles/r05/parity_report.json); OpenCV's interior is restated by the oracle, not pinned"),
                       "object_filter": ("host worker threads (border-following C++ in libunscreen_hip.so; same masks as the device kernels)"
                                         if args.object_filter == "host" and not args.sync_loop else
                                         "device (label-based HIP kernels inside the foreground's bounding box, usc_objects.hip): no CPU stage in the loop"),
                       "weights": "keyed-deterministic synthetic (no checkpoint offline)", "host_numa_rank0": numa},
        }
        if host_io:
            nb = H * W
            result["pcie_bytes_per_frame"] = {"up": 6 * nb + (nb if seg_agent is None else 0), "down": 4 * nb}
        if world == 1 and host_io:
            # the same loop with inputs resident in HBM and outputs left there (round 1's figure), for the record (not `value`)
            t1 = timed(lambda: run(frames[:Wm]), lambda: run(frames[first:first + K]))
            result["device_resident"] = {"value": round(K / t1, 2), "unit": "frames/s",
                                         "note": "same loop, frames already in HBM, alpha / fg not copied out"}
        if world == 1 and not args.sync_loop and args.object_filter != "host":
            # the same timed loop with remove_invalid_objects on host worker threads (round 1's default), for the record (not `value`)
            chosen_f, chosen_c = pipe.object_filter, pipe._filter_choice
            pipe.object_filter, pipe._filter_choice = "host", "host"
            if host_io:
                t1 = timed(lambda: run_host(0, Wm), lambda: run_host(first, K))
            else:
                t1 = timed(lambda: run(frames[:Wm]), lambda: run(frames[first:first + K]))
            result["host_object_filter"] = {"value": round(K / t1, 2), "unit": "frames/s",
                                            "note": "same loop and boundary as `value` with remove_invalid_objects on two host worker threads"}
            pipe.object_filter, pipe._filter_choice = chosen_f, chosen_c
        if world == 1 and not args.sync_loop and args.net == "hip" and K % 2 == 0 and K >= 2 * pipe.seg_batch and not args.no_interleave:
            # Opt-in figure, never `value`: this GPU's range cut into TWO contiguous sub-ranges - independent by the reference's own --range
            # semantics (tools/unscreen/bg_offline.py:34-35, 184-192: each starts its alpha_pre chain afresh) - whose frames share every
            # matting forward (usc_conv_desc.batch = 2, planned as for one image: each sub-range gets the bits run() gives it on its own,
            # tests/test_gpu_hostio.py).  It measures how much of the frame is launch granularity; the frames right after the cut differ
            # from a run over the whole range, as they do between any two --range runs.  Inputs resident in HBM.
            pipe2 = BgStepPipeline(cfg, H, W, device=dev, state_dict=sd, net_backend=args.net, seg_agent=seg_agent, seg_batch=pipe.seg_batch,
                                   object_filter=args.object_filter, interleave=2)
            half = K // 2
            ra, rb = frames[first:first + half], frames[first + half:first + K]

            def run2():
                pipe2.reset()
                pipe2.run_interleaved(iter(ra), iter(rb))
            for _ in range(2):
                run2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run2()
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t0
            result["interleaved_shards"] = {"value": round(K / t2, 2), "unit": "frames/s", "ranges": 2,
                                            "note": "NOT the headline: two contiguous sub-ranges of this GPU's frames (independent alpha_pre chains, as two "
                                                    "--range runs of the reference) share every matting forward; inputs in HBM; compare with `device_resident`"}
            del pipe2
            torch.cuda.empty_cache()
        if world == 1 and host_io and not args.sync_loop:
            # not timed: the host -> host loop must deliver the bytes of the loop over HBM-resident inputs (copy streams, the download
            # thread and the side streams may not overtake each other into buffers still in use) - two segmentation groups and a
            # ragged tail, compared byte for byte
            nchk = min(K, 2 * pipe.seg_batch + 3)
            kept = []
            pipe.reset()
            pipe.run(iter(frames[first:first + nchk]), on_result=lambda n, o: kept.append((o["alpha"].clone(), o["fg"].clone())))
            torch.cuda.synchronize()
            alpha_h[:nchk].zero_()
            pipe.reset()
            run_host(first, nchk)
            torch.cuda.synchronize()
            bad = [n for n in range(nchk) if not (torch.equal(alpha_h[n], kept[n][0].cpu()) and torch.equal(fg_h[n], kept[n][1].cpu()))]
            # and both against the synchronous loop: every stage of a frame in order on ONE stream (pipe.step)
            pipe.reset()
            bad_sync = []
            for n in range(nchk):
                fr_, _, bg_, bga_ = frames[first + n][0], None, frames[first + n][2], frames[first + n][3]
                o = pipe.step(fr_, frames[first + n][1] if seg_agent is None else None, bg_, bga_)
                if not (torch.equal(o["alpha"], kept[n][0]) and torch.equal(o["fg"], kept[n][1])):
                    bad_sync.append(n)
            result["self_check"] = {"frames": nchk, "host_loop_equals_device_loop": not bad, "pipelined_loop_equals_synchronous_loop": not bad_sync}
            if bad or bad_sync:
                print("bench.py: the loops disagree - host -> host vs device-resident on frames %s, pipelined vs synchronous on frames %s"
                      % (bad[:8], bad_sync[:8]), file=sys.stderr, flush=True)
                sys.exit(1)
        if not args.no_kernel_table:                         # (other ranks wait at the final barrier: not inside the timed region)
            rows, t_net, t_graph = kernel_table(pipe, frames[first], H, W)
            dom = max(rows, key=lambda r: r["us"])
            scale = (pipe.in_h * pipe.in_w) / (544 * 960)
            gflop = UNET_GFLOP_544x960 * scale
            tf = gflop / t_net / 1e3
            if args.net == "hip":
                # dominant kernels of the frame: conv_mfma_kernel / conv_wino_kernel (usc_conv.hip).  One "launch" here is one network
                # forward = num_layers conv launches; flops are the algorithmic 2*M*N*K of the reference module (torch flop counter,
                # SURVEY.md section 6) - the Winograd layers execute 2.25x fewer.  Time: HIP events around forwards whose layers are
                # launched one after the other on ONE stream with nothing beside them (the product's launch mode) - the quantity
                # profiles/r02/unet_layer_table.md (rocprofv3 kernel trace of the same launches, tools/dev/layer_table.py) sums per
                # layer.  ms_per_forward_graph: the same layers as one hipGraph replay (slower on this runtime).
                result["roofline"] = {"bound": "mfma", "kernel": "conv_mfma_kernel + conv_wino_kernel (all %d conv launches of one forward)" % pipe.net.num_layers,
                                      "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic(H, W),
                                      "alg_gflop_per_forward": round(gflop, 2), "ms_per_forward": round(t_net * 1e3, 3),
                                      "ms_per_forward_graph": round(t_graph * 1e3, 3),
                                      "timing": "serial single-stream launches, HIP events; the per-layer rocprofv3 kernel-trace table of the same launches is "
                                                "profiles/r05/unet_layer_table.md (tools/dev/layer_table.py)"}
                result["roofline_pixel"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": dom["frac_hbm_peak"], "traffic": None,
                                            "alg_bytes_per_launch": dom["alg_bytes"], "us_per_launch": dom["us"]}
            else:
                result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": dom["frac_hbm_peak"], "traffic": None,
                                      "alg_bytes_per_launch": dom["alg_bytes"], "us_per_launch": dom["us"]}
            result["kernels"] = rows
            what = ("matting UNet-18, hand-written f32 MFMA implicit-GEMM / Winograd convs (usc_conv.hip), %d fused launches" % pipe.net.num_layers
                    if args.net == "hip" else "matting UNet-18 via PyTorch-ROCm (MIOpen)")
            result["net"] = {"bound": "mfma", "what": what, "ms": round(t_net * 1e3, 3),
                             "achieved": round(gflop / t_net / 1e3, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(gflop / t_net / 1e3 / MFMA_F32_PEAK_TFLOPS, 4)}
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed at N = 1 only
            clip_cpu = SyntheticClip(H, W, args.cpu_frames, config_id=2, device="cpu")
            cf = []
            for t in range(args.cpu_frames):
                f = clip_cpu.frame(t)
                cf.append({k: f[k].numpy() for k in ("frame", "seg", "bg")})
            nthr = min(torch.get_num_threads(), usable_cpus())
            cpu_fps, cpu_dt = cpu_baseline(cfg, pipe, cf, clip_cpu.bg_always.numpy(), args.cpu_frames, threads=nthr)
            result["cpu_baseline"] = {"value": round(cpu_fps, 4), "unit": "frames/s", "cores": nthr,
                                      "kind": "port",
                                      "sample": f"{args.cpu_frames} frames {W}x{H}: C oracle (1 thread) for the pixel stages + the same "
                                                f"network(s) on torch-CPU ({nthr} threads), {cpu_dt:.1f} s",
                                      "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
            if not args.no_cpu_single_thread:                # single-thread next to all-core (one frame: ~10 - 25 s)
                fps1, dt1 = cpu_baseline(cfg, pipe, cf, clip_cpu.bg_always.numpy(), 1, threads=1)
                result["cpu_baseline"]["single_thread"] = {"value": round(fps1, 4), "unit": "frames/s", "cores": 1,
                                                           "sample": f"1 frame {W}x{H}, everything on one thread, {dt1:.1f} s"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
