"""Where does the host boundary of the frame loop lose time against the device-resident loop?  Same pipeline and clip as
bench.py's default; the loop is run (a) device-resident, (b) uploads only, (c) downloads only, (d) both, (e) both with the
copies themselves removed (events and stream waits only).

    python tools/dev/hostio_ablate.py [--frames 240] [--no-seg]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--no-seg", action="store_true")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--minimal", action="store_true", help="ONE HostFrameIO, the host -> host loop first (what bench.py does)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build_native()
    from video_unscreen_amd import hostio
    from video_unscreen_amd.pipeline import BgStepPipeline, default_config
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    dev = torch.device("cuda", 0)
    H, W, K = args.height, args.width, args.frames
    seg_agent = None
    if not args.no_seg:
        from video_unscreen_amd.binseg import HumanSegAgent
        from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
        seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
    clip = SyntheticClip(H, W, K, config_id=2, device=dev)
    clip.gen.manual_seed(20260002)
    bga = clip.bg_always
    dev_frames, host_frames = [], []
    for t in range(K):
        f = clip.frame(t)
        dev_frames.append((f["frame"], f["seg"] if seg_agent is None else None, f["bg"], bga))
        host_frames.append(hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), f["seg"].cpu() if seg_agent is None else None))
    alpha_h, fg_h = hostio.packed_outputs(K, H, W)
    alpha_s = torch.empty((K, H, W), dtype=torch.uint8).pin_memory()          # separate buffers: two downloads per frame (rounds 1 - 2)
    fg_s = torch.empty((K, H, W, 3), dtype=torch.uint8).pin_memory()

    class NoCopyIO(hostio.HostFrameIO):
        """events + waits as HostFrameIO, the copies themselves left out (staging slots are pre-filled once)"""

        def inputs(self, host_frames, bg_always, on_uploaded=None):
            for item in host_frames:
                i = self._n_in % self.depth
                self._n_in += 1
                if self.done[i] is not None and self.done[i] is not hostio._BUSY:
                    self.up.wait_event(self.done[i])
                self.done[i] = hostio._BUSY
                ev = torch.cuda.Event()
                ev.record(self.up)
                yield (self.frames[i], self.segs[i] if item[1] is not None else None, self.bgs[i], bg_always, ev)

        def collect(self, alpha_host=None, fg_host=None, target=None, on_downloaded=None):
            main = torch.cuda.current_stream(self.pipe.device)
            base = self._n_in

            def on_result(n, out):
                ev = torch.cuda.Event()
                ev.record(main)
                self.done[(base + n) % self.depth] = ev
                self.down.wait_event(ev)
                dn = torch.cuda.Event()
                dn.record(self.down)
                self.pipe.note_output_consumed(out, dn)
            return on_result

    def timed(fn, reps=2):
        best = 1e9
        for _ in range(reps):
            pipe.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            for q in ios:
                q.drain()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    io = hostio.HostFrameIO(pipe)                      # "hostwait": no barrier packet in either copy stream
    if args.minimal:
        for name, fn in (("uploads + downloads (first thing in the process)", lambda: pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_h, fg_h))),
                         ("device-resident", lambda: pipe.run(iter(dev_frames))),
                         ("uploads + downloads", lambda: pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_h, fg_h)))):
            for rep in range(3):
                pipe.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); io.drain(); torch.cuda.synchronize()
                t = time.perf_counter() - t0
                print(f"{name:50s} pass {rep}: {K / t:7.1f} frames/s  {t / K * 1e3:.3f} ms/frame", flush=True)
        extra = [hostio.HostFrameIO(pipe, down_mode="own") for _ in range(4)]       # eight more streams, never used
        name = "uploads + downloads, 8 idle streams created"
        for rep in range(3):
            pipe.reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_h, fg_h)); io.drain(); torch.cuda.synchronize()
            t = time.perf_counter() - t0
            print(f"{name:50s} pass {rep}: {K / t:7.1f} frames/s  {t / K * 1e3:.3f} ms/frame", flush=True)
        return
    io_o = hostio.HostFrameIO(pipe, down_mode="own")   # event-ordered copies (rounds 1 - 3a)
    io_p = hostio.HostFrameIO(pipe, down_mode="primed")
    io_s = hostio.HostFrameIO(pipe, down_mode="shared")
    io_x = hostio.HostFrameIO(pipe, down_mode="exclusive")

    ios = (io, io_o, io_p, io_s, io_x)

    def mark_consumed(io=io):                    # frees the staging slot like collect(), downloads nothing
        main = torch.cuda.current_stream(dev)
        base = io._n_in

        def on_result(n, out):
            ev = torch.cuda.Event()
            ev.record(main)
            io.done[(base + n) % io.depth] = ev
        return on_result

    nio = NoCopyIO(pipe)
    for k in range(nio.depth):
        nio.frames[k].copy_(dev_frames[k % K][0]); nio.bgs[k].copy_(dev_frames[k % K][2])
        if dev_frames[k % K][1] is not None:
            nio.segs[k].copy_(dev_frames[k % K][1])
    cases = [
        ("device-resident", lambda: pipe.run(iter(dev_frames))),
        ("uploads only", lambda: pipe.run(io.inputs(host_frames, bga), on_result=mark_consumed())),
        ("downloads only", lambda: pipe.run(iter(dev_frames), on_result=io.collect(alpha_h, fg_h))),
        ("uploads + downloads", lambda: pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_h, fg_h))),
        ("events + waits, no copies", lambda: pipe.run(nio.inputs(host_frames, bga), on_result=nio.collect())),
        ("up + down, event-ordered", lambda: pipe.run(io_o.inputs(host_frames, bga), on_result=io_o.collect(alpha_h, fg_h))),
        ("uploads only, event-ordered", lambda: pipe.run(io_o.inputs(host_frames, bga), on_result=mark_consumed(io_o))),
        ("downloads only, event-ordered", lambda: pipe.run(iter(dev_frames), on_result=io_o.collect(alpha_h, fg_h))),
        ("up + down, primed", lambda: pipe.run(io_p.inputs(host_frames, bga), on_result=io_p.collect(alpha_h, fg_h))),
        ("up + down, exclusive", lambda: pipe.run(io_x.inputs(host_frames, bga), on_result=io_x.collect(alpha_h, fg_h))),
        ("up + down, one copy stream", lambda: pipe.run(io_s.inputs(host_frames, bga), on_result=io_s.collect(alpha_h, fg_h))),
        ("up + down, alpha / fg as two copies", lambda: pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_s, fg_s))),
        ("uploads + downloads again", lambda: pipe.run(io.inputs(host_frames, bga), on_result=io.collect(alpha_h, fg_h))),
        ("device-resident again", lambda: pipe.run(iter(dev_frames))),
    ]
    for name, fn in cases:
        t, n = timed(fn), K
        for q in ios:
            q.drain()
            q.done = [None] * q.depth
        print(f"{name:28s} {n / t:7.1f} frames/s  {t / n * 1e3:.3f} ms/frame   (blocked so far: {io.slot_waited * 1e3:.0f} ms for staging slots, {hostio.PendingCopy.waited * 1e3:.0f} ms for ring entries)", flush=True)


if __name__ == "__main__":
    main()
