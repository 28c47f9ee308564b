"""Timeline of the 3840 x 2160 host -> host loop: per frame, when (GPU clock) its upload landed, its back half ended and its download ended;
where do the 0.7 ms per frame over the device-resident loop go?   python tools/dev/hostio_timeline.py [H W] [frames]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_native()
from video_unscreen_amd import hostio
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
K = int(sys.argv[3]) if len(sys.argv) > 3 else 120
dev = torch.device("cuda", 0)
seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
clip = SyntheticClip(H, W, K, config_id=2, device=dev)
clip.gen.manual_seed(20260002)
groups = []
for t in range(K):
    f = clip.frame(t)
    groups.append(hostio.pack_host_group(f["frame"].cpu(), f["bg"].cpu(), None))
a_h, f_h = hostio.packed_outputs(K, H, W)
class TimedIO(hostio.HostFrameIO):
    """HostFrameIO whose download thread keeps timing events and host timestamps of every copy"""
    log = []

    def _download_loop(self):
        torch.cuda.set_device(self.pipe.device)
        while True:
            job = self._jobs.get()
            if job is None:
                self._jobs.task_done()
                return
            ev, pairs, pend = job
            ev.synchronize()
            t_seen = time.perf_counter()
            with torch.cuda.stream(self.down):
                st = torch.cuda.Event(enable_timing=True); st.record(self.down)
                for dst, src in pairs:
                    dst.copy_(src, non_blocking=True)
                dn = torch.cuda.Event(enable_timing=True); dn.record(self.down)
            TimedIO.log.append((t_seen, time.perf_counter(), st, dn))
            pend._set(dn)
            self._jobs.task_done()

io = TimedIO(pipe)
main = torch.cuda.current_stream(dev)


def one_pass(record):
    ups, backs, dns, host = [], [], [], []
    t_host0 = time.perf_counter()

    def on_uploaded(ev, token):
        e = torch.cuda.Event(enable_timing=True); e.record(io.up); ups.append(e)
    col = io.collect(a_h, f_h, on_downloaded=lambda pend, token, n: dns.append(pend))

    def on_result(n, out):
        e = torch.cuda.Event(enable_timing=True); e.record(main); backs.append(e)
        host.append(time.perf_counter() - t_host0)
        col(n, out)
    base = torch.cuda.Event(enable_timing=True)
    pipe.reset(); torch.cuda.synchronize()
    base.record(main)
    t0 = time.perf_counter()
    pipe.run(io.inputs(groups, clip.bg_always, on_uploaded=on_uploaded if record else None), on_result=on_result if record else col)
    io.drain(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, base, ups, backs, dns, host

for _ in range(2):
    one_pass(False)
dt, base, ups, backs, dns, host = one_pass(False)
print("plain pass: %.3f ms / frame" % (dt / K * 1e3))
# download events need timing: wrap the thread's event creation
orig_loop = None
TimedIO.log.clear()
dt, base, ups, backs, dns, host = one_pass(True)
print("recorded pass: %.3f ms / frame" % (dt / K * 1e3))
tb = [base.elapsed_time(e) for e in backs]
tu = [base.elapsed_time(e) for e in ups]
ds = [base.elapsed_time(l[2]) for l in TimedIO.log]
de = [base.elapsed_time(l[3]) for l in TimedIO.log]
print("frame  upload landed  back half ended  (period)   host called on_result at   download started  ended")
for n in range(K):
    if n < 40 or n % 12 == 0:
        print("%4d   %9.2f      %9.2f        %6.2f     %9.2f              %9.2f  %9.2f" % (n, tu[n], tb[n], tb[n] - tb[n - 1] if n else tb[n], host[n] * 1e3, ds[n], de[n]))
# the same for the loop over HBM-resident inputs
dev_frames = [(clip.frame(t)["frame"], None, clip.frame(t)["bg"], clip.bg_always) for t in range(0)]
per = [tb[n] - tb[n - 1] for n in range(1, K)]
print("periods: mean %.3f, median %.3f, max %.3f ms; frames with period > 4 ms: %s" % (sum(per) / len(per), sorted(per)[len(per) // 2], max(per), [n for n in range(1, K) if per[n - 1] > 4.0]))

clip2 = SyntheticClip(H, W, K, config_id=2, device=dev)
clip2.gen.manual_seed(20260002)
dev_frames = []
for t in range(K):
    f = clip2.frame(t)
    dev_frames.append((f["frame"], None, f["bg"], clip2.bg_always))
for rep in range(3):
    backs, host = [], []
    base = torch.cuda.Event(enable_timing=True)
    pipe.reset(); torch.cuda.synchronize(); base.record(main); t0 = time.perf_counter()

    def on_result(n, out):
        e = torch.cuda.Event(enable_timing=True); e.record(main); backs.append(e); host.append(time.perf_counter() - t0)
    pipe.run(iter(dev_frames), on_result=on_result)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("device-resident pass: %.3f ms / frame" % (dt / K * 1e3))
tb = [base.elapsed_time(e) for e in backs]
print("frame  back half ended  (period)   host called on_result at")
for n in range(48):
    print("%4d   %9.2f        %6.2f     %9.2f" % (n, tb[n], tb[n] - tb[n - 1] if n else tb[n], host[n] * 1e3))
