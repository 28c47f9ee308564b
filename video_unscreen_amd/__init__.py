"""video_unscreen_amd: the unscreen hot path on MI355X (HIP kernels behind the reference's agent API).  See README.md / DESIGN.md."""
import os


def usable_cpus():
    """CPUs this process may really use: its affinity mask, capped by the cgroup CPU quota when there is one (a GPU box hands a
    process a share of the host: 128 torch threads on a 16-CPU share run slower than 16)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


_capped = False


def respect_cpu_share():
    """The host-side set-up of the networks (folding, weight packing, the Winograd transform of the weights in float64) runs on torch's CPU
    threads; torch sizes its pool by the HOST's cores, which on a cgroup-limited share oversubscribes it many times over.  Cap the pool at
    what the process may use (never raise it; USC_KEEP_TORCH_THREADS=1 leaves it alone).  Called once when the native library is loaded
    (_native.load()) - not at package import, which the decoder / encoder worker processes of utils/frameio.py also go through and which
    must stay free of torch."""
    global _capped
    if _capped or os.environ.get("USC_KEEP_TORCH_THREADS"):
        return
    _capped = True
    try:
        import torch
        n = usable_cpus()
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except Exception:      # noqa: BLE001 - a tuning step, never a reason to fail
        pass
