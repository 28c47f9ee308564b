"""Host <-> device plumbing for the NumPy-facing API (the reference's agents take and return NumPy images)."""
import numpy as np
import torch


def device(index=0):
    if not torch.cuda.is_available():
        raise RuntimeError("video_unscreen_amd needs a ROCm GPU: the HIP path has no CPU fallback")
    return torch.device("cuda", index)


def to_dev(a, dev=None):
    """uint8 / float32 ndarray -> CUDA tensor (copy)."""
    dev = dev if dev is not None else device()
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a).to(dev, non_blocking=False)


def to_host(t):
    return t.detach().cpu().numpy()
