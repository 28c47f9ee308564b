"""VMattingAgent: same constructor and forward() as the reference (unscreen/vmatting/agent.py:11-42, 134-163).

Pre- and post-processing are single HIP kernels (usc_vmat_pre / usc_vmat_post); the network is the folded
conv plan of net.py on PyTorch-ROCm.  forward_device() keeps everything on the GPU; forward() is the
reference's NumPy-in / NumPy-out call."""
import torch

from .. import _native as nv
from ..utils._dev import device, to_dev, to_host
from .net import MattingNet, synthetic_state_dict


class VMattingAgent():

    def __init__(self, model_path, input_long_side=960, cuda_device=0, dtype=torch.float32):
        if cuda_device is None or cuda_device < 0:
            raise RuntimeError("VMattingAgent: cuda_device < 0 (CPU) is not offered - the HIP path has no CPU fallback")
        self.cuda_device = cuda_device
        self.device = device(cuda_device)
        if model_path is None or model_path == "synthetic":
            weights = synthetic_state_dict(0)        # no trained checkpoint exists offline
        else:
            weights = torch.load(model_path, map_location='cpu')['state_dict']
        self.model = MattingNet(weights, device=self.device, dtype=dtype)
        self.division = 32
        if input_long_side % self.division != 0:
            input_long_side = (input_long_side // self.division + 1) * self.division
        self.input_long_side = input_long_side

    def input_size(self, h, w):
        return nv.get_target_size(h, w, self.input_long_side, self.division)

    def forward_device(self, img, alpha_pre, trimap, out=None, want_float=False):
        """img (H, W, 3) u8, alpha_pre (H, W) u8, trimap (H, W) u8 CUDA tensors -> alpha (H, W) u8 [, float alpha]."""
        h, w = trimap.shape
        in_h, in_w = self.input_size(h, w)
        x7 = nv.vmat_pre(img, alpha_pre, trimap, in_h, in_w, layout=nv.NHWC)
        pred = self.model(x7)
        return nv.vmat_post(pred.view(in_h, in_w), trimap, out_u8=out, want_float=want_float)

    def forward(self, img, alpha_pre, trimap):
        with torch.cuda.device(self.device):
            return to_host(self.forward_device(to_dev(img, self.device), to_dev(alpha_pre, self.device),
                                               to_dev(trimap, self.device)))
