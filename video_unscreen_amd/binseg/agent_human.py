"""HumanSegAgent: same constructor and forward() as the reference (unscreen/binseg/agent_human.py:128-157, 262-312).

forward(img) = affine-warp the whole frame into the 473 x 473 network input, SCHP human parsing, bilinear
(align_corners) upsample of the 20 class logits, inverse affine warp back to the frame, argmax > 0 -> 255.
Pre- and post-processing are HIP kernels (csrc/usc_seg.hip); every convolution of the network runs on the
f32-MFMA kernel (schp_net.HipSchpNet).  Like the reference, this agent has no CPU path (agent_human.py:141, 295).
"""
import torch

from .. import _native as nv
from ..utils._dev import device, to_dev, to_host
from .schp_net import NUM_CLASSES, HipSchpNet, schp_synthetic_state_dict, strip_module_prefix

INPUT_SIZE = (473, 473)            # dataset_settings['lip']['input_size'] (agent_human.py:19-31)


class HumanSegAgent():
    def __init__(self, model_path=None, input_long_side=912, downscale=1, crop_h=473, crop_w=473,
                 stride_ratio=1 / 2., flip=True, cuda_device=0, state_dict=None):
        self.input_size = INPUT_SIZE
        self.model_path = model_path
        self.cuda_device = cuda_device if cuda_device is not None and cuda_device >= 0 else 0
        self.device = device(self.cuda_device)
        if state_dict is not None:                     # weights handed over in memory (broadcast from rank 0: dist.broadcast_networks)
            sd = state_dict
        elif model_path is None or model_path == "synthetic":
            sd = schp_synthetic_state_dict(0)          # no trained checkpoint exists offline
        else:
            sd = strip_module_prefix(torch.load(model_path, map_location="cpu"))   # DataParallel keys: module.*
        self.state_dict = sd
        self.plan_batch = 1
        self._nets = {}                       # batch -> network, all planned for self.plan_batch
        # kept for signature compatibility; the reference's forward() does not use them either (agent_human.py:277-312)
        self.crop_h, self.crop_w, self.flip = crop_h, crop_w, flip
        self.input_long_side, self.stride_ratio, self.division = input_long_side, stride_ratio, 1
        self._ws = None

    def set_plan_batch(self, n):
        """Plan the conv kernels' tiles / K partitions for `n` frames per forward (what the caller will mostly run).  All
        networks of one plan give bit-identical masks per frame whatever the batch they are run at."""
        n = max(1, int(n))
        if n != self.plan_batch:
            self.plan_batch, self._nets = n, {}

    def model_for(self, batch):
        if batch not in self._nets:
            self._nets[batch] = HipSchpNet(self.state_dict, INPUT_SIZE[0], INPUT_SIZE[1], device=self.device, batch=batch,
                                           plan_batch=self.plan_batch)
        return self._nets[batch]

    @property
    def net(self):
        return self.model_for(1)

    def forward_device(self, img, out=None, return_logits=False):
        """img: (H, W, 3) uint8 BGR CUDA tensor -> (H, W) uint8 mask {0, 255} on the device."""
        H, W, _ = img.shape
        nv.humanseg_pre(img, INPUT_SIZE[0], INPUT_SIZE[1], out=self.net.x4)
        logits = self.net.forward(None)
        need = nv.load().usc_humanseg_post_ws_bytes(NUM_CLASSES, INPUT_SIZE[0], INPUT_SIZE[1])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        mask = nv.humanseg_post(logits, NUM_CLASSES, H, W, INPUT_SIZE[0], INPUT_SIZE[1], out=out, ws=self._ws)
        return (mask, logits) if return_logits else mask

    def forward_device_batch(self, imgs, outs=None):
        """Several frames through ONE network forward (frames are independent for this agent).  Every mask is bit for bit
        what forward_device gives for that frame; the win is half (a third, ...) of the kernel launches per frame."""
        n = len(imgs)
        if n == 1 and self.plan_batch == 1:
            return [self.forward_device(imgs[0], out=None if outs is None else outs[0])]
        # a short group (the tail of a clip) rides in the network the agent is planned for, its last frame repeated: no second
        # network has to be built for it, and - per-image results not depending on the batch - its masks are the same bits
        nb = max(n, self.plan_batch)
        net = self.model_for(nb)
        for b in range(nb):
            nv.humanseg_pre(imgs[min(b, n - 1)], INPUT_SIZE[0], INPUT_SIZE[1], out=net.image(net.x4, b))
        logits = net.forward(None)
        need = nv.load().usc_humanseg_post_ws_bytes(NUM_CLASSES, INPUT_SIZE[0], INPUT_SIZE[1])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        masks = []
        for b, img in enumerate(imgs):
            H, W, _ = img.shape
            masks.append(nv.humanseg_post(net.image(logits, b), NUM_CLASSES, H, W, INPUT_SIZE[0], INPUT_SIZE[1],
                                          out=None if outs is None else outs[b], ws=self._ws))
        return masks

    def forward(self, img):
        """img (h, w, 3) uint8 BGR -> (h, w) uint8, 255 = person (reference agent_human.py:262-312)."""
        with torch.cuda.device(self.device):
            return to_host(self.forward_device(to_dev(img, self.device)))
