"""Device-resident bg_step (stage 3) frame loop: tools/unscreen/bg_offline.py:145-176 of the reference.

Per frame, with every image staying in HBM between stages:

    bg, gate   = usc_bgdiff_mask_u8(frame, bg_i, bg_always, seg)        bg_offline.py:150-160
    alpha_pre  = gate on the first frame                                 :162-163
    alphaor    = remove_invalid_objects(gate)      (host stage)          :165
    trimap     = usc_trimap_u8(alphaor)                                  :166
    alpha      = usc_vmat_pre -> matting net -> usc_vmat_post            :167
    fg         = usc_get_fg_u8(frame, alpha, bg pasted where alpha == 0) :171-172
    alpha_pre  = alpha                                                   :176

`step()` is the plain synchronous form.  `run()` software-pipelines it: the front half of frame t+1 (difference
mask, the 1 byte/pixel round trip through the host object filter, trimap) runs on its own HIP stream and a
worker thread while the back half of frame t (network + foreground) runs on the main stream - legal because the
only cross-frame dependency is alpha_pre -> network input.
"""
import copy
import queue
import threading

import numpy as np
import torch

from . import _native as nv
from .vmatting.net import MattingNet, synthetic_state_dict

_DEFAULT_CFG = {
    "fg_exist_thr": 0.001,
    "bg_mask": {"thr": 25, "fusion_weight": 0.1},
    "objectremoval": {"score_map_center": {"landscape": [0.5, 0.5], "portrait": [0.6, 0.5]},
                      "saliency_thr": 0.00001, "consensus_thr": 0.5},
    "trimap": {"input_long_side": 960, "kernelsize": 3, "iters": 5, "color_winsize": [10, 100, 180]},
    "vmatting": {"model_path": "synthetic", "input_long_side": 960},
}


def default_config():
    """The bg_step subset of the reference's configs/bg.json."""
    return copy.deepcopy(_DEFAULT_CFG)


class _Slot:
    """Per-frame device buffers of the front half (double-buffered by run())."""

    def __init__(self, H, W, dev):
        u8 = dict(dtype=torch.uint8, device=dev)
        self.bg = torch.empty((H, W, 3), **u8)
        self.gate = torch.empty((H, W), **u8)
        self.alphaor = torch.empty((H, W), **u8)
        self.trimap = torch.empty((H, W), **u8)
        self.seg = torch.empty((H, W), **u8)
        self.host = torch.empty((H, W), dtype=torch.uint8).pin_memory()
        self.ready = torch.cuda.Event()
        self.copied = torch.cuda.Event()
        self.released = torch.cuda.Event()            # recorded behind the back half of the frame that used this slot
        self.host_done = threading.Event()            # set by the host worker that filtered this slot
        self.frame = None
        self.mset = None                              # set of segmentation-mask buffers this frame is the last user of (run())


class BgStepPipeline:
    def __init__(self, cfg, H, W, device="cuda:0", dtype=torch.float32, state_dict=None, net_backend="hip",
                 seg_agent=None, seg_batch=1, object_filter="auto", interleave=1):
        """seg_agent: optional HumanSegAgent.  When given, the coarse segmentation mask of every frame is computed on
        the device (reference bg_offline.py:63 `segagent.forward(frame)`, the untracked branch of stage 1) instead of
        being read from the caller; it runs in the front half, concurrently with the previous frame's matting net.
        seg_batch: run() segments that many frames ahead per network forward (frames are independent for the
        segmentation; the masks are bit-identical to per-frame forwards, the launches per frame drop).
        object_filter: "device" = remove_invalid_objects from connected-component labels on the GPU (0.5 ms at 1080p, the frame
        never leaves HBM, no host synchronisation); "host" = the border-following C++ stage (1 byte / pixel down and up,
        1.3 ms on one core, on worker threads inside run()).  Both give identical masks.  "auto" (default) = "device": since the
        device filter works inside the foreground's bounding box with LDS-combined sums it is the faster loop (round 2, 1080p:
        337 frames/s all on the device against 324 with the host stage), and it leaves no CPU thread in the per-rank loop."""
        self.cfg = cfg
        self.H, self.W = H, W
        if interleave not in (1, 2):
            raise ValueError("interleave: 1 (one frame range) or 2 (two independent frame ranges through one matting forward)")
        self.interleave = interleave
        self._enqueue_lock = threading.Lock()
        self.device = torch.device(device)
        nv.load()
        self._sd = state_dict if state_dict is not None else synthetic_state_dict(0)
        self.net_backend = net_backend
        self._cpu_net = None
        self.beta = cfg["bg_mask"]["fusion_weight"]
        self.thr = cfg["bg_mask"]["thr"]
        tcfg = cfg["trimap"]
        self.long_side, self.ksize, self.iters = tcfg["input_long_side"], tcfg["kernelsize"], tcfg["iters"]
        L = cfg["vmatting"]["input_long_side"]
        if L % 32 != 0:
            L = (L // 32 + 1) * 32
        self.in_h, self.in_w = nv.get_target_size(H, W, L, 32)
        centers = cfg["objectremoval"]["score_map_center"]
        self.center = centers["landscape"] if W > H else centers["portrait"]
        self.sal_thr = cfg["objectremoval"]["saliency_thr"]
        self.cons_thr = cfg["objectremoval"]["consensus_thr"]
        if object_filter not in ("auto", "device", "host"):
            raise ValueError(f"unknown object_filter {object_filter!r}")
        self._filter_choice = object_filter
        self.object_filter = "device" if object_filter == "auto" else object_filter
        self.filter = nv.ObjectFilter()
        self.dfilter = nv.DeviceObjectFilter()
        dev = self.device
        self.ws_bgdiff = torch.empty(nv.load().usc_bgdiff_ws_bytes(H, W), dtype=torch.uint8, device=dev)
        self.ws_trimap = torch.empty(nv.load().usc_trimap_ws_bytes(H, W, self.long_side), dtype=torch.uint8, device=dev)
        if net_backend == "hip":          # hand-written f32-MFMA convs, one C call per forward (vmatting/net_hip.py)
            if dtype != torch.float32:
                raise ValueError("the HIP conv path computes in float32")
            from .vmatting.net_hip import HipMattingNet
            self.net = HipMattingNet(self._sd, self.in_h, self.in_w, device=dev)
            self.x7 = self.net.x8
            self._layout = nv.NHWC8
            # interleave = 2 (run_interleaved): two images per set of launches - planned as for one image, so each image gets the bits the
            # single-image network gives it (usc_conv_desc.batch / plan_batch)
            self.net2 = HipMattingNet(self._sd, self.in_h, self.in_w, device=dev, batch=2) if interleave == 2 else None
        elif net_backend == "miopen":     # the same plan through PyTorch-ROCm / MIOpen
            self.net = MattingNet(self._sd, device=dev, dtype=dtype)
            self.x7 = torch.empty((1, 7, self.in_h, self.in_w), dtype=torch.float32, device=dev,
                                  memory_format=torch.channels_last)
            self._layout = nv.NHWC
        else:
            raise ValueError(f"unknown net_backend {net_backend!r}")
        if interleave == 2 and net_backend != "hip":
            raise ValueError("interleave = 2 needs the HIP conv path")
        self.slots = [_Slot(H, W, dev) for _ in range(3 * interleave)]
        self.seg_agent = seg_agent
        self.seg_batch = max(1, int(seg_batch))
        if seg_agent is not None and hasattr(seg_agent, "set_plan_batch"):
            seg_agent.set_plan_batch(self.seg_batch)       # step() and run() then segment a frame to the same bits
        self._seg_ring = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(3 * self.seg_batch)] if seg_agent is not None else []
        self._seg_set_done = [torch.cuda.Event() for _ in range(3)]      # the last frame of the group that used a set has left the back half
        # outputs live in a ring of 8 (alpha of frame t is alpha_pre of frame t + 1; a consumer on another stream - the
        # download stream of hostio.HostFrameIO - may still be reading frame t - 1): no allocation inside the frame loop.  A ring
        # entry is ONE buffer [alpha (H W) | fg (H W 3)]: a consumer that wants both on the host fetches them with one copy (two
        # device -> host copies per frame cost the conv stream 0.24 ms per frame, one packed copy 0.05: tools/dev/d2h_batch_probe.py)
        from .hostio import out_layout
        fg_off, total = out_layout(H, W)                     # both parts on 16-byte boundaries
        self._out_buf = torch.empty((8, total), dtype=torch.uint8, device=dev)
        self._out_ring = [{"alpha": self._out_buf[i, :H * W].view(H, W), "fg": self._out_buf[i, fg_off:fg_off + 3 * H * W].view(H, W, 3),
                           "packed": self._out_buf[i], "consumed": None} for i in range(8)]
        self._out_i = 0
        self._alpha = [None, None]                            # alpha_pre of the frame range(s): index = lane (one lane unless run_interleaved)

    def cpu_net(self):
        """The same folded plan on the CPU in float32 (used by tests / smoke as the network's oracle)."""
        if self._cpu_net is None:
            self._cpu_net = MattingNet(self._sd, device="cpu", dtype=torch.float32)
        return self._cpu_net

    @property
    def alpha_pre(self):
        return self._alpha[0]

    @alpha_pre.setter
    def alpha_pre(self, value):
        self._alpha[0] = value

    def reset(self):
        self._alpha = [None, None]

    # ------------------------------------------------------------------ stages
    def _front_device(self, slot, frame, seg, bg_i, bg_always, seg_ready=False):
        if self.seg_agent is not None and not seg_ready:
            seg = self.seg_agent.forward_device(frame, out=slot.seg)
        elif seg_ready:
            slot.seg = seg
        nv.bgdiff_mask(frame, bg_i, bg_always, seg, self.beta, self.thr, out_bg=slot.bg, out_alpha=slot.gate,
                       ws=self.ws_bgdiff)
        if self.object_filter == "host":
            slot.host.copy_(slot.gate, non_blocking=True)
            slot.copied.record()

    def _front_filter_device(self, slot):
        # alphaor = filtered copy of the gate; the gate itself stays (it is alpha_pre of the first frame)
        self.dfilter(slot.alphaor, None, self.center, self.sal_thr, self.cons_thr, src=slot.gate)
        nv.trimap(slot.alphaor, self.long_side, self.ksize, self.iters, out=slot.trimap, ws=self.ws_trimap)
        slot.ready.record()

    def _front_host(self, slot):
        slot.copied.synchronize()
        self.filter(slot.host.numpy(), None, self.center, self.sal_thr, self.cons_thr)
        with self._enqueue_lock:                           # the trimap kernels of two workers share one scratch buffer
            slot.alphaor.copy_(slot.host, non_blocking=True)
            nv.trimap(slot.alphaor, self.long_side, self.ksize, self.iters, out=slot.trimap, ws=self.ws_trimap)
            slot.ready.record()

    def _back(self, slot, frame, want_float=False, lane=0):
        alpha_pre = self._alpha[lane] if self._alpha[lane] is not None else slot.gate       # bg_offline.py:162-163
        nv.vmat_pre(frame, alpha_pre, slot.trimap, self.in_h, self.in_w, layout=self._layout, out=self.x7)
        pred = self.net(self.x7)
        return self._post(slot, frame, pred, want_float, lane)

    def _back2(self, slot_a, slot_b):
        """The back halves of one frame of each of two independent frame ranges with ONE matting forward over both (run_interleaved)."""
        h = self.in_h
        for lane, slot in ((0, slot_a), (1, slot_b)):
            alpha_pre = self._alpha[lane] if self._alpha[lane] is not None else slot.gate
            nv.vmat_pre(slot.frame, alpha_pre, slot.trimap, self.in_h, self.in_w, layout=self._layout, out=self.net2.x8[lane * h:(lane + 1) * h])
        pred = self.net2()
        return [self._post(slot, slot.frame, pred[lane], False, lane) for lane, slot in ((0, slot_a), (1, slot_b))]

    def _post(self, slot, frame, pred, want_float, lane):
        ring = self._out_ring[self._out_i % len(self._out_ring)]
        self._out_i += 1
        if ring["consumed"] is not None:                   # a reader on another stream (note_output_consumed) has finished
            done = ring["consumed"]
            if hasattr(done, "wait_on"):                   # hostio.PendingCopy: a copy a helper thread enqueues (blocks until it has)
                done.wait_on(torch.cuda.current_stream(self.device))
            else:
                torch.cuda.current_stream(self.device).wait_event(done)
            ring["consumed"] = None
        # postprocess + bg[alpha == 0] = frame[alpha == 0] + get_fg (bg_offline.py:165-173) as one launch
        alpha, alpha_f, fg = nv.vmat_post_fg(pred.view(self.in_h, self.in_w), slot.trimap, frame, slot.bg, paste_thr=1,
                                             out_u8=ring["alpha"], out_fg=ring["fg"], want_float=want_float)
        self._alpha[lane] = alpha
        return {"alpha": alpha, "alpha_f32": alpha_f, "fg": fg, "trimap": slot.trimap, "bg": slot.bg,
                "gate": slot.gate, "alphaor": slot.alphaor, "pred": pred, "seg": slot.seg, "packed": ring["packed"], "_ring": ring}

    def note_output_consumed(self, out, event):
        """A consumer that reads out["alpha"] / out["fg"] on another stream hands back the event that ends its reads (or an
        object with wait_on(stream), hostio.PendingCopy); the ring entry is not rewritten (eight frames later) before that event."""
        out["_ring"]["consumed"] = event

    # ------------------------------------------------------------------ synchronous form
    def step(self, frame, seg, bg_i, bg_always, want_float=False):
        """One frame, all stages in order on the current stream (one host sync for the object filter)."""
        slot = self.slots[0]
        self._front_device(slot, frame, seg, bg_i, bg_always)
        if self.object_filter == "device":
            self._front_filter_device(slot)
        else:
            self._front_host(slot)
        return self._back(slot, frame, want_float=want_float)

    # ------------------------------------------------------------------ pipelined form
    def run(self, inputs, on_result=None):
        """inputs: iterable of (frame, seg, bg_i, bg_always[, ready_event]) CUDA tensors; ready_event (optional): the
        event after which the tensors may be read - inputs staged on a copy stream (hostio.HostFrameIO); without it they
        are taken to be produced on the current stream.  Calls on_result(index, dict) per frame (the dict's tensors are
        only valid until two more frames have been submitted).  Returns the frame count."""
        chosen = self.object_filter
        try:
            return self._run(inputs, on_result)
        finally:
            self.object_filter = chosen

    def run_interleaved(self, inputs_a, inputs_b, on_result=None):
        """Two INDEPENDENT frame ranges of equal length (two sub-ranges of this GPU's shard: each starts its own alpha_pre chain, exactly
        like two of the reference's --range runs, tools/unscreen/bg_offline.py:34-35, 184-192) through the same pipelined loop, frame i of
        both ranges in one matting forward (interleave = 2 at construction).  on_result(lane, i, out).  Every frame of a range gets
        the bits BgStepPipeline.run() gives that range on its own; what is gained is launch granularity only (tools/dev/batch2_probe.py:
        6 % of the matting forward).  Returns the number of frames processed (both ranges)."""
        if self.interleave != 2:
            raise ValueError("run_interleaved needs BgStepPipeline(..., interleave=2)")
        if self.seg_agent is not None and self.seg_batch < 4:
            raise ValueError("run_interleaved with a segmentation agent needs seg_batch >= 4 (four frames wait between gate and matting net)")

        def alternate():
            for a, b in zip(inputs_a, inputs_b):
                yield a
                yield b
        chosen = self.object_filter
        try:
            cb = (lambda n, out: on_result(n & 1, n >> 1, out)) if on_result is not None else None
            return self._run(alternate(), cb, lanes=2)
        finally:
            self.object_filter = chosen

    def _run(self, inputs, on_result, lanes=1):
        main = torch.cuda.current_stream(self.device)
        # the pipeline's side streams live as long as it does: the conv graphs are instantiated once per stream they run on
        if getattr(self, "_front_stream", None) is None:
            self._front_stream = torch.cuda.Stream(self.device, priority=getattr(self, "front_priority", 0))
        front = self._front_stream
        work = queue.Queue()
        err = []

        def worker():
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(front):
                while True:
                    slot = work.get()
                    if slot is None:
                        return
                    try:
                        self._front_host(slot)
                    except Exception as e:      # surfaced to the main thread
                        err.append(e)
                    slot.host_done.set()

        on_device = self.object_filter == "device"
        # two host workers: at most two frames are between the gate and the matting net, and at 3840x2160 the border following
        # of one frame (5 ms on one core) is longer than the frame's GPU time; results are still consumed in frame order
        workers = [] if on_device else [threading.Thread(target=worker, daemon=True) for _ in range(2)]
        if workers:
            self.filter(np.zeros((self.H, self.W), np.uint8), None, self.center, self.sal_thr, self.cons_thr)   # score map built once, here
        for th in workers:
            th.start()
        n = 0
        pending = []
        submitted = 0
        try:
            it = iter(inputs)
            ahead = []                                      # inputs whose segmentation mask is already enqueued
            batched = self.seg_agent is not None and self.seg_batch > 1
            if batched and getattr(self, "_seg_stream", None) is None:
                self._seg_stream = torch.cuda.Stream(self.device, priority=getattr(self, "seg_priority", 0))
            seg_stream = self._seg_stream if batched else None
            state = {"exhausted": False, "ngroups": 0}

            def launch_group():
                """Fetch the next seg_batch inputs and enqueue their segmentation forward on the segmentation stream, where it
                runs beside the stage-3 work of the group before it (two MFMA-bound streams fill each other's launch gaps
                and tails).  Three alternating sets of mask buffers: a set is rewritten three groups later, when every
                frame of its group has left the pipeline (the group in flight + the one ahead + at most 2 pending frames)."""
                group = []
                while not state["exhausted"] and len(group) < self.seg_batch:
                    try:
                        group.append(next(it))
                    except StopIteration:
                        state["exhausted"] = True
                if not group:
                    return []
                mset = state["ngroups"] % 3
                first = mset * self.seg_batch
                state["ngroups"] += 1
                group = [tuple(g) if len(g) == 5 else tuple(g) + (None,) for g in group]
                if any(g[4] is None for g in group):
                    seg_stream.wait_stream(main)            # frames produced on the main stream
                for g in group:
                    if g[4] is not None:
                        seg_stream.wait_event(g[4])         # frames staged on a copy stream
                # the masks of the group three back have been read (nothing else orders this stream behind the main stream when
                # the inputs come from a copy stream; a no-op until the set has been used once)
                seg_stream.wait_event(self._seg_set_done[mset])
                with torch.cuda.stream(seg_stream):
                    masks = self.seg_agent.forward_device_batch([g[0] for g in group], outs=self._seg_ring[first:first + len(group)])
                    ev = torch.cuda.Event()
                    ev.record(seg_stream)
                return [(g[0], m, g[2], g[3], ev, g[4], mset if k == len(group) - 1 else None) for k, (g, m) in enumerate(zip(group, masks))]

            nxt = None
            while True:
                while (ahead or nxt or not state["exhausted"]) and len(pending) < 2 * lanes:
                    if not ahead:
                        if batched:
                            ahead = nxt if nxt is not None else launch_group()
                            nxt = launch_group()            # one group ahead
                        else:
                            try:
                                g = tuple(next(it))
                                ahead = [g[:4] + (None, g[4] if len(g) == 5 else None, None)]
                            except StopIteration:
                                state["exhausted"] = True
                        if not ahead:
                            break
                    frame, seg, bg_i, bg_always, seg_ev, up_ev, slot_mset = ahead.pop(0)
                    if seg_ev is not None:
                        front.wait_event(seg_ev)
                    slot = self.slots[submitted % len(self.slots)]
                    slot.frame = frame
                    slot.mset = slot_mset
                    # the slot's buffers (gate, trimap, bg ...) are still read by the back half of the frame three back; with
                    # staged inputs nothing else keeps the front stream behind the main stream
                    front.wait_event(slot.released)
                    if up_ev is not None:
                        front.wait_event(up_ev)             # inputs staged on a copy stream
                    else:
                        front.wait_stream(main)             # inputs produced on the main stream
                    with torch.cuda.stream(front):
                        self._front_device(slot, frame, seg, bg_i, bg_always, seg_ready=batched)
                        if on_device:
                            self._front_filter_device(slot)
                    if not on_device:
                        slot.host_done.clear()
                        work.put(slot)
                    pending.append(slot)
                    submitted += 1
                if not pending:
                    break
                if lanes == 2 and len(pending) < 2 and (ahead or nxt or not state["exhausted"]):
                    continue                                # (the partner frame of the other range is still to be submitted)
                group = [pending.pop(0) for _ in range(min(lanes, len(pending)))]
                for slot in group:
                    if not on_device:
                        slot.host_done.wait()
                    if err:
                        raise err[0]
                    main.wait_event(slot.ready)
                outs = self._back2(*group) if len(group) == 2 else [self._back(group[0], group[0].frame, lane=n % lanes)]
                for slot, out in zip(group, outs):
                    if on_result is not None:
                        on_result(n, out)
                    slot.released.record(main)              # behind whatever on_result enqueued on this stream
                    if slot.mset is not None:
                        self._seg_set_done[slot.mset].record(main)
                    n += 1
        finally:
            for th in workers:
                work.put(None)
            for th in workers:
                th.join()
        return n
