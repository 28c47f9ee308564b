"""Host boundary of the frame loops: decoded frames in pinned host memory go in, alpha + foreground
come back in pinned host memory - what the reference's loop does around every agent (`.cuda()` up, `.cpu().numpy()` down:
vmatting/agent.py:154-161, binseg/agent_human.py:295-303) and what its tools do around the loop (frame_list in RAM,
bg_offline.py:36; cv2.imwrite of alpha and fg, :169,175).

Uploads run on their own HIP stream a bounded number of frames ahead of the compute streams, downloads on a third stream
behind them; PCIe is full duplex, so at 1080p (14.5 MB up, 8.3 MB down per frame) both directions hide behind the 3 ms of
GPU work per frame.  Nothing here touches pixels: it is plumbing around BgStepPipeline.run().

Who waits for whom (round 3, tools/dev/d2h_dep_probe.py): a copy costs the conv streams nothing as long as the copy stream's queue
holds nothing but copies - 80 matting forwards with an independent 8.3 MB download each run at 1.224 ms per forward, the same as with
no copy.  The same copy behind `stream.wait_event(compute_event)` costs 0.03 - 0.19 ms per forward: with a barrier packet in front of
it the runtime moves the bytes with a blit KERNEL in that queue, and a blit wave on a CU keeps a one-workgroup-per-CU conv launch off
it.  So the default mode ("hostwait") keeps barrier packets out of both copy streams: a helper thread waits for the compute event on
the HOST and only then enqueues the download; an upload into a staging slot waits on the host for the slot's previous frame (40
frames back), and uploads are PACED - one per finished frame, a look-ahead of two segmentation groups staged when the run starts -
because a burst of queued copies (12 x 50 MB when a group is pulled at 3840 x 2160) books the DMA engines, the runtime then moves
the next copy with a kernel after all, and the download behind the burst took 8 ms instead of 0.6 (tools/dev/hostio_timeline.py).
The event-ordered form is kept as down_mode="own".  Measured (profiles/r03/hostio_ablate*.txt): 1080p 416 frames/s host -> host
against 419 device-resident (event-ordered: 393); 3840 x 2160 368 against 385 (event-ordered: 273)."""
import collections
import queue
import threading
import time

import torch


_BUSY = object()          # staging slot handed to the pipeline, result not collected yet


def pinned_like(shape, dtype=torch.uint8):
    return torch.empty(shape, dtype=dtype).pin_memory()


def _a16(v):
    return (v + 15) & ~15


def in_layout(H, W):
    """Byte offsets of one staged input group [frame | bg_i | seg]: (bg offset, seg offset, total); every part starts on a 16-byte
    boundary (the kernels read 12- and 16-byte groups)."""
    n = H * W
    bg = _a16(3 * n)
    seg = _a16(bg + 3 * n)
    return bg, seg, _a16(seg + n)


def out_layout(H, W):
    """Byte offsets of one output group [alpha | fg]: (fg offset, total) - the layout of BgStepPipeline's output ring."""
    n = H * W
    fg = _a16(n)
    return fg, _a16(fg + 3 * n)


def _host_span(parts, offsets):
    """parts: contiguous uint8 CPU tensors, offsets: where each should start relative to the first.  If they are views of ONE
    storage at exactly those distances, the flat view from the first to the end of the last (one copy moves the group); else None."""
    try:
        base = parts[0].untyped_storage().data_ptr()
        p0 = parts[0].data_ptr()
        for t, o in zip(parts, offsets):
            if t.dtype != torch.uint8 or not t.is_contiguous() or t.untyped_storage().data_ptr() != base or t.data_ptr() != p0 + o:
                return None
        return parts[0].as_strided((offsets[len(parts) - 1] + parts[-1].numel(),), (1,))
    except (RuntimeError, AttributeError):
        return None


def pack_host_group(frame, bg, seg=None, pin=True):
    """One page-locked buffer laid out as in_layout() holding copies of the given uint8 tensors -> (frame, seg | None, bg) views of
    it: HostFrameIO uploads such a group with a single copy."""
    H, W = frame.shape[:2]
    bg_off, seg_off, total = in_layout(H, W)
    buf = torch.zeros(total, dtype=torch.uint8)
    if pin:
        buf = buf.pin_memory()
    n = H * W
    fv, bv = buf[:3 * n].view(H, W, 3), buf[bg_off:bg_off + 3 * n].view(H, W, 3)
    fv.copy_(frame)
    bv.copy_(bg)
    sv = None
    if seg is not None:
        sv = buf[seg_off:seg_off + n].view(H, W)
        sv.copy_(seg)
    return fv, sv, bv


def packed_outputs(K, H, W, pin=True):
    """(alpha (K, H, W), fg (K, H, W, 3)) views of one page-locked (K, out_layout total) buffer: frame k's alpha and fg sit as in the
    pipeline's output ring, so HostFrameIO.collect() fetches them with a single copy."""
    fg_off, total = out_layout(H, W)
    buf = torch.empty((K, total), dtype=torch.uint8)
    if pin:
        buf = buf.pin_memory()
    n = H * W
    return buf[:, :n].unflatten(1, (H, W)), buf[:, fg_off:fg_off + 3 * n].unflatten(1, (H, W, 3))


class PendingCopy:
    """Completion of a copy that a helper thread enqueues once a compute event has fired.  Stands in for the copy's event:
    query() / synchronize() as torch.cuda.Event, wait_on(stream) for a stream that has to run after the copy."""

    def __init__(self):
        self._issued = threading.Event()
        self._event = None
        self._error = None

    def _set(self, event=None, error=None):
        self._event, self._error = event, error
        self._issued.set()

    def _get(self):
        self._issued.wait()
        if self._error is not None:
            raise RuntimeError("HostFrameIO: the download thread failed") from self._error
        return self._event

    def query(self):
        return self._issued.is_set() and self._get().query()

    def synchronize(self):
        self._get().synchronize()

    def wait_on(self, stream):
        if not self._issued.is_set():
            t0 = time.perf_counter()
            ev = self._get()
            PendingCopy.waited += time.perf_counter() - t0       # how long loops blocked on copies not yet enqueued (diagnostics)
        else:
            ev = self._get()
        stream.wait_event(ev)

    waited = 0.0


class HostFrameIO:
    """Ring of device staging slots + copy streams around a pipeline's run().

        io = HostFrameIO(pipe, depth)
        pipe.run(io.inputs(host_frames, bg_always_dev), on_result=io.collect(alpha_host, fg_host))
        io.drain()

    host_frames yields (frame, seg | None, bg_i) pinned uint8 CPU tensors ((H, W, 3), (H, W) | None, (H, W, 3));
    alpha_host[i] / fg_host[i] are pinned CPU tensors that receive frame i's outputs.  depth must exceed the number of
    frames the pipeline holds in flight (run() looks two segmentation groups + two frames ahead)."""

    def __init__(self, pipe, depth=None, down_mode="hostwait"):
        """down_mode: how the device -> host copies of a frame are ordered behind the frame's kernels (tools/dev/hostio_ablate.py
        measures them): "hostwait" = a helper thread waits for the compute event, then enqueues the copy on a download stream that
        never sees a barrier packet, and an upload waits on the host for its staging slot (module docstring);
        "own" = on the download stream behind `wait_event(compute event)`; "primed" = the same with a 64-byte host -> device copy in front of
        them; "shared" = on the upload stream (one copy stream for both directions); "exclusive" = two streams, but a download waits
        for the last upload issued and the next upload for the last download (one PCIe direction at a time)."""
        self.pipe = pipe
        self.down_mode = down_mode
        self._tiny_h = torch.zeros(64, dtype=torch.uint8).pin_memory()
        self._tiny_d = torch.zeros(64, dtype=torch.uint8, device=pipe.device)
        dev = pipe.device
        H, W = pipe.H, pipe.W
        # "hostwait": `lookahead` frames are staged ahead of the frame loop (two segmentation groups + the frames between gate and
        # matting net + slack), and a slot is rewritten `depth` frames later - by then its frame must have left the GPU, which trails
        # the host by up to the pipeline's output ring (8): depth = lookahead + 12.  Event-ordered modes stage on demand.
        paced = down_mode == "hostwait"
        self.depth = depth if depth is not None else (2 * pipe.seg_batch + 16 if paced else 2 * pipe.seg_batch + 6)
        self.lookahead = max(1, min(2 * pipe.seg_batch + 4, self.depth - 10))
        self._feed = None
        n = H * W
        # one staging buffer per slot, [frame | bg_i | seg] (in_layout): host groups laid out the same way (pack_host_group) arrive
        # with ONE copy
        self._bg_off, self._seg_off, total = in_layout(H, W)
        self.stage = [torch.empty(total, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        self.frames = [t[:3 * n].view(H, W, 3) for t in self.stage]
        self.bgs = [t[self._bg_off:self._bg_off + 3 * n].view(H, W, 3) for t in self.stage]
        self.segs = [t[self._seg_off:self._seg_off + n].view(H, W) for t in self.stage]
        self.done = [None] * self.depth              # event: the frame staged in this slot has left the pipeline
        self.up = torch.cuda.Stream(dev)             # (normal priority: high-priority copy streams measured 4 % slower at 3840 x 2160)
        self.down = torch.cuda.Stream(dev)
        self._n_in = 0
        self._last_down = None
        self.slot_waited = 0.0                       # seconds inputs() blocked for a staging slot (diagnostics)
        self._jobs = None                            # "hostwait": (compute event, [(dst, src)], PendingCopy) for the download thread
        self._thread = None
        self._error = None                           # the first exception of the download thread: re-raised by drain() and close()

    # ------------------------------------------------------------------ "hostwait": the download thread
    def _download_loop(self):
        torch.cuda.set_device(self.pipe.device)
        while True:
            job = self._jobs.get()
            if job is None:
                self._jobs.task_done()
                return
            ev, pairs, pend = job
            try:
                ev.synchronize()                     # on the host: the download stream's queue holds copies only
                with torch.cuda.stream(self.down):
                    for dst, src in pairs:
                        dst.copy_(src, non_blocking=True)
                    dn = torch.cuda.Event()
                    dn.record(self.down)
                pend._set(dn)
            except BaseException as e:               # handed to whoever waits on the copy - and kept: nothing waits on a clip's last copies
                pend._set(error=e)
                if self._error is None:
                    self._error = e
            finally:
                self._jobs.task_done()

    def _enqueue_download(self, ev, pairs):
        if self._thread is None or not self._thread.is_alive():
            self._jobs = queue.Queue()
            self._thread = threading.Thread(target=self._download_loop, name="usc-download", daemon=True)
            self._thread.start()
        pend = PendingCopy()
        self._jobs.put((ev, pairs, pend))
        return pend

    def _raise_pending(self):
        if self._error is not None:
            err, self._error = self._error, None
            raise RuntimeError("HostFrameIO: the download thread failed") from err

    def close(self):
        """Stops the download thread (after the copies it still holds); raises what it failed with, if anything."""
        if self._thread is not None and self._thread.is_alive():
            self._jobs.put(None)
            self._thread.join()
        self._thread = None
        self._raise_pending()

    def inputs(self, host_frames, bg_always, on_uploaded=None):
        """Generator for BgStepPipeline.run(): uploads each host frame into the next staging slot on the upload stream and
        yields (frame, seg, bg_i, bg_always, uploaded_event) device tensors.  host_frames yields (frame, seg | None, bg_i) or
        (frame, seg | None, bg_i, token); on_uploaded(event, token) is told when the host buffers may be reused (a FrameReader
        slot to release).

        "hostwait" paces the uploads: `lookahead` frames go up when the run starts, after that ONE per finished frame (collect()'s
        callback feeds the next), so the pipeline's pulls - 12 frames at once for a segmentation group - find their frames staged and
        never put a burst of copies in front of a download (tools/dev/hostio_timeline.py: with 12 x 50 MB uploads queued at
        3840 x 2160 a download took 8 ms instead of 0.6 and the conv kernels beside it ran at half speed - the runtime falls back to
        copy kernels when the DMA engines are booked)."""
        it = iter(host_frames)
        staged = collections.deque()

        def stage_one():
            try:
                item = next(it)
            except StopIteration:
                return False
            frame_h, seg_h, bg_h = item[:3]
            i = self._n_in % self.depth
            self._n_in += 1
            if self.done[i] is _BUSY:
                raise RuntimeError("HostFrameIO: depth %d is smaller than the number of frames the pipeline holds in flight" % self.depth)
            if self.done[i] is not None:             # the slot's previous frame (depth frames back) has been consumed
                if self.down_mode == "hostwait":
                    if not self.done[i].query():
                        t0 = time.perf_counter()
                        self.done[i].synchronize()   # on the host: no barrier packet in the upload stream's queue
                        self.slot_waited += time.perf_counter() - t0
                else:
                    self.up.wait_event(self.done[i])
            if self.down_mode == "exclusive" and self._last_down is not None:
                self.up.wait_event(self._last_down)  # never an upload beside a download (one PCIe direction at a time)
            self.done[i] = _BUSY
            span = _host_span([frame_h, bg_h] + ([seg_h] if seg_h is not None else []), [0, self._bg_off, self._seg_off])
            with torch.cuda.stream(self.up):
                if span is not None:                 # the group is one contiguous run of host memory in the staging order
                    self.stage[i][:span.numel()].copy_(span, non_blocking=True)
                else:
                    self.frames[i].copy_(frame_h, non_blocking=True)
                    self.bgs[i].copy_(bg_h, non_blocking=True)
                    if seg_h is not None:
                        self.segs[i].copy_(seg_h, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.up)
            self._last_up = ev
            if on_uploaded is not None:
                on_uploaded(ev, item[3] if len(item) > 3 else None)
            staged.append((self.frames[i], self.segs[i] if seg_h is not None else None, self.bgs[i], bg_always, ev))
            return True

        if self.down_mode != "hostwait":             # event-ordered modes: a frame goes up when the pipeline asks for it
            while stage_one():
                yield staged.popleft()
            return
        self._feed = stage_one                       # collect() calls it once per finished frame
        try:
            for _ in range(self.lookahead):
                if not stage_one():
                    break
            while staged or stage_one():             # (a consumer that never reports results just gets its frames on demand)
                yield staged.popleft()
        finally:
            if self._feed is stage_one:              # (an abandoned generator may be finalised after the next run has started)
                self._feed = None

    def collect(self, alpha_host=None, fg_host=None, target=None, on_downloaded=None):
        """on_result callback for run(): copies frame n's alpha and fg to alpha_host[n] / fg_host[n] - or to the pair of host
        tensors target(n) returns, with a token as third element - on the download stream and marks the frame's staging slot free;
        on_downloaded(event, token, n) is told when the copies have been enqueued (event = their completion).  The pipeline's
        outputs live in its own ring of eight, whose reuse waits for these copies."""
        main = torch.cuda.current_stream(self.pipe.device)
        base = self._n_in                            # inputs() of this run has not started yet: slots continue from here

        def on_result(n, out):
            ev = torch.cuda.Event()
            ev.record(main)
            self.done[(base + n) % self.depth] = ev  # frames are consumed in order: the slot input n was staged in
            if target is not None:
                a_h, f_h, token = target(n)
            else:
                a_h, f_h, token = alpha_host[n], fg_host[n], None
            span = _host_span([a_h, f_h], [0, out["fg"].data_ptr() - out["alpha"].data_ptr()]) if "packed" in out else None
            if self.down_mode == "hostwait":
                pairs = [(span, out["packed"][:span.numel()])] if span is not None else [(a_h, out["alpha"]), (f_h, out["fg"])]
                dn = self._enqueue_download(ev, pairs)
                self.pipe.note_output_consumed(out, dn)
                self._last_down = dn
                if on_downloaded is not None:
                    on_downloaded(dn, token, n)
                if self._feed is not None:
                    self._feed()                     # one frame finished, one more goes up (inputs())
                return
            down = self.up if self.down_mode == "shared" else self.down
            down.wait_event(ev)
            if self.down_mode == "exclusive" and getattr(self, "_last_up", None) is not None:
                down.wait_event(self._last_up)
            with torch.cuda.stream(down):
                if self.down_mode == "primed":
                    self._tiny_d.copy_(self._tiny_h, non_blocking=True)
                if span is not None:                 # alpha and fg are neighbours on the host as they are in the pipeline's ring entry
                    span.copy_(out["packed"][:span.numel()], non_blocking=True)
                else:
                    a_h.copy_(out["alpha"], non_blocking=True)
                    f_h.copy_(out["fg"], non_blocking=True)
                dn = torch.cuda.Event()
                dn.record(down)
            self.pipe.note_output_consumed(out, dn)
            self._last_down = dn
            if on_downloaded is not None:
                on_downloaded(dn, token, n)
        return on_result

    def drain(self):
        """Every download has been enqueued and has finished.  A failure of the download thread surfaces here even when no later
        frame waited on the failed copy (the last frames of a clip)."""
        if self._jobs is not None:
            self._jobs.join()                        # every download has been enqueued
        self.up.synchronize()
        self.down.synchronize()
        self._raise_pending()


def stream_clip(pipe, reader, writer, out_paths, bg_always, depth=None):
    """bg_step stage 3 over files: FrameReader (decoder processes -> page-locked shared-memory slots) -> uploads -> the pipelined
    frame loop -> downloads -> FrameWriter (encoder processes), all bounded and overlapped; nothing but queue bookkeeping runs on
    the loop's thread.  reader yields (frame, bg_i, segmask) groups - the order of the staging layout, so a reader slot goes up with
    one copy and a writer slot [alpha | fg] comes down with one; out_paths(n) -> (alpha path, fg path).  Returns the frame count."""
    import collections
    from .utils.frameio import DeferredSubmit
    io = HostFrameIO(pipe, depth)
    deferred = DeferredSubmit(writer)
    uploaded = collections.deque()                     # (event, reader slot) in upload order

    def reap():
        while uploaded and uploaded[0][0].query():
            reader.release(uploaded.popleft()[1])

    def wait_release():
        # the reader has no free slot and no decode in flight (the host ran ahead of the GPU: nothing in the loop synchronises
        # since the object filter moved to the device): block on the OLDEST upload, then hand its slot back
        if not uploaded:
            raise RuntimeError("stream_clip: the reader is out of slots but no upload is pending (reader depth %d is smaller than "
                               "the frames the pipeline looks ahead)" % reader.ring.depth)
        uploaded[0][0].synchronize()
        reap()

    reader.wait_release = wait_release

    def host_frames():
        for slot, (f, b, m) in reader:
            reap()
            yield f, m, b, slot

    def target(n):
        slot, (a, f) = writer.acquire()
        return a, f, slot

    try:
        n = pipe.run(io.inputs(host_frames(), bg_always, on_uploaded=lambda ev, slot: uploaded.append((ev, slot))),
                     on_result=io.collect(target=target, on_downloaded=lambda ev, slot, k: deferred.submit_after(ev, slot, out_paths(k))))
        io.drain()
        while uploaded:
            uploaded[0][0].synchronize()
            reader.release(uploaded.popleft()[1])
    except BaseException:
        # the helper threads are stopped either way; their own errors must not mask the one that is already propagating
        for closer in (deferred.close, io.close):
            try:
                closer()
            except Exception:      # noqa: BLE001
                pass
        raise
    deferred.close()               # (raises if a frame could not be handed to the writer)
    io.close()
    return n
