"""Frame I/O that keeps up with the GPU loop (reference unscreen/utils/fileio.py:31-62: a Pool(48) of
cv2.imread that materialises the whole clip in RAM, then cv2.imwrite per frame on the main thread).

    FrameReader   several decoder PROCESSES (JPEG via Pillow, or raw .npy - no codec) write straight into the slots of one
                  shared-memory ring; the consumer gets the slots as torch views of that ring - page-locked (hipHostRegister)
                  when a GPU is present, so hostio.HostFrameIO uploads from them without a staging copy.  Bounded: a slot is
                  handed back with release() and only then reused; at most `depth` frame groups are ever in memory (the
                  reference holds the clip: 1.9 GB at 1080p x 300, 7.5 GB at 4K).
    FrameWriter   the mirror image: the consumer fills a free slot (the download target of the D2H copy), submit() hands it to
                  an encoder process (JPEG / .npy), which frees the slot when the file is written.

Frames are delivered and written in order.  A "group" is what one loop iteration needs: e.g. (frame, segmask, background)
in, (alpha, foreground) out."""
import multiprocessing as mp
import os
import queue
import threading
from multiprocessing import shared_memory

import numpy as np
import torch

from .._io_workers import decode_worker as _decode_worker, encode_worker as _encode_worker, part_offsets

_CTX = mp.get_context("spawn")         # no fork: the parent may hold a HIP context


class _light_children:
    """A spawned child re-imports the parent's main script (as __mp_main__) before it runs its target - for tools/unscreen/*.py that
    is `import torch` + the whole package, seconds per worker.  The workers need none of it (_io_workers.py: numpy + Pillow), so the
    script's path is hidden from multiprocessing while they are started."""

    def __enter__(self):
        import sys
        self.main = sys.modules.get("__main__")
        self.saved = {}
        for k in ("__file__", "__spec__"):
            if self.main is not None and hasattr(self.main, k):
                self.saved[k] = getattr(self.main, k)
                setattr(self.main, k, None)

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(self.main, k, v)


class _Ring:
    def __init__(self, shapes, depth, pin):
        self.shapes = [tuple(s) for s in shapes]
        self.offsets, self.slot_bytes = part_offsets(self.shapes)         # every array of a slot on a 16-byte boundary
        self.depth = depth
        self.nbytes = self.slot_bytes * depth
        self.shm = shared_memory.SharedMemory(create=True, size=self.nbytes)
        self.np = np.ndarray((self.nbytes,), np.uint8, buffer=self.shm.buf)
        self.t = torch.from_numpy(self.np)
        self.pinned = False
        if pin and torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.t.data_ptr(), self.nbytes, 0)
            self.pinned = int(rc) == 0

    def views(self, slot):
        out = []
        for s, o in zip(self.shapes, self.offsets):
            off, n = slot * self.slot_bytes + o, int(np.prod(s))
            out.append(self.t[off:off + n].view(s))
        return out

    def close(self):
        if self.pinned:
            torch.cuda.cudart().cudaHostUnregister(self.t.data_ptr())
            self.pinned = False
        self.t = None
        self.np = None
        try:
            self.shm.close()
            self.shm.unlink()
        except Exception:
            pass


class FrameReader:
    """for slot, tensors in reader: ...; reader.release(slot)

    groups: list of path tuples, one tuple per loop iteration (e.g. (frame_i.jpg, segmask_i.jpg, bg_i.jpg));
    shapes: the array shape of each element of a tuple ((H, W, 3), (H, W), (H, W, 3)); a 2-D shape takes channel 0 of a
    3-channel file.  workers decoder processes, `depth` slots."""

    def __init__(self, groups, shapes, workers=8, depth=32, pin=True, wait_release=None):
        """wait_release: called (no arguments) when the consumer asks for the next group while every slot has been delivered and
        none handed back - it must block until it has release()d at least one slot (hostio.stream_clip waits for the oldest
        upload).  Without it that state raises instead of waiting for ever."""
        self.groups = list(groups)
        self.wait_release = wait_release
        self.outstanding = 0                               # decode tasks handed to the workers and not reported back yet
        self.ring = _Ring(shapes, depth, pin)
        self.tasks, self.done = _CTX.Queue(), _CTX.Queue()
        gray = [len(s) == 2 for s in self.ring.shapes]
        self.procs = [_CTX.Process(target=_decode_worker, daemon=True,
                                   args=(self.ring.shm.name, self.ring.nbytes, self.ring.slot_bytes, self.ring.shapes, gray, self.tasks, self.done))
                      for _ in range(max(1, workers))]
        with _light_children():
            for p in self.procs:
                p.start()
        self.free = list(range(depth))
        self.next_submit = 0
        self.next_deliver = 0
        self.ready = {}
        self._fill()

    def _fill(self):
        while self.free and self.next_submit < len(self.groups):
            slot = self.free.pop()
            self.tasks.put((self.next_submit, slot, self.groups[self.next_submit]))
            self.next_submit += 1
            self.outstanding += 1

    def __iter__(self):
        return self

    def __next__(self):
        if self.next_deliver >= len(self.groups):
            raise StopIteration
        while self.next_deliver not in self.ready:
            if self.outstanding == 0:
                # nothing is being decoded: every slot sits with the consumer.  Only a release() can make progress.
                if self.wait_release is None:
                    raise RuntimeError("FrameReader: all %d slots are delivered and none was released - release() a slot before "
                                       "asking for the next group (or pass wait_release=)" % self.ring.depth)
                before = self.outstanding
                self.wait_release()
                if self.outstanding == before:
                    raise RuntimeError("FrameReader: wait_release() returned without releasing a slot")
                continue
            try:
                seq, slot, err = self.done.get(timeout=2.0)
            except queue.Empty:
                dead = [p.pid for p in self.procs if not p.is_alive()]
                if dead:                                   # a decoder that died (e.g. killed for memory) will never report its task
                    self.close()
                    raise RuntimeError("FrameReader: decoder process(es) %s died with %d task(s) outstanding" % (dead, self.outstanding))
                continue
            self.outstanding -= 1
            if err:
                self.close()
                raise RuntimeError("FrameReader: " + err)
            self.ready[seq] = slot
        slot = self.ready.pop(self.next_deliver)
        self.next_deliver += 1
        return slot, self.ring.views(slot)

    def release(self, slot):
        self.free.append(slot)
        self._fill()

    def close(self):
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self.procs = []
        self.ring.close()

    def __del__(self):
        try:
            if self.procs:
                self.close()
        except Exception:
            pass


class FrameWriter:
    """slot, tensors = writer.acquire(); <fill tensors (e.g. as D2H targets), make sure the copies have finished>;
    writer.submit(slot, paths); ... writer.close() waits for every file."""

    def __init__(self, shapes, workers=8, depth=32, pin=True):
        self.ring = _Ring(shapes, depth, pin)
        self.tasks, self.done = _CTX.Queue(), _CTX.Queue()
        self.procs = [_CTX.Process(target=_encode_worker, daemon=True,
                                   args=(self.ring.shm.name, self.ring.nbytes, self.ring.slot_bytes, self.ring.shapes, self.tasks, self.done))
                      for _ in range(max(1, workers))]
        with _light_children():
            for p in self.procs:
                p.start()
        self.free = list(range(depth))
        self.inflight = 0
        self.errors = []
        self.check = None                                 # optional callable that raises a pending failure of whoever feeds submit()
        self.pending = None                               # optional callable: slots that someone (DeferredSubmit) still holds and will hand over
        self._lock = threading.Lock()                     # submit() / give_back() may come from DeferredSubmit's thread

    def _reap(self, block, timeout=None):
        try:
            slot, err = self.done.get(block=block, timeout=timeout)
        except queue.Empty:
            return False
        with self._lock:
            self.inflight -= 1
        self.free.append(slot)
        if err:
            self.errors.append(err)
        return True

    def acquire(self):
        """A free slot.  Blocks while every slot is being encoded - but never for ever: the wait is cut into 2 s pieces, and between them
        an encoder's error, a dead encoder process, a failure `check` reports (DeferredSubmit: a copy that failed before its slot reached
        the encoders) or "no slot is with the encoders at all" (slots acquired and never submitted or given back) raise."""
        while self._reap(False):
            pass
        while not self.free:
            self._raise_errors()
            if self.inflight == 0 and (self.pending is None or self.pending() == 0) and self.done.empty():
                raise RuntimeError("FrameWriter: all %d slots are acquired and none is being written - submit() or give_back() a slot "
                                   "before asking for the next" % self.ring.depth)
            if not self._reap(True, timeout=2.0):
                dead = [p.pid for p in self.procs if not p.is_alive()]
                if dead:
                    raise RuntimeError("FrameWriter: encoder process(es) %s died with %d slot(s) outstanding" % (dead, self.inflight))
        self._raise_errors()
        slot = self.free.pop()
        return slot, self.ring.views(slot)

    def _raise_errors(self):
        if self.check is not None:
            self.check()
        if self.errors:
            raise RuntimeError("FrameWriter: " + self.errors[0])

    def give_back(self, slot, err=None):
        """A slot that was acquired but will not be written (its copy failed, or an earlier failure stopped the hand-over): it comes back
        through the encoders' reply queue (thread-safe, and the one place acquire() waits on), with `err` recorded like an encoder's."""
        with self._lock:
            self.inflight += 1
        self.done.put((slot, err))

    def submit(self, slot, paths):
        with self._lock:
            self.inflight += 1
        self.tasks.put((slot, tuple(paths)))

    def close(self):
        while self.inflight:
            self._reap(True)
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self.procs = []
        self.ring.close()
        if self.errors:
            raise RuntimeError("FrameWriter: " + self.errors[0])

    def __del__(self):
        try:
            if self.procs:
                self.close()
        except Exception:
            pass


class DeferredSubmit:
    """Hands a writer slot to the encoders once the GPU copies into it have finished, without blocking the frame loop: a helper
    thread waits on the copy's event.  After the first failure (a failed copy, a submit() that raises) nothing more is written, every
    later slot goes straight back to the writer (`give_back`), and the error is raised by the NEXT submit_after() / writer.acquire() on
    the loop's thread - not only by close(), which a loop blocked in acquire() would never reach."""

    def __init__(self, writer):
        self.writer = writer
        self.q = queue.Queue()
        self.error = None                      # the first exception of the helper thread (a failed copy, a dead encoder)
        self.outstanding = 0                   # slots handed to submit_after() that the helper thread has not passed on yet
        self._count = threading.Lock()
        if hasattr(writer, "check"):
            writer.check = self.raise_pending
            writer.pending = lambda: self.outstanding
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _give_back(self, slot, err):
        give_back = getattr(self.writer, "give_back", None)
        if give_back is not None:
            try:
                give_back(slot, err)
            except Exception:                  # noqa: BLE001 - the first error is the one that is reported
                pass

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            event, slot, paths = item
            if self.error is not None:
                self._give_back(slot, None)    # (keep draining the queue: close() must not block on it)
            else:
                try:
                    event.synchronize()
                    self.writer.submit(slot, paths)
                except BaseException as e:     # noqa: BLE001 - handed to the loop's thread
                    self.error = e
                    self._give_back(slot, None)
            with self._count:
                self.outstanding -= 1

    def raise_pending(self):
        if self.error is not None:
            raise RuntimeError("DeferredSubmit: handing a finished frame to the writer failed") from self.error

    def submit_after(self, event, slot, paths):
        self.raise_pending()
        with self._count:
            self.outstanding += 1
        self.q.put((event, slot, paths))

    def close(self):
        self.q.put(None)
        self.th.join()
        if hasattr(self.writer, "check"):
            self.writer.check = self.writer.pending = None
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError("DeferredSubmit: handing a finished frame to the writer failed") from err


class AsyncSaver:
    """What the green / bg tools do with a finished frame, off the frame loop.  The reference writes three or four JPEGs per frame with
    cv2.imwrite on the main thread (tools/unscreen/green.py:119-131, tools/unscreen/bg.py:93-103: ~25 ms of encoding per 1080p frame, ten
    times the GPU's share of the frame); here save() copies the frame's device tensors into a free slot of a FrameWriter ring on the
    CURRENT stream (page-locked: the copies are asynchronous and ordered behind the kernels that produce the tensors, ahead of the next
    frame's), and a helper thread hands the slot to the encoder processes once the copies have landed.  The files are the ones
    save_img() writes from the same arrays.  paths[i] = None: array i is not written for this frame."""

    def __init__(self, shapes, workers=8, depth=12):
        self.writer = FrameWriter(shapes, workers=workers, depth=depth)
        self.deferred = DeferredSubmit(self.writer)

    def save(self, tensors, paths):
        self.deferred.raise_pending()                          # a frame that could not be handed over fails the NEXT save, not only close()
        slot, views = self.writer.acquire()                    # (blocks only when every slot is still being encoded; bounded, see acquire)
        for v, t, p in zip(views, tensors, paths):
            if p is not None:
                v.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.deferred.submit_after(ev, slot, tuple(paths))

    def close(self):
        """Waits for every file; re-raises the first failure of a copy or an encoder."""
        try:
            self.deferred.close()
        finally:
            self.writer.close()


class AsyncLoader:
    """for tensors in AsyncLoader(groups, shapes, device): the files of each group as DEVICE tensors, in order.  Decoder processes fill the
    slots of a page-locked ring (FrameReader), each slot goes up with asynchronous copies on the current stream and is handed back to the
    decoders once its copies have landed - the reference reads every file with cv2.imread on the loop's own thread
    (tools/replace/replace.py:32-35, 61-64).  At most `depth` groups are decoded ahead."""

    def __init__(self, groups, shapes, device, workers=8, depth=16):
        self.device = torch.device(device)
        self.pending = []                                      # (event after a slot's uploads, slot), oldest first
        self.reader = FrameReader(groups, shapes, workers=workers, depth=depth, wait_release=self._wait_oldest)

    def _reap(self):
        while self.pending and self.pending[0][0].query():
            self.reader.release(self.pending.pop(0)[1])

    def _wait_oldest(self):
        ev, slot = self.pending.pop(0)
        ev.synchronize()
        self.reader.release(slot)

    def __iter__(self):
        try:
            for slot, views in self.reader:
                devs = [torch.empty(v.shape, dtype=v.dtype, device=self.device).copy_(v, non_blocking=True) for v in views]
                ev = torch.cuda.Event()
                ev.record()
                self.pending.append((ev, slot))
                self._reap()
                yield devs
        finally:
            self.close()

    def close(self):
        if self.reader is not None:
            torch.cuda.synchronize(self.device)                # (no copy may still read the ring when it is unmapped)
            self.reader.close()
            self.reader = None
