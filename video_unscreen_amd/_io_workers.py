"""Decoder / encoder processes of utils/frameio.py.  Lives at the package top (not under utils/) and imports numpy + Pillow only:
a spawned worker is up in ~0.2 s instead of the seconds `import torch` takes (sixteen of them at once on the box's CPU share)."""
from multiprocessing import shared_memory

import numpy as np

from ._imgcodec import read_img, save_img


def part_offsets(shapes):
    """Byte offset of every array of a slot + the slot size: each array starts on a 16-byte boundary (the layout hostio.in_layout /
    out_layout give the device staging buffers, so a whole slot moves with one copy)."""
    offs, o = [], 0
    for s in shapes:
        offs.append(o)
        o = (o + int(np.prod(s)) + 15) & ~15
    return offs, o


def _attach(name, nbytes):
    shm = shared_memory.SharedMemory(name=name)
    return shm, np.ndarray((nbytes,), np.uint8, buffer=shm.buf)


def decode_worker(shm_name, nbytes, slot_bytes, shapes, gray, tasks, done):
    shm, ring = _attach(shm_name, nbytes)
    try:
        while True:
            t = tasks.get()
            if t is None:
                break
            seq, slot, paths = t
            try:
                offs, _ = part_offsets(shapes)
                for p, shp, g, o in zip(paths, shapes, gray, offs):
                    n = int(np.prod(shp))
                    off = slot * slot_bytes + o
                    img = read_img(p)
                    if g and img.ndim == 3:
                        img = img[..., 0]                                  # masks stored as 3-channel JPEGs (bg_offline.py:134-141)
                    if img.shape != tuple(shp):
                        raise ValueError("%s: shape %s, expected %s" % (p, img.shape, tuple(shp)))
                    ring[off:off + n] = np.ascontiguousarray(img).reshape(-1)
                done.put((seq, slot, None))
            except Exception as e:                                         # surfaced to the consumer
                done.put((seq, slot, repr(e)))
    finally:
        del ring
        shm.close()


def encode_worker(shm_name, nbytes, slot_bytes, shapes, tasks, done):
    shm, ring = _attach(shm_name, nbytes)
    try:
        while True:
            t = tasks.get()
            if t is None:
                break
            slot, paths = t
            err = None
            try:
                offs, _ = part_offsets(shapes)
                for p, shp, o in zip(paths, shapes, offs):
                    if p is None:                                          # (this array of the slot is not written for this frame)
                        continue
                    n = int(np.prod(shp))
                    off = slot * slot_bytes + o
                    save_img(ring[off:off + n].reshape(shp), p)
            except Exception as e:
                err = repr(e)
            done.put((slot, err))
    finally:
        del ring
        shm.close()
