"""Device-resident replace loop: tools/replace/replace.py:28-82 of the reference (BASELINE.json configs[4], the compositing half).

Pass 1 (replace.py:28-41 comp_dx_dy), per frame:  centre of the target's alpha mask and of the source's (resized to the target's
    size): get_center = RGB2GRAY + m10 / m00, m01 / m00 (utils/imgprocess.py:7-14).  The integer moments of every frame land in one
    device table; ONE host read at the end turns them into the reference's int(m10 / (m00 + 1e-6)) centres and their mean offset.
    Across ranks the offsets are sums: one all-reduce of (sum dx, sum dy, count) (tools/replace/replace.py of this repo).
Pass 2 (replace.py:59-82), per frame, ONE launch (usc_replace_compose_u8):
    res = u8(fg' * m' + bg * (1 - m')),  fg', m' = rescale_fg(shift_fg(., dx, dy), 1.2);  compare = [resize(src) | res].
"""
import numpy as np
import torch

from . import _native as nv


class ReplacePipeline:
    def __init__(self, H, W, bg_image, device="cuda:0", scale=1.2, max_frames=1):
        self.H, self.W, self.scale = H, W, scale
        self.device = torch.device(device)
        # replace.py:66 resizes the background to the target's size on every frame; after the first time that is a same-size copy
        self.bg = nv.resize(bg_image.to(self.device), H, W, 1)
        self.moments = torch.zeros((max_frames, 2, 3), dtype=torch.int64, device=self.device)
        self._src_mask = None

    # ---- pass 1
    def centroid_pair(self, row, dst_mask, src_mask):
        """Enqueue get_center(dst_mask) and get_center(adaptive_resize(src_mask, dst_mask)) of frame `row` (no host synchronisation)."""
        if src_mask.shape[:2] != (self.H, self.W):
            if self._src_mask is None:
                self._src_mask = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=self.device)
            src_mask = nv.resize(src_mask, self.H, self.W, 1, out=self._src_mask)
        nv.gray_moments_pair(dst_mask, src_mask, self.moments[row], rgb_order=True)

    def offset_sums(self, n):
        """(sum of src_cx - dst_cx, sum of src_cy - dst_cy, n) over rows 0 .. n-1: the one host read of pass 1."""
        m = self.moments[:n].cpu().numpy().astype(np.float64)                   # (n, 2, 3): m00, m10, m01
        cx = (m[..., 1] / (m[..., 0] + 1e-6)).astype(np.int64)                  # int(m10 / (m00 + 1e-6)), values >= 0
        cy = (m[..., 2] / (m[..., 0] + 1e-6)).astype(np.int64)
        return float((cx[:, 1] - cx[:, 0]).sum()), float((cy[:, 1] - cy[:, 0]).sum()), float(n)

    # ---- pass 2
    @staticmethod
    def shift_matrix(dx, dy):
        return np.float32([[1, 0, dx], [0, 1, dy]]).astype(np.float64)          # shift_fg builds it in float32 (imgprocess.py:62)

    def compose(self, fg, mask, src, M, res=None, compare=None, want_res=True):
        return nv.replace_compose(fg, mask, M, self.bg, src=src, scale=self.scale, res=res, compare=compare, want_res=want_res)

    def compose_chain(self, fg, mask, src, M):
        """The same frame through the stand-alone kernels (what the tool ran before the fused launch; kept as its cross-check)."""
        h, w = self.H, self.W
        fg_s = nv.rescale_fg(nv.warp_affine(fg, M, h, w), self.scale)
        m_s = nv.rescale_fg(nv.warp_affine(mask, M, h, w), self.scale)
        res = nv.blend(fg_s, m_s, self.bg)
        return res, (torch.cat((nv.resize(src, h, w, 1), res), 1) if src is not None else None)


def pack_pair(a, b, pin=True):
    """Two same-shape uint8 images as one (2, H, W, 3) host tensor (page-locked): ReplaceHostLoop moves such a pair with ONE copy
    (a 12.4 MB transfer runs at about twice the rate of two 6.2 MB ones: tools/dev/hostio_probe.py)."""
    t = torch.empty((2,) + tuple(a.shape), dtype=torch.uint8)
    if pin:
        t = t.pin_memory()
    t[0].copy_(a)
    t[1].copy_(b)
    return t


class ReplaceHostLoop:
    """The two passes over a clip whose frames start and end in page-locked host memory: uploads on one copy stream, kernels on a
    compute stream, downloads on a second copy stream, `ring` groups in flight; the target's alpha masks stay in HBM between the
    passes (they are read by both: K x 6.2 MB at 1080p), so a frame crosses PCIe as 12 bytes per pixel up (two masks, fg, source
    frame) and 6 down (the side-by-side image, whose right half IS res - replace.py:79-82 writes both from the same array).
    Per-frame inputs are (a, b) tuples, pack_pair(a, b) tensors, or - fastest - ONE (K, 2, H, W, 3) page-locked tensor per pass
    (`pack_clip`): then `group` frames move per copy (tools/dev/replace_group_probe.py, both passes at 1080p: 1213 / 1238 / 1042 / 2065 / 2043
    frames/s at 1 / 2 / 4 / 8 / 12 frames per copy - from eight frames per group on the uploads run at 51 GB/s with the downloads beside them; the copies themselves are
    as fast at any size (tools/dev/h2d_streams_probe.py): what a larger group buys is fewer event hand-overs per frame for the three-slot ring to hide)."""

    def __init__(self, pipe, nframes, src_hw=None, ring=3, group=8):
        self.pipe, self.K, self.ring, self.group = pipe, nframes, ring, max(1, group)
        H, W, dev = pipe.H, pipe.W, pipe.device
        sh, sw = src_hw if src_hw is not None else (H, W)
        u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)                      # noqa: E731
        G = self.group
        self.bank = u8(nframes, 2, H, W, 3)                  # [i, 0] = the target's alphamask (both passes), [i, 1] = the source's when packed
        self.s_mask = [u8(sh, sw, 3) for _ in range(ring)]
        self.s_in = [u8(G, 2, H, W, 3) for _ in range(ring)] # [fg | source frame] of one size, G frames
        self.s_src = [u8(sh, sw, 3) for _ in range(ring)]
        self.s_cmp = [u8(G, H, 2 * W, 3) for _ in range(ring)]
        self.up, self.comp, self.down = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        mk = lambda: [torch.cuda.Event() for _ in range(ring)]                             # noqa: E731
        self.e_ready, self.e_used, self.e_out = mk(), mk(), mk()

    def _start(self):
        cur = torch.cuda.current_stream()
        for s in (self.up, self.comp, self.down):
            s.wait_stream(cur)

    def _finish(self):
        cur = torch.cuda.current_stream()
        for s in (self.up, self.comp, self.down):
            cur.wait_stream(s)

    def _groups(self, items):
        """(first frame, frame count, clip tensor | None): G frames of a (K, 2, H, W, 3) clip tensor at a time, else one frame per step."""
        n = len(items)
        if torch.is_tensor(items) and items.dim() == 5:
            return [(g, min(self.group, n - g), items) for g in range(0, n, self.group)]
        return [(i, 1, None) for i in range(n)]

    def pass1(self, masks_h):
        """masks_h[i] = (target alphamask (H, W, 3), source alphamask (sh, sw, 3)) or their pack_pair; or one (K, 2, H, W, 3) clip tensor
        -> (sum dx, sum dy, n)."""
        p, R = self.pipe, self.ring
        self._start()
        for step, (g0, cnt, clip) in enumerate(self._groups(masks_h)):
            k = step % R
            if step >= R:
                self.up.wait_event(self.e_used[k])
            with torch.cuda.stream(self.up):
                if clip is not None:
                    self.bank[g0:g0 + cnt].copy_(clip[g0:g0 + cnt], non_blocking=True)
                    sm = None
                elif torch.is_tensor(masks_h[g0]):
                    self.bank[g0].copy_(masks_h[g0], non_blocking=True)
                    sm = None
                else:
                    self.bank[g0, 0].copy_(masks_h[g0][0], non_blocking=True)
                    sm = self.s_mask[k]
                    sm.copy_(masks_h[g0][1], non_blocking=True)
                self.e_ready[k].record(self.up)
            self.comp.wait_event(self.e_ready[k])
            with torch.cuda.stream(self.comp):
                for i in range(g0, g0 + cnt):
                    p.centroid_pair(i, self.bank[i, 0], sm if sm is not None else self.bank[i, 1])
                self.e_used[k].record(self.comp)
        self._finish()
        return p.offset_sums(len(masks_h))

    def pass2(self, inputs_h, M, compare_h):
        """inputs_h[i] = (fg (H, W, 3), source frame (sh, sw, 3)) or their pack_pair, or one (K, 2, H, W, 3) clip tensor; compare_h
        (K, H, 2 W, 3) pinned receives [source | res]."""
        p, R = self.pipe, self.ring
        self._start()
        for step, (g0, cnt, clip) in enumerate(self._groups(inputs_h)):
            k = step % R
            if step >= R:
                self.up.wait_event(self.e_used[k])
                self.comp.wait_event(self.e_out[k])                     # the slot's previous composites have been copied out
            with torch.cuda.stream(self.up):
                if clip is not None:
                    self.s_in[k][:cnt].copy_(clip[g0:g0 + cnt], non_blocking=True)
                    pairs = [(self.s_in[k][j, 0], self.s_in[k][j, 1]) for j in range(cnt)]
                elif torch.is_tensor(inputs_h[g0]):
                    self.s_in[k][0].copy_(inputs_h[g0], non_blocking=True)
                    pairs = [(self.s_in[k][0, 0], self.s_in[k][0, 1])]
                else:
                    fg, fr = self.s_in[k][0, 0], self.s_src[k]
                    fg.copy_(inputs_h[g0][0], non_blocking=True)
                    fr.copy_(inputs_h[g0][1], non_blocking=True)
                    pairs = [(fg, fr)]
                self.e_ready[k].record(self.up)
            self.comp.wait_event(self.e_ready[k])
            with torch.cuda.stream(self.comp):
                for j, (fg, fr) in enumerate(pairs):
                    p.compose(fg, self.bank[g0 + j, 0], fr, M, compare=self.s_cmp[k][j], want_res=False)
                self.e_used[k].record(self.comp)
            self.down.wait_event(self.e_used[k])
            with torch.cuda.stream(self.down):
                compare_h[g0:g0 + cnt].copy_(self.s_cmp[k][:cnt], non_blocking=True)
                self.e_out[k].record(self.down)
        self._finish()


def pack_clip(pairs, pin=True):
    """[(a_i, b_i)] same-shape uint8 images -> one (K, 2, H, W, 3) host tensor (page-locked): ReplaceHostLoop then moves several frames per copy."""
    a0 = pairs[0][0]
    t = torch.empty((len(pairs), 2) + tuple(a0.shape), dtype=torch.uint8)
    if pin:
        t = t.pin_memory()
    for i, (a, b) in enumerate(pairs):
        t[i, 0].copy_(a)
        t[i, 1].copy_(b)
    return t
