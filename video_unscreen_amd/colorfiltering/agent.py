"""ColorFilteringAgent: per-channel HSV mixture models of screen and foreground -> alpha (reference
unscreen/colorfiltering/agent.py:9-354, same constructor / method names / return values).

Split by where the work is:
  * fitting (get_color_prior, fit_bg_gmms, fit_fg_gmms): <= 10 000 scalar samples per channel through sklearn's EM,
    exactly the reference's dependency - host work on the 540x960 HSV image, only on training frames
    (every `colorfiltering_update_duration` frames);
  * inference on every frame (HIP, libunscreen_hip.so): BGR->HSV, bilinear resize, the mixture likelihood ratio
    (each 1-D mixture collapses to a 256-entry table: usc_gmm_lut_host + usc_gmm_alpha_u8), the adaptive threshold
    (usc_cf_threshold_u8), close/open (usc_morph_u8) and the resize back.
"""
import numpy as np
import torch

from .. import _native as nv
from ..utils._dev import to_dev, to_host


class _FitPool:
    """Six processes that run GaussianMixture.fit (video_unscreen_amd/_fit_worker.py); started at the first training frame."""

    def __init__(self, n):
        self.n = n
        self.conns, self.procs = [], []
        self._start()

    def _start(self):
        import multiprocessing as mp
        from .._fit_worker import fit_worker
        from ..utils.frameio import _light_children
        ctx = mp.get_context("spawn")                         # no fork: the parent holds a HIP context
        limits = {}
        try:                                                  # the thread counts sklearn's OpenMP / BLAS code sees in THIS process (torch has set them)
            import sklearn.cluster                            # noqa: F401
            from threadpoolctl import threadpool_info
            for info in threadpool_info():
                api = info.get("user_api")
                if api in ("openmp", "blas"):
                    limits[api] = min(limits.get(api, 1 << 30), int(info["num_threads"]))
        except Exception:                                     # noqa: BLE001 - without threadpoolctl the workers keep their defaults
            limits = {}
        with _light_children():
            for _ in range(self.n):
                a, b = ctx.Pipe()
                p = ctx.Process(target=fit_worker, args=(b, limits), daemon=True)
                p.start()
                b.close()
                self.conns.append(a)
                self.procs.append(p)

    def _fail(self, msg):
        """An answer is missing or an error came back: the other workers of the round may still hold (or be about to send) THIS round's
        replies, which a caller that catches the error and fits again would read as the next round's.  The pool is torn down (pipes
        closed, processes ended) and started afresh before the error is raised, so no stale reply survives."""
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        self.close()
        self._start()
        raise RuntimeError(msg)

    def fit(self, jobs, timeout=120.0):
        """jobs: [(mixture, packed samples, stride)] -> the fitted mixtures, in order (job i on worker i % n)."""
        n = len(self.conns)
        out = [None] * len(jobs)
        for base in range(0, len(jobs), n):
            chunk = jobs[base:base + n]
            for i, job in enumerate(chunk):
                self.conns[i].send(job)
            for i in range(len(chunk)):
                if not self.conns[i].poll(timeout):
                    self._fail("ColorFilteringAgent: an EM worker process did not answer within %g s (alive: %s)" % (timeout, self.procs[i].is_alive()))
                try:
                    gmm, err = self.conns[i].recv()
                except (EOFError, OSError) as e:
                    self._fail("ColorFilteringAgent: an EM worker process went away (%r)" % (e,))
                if err:
                    self._fail("ColorFilteringAgent: GaussianMixture.fit failed in a worker: " + err)
                out[base + i] = gmm
        return out

    def close(self):
        for c in self.conns:
            try:
                c.send(None)
                c.close()
            except Exception:                                 # noqa: BLE001
                pass
        for p in self.procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        self.conns, self.procs = [], []


class ColorFilteringAgent():

    def __init__(self, input_long_side=960, bg_ncomp=(3, 5, 5), fg_ncomp=(10, 10, 10), max_num_samples=10000,
                 color_prior_winsize=30, use_opencv_gmm=False, random_state=None, fit_workers=6):
        """fit_workers (not in the reference): processes the six independent EM fits of a training round run in (0: in this
        interpreter, one after the other, as the reference).  Same results either way."""
        assert isinstance(input_long_side, int)
        assert len(bg_ncomp) == 3 and len(fg_ncomp) == 3
        assert isinstance(max_num_samples, int) and max_num_samples > 2
        assert isinstance(color_prior_winsize, int) and color_prior_winsize > 0
        if use_opencv_gmm:
            raise NotImplementedError("use_opencv_gmm=True needs cv2.ml.EM; this build fits with sklearn (the reference default)")
        self.input_long_side = input_long_side
        self.bg_ncomp = bg_ncomp
        self.fg_ncomp = fg_ncomp
        self.max_num_samples = max_num_samples
        self.color_prior_winsize = color_prior_winsize
        self.use_opencv_gmm = use_opencv_gmm
        self.random_state = random_state          # None = unseeded, as the reference (agent.py:101-110)
        self.fit_workers = int(fit_workers)
        self._pool = None
        self._luts = None
        self._ws = None
        self._cnt2 = None
        self._bg_cache = None
        self.last_bg_color = None
        self.reset_gmms()

    def is_trained(self):
        return self._is_trained

    def reset_gmms(self):
        """Fresh spherical, warm-started mixtures for the three HSV channels of screen and foreground (agent.py:81-111)."""
        from sklearn import mixture
        mk = lambda k: mixture.GaussianMixture(n_components=k, covariance_type='spherical', warm_start=True,  # noqa: E731
                                               random_state=self.random_state)
        self.bg_gmms = [mk(self.bg_ncomp[i]) for i in range(3)]
        self.fg_gmms = [mk(self.fg_ncomp[i]) for i in range(3)]
        self._is_trained = False
        self._luts = None

    # ------------------------------------------------------------------ host: sampling + EM (training frames only)
    def _subsample(self, samples):
        if len(samples) > self.max_num_samples:
            samples = samples[::len(samples) // self.max_num_samples]
        return samples

    def get_color_prior(self, img_hsv, mask, color_prior_winsize=None):
        """Pixels whose hue lies strictly inside (peak - w//2, peak + w//2) of the masked hue histogram (agent.py:113-149)."""
        if color_prior_winsize is None:
            color_prior_winsize = self.color_prior_winsize
        hue = img_hsv[:, :, 0]
        samples = self._subsample(hue[mask].astype(np.float64))
        hist, _ = np.histogram(samples, 256, [0, 256])
        peak = int(np.argmax(hist))
        half = color_prior_winsize // 2
        return (hue.astype(np.int64) > peak - half) & (hue.astype(np.int64) < peak + half)

    def _samples(self, img_hsv, mask):
        return [self._subsample(img_hsv[:, :, i][mask].astype(np.float64))[..., np.newaxis] for i in range(3)]

    def _fit(self, gmms, img_hsv, mask):
        for i, samples in enumerate(self._samples(img_hsv, mask)):
            gmms[i].fit(samples)
        self._is_trained = True
        self._luts = None

    def _fg_mask(self, mask, mask_by_prior):
        keep = mask & ~mask_by_prior
        return keep if keep.sum() > max(self.fg_ncomp) * 5 else mask

    def fit_both(self, img_hsv, bg_mask, bg_prior, fg_mask, fg_prior):
        """fit_bg_gmms(img_hsv, bg_mask, bg_prior) then fit_fg_gmms(img_hsv, fg_mask, fg_prior) - the six fits are independent of each
        other, so with fit_workers > 0 they run side by side in the worker processes (the fitted objects replace bg_gmms / fg_gmms)."""
        if self.fit_workers <= 0:
            self.fit_bg_gmms(img_hsv, bg_mask, bg_prior)
            self.fit_fg_gmms(img_hsv, fg_mask, fg_prior)
            return
        if self._pool is None:
            self._pool = _FitPool(self.fit_workers)
        def job(gmm, view):                                   # (mixture, samples packed, stride in elements of the parent's view)
            return gmm, np.ascontiguousarray(view[:, 0]), max(1, view.strides[0] // view.itemsize)
        jobs = [job(g, v) for g, v in zip(self.bg_gmms, self._samples(img_hsv, bg_mask & bg_prior))] + \
            [job(g, v) for g, v in zip(self.fg_gmms, self._samples(img_hsv, self._fg_mask(fg_mask, fg_prior)))]
        # the slowest first (ten components against three / five): with fewer workers than jobs the long ones do not queue behind short ones
        fitted = self._pool.fit(jobs[3:] + jobs[:3])
        self.fg_gmms, self.bg_gmms = list(fitted[:3]), list(fitted[3:])
        self._is_trained = True
        self._luts = None

    def close(self):
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                     # noqa: BLE001
            pass

    def fit_bg_gmms(self, img_hsv, mask, mask_by_prior=None):
        """Screen model from (coarse background) & (hue prior) (agent.py:151-176)."""
        if mask_by_prior is None:
            mask_by_prior = self.get_color_prior(img_hsv, mask)
        self._fit(self.bg_gmms, img_hsv, mask & mask_by_prior)

    def fit_fg_gmms(self, img_hsv, mask, mask_by_prior=None):
        """Foreground model; screen-hued pixels are dropped when enough others remain (agent.py:178-199)."""
        if mask_by_prior is None:
            mask_by_prior = self.get_color_prior(img_hsv, ~mask, self.color_prior_winsize // 5)
        self._fit(self.fg_gmms, img_hsv, self._fg_mask(mask, mask_by_prior))

    # ------------------------------------------------------------------ device: inference
    def get_prob_by_gmm(self, samples, gmm):
        """Mixture likelihood of uint8-valued samples (agent.py:201-230): a gather from the 256-entry table."""
        lut = nv.gmm_lut(gmm.means_.squeeze(), np.sqrt(gmm.covariances_.squeeze()), gmm.weights_.squeeze())
        return torch.from_numpy(lut[np.asarray(samples).reshape(-1).astype(np.int64)])

    def _device_luts(self, device):
        if self._luts is None or self._luts.device != device:
            tabs = [nv.gmm_lut(g.means_.squeeze(), np.sqrt(g.covariances_.squeeze()), g.weights_.squeeze())
                    for g in (self.bg_gmms + self.fg_gmms)]
            self._luts = torch.from_numpy(np.stack(tabs)).to(device)
        return self._luts

    def get_alpha_by_gmm_device(self, img_hsv, want_confidence=True):
        if not want_confidence:
            return nv.gmm_alpha(img_hsv, self._device_luts(img_hsv.device)), None
        alpha, prob = nv.gmm_alpha(img_hsv, self._device_luts(img_hsv.device), want_prob=True)
        return alpha, float(torch.std(prob))

    def get_alpha_by_gmm(self, img_hsv):
        """(alpha uint8 (h, w), confidence) (agent.py:232-257; the reference returns the unbound `.item` as
        confidence - here it is the float it meant)."""
        alpha, conf = self.get_alpha_by_gmm_device(to_dev(img_hsv))
        return to_host(alpha), conf

    def postprocess_device(self, alpha, mask, thr_ratio=0.8):
        if self._ws is None or self._ws.device != alpha.device:
            self._ws = torch.empty(256, dtype=torch.uint8, device=alpha.device)
        nv.cf_threshold(alpha, mask, thr_ratio, ws=self._ws)
        alpha = nv.morph(nv.morph(alpha, 3, 2, nv.DILATE), 3, 2, nv.ERODE)
        return nv.morph(nv.morph(alpha, 3, 2, nv.ERODE), 3, 2, nv.DILATE)

    def postprocess(self, alpha, mask, thr_ratio=0.8):
        """Adaptive threshold, then close and open (agent.py:259-283)."""
        return to_host(self.postprocess_device(to_dev(np.array(alpha, copy=True)), to_dev(mask), thr_ratio))

    def forward_device(self, img, mask, iters=1, want_confidence=False):
        """img (H, W, 3), mask (H, W) uint8 CUDA -> (alpha CUDA (H, W), bg_img CUDA (H, W, 3), confidence).  The confidence
        (std of the probability map, one more reduction and a host sync) is only computed on request; no caller of the
        reference uses it (green.py:99 discards it)."""
        n = mask.numel()
        if self._cnt2 is None or self._cnt2.device != mask.device:
            self._cnt2 = torch.zeros(2, dtype=torch.int64, device=mask.device)
        nv.count_ge(mask, 129, out=self._cnt2[0:1])
        nv.count_ge(mask, 128, out=self._cnt2[1:2])
        n_ge129, n_ge128 = self._cnt2.tolist()                      # one host round trip for both counts
        self.last_bg_color = None                            # set below once a fitted background colour is returned
        if n_ge129 < max(self.fg_ncomp) * 5:                 # no foreground
            return mask, img, 1.0
        if n - n_ge128 < max(self.bg_ncomp) * 5:             # no background
            return mask, torch.zeros_like(img), 1.0
        ori_h, ori_w = mask.shape
        th, tw = nv.get_target_size(ori_h, ori_w, self.input_long_side)
        img_hsv = nv.resize(nv.cvt_color(img, nv.BGR2HSV), th, tw, 1)
        mask = nv.resize(mask, th, tw, 1)
        if iters == 0:
            alpha, confidence = self.get_alpha_by_gmm_device(img_hsv, want_confidence)
            alpha = self.postprocess_device(alpha, mask)
        else:
            hsv_host = to_host(img_hsv)
            for _ in range(iters):
                mask_host = to_host(mask)
                bg_prior = self.get_color_prior(hsv_host, mask_host < 128, self.color_prior_winsize)
                fg_prior = self.get_color_prior(hsv_host, mask_host < 128, self.color_prior_winsize // 5)
                self.fit_both(hsv_host, mask_host < 128, bg_prior, mask_host > 128, fg_prior)
                alpha, confidence = self.get_alpha_by_gmm_device(img_hsv, want_confidence)
                alpha = self.postprocess_device(alpha, mask)
                mask = (alpha > 128).to(torch.uint8) * 255
                nfg = int(nv.count_ge(mask, 129).item())
                if nfg < max(self.fg_ncomp) * 5 or mask.numel() - nfg < max(self.bg_ncomp) * 5:
                    break
        alpha = nv.resize(alpha, ori_h, ori_w, 1)
        hsv = tuple(int(np.mean(g.means_[0, 0])) for g in self.bg_gmms)
        if self._bg_cache is None or self._bg_cache[0] != (hsv, ori_h, ori_w, str(img.device)):
            bg_hsv = torch.tensor([[list(hsv)]], dtype=torch.uint8, device=img.device)
            bg_bgr = nv.cvt_color(bg_hsv, nv.HSV2BGR)
            self._bg_cache = ((hsv, ori_h, ori_w, str(img.device)), bg_bgr.expand(ori_h, ori_w, 3).contiguous(), bg_bgr[0, 0].cpu().numpy())
        self.last_bg_color = self._bg_cache[2]                      # host copy of the background colour (B, G, R)
        return alpha, self._bg_cache[1].clone(), confidence

    def forward(self, img, mask, iters=1):
        """NumPy API of the reference (agent.py:285-354)."""
        a, b, c = self.forward_device(to_dev(img), to_dev(mask), iters, want_confidence=True)
        return to_host(a), to_host(b), c
