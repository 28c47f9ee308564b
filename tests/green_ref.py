"""Test-side restatement of the green-mode host control flow (ColorFilteringAgent.forward,
reference unscreen/colorfiltering/agent.py:285-354) on top of the CPU oracle: the checker for the HIP agent."""
import numpy as np
from sklearn import mixture


def green_screen_clip(H, W, seed, screen_bgr=(60, 200, 40)):
    """A synthetic green-screen frame: noisy uniform screen + a textured elliptical 'person'. Returns (frame, coarse mask)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    inside = ((yy - H * 0.55) / (H * 0.35)) ** 2 + ((xx - W * 0.5) / (W * 0.18)) ** 2 < 1
    frame = np.empty((H, W, 3), np.float64)
    frame[:] = np.asarray(screen_bgr, np.float64) + rng.normal(0, 4, (H, W, 3))
    person = np.stack([120 + 60 * np.sin(xx / 7.0), 90 + 50 * np.cos(yy / 9.0), 150 + 40 * np.sin((xx + yy) / 11.0)], -1)
    person += rng.normal(0, 6, (H, W, 3))
    frame[inside] = person[inside]
    frame = np.clip(np.round(frame), 0, 255).astype(np.uint8)
    coarse = ((yy - H * 0.55) / (H * 0.38)) ** 2 + ((xx - W * 0.5) / (W * 0.2)) ** 2 < 1
    return frame, coarse.astype(np.uint8) * 255


class OracleColorFilter:
    def __init__(self, oracle, input_long_side=960, bg_ncomp=(3, 5, 5), fg_ncomp=(10, 10, 10), max_num_samples=10000,
                 color_prior_winsize=30, random_state=0):
        self.o = oracle
        self.L, self.bg_ncomp, self.fg_ncomp = input_long_side, bg_ncomp, fg_ncomp
        self.maxn, self.win = max_num_samples, color_prior_winsize
        mk = lambda k: mixture.GaussianMixture(n_components=k, covariance_type='spherical', warm_start=True,  # noqa: E731
                                               random_state=random_state)
        self.bg = [mk(k) for k in bg_ncomp]
        self.fg = [mk(k) for k in fg_ncomp]

    def _sub(self, s):
        return s[::len(s) // self.maxn] if len(s) > self.maxn else s

    def prior(self, hsv, mask, win):
        s = self._sub(hsv[:, :, 0][mask].astype(np.float64))
        hist, _ = np.histogram(s, 256, [0, 256])
        peak = np.argmax(hist)
        return (hsv[:, :, 0] > peak - win // 2) & (hsv[:, :, 0] < peak + win // 2)

    def luts(self, gmms):
        return np.stack([self.o.gmm_lut(g.means_.squeeze(), np.sqrt(g.covariances_.squeeze()), g.weights_.squeeze()) for g in gmms])

    def forward(self, img, mask, iters):
        o = self.o
        if (mask > 128).sum() < max(self.fg_ncomp) * 5:
            return mask, img
        if (mask < 128).sum() < max(self.bg_ncomp) * 5:
            return mask, np.zeros_like(img)
        H, W = mask.shape
        th, tw = o.get_target_size(H, W, self.L)
        hsv = o.resize_linear(o.bgr2hsv(img), th, tw)
        mask = o.resize_linear(mask, th, tw)
        for _ in range(max(iters, 1)):
            if iters > 0:
                bgp = self.prior(hsv, mask < 128, self.win)
                fgp = self.prior(hsv, mask < 128, self.win // 5)
                m = (mask < 128) & bgp
                for i in range(3):
                    self.bg[i].fit(self._sub(hsv[:, :, i][m].astype(np.float64))[..., None])
                m = mask > 128
                if (m & (1 - fgp)).sum() > max(self.fg_ncomp) * 5:
                    m = m & (1 - fgp).astype(bool)
                for i in range(3):
                    self.fg[i].fit(self._sub(hsv[:, :, i][m].astype(np.float64))[..., None])
            alpha = o.gmm_alpha(hsv, self.luts(self.bg), self.luts(self.fg))
            alpha = o.cf_postprocess(alpha, mask)
            if iters == 0:
                break
            mask = ((alpha > 128) * 255).astype(np.uint8)
            if (mask > 128).sum() < max(self.fg_ncomp) * 5 or (mask < 128).sum() < max(self.bg_ncomp) * 5:
                break
        alpha = o.resize_linear(alpha, H, W)
        bg_hsv = np.zeros((H, W, 3), np.uint8)
        for i in range(3):
            bg_hsv[:, :, i] = int(np.mean(self.bg[i].means_[0, 0]))
        return alpha, o.hsv2bgr(bg_hsv)
