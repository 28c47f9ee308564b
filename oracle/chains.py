"""Whole-frame CPU chains of the green and bg-online loops on the oracle (TEST INFRASTRUCTURE, like everything under oracle/): the
reference's loop bodies (tools/unscreen/green.py:70-138, tools/unscreen/bg.py:45-104) restated stage by stage on oracle/usc_oracle.c,
with the networks supplied by the caller as torch-CPU callables (the folded plans: DeepLabNetTorch, SchpNetTorch, StmNetTorch,
MattingNet).  Used by the GPU tests as the checker (tests/green_ref.py re-exports OracleColorFilter) and by bench.py's cpu_baseline
leg of `--mode green` / `--mode bg_online`; the bg_step chain is __graft_entry__.oracle_frame."""
import numpy as np
import torch
from sklearn import mixture


class OracleColorFilter:
    """ColorFilteringAgent.forward (unscreen/colorfiltering/agent.py:285-354) on the oracle; the EM fit is the reference's sklearn call."""

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

    def trained(self):
        return all(hasattr(g, "means_") for g in self.bg + self.fg)


def _matting(o, net, frame, alpha_pre, tri, long_side=960):
    in_h, in_w = o.vmat_input_size(frame.shape[0], frame.shape[1], long_side)
    x7 = o.vmat_pre(frame, alpha_pre, tri, in_h, in_w)
    return o.vmat_post(net(torch.from_numpy(x7)[None]).numpy()[0, 0], tri)


def seg_agent(o, frame, deeplab_cpu, long_side, crop=513):
    """SegAgent.forward (unscreen/binseg/agent.py:239-260): crops -> DeepLabv3+ on torch-CPU -> merge -> argmax."""
    H, W = frame.shape[:2]
    th, tw = o.segagent_target_size(H, W, long_side)
    locs = o.grid_crop_locations(th, tw)
    lg = None
    for i, loc in enumerate(locs):
        x = torch.from_numpy(o.segagent_pre(frame, th, tw, loc)[..., :3]).permute(2, 0, 1)[None]
        y = deeplab_cpu(x)[0].permute(1, 2, 0).numpy()
        if lg is None:
            lg = np.zeros((len(locs),) + y.shape, np.float32)
        lg[i] = y
    return o.segagent_post(lg, locs, crop, crop, th, tw, H, W)


def green_frame(o, cfg, st, frame, cf, deeplab_cpu, matting_cpu):
    """One iteration of tools/unscreen/green.py:70-138; st = {"alpha_pre", "tracking", "fid"} (updated in place) -> dict."""
    H, W = frame.shape[:2]
    if st.get("alpha_pre") is None:
        st.update(alpha_pre=np.zeros((H, W), np.uint8), tracking=False, fid=0)
    tracking = st["tracking"]
    segmask = st["alpha_pre"].copy() if tracking else seg_agent(o, frame, deeplab_cpu, cfg["binseg"]["input_long_side"])
    if not (tracking or o.exist_foreground(segmask, cfg["fg_exist_thr"])):
        alpha, fg, bg = np.zeros_like(segmask), np.zeros_like(frame), frame
    else:
        retrain = st["fid"] % cfg["colorfiltering_update_duration"] == 0 or not tracking or not cf.trained()
        alphacf, bgimg = cf.forward(frame.copy(), segmask.copy(), cfg["colorfiltering_train_iters"] if retrain else 0)
        bg_color = bgimg[0, 0]
        alphaor = o.remove_invalid_objects(cfg, alphacf.copy(), None if tracking else segmask.copy())
        tri = o.trimap_withbg(alphaor, frame, bg_color, tuple(cfg["trimap"]["color_winsize"]))
        alpha_mat = _matting(o, matting_cpu, frame, st["alpha_pre"], tri, cfg["vmatting"]["input_long_side"])
        alpha = o.color_correct(frame, alpha_mat, bg_color)
        fg, bg = o.get_fg(frame, alpha, bgimg.copy(), paste_thr=128)
    st["tracking"] = bool(o.exist_foreground(alpha, cfg["fg_exist_thr"]))
    st["alpha_pre"] = alpha
    st["fid"] += 1
    return {"alpha": alpha, "fg": fg, "bg": bg, "segmask": segmask, "tracking": tracking}


def stm_track(o, stm_cpu, prev_frame, frame, seed, long_side=960):
    """STMAgent.forward([prev_frame, frame], seed)[-1] (unscreen/stm/agent.py:203-221) for two frames: memorize the first with the
    seed mask, segment the second against that one memory entry."""
    H, W = frame.shape[:2]
    in_h, in_w = o.get_target_size(H, W, long_side, 16)
    nchw = lambda a: torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None]          # noqa: E731
    x0 = o.stm_pre(prev_frame, seed, in_h, in_w)
    prob = np.stack([x0[..., 4], x0[..., 3]], -1)
    k, v = stm_cpu.memorize(nchw(x0[..., :3]), nchw(prob))
    xq = o.stm_pre(frame, None, in_h, in_w)
    p2 = stm_cpu.decoder_logits(nchw(xq[..., :3]), k.unsqueeze(2), v.unsqueeze(2))[0].permute(1, 2, 0).numpy()
    return o.stm_post(p2, in_h, in_w, H, W)


def bg_online_frame(o, cfg, st, frame, schp_cpu, stm_cpu, matting_cpu):
    """One iteration of tools/unscreen/bg.py:47-104; st = {"alpha_pre", "tracking", "prev_frame"} (updated in place) -> dict."""
    H, W = frame.shape[:2]
    if st.get("alpha_pre") is None:
        st.update(alpha_pre=np.zeros((H, W), np.uint8), tracking=False, prev_frame=None)
    tracking = st["tracking"]
    if tracking:
        seed = np.where(st["alpha_pre"] >= 128, 255, st["alpha_pre"]).astype(np.uint8)
        segmask = stm_track(o, stm_cpu, st["prev_frame"], frame, seed, cfg["stm"].get("input_long_side", 960))
    else:
        x = torch.from_numpy(o.humanseg_pre(frame))[None]
        segmask = o.humanseg_post(schp_cpu(x)[0].numpy(), H, W)
    if not o.exist_foreground(segmask, cfg["fg_exist_thr"]):
        alpha, fg = np.zeros_like(segmask), np.zeros_like(frame)
    else:
        L = cfg["vmatting"]["input_long_side"]
        tcfg = cfg["trimap"]
        tri1 = o.trimap(o.remove_invalid_objects(cfg, segmask.copy()), tcfg["input_long_side"], tcfg["kernelsize"], tcfg["iters"])
        a1 = _matting(o, matting_cpu, frame, st["alpha_pre"], tri1, L)
        bgh = o.get_bg(a1, frame)
        hole = o.morph(np.where(a1 > 128, 255, 0).astype(np.uint8), 3, 2, "dilate")
        bgimg = np.stack([o.regionfill(bgh[..., c], hole) for c in range(3)], -1).astype(np.uint8)      # scipy spsolve per channel
        _, ensm = o.bgdiff_mask(frame, bgimg, None, a1, 1.0, cfg["bg_mask"]["thr"])
        tri2 = o.trimap(o.remove_invalid_objects(cfg, ensm.copy()), tcfg["input_long_side"], tcfg["kernelsize"], tcfg["iters"])
        alpha = _matting(o, matting_cpu, frame, st["alpha_pre"], tri2, L)
        fg, _ = o.get_fg(frame, alpha, bgimg, paste_thr=1)
    st["alpha_pre"] = alpha
    st["tracking"] = bool(o.exist_foreground(alpha, cfg["fg_exist_thr"]))
    st["prev_frame"] = frame
    return {"alpha": alpha, "fg": fg, "segmask": segmask, "tracking": tracking}
