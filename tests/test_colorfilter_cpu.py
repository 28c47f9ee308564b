"""ColorFilteringAgent's EM fits (reference unscreen/colorfiltering/agent.py:151-199): the six mixtures of a training round fitted in the
worker processes (fit_workers = 6, the default) must be the in-process fits (fit_workers = 0, the reference's order) bit for bit,
warm starts included."""
import numpy as np


def _round(agent, hsv, bg_mask, fg_mask):
    bg_prior = agent.get_color_prior(hsv, bg_mask, agent.color_prior_winsize)
    fg_prior = agent.get_color_prior(hsv, bg_mask, agent.color_prior_winsize // 5)
    agent.fit_both(hsv, bg_mask, bg_prior, fg_mask, fg_prior)


def test_fit_workers_give_the_in_process_fits():
    from video_unscreen_amd.colorfiltering.agent import ColorFilteringAgent
    rng = np.random.RandomState(3)
    H, W = 135, 240
    hsv = np.empty((H, W, 3), np.uint8)
    hsv[..., 0] = np.clip(rng.normal(60, 4, (H, W)), 0, 179)               # a green screen ...
    hsv[..., 1] = np.clip(rng.normal(200, 20, (H, W)), 0, 255)
    hsv[..., 2] = np.clip(rng.normal(180, 25, (H, W)), 0, 255)
    fg = np.zeros((H, W), bool)
    fg[30:110, 80:170] = True                                              # ... and a subject of other colours
    hsv[fg] = rng.randint(0, 256, (int(fg.sum()), 3))
    hsv[..., 0] = np.minimum(hsv[..., 0], 179)
    a = ColorFilteringAgent(random_state=0, fit_workers=0)
    b = ColorFilteringAgent(random_state=0, fit_workers=6)
    try:
        for it in range(3):                                                # the second and third round are warm-started
            m = fg if it < 2 else np.roll(fg, 5, 1)
            _round(a, hsv, ~m, m)
            _round(b, hsv, ~m, m)
            for ga, gb in zip(a.bg_gmms + a.fg_gmms, b.bg_gmms + b.fg_gmms):
                assert np.array_equal(ga.means_, gb.means_) and np.array_equal(ga.covariances_, gb.covariances_)
                assert np.array_equal(ga.weights_, gb.weights_) and ga.n_iter_ == gb.n_iter_
        assert b.is_trained() and b._pool is not None and len(b._pool.procs) == 6
    finally:
        b.close()
    assert b._pool is None
