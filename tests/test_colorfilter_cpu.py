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


def test_fit_pool_starts_afresh_after_a_failed_round():
    """a round in which one worker fails (or times out) left the other workers' replies in their pipes - a caller that
    caught the error and fitted again received the PREVIOUS round's mixtures.  The pool is recreated before the error is raised."""
    import pytest
    from sklearn import mixture
    from video_unscreen_amd.colorfiltering.agent import _FitPool
    rng = np.random.RandomState(5)
    mk = lambda: mixture.GaussianMixture(n_components=2, covariance_type='spherical', warm_start=True, random_state=0)   # noqa: E731
    a = np.concatenate([rng.normal(10, 1, 400), rng.normal(50, 2, 400)])
    b = np.concatenate([rng.normal(100, 1, 400), rng.normal(200, 2, 400)])
    pool = _FitPool(2)
    try:
        old = list(pool.procs)
        with pytest.raises(RuntimeError, match="failed in a worker"):
            pool.fit([(mk(), np.zeros(1), 1), (mk(), a, 1)])              # job 0 fails (one sample, two components); job 1's reply is still in its pipe
        assert len(pool.procs) == 2 and not any(p in old for p in pool.procs)
        got = pool.fit([(mk(), b, 1), (mk(), b, 1)])                      # used to return job 1 of the failed round for worker 1
        want = mk().fit(b[..., np.newaxis])
        for g in got:
            assert np.array_equal(g.means_, want.means_)
        with pytest.raises(RuntimeError, match="did not answer"):
            pool.fit([(mk(), np.tile(rng.normal(0, 1, 200000), 1), 1)], timeout=0.0)
        got = pool.fit([(mk(), b, 1)])
        assert np.array_equal(got[0].means_, want.means_)
    finally:
        pool.close()
