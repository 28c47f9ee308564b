"""EM worker processes of colorfiltering/agent.py: the six scalar mixtures of a training round (screen and foreground x H, S, V) are
independent sklearn fits (reference unscreen/colorfiltering/agent.py:151-199, sklearn's GaussianMixture.fit on <= 10 000 samples each) -
~6 ms each in one interpreter, one after the other 37 ms per round, three rounds per training frame.  Each worker holds nothing: a job
is (the mixture object as it stands - it is warm-started -, the samples packed, their stride in the parent), the answer the fitted object; the arithmetic is sklearn's own in
either place, so the result is the in-process fit's bit for bit (a seeded random_state is re-derived inside every fit() call).
Lives at the package top and imports nothing of torch: a spawned worker is up as soon as sklearn is imported."""
import numpy as np


def fit_worker(conn, limits):
    """limits: {threadpoolctl user_api: threads} of the parent - sklearn's k-means initialisation sums per OpenMP thread, so a fit is
    only the parent's fit bit for bit with the parent's thread counts (and a worker that sized its pools by the host's cores would
    oversubscribe a cgroup share six times over)."""
    import sklearn.mixture                                     # noqa: F401  (imported before the first job: the first fit is not the slow one)
    import sklearn.cluster                                     # noqa: F401  (k-means: the initialisation; loads the OpenMP runtime the limits apply to)
    if limits:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=limits)
    while True:
        try:
            job = conn.recv()
        except EOFError:
            return
        if job is None:
            return
        gmm, compact, step = job
        try:
            # the parent's sample array is a strided view (every step-th masked pixel, agent.py:_subsample) and NumPy / sklearn sum a
            # strided array in another order than a packed one (1e-11 on the means): the same layout is rebuilt here
            if step > 1:
                buf = np.zeros(len(compact) * step, np.float64)
                buf[::step] = compact
                samples = buf[::step][..., np.newaxis]
            else:
                samples = compact[..., np.newaxis]
            gmm.fit(samples)
            conn.send((gmm, None))
        except Exception as e:                                 # noqa: BLE001 - reported to the caller, which raises it
            conn.send((None, "%s: %s" % (type(e).__name__, e)))
