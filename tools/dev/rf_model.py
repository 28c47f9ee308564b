"""NumPy model of the Laplace solver of csrc/usc_regionfill.hip (one channel, float64): preconditioned CG with one multigrid V-cycle
per iteration - 2 x 2 aggregation, piecewise-constant P, Chebyshev-weighted Jacobi (2 pre, 2 post), 5 sweeps on the coarsest level.
Used to price changes of the preconditioner by their iteration count before any kernel is touched.

    python tools/dev/rf_model.py [gpurun_out/bgonline_hole.npz]      (tools/dev/dump_bgonline_hole.py writes that file on the GPU box)

coarse = "rediscretised": the shipped form - a coarse cell is in the mask iff any child is, the coarse operator is the same masked
         5-point stencil on the coarse grid (n z - sum of in-mask neighbours);
coarse = "galerkin":      P^T A P / 2 per level - per cell a diagonal and four edge weights, summed from the children's: a coarse cell
         that is only partly inside the mask gets the diagonal and the couplings its children really have (the ragged Dirichlet
         boundary is represented exactly in the variational sense; the / 2 keeps the over-correction the shipped form relies on)."""
import sys
import numpy as np

OA, OB, OC = 1.3895, 0.5617, 0.8


def shifts(a):
    """the four neighbours' values (up, down, left, right), zero outside the image"""
    p = np.pad(a, 1)
    return p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]


class Level:
    def __init__(self, mask, d, w):
        self.mask, self.d, self.w = mask, d, w          # d: diagonal, w[k]: weight towards neighbour k (0 where that neighbour is not in the mask)
        self.dinv = np.where(mask, 1.0 / np.where(d > 0, d, 1), 0.0)

    def A(self, z):
        s = shifts(z)
        return np.where(self.mask, self.d * z - sum(self.w[k] * s[k] for k in range(4)), 0.0)


def level0(mask):
    H, W = mask.shape
    n = np.full((H, W), 4.0)
    n[0, :] -= 1; n[-1, :] -= 1; n[:, 0] -= 1; n[:, -1] -= 1
    w = [np.where(mask, m.astype(np.float64), 0.0) for m in shifts(mask)]
    return Level(mask, np.where(mask, n, 0.0), w)


def pad2(a):
    H, W = a.shape
    return np.pad(a, ((0, H & 1), (0, W & 1)))


def coarsen(lv, mode, scale=0.5):
    mask = pad2(lv.mask)
    Hc, Wc = mask.shape[0] // 2, mask.shape[1] // 2
    mc = mask.reshape(Hc, 2, Wc, 2).any(axis=(1, 3))
    if mode == "rediscretised":
        return level0(mc)
    d, w = pad2(lv.d), [pad2(x) for x in lv.w]
    blk = lambda a: a.reshape(Hc, 2, Wc, 2)                     # noqa: E731  [Y, dy, X, dx]
    D, Wt = blk(d), [blk(x) for x in w]
    # internal edges of an aggregate: up-edges of its lower children, down-edges of its upper children, left of the right, right of the left
    internal = Wt[0][:, 1].sum(-1) + Wt[1][:, 0].sum(-1) + Wt[2][:, :, :, 1].sum(1) + Wt[3][:, :, :, 0].sum(1)
    dc = (D.sum(axis=(1, 3)) - internal) * scale
    wc = [Wt[0][:, 0].sum(-1) * scale, Wt[1][:, 1].sum(-1) * scale, Wt[2][:, :, :, 0].sum(1) * scale, Wt[3][:, :, :, 1].sum(1) * scale]
    return Level(mc, np.where(mc, dc, 0.0), [np.where(mc, x, 0.0) for x in wc])


def restrict(lv, res):
    r = pad2(np.where(lv.mask, res, 0.0))
    return r.reshape(r.shape[0] // 2, 2, r.shape[1] // 2, 2).sum(axis=(1, 3))


def prolong(lv, zc):
    H, W = lv.mask.shape
    return np.where(lv.mask, np.repeat(np.repeat(zc, 2, 0), 2, 1)[:H, :W], 0.0)


def vcycle(levels, l, r, sweeps=1):
    lv = levels[l]
    if isinstance(sweeps, (list, tuple)):
        sw_all, sweeps = sweeps, sweeps[min(l, len(sweeps) - 1)]
    else:
        sw_all = sweeps
    if l == len(levels) - 1:
        z = OC * lv.dinv * r
        for _ in range(4):
            z = z + OC * lv.dinv * (r - lv.A(z))
        return z
    z = OA * lv.dinv * r
    z = z + OB * lv.dinv * (r - lv.A(z))
    for _ in range(sweeps - 1):
        z = z + OA * lv.dinv * (r - lv.A(z))
        z = z + OB * lv.dinv * (r - lv.A(z))
    rc = restrict(lv, r - lv.A(z))
    z = z + prolong(lv, vcycle(levels, l + 1, np.where(levels[l + 1].mask, rc, 0.0), sw_all))
    for _ in range(sweeps - 1):
        z = z + OB * lv.dinv * (r - lv.A(z))
        z = z + OA * lv.dinv * (r - lv.A(z))
    z = z + OB * lv.dinv * (r - lv.A(z))
    z = z + OA * lv.dinv * (r - lv.A(z))
    return z


def solve(img, mask, coarse="rediscretised", rtol=1e-7, sweeps=1, max_iters=400, scale=0.5):
    levels = [level0(mask)]
    while levels[-1].mask.shape[0] > 8 or levels[-1].mask.shape[1] > 8:
        levels.append(coarsen(levels[-1], coarse, scale))
    A = levels[0]
    x = np.where(mask, img, 0.0)
    known = np.where(mask, 0.0, img)
    b = np.where(mask, sum(shifts(known)), 0.0)
    r = b - A.A(x)
    r0 = float((r * r).sum())
    p = rz_old = None
    for it in range(1, max_iters + 1):
        z = vcycle(levels, 0, r, sweeps)
        rz = float((r * z).sum())
        p = z if p is None else z + (rz / rz_old) * p
        Ap = A.A(p)
        alpha = rz / float((p * Ap).sum())
        x += alpha * p
        r -= alpha * Ap
        rz_old = rz
        if it % 2 == 0 and float((r * r).sum()) <= rtol * rtol * r0:
            return x, it
    return x, max_iters


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bgonline_hole.npz"
    d = np.load(path)
    H, W = (int(v) for v in d["shape"])
    mask = np.unpackbits(d["hole"])[:H * W].reshape(H, W).astype(bool)
    img = d["bg"].astype(np.float64)
    print("hole fraction %.3f, the kernels needed %d iterations" % (mask.mean(), int(d["iters"])))
    ref = None
    for coarse, sweeps in (("rediscretised", 1), ("galerkin", 1), ("galerkin", (1, 2)), ("galerkin", (1, 1, 2))):
        x, it = solve(img, mask, coarse, sweeps=sweeps)
        if ref is None:
            ref = x
        print("%-14s sweeps %s: %3d iterations, max |x - x_first| = %.2e" % (coarse, sweeps, it, float(np.abs(x - ref)[mask].max())), flush=True)
