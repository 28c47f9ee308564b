"""NumPy emulation of conv_wino_kernel's index arithmetic (csrc/usc_conv.hip): Winograd F(2x2, 3x3) as 16 per-position GEMMs
over blocks of 8 x 4 output tiles.  Checks the tile / patch geometry, the transform matrices, the packed weight layout and the
output phases against a direct float64 convolution, on the CPU, before the kernel ever runs.

    python tools/dev/wino_emulate.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def emulate(x, U_packed, Cin, Cout, H, W, batch=1):
    """x: (batch*H, W, Cin) float32 NHWC; U_packed as pack_wino() lays it out.  Returns (batch*H, W, Cout) float32."""
    U = U_packed.reshape(Cin // 8, 16, 2, Cout, 4)                    # [c][p][h][n][q] : cin = 8c + 4h + q
    out = np.zeros((batch * H, W, Cout), np.float32)
    TH, TW = (H + 1) // 2, (W + 1) // 2
    BH, BW = (TH + 3) // 4, (TW + 7) // 8
    for img in range(batch):
        for by in range(BH):
            for bx in range(BW):
                y0, x0 = 8 * by - 1, 16 * bx - 1
                patch = np.zeros((10, 18, Cin), np.float32)
                for py in range(10):
                    for px in range(18):
                        y, xx = y0 + py, x0 + px
                        if 0 <= y < H and 0 <= xx < W:
                            patch[py, px] = x[img * H + y, xx]
                acc = np.zeros((16, 32, Cout), np.float32)           # [position][tile][cout]
                for c in range(Cin // 8):
                    d = np.zeros((32, 4, 4, 8), np.float32)
                    for t in range(32):
                        ty, tx = t // 8, t % 8
                        d[t] = patch[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, 8 * c:8 * c + 8]
                    T = np.einsum('ai,tijk->tajk', BT.astype(np.float32), d)
                    V = np.einsum('tajk,bj->tabk', T, BT.astype(np.float32))   # [tile][a][b][8 cin]
                    Uc = U[c].transpose(0, 1, 3, 2).reshape(16, 8, Cout)       # [p][h*4+q][n]
                    for p in range(16):
                        acc[p] += V[:, p // 4, p % 4, :] @ Uc[p]
                M = acc.reshape(4, 4, 32, Cout)
                Y = np.einsum('ia,abtn,jb->ijtn', AT.astype(np.float32), M, AT.astype(np.float32))
                for t in range(32):
                    ty, tx = 4 * by + t // 8, 8 * bx + t % 8
                    for i in range(2):
                        for j in range(2):
                            oy, ox = 2 * ty + i, 2 * tx + j
                            if oy < H and ox < W:
                                out[img * H + oy, ox] = Y[i, j, t]
    return out


def main():
    from video_unscreen_amd.vmatting.net_hip import pack_wino
    rng = np.random.default_rng(0)
    for (H, W, Cin, Cout, batch) in ((17, 30, 8, 32, 1), (12, 20, 16, 64, 2), (9, 7, 32, 32, 1)):
        x = rng.standard_normal((batch, Cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.2).astype(np.float32)
        want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
        xn = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(batch * H, W, Cin)
        got = emulate(xn, pack_wino(torch.from_numpy(w)).numpy(), Cin, Cout, H, W, batch)
        got = got.reshape(batch, H, W, Cout).transpose(0, 3, 1, 2)
        err = np.abs(got - want).max() / np.abs(want).max()
        print(f"{H}x{W} Cin {Cin} Cout {Cout} batch {batch}: max rel err {err:.2e}")
        assert err < 2e-6


if __name__ == "__main__":
    main()
