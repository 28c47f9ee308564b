"""Test-side restatement of the green-mode host control flow (ColorFilteringAgent.forward,
reference unscreen/colorfiltering/agent.py:285-354) on top of the CPU oracle: the checker for the HIP agent."""
import numpy as np


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


from oracle.chains import OracleColorFilter  # noqa: E402,F401  (the checker's colour-filter flow lives with the other oracle chains)
