"""read_img / save_img on numpy + Pillow only: importable by the frame I/O worker processes (utils/frameio.py) without pulling
in torch or the HIP library.  Re-exported by utils/fileio.py under the reference's names (reference unscreen/utils/fileio.py:31-62)."""
import numpy as np
from PIL import Image


def read_img(path):
    """3-channel uint8 BGR array (what cv2.imread returns), or a .npy array as is."""
    if path.endswith(".npy"):
        return np.load(path)
    im = Image.open(path).convert("RGB")
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def save_img(img, save_path, downsacle=1):
    assert isinstance(downsacle, int)
    if save_path.endswith(".npy"):
        np.save(save_path, img)
        return
    im = Image.fromarray(img if img.ndim == 2 else np.ascontiguousarray(img[:, :, ::-1]))
    if downsacle != 1:
        im = im.resize((im.width // downsacle, im.height // downsacle), Image.BILINEAR)
    im.save(save_path, quality=95)
