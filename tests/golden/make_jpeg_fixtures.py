"""Writes tests/golden/jpeg_fixtures.npz: small JPEG files of every kind the loader's reader handles,
encoded with Pillow (libjpeg-turbo), each with the RGB pixels Pillow decodes from it.  Run once; the
.npz is committed.  Pillow is only needed here, not by the tests."""
import io
import os

import numpy as np
from PIL import Image


def picture(width, height, seed):
    """Smooth colour gradients, a few hard edges and some noise: exercises DC prediction, long zero
    runs and high-frequency coefficients."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    r = 128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0)
    g = 255.0 * x / max(width - 1, 1)
    b = 255.0 * y / max(height - 1, 1)
    img = np.stack([r, g, b], axis=2)
    img[height // 3:height // 2, width // 4:width // 2] = [250, 20, 30]
    img[::5, ::7] = [0, 255, 0]
    img += rng.normal(0.0, 6.0, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {
    # name: (width, height, mode, save options)
    "baseline_444": (48, 40, "RGB", dict(quality=90, subsampling=0)),
    "baseline_422": (50, 37, "RGB", dict(quality=85, subsampling=1)),
    "baseline_420": (61, 43, "RGB", dict(quality=80, subsampling=2)),
    "baseline_420_odd": (17, 9, "RGB", dict(quality=95, subsampling=2)),
    "grey": (33, 29, "L", dict(quality=88)),
    "progressive_420": (64, 48, "RGB", dict(quality=75, subsampling=2, progressive=True)),
    "progressive_444": (37, 53, "RGB", dict(quality=92, subsampling=0, progressive=True)),
    "progressive_grey": (40, 24, "L", dict(quality=60, progressive=True)),
    "restart_420": (72, 40, "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=2)),
    "optimised_low_quality": (56, 56, "RGB", dict(quality=25, subsampling=2, optimize=True)),
    "tiny": (1, 1, "RGB", dict(quality=90, subsampling=2)),
}


def main():
    out = {}
    for k, (name, (w, h, mode, options)) in enumerate(CASES.items()):
        img = picture(w, h, k)
        pil = Image.fromarray(img if mode == "RGB" else img[..., 1], mode)
        buf = io.BytesIO()
        pil.save(buf, "JPEG", **options)
        data = buf.getvalue()
        decoded = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        out[name + "_file"] = np.frombuffer(data, dtype=np.uint8)
        out[name + "_rgb"] = decoded
        print(name, len(data), "bytes", decoded.shape)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_fixtures.npz"), **out)


main()
