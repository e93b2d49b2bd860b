"""Writes tests/golden/jpeg_cases.npz: small JPEG files (as uint8 arrays) + their decode by Pillow's bundled libjpeg-turbo (BGR, as
cv2.imread(path, IMREAD_COLOR) returns it) + the geometry capf_jpeg_info must report.  Build container only:
    python oracle/make_jpeg_goldens.py [--check]
Pillow (12.2, libjpeg-turbo: PIL.features.version('jpg')) is the third-party decoder in this image; the reference's own decoder is OpenCV's
imread (ContextPose/mvn/datasets/human36m.py:292-295), which is the same libjpeg default path (islow IDCT, fancy upsampling)."""
import io
import os
import sys

import numpy as np
from PIL import Image, features

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {   # name: (H, W, mode, subsampling, quality, save kwargs)
    "rgb444_q90": (40, 48, "RGB", 0, 90, {}),
    "rgb420_q75_odd": (37, 53, "RGB", 2, 75, {}),
    "rgb422_q50": (50, 45, "RGB", 1, 50, {}),
    "rgb420_q95_opt": (41, 29, "RGB", 2, 95, dict(optimize=True)),
    "rgb420_q85_rst": (48, 64, "RGB", 2, 85, dict(restart_marker_blocks=2)),
    "rgb444_q30": (33, 47, "RGB", 0, 30, {}),
    "grey_q80": (35, 42, "L", 0, 80, {}),
    "rgb420_q100_sat": (32, 32, "RGB", 2, 100, {}),
}


def synth(name, H, W, mode):
    rng = np.random.Generator(np.random.Philox(key=[sum(map(ord, name)), 20261003]))
    y, x = np.mgrid[0:H, 0:W]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 5.0 + y / 9.0), (x * 3 + y * 5) % 256], -1)
    img = img + rng.normal(0, 12, img.shape)
    if name.endswith("_sat"):                       # saturated blocks: exercises the range limit of the IDCT and of the colour conversion
        img = np.where((x // 8 + y // 8)[..., None] % 2 == 0, 255.0, 0.0) + rng.normal(0, 40, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img if mode == "RGB" else img[..., 0]


def make():
    rec = {"pillow_version": np.array(Image.__version__), "libjpeg_turbo": np.array(str(features.version("jpg")))}
    for name, (H, W, mode, sub, q, kw) in CASES.items():
        im = Image.fromarray(synth(name, H, W, mode), mode)
        buf = io.BytesIO()
        if mode == "RGB":
            im.save(buf, "JPEG", quality=q, subsampling=sub, **kw)
        else:
            im.save(buf, "JPEG", quality=q, **kw)
        data = buf.getvalue()
        dec = Image.open(io.BytesIO(data))
        rgb = np.asarray(dec.convert("RGB"))
        rec[name + ":jpeg"] = np.frombuffer(data, np.uint8)
        rec[name + ":bgr"] = np.ascontiguousarray(rgb[..., ::-1])
        rec[name + ":info"] = np.array([W, H, 3 if mode == "RGB" else 1, {0: 1, 1: 2, 2: 2}[sub] if mode == "RGB" else 1,
                                        {0: 1, 1: 1, 2: 2}[sub] if mode == "RGB" else 1], np.int32)
    return rec


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    rec = make()
    if "--check" in sys.argv:
        old = np.load(path, allow_pickle=False)
        bad = [k for k in rec if k not in ("pillow_version", "libjpeg_turbo") and not np.array_equal(rec[k], old[k])]
        print("check jpeg_cases:", "OK" if not bad and set(rec) == set(old.files) else f"MISMATCH {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(path, **rec)
    print(f"-> {path} ({os.path.getsize(path) / 1024:.0f} KiB, Pillow {Image.__version__}, libjpeg {features.version('jpg')})")
