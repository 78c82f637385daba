"""Writes the small JPEG fixtures under tests/golden/jpeg/ and their expected decodes (jpeg_expected.npz).

The files come from Pillow's encoder (libjpeg-turbo); the expected pixels are what the oracle's N12
restatement (oracle/ingest.py) decodes, checked here to equal Pillow's own decoder byte for byte before
anything is written.  Run from the repo root:  python tests/golden/make_jpeg_fixtures.py
"""
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, ".")
from oracle import ingest as oi

HERE = os.path.join("tests", "golden", "jpeg")
rng = np.random.default_rng(1023)


def textured(*shape):
    a = rng.integers(0, 256, shape).astype(np.float64)
    for ax in (0, 1):
        for _ in range(2):
            a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax) + np.roll(a, 2, ax)) / 4
    return ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)


CASES = [  # name, shape, quality, subsampling, extra encoder options
    ("gray_q75", (40, 56), 75, 0, {}),
    ("c444_q90", (33, 47, 3), 90, 0, {}),
    ("c422_q80", (40, 51, 3), 80, 1, {}),
    ("c420_q85", (45, 62, 3), 85, 2, {}),
    ("c420_q50_restart", (48, 64, 3), 50, 2, {"restart_marker_blocks": 3}),
    ("c420_q95_optimized", (31, 29, 3), 95, 2, {"optimize": True}),
]
expected = {}
for name, shape, q, sub, kw in CASES:
    b = io.BytesIO()
    Image.fromarray(textured(*shape)).save(b, "JPEG", quality=q, subsampling=sub, **kw)
    data = b.getvalue()
    dec = oi.jpeg_decode(data)
    assert np.array_equal(dec, np.asarray(Image.open(io.BytesIO(data)))), name
    with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
        f.write(data)
    expected[name] = dec
    print(name, len(data), "bytes", dec.shape)
np.savez_compressed(os.path.join(HERE, "jpeg_expected.npz"), **expected)
