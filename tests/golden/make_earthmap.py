"""Decodes the reference's texture asset (images/earthmap.jpg, read by texture.rs:94-96 through image 0.25.2 +
zune-jpeg 0.4.13) ONCE with PIL/libjpeg and stores the decoded RGB8 raster as a lossless PNG, so that the oracle
and the GPU path read identical texels wherever they run.  The JPEG is baseline 4:4:4, so decoders differ by
at most +-1 LSB (IDCT / colour-conversion rounding); which decoder the reference would have used is therefore
"parity unpinned" at the 1-LSB level.  Decoded raster sha256 prefix: a8cdc92a168d554d.

Run in the authoring container only (needs /root/reference):  python tests/golden/make_earthmap.py
"""
import hashlib
import os

import numpy as np
from PIL import Image

src = "/root/reference/images/earthmap.jpg"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "earthmap_rgb8.png")
a = np.asarray(Image.open(src).convert("RGB"))
assert a.shape == (512, 1024, 3)
Image.fromarray(a).save(dst, optimize=True)
b = np.asarray(Image.open(dst).convert("RGB"))
assert (a == b).all()
print(dst, os.path.getsize(dst), hashlib.sha256(a.tobytes()).hexdigest()[:16])
