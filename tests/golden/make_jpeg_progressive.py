#!/usr/bin/env python3
"""Writes the progressive-JPEG fixtures of tests/test_textures.py: pairs of files with the SAME quantised DCT coefficients,
one coded as a progressive JPEG (SOF2: DC / AC first scans, successive-approximation refinements, interleaved and
single-component scans) and one as a baseline JPEG, both by Pillow / libjpeg from synthetic pictures.  A decoder that is
right on baseline files must give the same pixels for both.  Run once; the outputs are committed."""
import io, os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(21)
yy, xx = np.mgrid[0:37, 0:53]
smooth = np.stack([(xx * 4 + yy) % 256, (yy * 6) % 256, (xx * yy) % 256], 2).astype(np.uint8)
blobs = (rng.integers(0, 256, (6, 9, 3)).repeat(8, 0).repeat(8, 1)[:45, :70] * 0.8 + rng.integers(0, 50, (45, 70, 3))).astype(np.uint8)
noisy = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
cases = [("a", smooth, dict(quality=85, subsampling=2)),            # 4:2:0
         ("b", blobs, dict(quality=60, subsampling=1)),             # 4:2:2
         ("c", noisy, dict(quality=40, subsampling=0)),             # 4:4:4, dense high frequencies -> long refinement scans
         ("d", blobs[..., 1], dict(quality=75)),                    # greyscale
         ("e", smooth, dict(quality=95, subsampling=2, restart_marker_rows=1))]
for name, img, kw in cases:
    im = Image.fromarray(img)
    for kind, prog in (("prog", True), ("base", False)):
        buf = io.BytesIO()
        im.save(buf, "JPEG", progressive=prog, optimize=prog, **kw)
        data = buf.getvalue()
        assert (b"\xff\xc2" in data) == prog
        open(os.path.join(HERE, "jpeg_%s_%s.jpg" % (kind, name)), "wb").write(data)
        print(name, kind, len(data), "bytes", "restart markers" if b"\xff\xd0" in data else "")
