"""Generates tests/golden/asset_*.png from the reference's shipped assets.

Run in the build container only (needs /root/reference).  The fixtures are DATA: the
PIL/libjpeg-turbo-decoded pixels of assets/main.jpg, assets/additional.jpg (inputs) and
assets/depth.jpg (the reference's shipped result of the hot path, Readme.md:28-34) and
assets/foreground.jpg (the shipped result of the README's second half -- invert_depth,
select_foreground, apply_mask on depth.jpg -- Readme.md:18-24, :36-37), stored losslessly
as PNG so that tests on the GPU box (where /root/reference does not exist) see the very same
bytes.  Decoder: Pillow (libjpeg-turbo); the reference decodes with jpeg-decoder 0.1.22
(Cargo.lock), which may differ by +-1..2 per channel -> the asset test is a tolerance test.
"""
import os
import sys

import numpy as np
from PIL import Image, features

REF = "/root/reference/assets"
OUT = os.path.dirname(os.path.abspath(__file__))

for name, mode in (("main", "RGB"), ("additional", "RGB"), ("depth", "L"), ("foreground", "RGB")):
    im = Image.open(os.path.join(REF, name + ".jpg")).convert(mode)
    a = np.asarray(im)
    Image.fromarray(a).save(os.path.join(OUT, f"asset_{name}.png"), optimize=True)
    print(name, a.shape, a.dtype, "mean", a.mean())
print("Pillow", Image.__version__, "libjpeg-turbo", features.version("libjpeg_turbo"),
      file=sys.stderr)
