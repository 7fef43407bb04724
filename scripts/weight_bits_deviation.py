#!/usr/bin/env python3
"""What D3D's freedom to snap texel coordinates to 8 fractional bits could change: the CPU oracle with exact fp32 filter weights
against its optional 8-bit mode (mvo_set_options), whole pipeline at a config (default C3).  CPU only; prints the deviation of the
cube maps the view march writes.  usage: python scripts/weight_bits_deviation.py [CONFIG] [bits]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import raycaster as rc, scene
from oracle import mv_oracle as o
from tests.helpers import oracle_pipeline
from tests.conftest import half_to_float

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = scene.CONFIGS[name]
vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
M = 1024 if cfg["N"] >= 128 else 64
ref = oracle_pipeline(o, cfg, vols, M)
o.load().mvo_set_options(bits, 0)
try:
    q = oracle_pipeline(o, cfg, vols, M)
finally:
    o.load().mvo_set_options(0, 0)
assert np.array_equal(q["info"], ref["info"])
worst, tot, n, beyond = 0.0, 0.0, 0, 0
for v in ref["cube"]:
    mip = int(ref["info"][v, 0])
    a = half_to_float(o.cube_face_view(ref["cubeMaps"][v], cfg["N"], mip))
    b = half_to_float(o.cube_face_view(q["cubeMaps"][v], cfg["N"], mip))
    d = np.abs(a - b)
    worst = max(worst, float(d.max())); tot += float(d.sum()); n += d.size; beyond += int((d.max(axis=-1) > 1e-3).sum())
texels = n // 4
print(f"{name}: {bits}-bit filter weights vs exact fp32: max |cube map diff| {worst:.4g}, mean {tot / n:.3g}, "
      f"texels beyond 1e-3: {beyond} of {texels} ({100.0 * beyond / texels:.2f} %); "
      f"view samples {ref['stats']['view_samples']} -> {q['stats']['view_samples']}")
