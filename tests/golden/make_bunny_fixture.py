"""Generates tests/golden/bunny_mesh.npz: the triangle mesh the reference's demo scenes rasterise into the depth / shadow maps
(`-mesh Assets/bunny.obj 0 -9 0 1.8`, Bin/all16.bat:1 and all64.bat:1; the Stanford bunny), as DATA: vertex positions and triangle
indices exactly as the product's own OBJ reader (mvh_load_obj) returns them.  The OBJ lives under /root/reference/Bin/Assets in
the authoring container only; the GPU box gets this fixture.  Run from the repo root: python tests/golden/make_bunny_fixture.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multivolumes_amd import raycaster as rc

SRC = "/root/reference/Bin/Assets/bunny.obj"
pos, idx, lo, hi = rc.load_obj(SRC)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bunny_mesh.npz")
np.savez_compressed(out, positions=pos.astype(np.float32), indices=idx.astype(np.uint32), aabb_min=lo, aabb_max=hi)
print(out, pos.shape, idx.shape, lo, hi, os.path.getsize(out))
