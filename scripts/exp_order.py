import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024)
for k in range(V): c.LoadVolumeData(k, rc.synth_volume(N, k))
fr = scene.setup_raycaster(c, cfg)
lib = _lib.load()
def frames(n):
    c.ResetStats(); c.SetProfiling(True)
    for _ in range(n):
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.Render(composite=True)
    c.SetProfiling(False); st = c.GetStats()
    return st["ms_march_view"] / st["n_march_view"]
for _ in range(V): c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.Render(composite=True)
print("list order      ms", frames(20))
costs = np.zeros(400000, np.uint16); n = C.c_uint32()
lib.mv_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
lib.mv_debug_set_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert lib.mv_debug_tile_costs(c._ctx, costs.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
cc = costs[:n.value].astype(np.int64)
print("tiles", n.value, "cost hist", np.histogram(cc, bins=[0,1,8,16,32,64,96,120,129])[0].tolist(), "sum", cc.sum(), "mean", cc.mean())
order = np.argsort(-cc, kind="stable").astype(np.uint32)
assert lib.mv_debug_set_tile_order(c._ctx, order.ctypes.data_as(C.c_void_p), n.value) == 0
print("longest first   ms", frames(20))
order2 = np.argsort(cc, kind="stable").astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, order2.ctypes.data_as(C.c_void_p), n.value)
print("shortest first  ms", frames(20))
rng = np.random.default_rng(0); order3 = rng.permutation(n.value).astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, order3.ctypes.data_as(C.c_void_p), n.value)
print("random          ms", frames(20))
# drop zero-cost tiles to the end, long first, but keep neighbours together in chunks of 32 tiles
import math
nn = n.value
def stride_perm(frac):
    s = int(nn * frac) | 1
    while math.gcd(s, nn) != 1: s += 2
    return ((np.arange(nn, dtype=np.int64) * s) % nn).astype(np.uint32)
for frac in (0.6180339887, 0.381966, 0.1234, 0.05, 0.01):
    o = stride_perm(frac)
    assert len(np.unique(o)) == nn
    lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn)
    print("stride %.4f     ms" % frac, frames(20))
# sorted by cost, dealt round-robin into 64 queues (mix long and short everywhere)
srt = np.argsort(-cc, kind="stable")
o = np.concatenate([srt[i::64] for i in range(64)]).astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn); print("cost-dealt/64    ms", frames(20))
# random within chunks of 256 consecutive tiles, chunks random
idx = np.arange(nn); chunks = [idx[i:i+256] for i in range(0, nn, 256)]
rng.shuffle(chunks); o = np.concatenate(chunks).astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn); print("chunk256 random ms", frames(20))
chunks = [idx[i:i+16] for i in range(0, nn, 16)]
rng.shuffle(chunks); o = np.concatenate(chunks).astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn); print("chunk16 random  ms", frames(20))
o = rng.permutation(nn).astype(np.uint32)
lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn); print("random again    ms", frames(20))
