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
lib.mv_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
lib.mv_debug_set_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
def frames(n):
    c.ResetStats(); c.SetProfiling(True)
    for _ in range(n):
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.Render(composite=True)
    c.SetProfiling(False); st = c.GetStats()
    return st["ms_march_view"] / st["n_march_view"]
for _ in range(V): c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.Render(composite=True)
print("list order           ms %.4f" % frames(20))
costs = np.zeros(400000, np.uint16); n = C.c_uint32()
assert lib.mv_debug_tile_costs(c._ctx, costs.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
nn = n.value; cc = costs[:nn].astype(np.int64); idx = np.arange(nn)
def run(name, o):
    o = np.asarray(o, np.uint32); assert len(np.unique(o)) == nn
    lib.mv_debug_set_tile_order(c._ctx, o.ctypes.data_as(C.c_void_p), nn); print("%-20s ms %.4f" % (name, frames(20)))
for thr in (80, 64, 48):
    long_ = idx[cc >= thr]; rest = idx[cc < thr]
    run("long>=%d first" % thr, np.concatenate([long_, rest]))
    # interleave long tiles evenly into the first half of the dispatch order
    k = len(long_); m = max(1, (nn // 2) // max(1, k)); out = []; ri = 0
    for i, t in enumerate(long_):
        out.append(t); take = rest[ri:ri + m - 1]; out.extend(take); ri += len(take)
    out.extend(rest[ri:]); run("long>=%d spread/half" % thr, out)
zero = idx[cc == 0]; nz = idx[cc > 0]
run("empty tiles last", np.concatenate([nz, zero]))
lib.mv_debug_set_tile_order(c._ctx, None, 0); print("list order again     ms %.4f" % frames(20))
for thr in (40, 32, 24, 16):
    run("long>=%d first" % thr, np.concatenate([idx[cc >= thr], idx[cc < thr]]))
for bins in ((64, 32), (64, 48, 32), (80, 64, 48, 32, 16), (64, 32, 16, 8)):
    parts = []; hi = 1 << 30
    for b in bins: parts.append(idx[(cc >= b) & (cc < hi)]); hi = b
    parts.append(idx[cc < hi]); run("bins %s" % (bins,), np.concatenate(parts))
run("full sort desc", np.argsort(-cc, kind="stable"))
