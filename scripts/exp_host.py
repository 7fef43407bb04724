"""Where does the host spend its time per frame (enqueue cost of each ABI call) and how far ahead of the GPU does it run?"""
import sys, os, time
if "--torch" in sys.argv:
    import torch; torch.cuda.init(); print("torch first", torch.version.hip)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
flags = 0 if "--no-pipeline" in sys.argv else _lib.MV_FLAG_PIPELINE_FRAMES
c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=flags)
for k in range(V): c.LoadVolumeData(k, rc.synth_volume(N, k))
fr = scene.setup_raycaster(c, cfg)
def frame():
    c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.Render(composite=True)
for _ in range(40): frame()
c.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): frame()
t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
print("enqueue %.1f us/frame   total %.1f us/frame" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
lib = c._lib; ctx = c._ctx
acc = dict(update=0.0, host_math=0.0, cull=0.0, light=0.0, view=0.0, comp=0.0)
c.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); b = time.perf_counter(); acc["update"] += b - a
    a = time.perf_counter(); lib.mv_cull(ctx); b = time.perf_counter(); acc["cull"] += b - a
    a = time.perf_counter(); lib.mv_march_light(ctx); b = time.perf_counter(); acc["light"] += b - a
    a = time.perf_counter(); lib.mv_march_view(ctx); b = time.perf_counter(); acc["view"] += b - a
    a = time.perf_counter(); lib.mv_composite(ctx, 0, 0); b = time.perf_counter(); acc["comp"] += b - a
t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
print("per call (us):", {k: round(v / n * 1e6, 1) for k, v in acc.items()}, " enqueue %.1f total %.1f us/frame" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
