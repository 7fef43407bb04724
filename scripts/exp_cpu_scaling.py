"""How the two CPU legs of bench.py's cpu_baseline scale with host threads on this box: the oracle (OpenMP) and the reference's compiled
CSRayMarchV through oracle/_ref (pthread group workers), C2's view march.  Also prints what the box gives the process (affinity, cgroup quota).
usage: python scripts/exp_cpu_scaling.py [config = C2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multivolumes_amd import scene
from multivolumes_amd import raycaster as rc
from tests.helpers import oracle_pipeline
from oracle import mv_oracle as oracle, dxref

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError: pass
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(scene.CONFIGS[name]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
ref = oracle_pipeline(oracle, cfg, vols, 64, frames=1)
n = ref["stats"]["view_samples"]
lib = oracle.load()
top = lib.mvo_num_threads()
args = (V, N, L, cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"], ref["info"], ref["cube"])
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > 2 * top: break
    lib.mvo_set_num_threads(th)
    oracle.march_view(*args)
    t0 = time.perf_counter(); oracle.march_view(*args); dt = time.perf_counter() - t0
    line = f"{th:4d} threads: oracle {n / dt / 1e6:8.2f} Msamples/s"
    if dxref.available():
        dxref.load().dxref_set_threads(th)
        t0 = time.perf_counter(); dxref.march_view(*args); dr = time.perf_counter() - t0
        line += f"   compiled reference {n / dr / 1e6:8.2f} Msamples/s"
    print(line, flush=True)
lib.mvo_set_num_threads(top)
