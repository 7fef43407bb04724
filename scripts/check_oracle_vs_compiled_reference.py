"""The oracle against the reference's shipped compiled shaders (oracle/_ref, see oracle/dxil) at the BASELINE configs' full sizes, on the
CPU: run where /root/reference exists (`make -C oracle ref` first).  tests/test_oracle_vs_reference_dxil.py holds the small cases.
Since round 3 the fixed-function stages inside oracle/_ref are oracle/dxil/d3d_sampler.c, a statement independent of the oracle's
sampler: "identical" below therefore also says that the two sampler statements agree on every fetch of the frame.  The last lines
report what the device-dependent switches of that statement move (8-bit texel coordinates, truncating float11 / float10 stores)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mv_oracle as o, dxref
from multivolumes_amd import scene, raycaster as rc
from tests.helpers import default_descs, host_frame, oracle_pipeline
o.load()
assert dxref.available()
for name in sys.argv[1:] or ["C2", "C3"]:
    cfg = dict(scene.CONFIGS[name]); V, N, L, M = cfg["V"], cfg["N"], cfg["L"], 256
    t0 = time.time()
    vols = [rc.synth_volume(N, k) for k in range(V)]
    descs = default_descs(V, V, N)
    _, vp, svp, cb, po = host_frame(cfg, 0)
    depth, shadow = scene.depth_and_shadow(cfg, vp, svp, M)
    i0, v0, c0 = o.cull(V, po, descs, cb, cfg["maxRay"])
    i1, v1, c1 = dxref.cull(V, po, descs, cb, cfg["maxRay"])
    print(f"{name}: cull identical: {np.array_equal(i0, i1) and np.array_equal(np.sort(v0), v1) and np.array_equal(np.sort(c0), c1)}  ({len(v0)} visible, {len(c0)} on the cube-map scheme)", flush=True)
    lms = None
    ref_lm = np.zeros((V, L, L, L), np.uint32)
    nd = 0
    for f in range(2):
        _, _, _, cbf, pof = host_frame(cfg, f)
        vid, lms, n = o.march_light(V, N, L, M, vols, descs, pof, cbf, shadow, None, v0, cfg["maxLight"], lms)
        ref_lm = dxref.march_light(V, N, L, M, vols, descs, pof, cbf, shadow, None, v0, cfg["maxLight"], ref_lm)
        nd += int((ref_lm[vid] != lms[vid]).sum())
    print(f"{name}: light march, 2 frames x {L}^3 voxels: {nd} voxels differ  ({time.time() - t0:.0f} s)", flush=True)
    cm, cd, st = o.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, depth, po, cb, i0, c0)
    rm, rd = dxref.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, depth, po, cb, i0, c0)
    print(f"{name}: view march, {st['view_samples']} samples, {st['rays']} rays: {int((rm != cm).sum())} of {int(sum(6 * (N >> int(i0[v, 0])) ** 2 * 4 for v in c0))} marched halves differ, "
          f"{int((rd != cd).sum())} cube depths differ  ({time.time() - t0:.0f} s)", flush=True)
    # the reference's second statement of cull + view march, the work-graph library Bin/LibRayMarch.cso (both nodes, scheduled as the
    # reference dispatches the graph), against the oracle's LibRayMarch mode (no ComputeTargetHit clamp)
    o.load().mvo_set_options(0, 1)
    cml, cdl, _ = o.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, depth, po, cb, i0, c0)
    o.load().mvo_set_options(0, 0)
    dxref.load().dxref_set_threads(o.host_cpu_budget())
    il, vl, recs, rml, rdl = dxref.lib_ray_march(V, N, L, cfg["W"], cfg["H"], vols, lms, depth, po, descs, cb, cfg["maxRay"])
    print(f"{name}: work-graph library: VolumeInfo / visible list identical: {np.array_equal(il, i0) and np.array_equal(vl, np.sort(v0))}; {len(recs)} RayMarchRecords == the cube-map volumes: "
          f"{np.array_equal(recs[:, 2], np.sort(c0))}; {int((rml != cml).sum())} marched halves and {int((rdl != cdl).sum())} cube depths differ from the oracle's LibRayMarch mode "
          f"(which differs from the CSRayMarchV statement in {int((cml != cm).sum())} halves)  ({time.time() - t0:.0f} s)", flush=True)
    W2, H2 = 256, 256   # the composite runs one Python-driven invocation per fragment: a quarter-size viewport
    cfg2 = dict(cfg, W=W2, H=H2)
    _, vp2, svp2, cb2, po2 = host_frame(cfg2, 0)
    depth2, _ = scene.depth_and_shadow(cfg2, vp2, svp2, M)
    i2, v2, c2 = o.cull(V, po2, descs, cb2, cfg["maxRay"])
    cm2, cd2, _ = o.march_view(V, N, L, W2, H2, vols, lms, depth2, po2, cb2, i2, c2)
    args = (V, N, L, W2, H2, vols, lms, depth2, po2, cb2, i2, np.sort(v2), cm2, cd2)
    img, _ = o.composite(*args)
    got = dxref.composite(*args)
    print(f"{name}: composite at {W2}x{H2}: {int((got != img).sum())} of {img.size} halves differ, {int((img[..., 3] != 0).sum())} covered pixels  ({time.time() - t0:.0f} s)", flush=True)
    # what D3D lets a device do differently in the fixed-function stages, through the reference's compiled shaders
    dxref.set_sampler_options(frac_bits=8)
    sm, _ = dxref.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, depth, po, cb, i0, c0)
    dxref.set_sampler_options()
    a = np.concatenate([o.cube_face_view(rm[v], N, int(i0[v, 0])).reshape(-1) for v in c0]).view(np.float16).astype(np.float32)
    b = np.concatenate([o.cube_face_view(sm[v], N, int(i0[v, 0])).reshape(-1) for v in c0]).view(np.float16).astype(np.float32)
    dd = np.abs(a - b)
    print(f"{name}: texel coordinates snapped to 8 fractional bits (the D3D minimum): max |d| {dd.max():.3g}, mean {dd.mean():.3g}, {100.0 * (dd > 1e-3).mean():.2f} % of the marched halves beyond 1e-3, {100.0 * (dd > 0).mean():.1f} % moved at all", flush=True)
    dxref.set_sampler_options(r11_truncate=True)
    _, _, _, cbf, pof = host_frame(cfg, 0)
    lt = dxref.march_light(V, N, L, M, vols, descs, pof, cbf, shadow, None, v0, cfg["maxLight"], np.zeros((V, L, L, L), np.uint32))
    dxref.set_sampler_options()
    lr = dxref.march_light(V, N, L, M, vols, descs, pof, cbf, shadow, None, v0, cfg["maxLight"], np.zeros((V, L, L, L), np.uint32))
    vid = int(v0[0])
    t_, r_ = lt[vid].astype(np.int64), lr[vid].astype(np.int64)
    moved = [float((((r_ >> sh) & ((1 << bits) - 1)) != ((t_ >> sh) & ((1 << bits) - 1))).mean()) for sh, bits in ((0, 11), (11, 11), (22, 10))]
    print(f"{name}: float11 / float10 stores truncated instead of rounded to nearest even: {100 * moved[0]:.0f} % / {100 * moved[1]:.0f} % / {100 * moved[2]:.0f} % of the R / G / B fields of a light map one unit lower  ({time.time() - t0:.0f} s)", flush=True)
