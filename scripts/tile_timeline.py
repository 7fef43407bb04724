#!/usr/bin/env python3
"""Where a lone view-march launch (tiles) spends its time -- needs a -DMV_TILE_TIMELINE build loaded through MV_LIB_PATH: per-wavefront stamps
(100 MHz wall clock) of the tiles in the first `marchWaves` dispatch slots, i.e. the longest ones, dispatched first: start, loop start, end,
iterations.  Prints the launch's span, how late wavefronts start, set-up time, time per iteration by cost bin, and who ends last."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multivolumes_amd import MultiRayCaster, scene
from multivolumes_amd import raycaster as rc
from multivolumes_amd import distributed as D

cfg = scene.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=1024) as c:
    for k in range(cfg["V"]):
        c.LoadVolumeData(k, rc.synth_volume(cfg["N"], k))
    fr = scene.setup_raycaster(c, cfg)
    ptr, n = c.GetBuffer(8)
    view = D._device_view(ptr, n)
    frames, raws = [], []
    for f in range(40):
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.synchronize(); view.zero_(); torch.cuda.synchronize()
        c.Render()
        c.synchronize()
        if f >= 24:
            t = view.cpu().numpy().view(np.uint64).reshape(-1, 8).astype(np.int64)
            frames.append(t[t[:, 2] > 0]); raws.append(t)
    # the same tiles on different CUs (the diagnostic build rotates the first 4096 slots by the frame): what is the tile's, what is the CU's?
    T = np.concatenate(frames)
    F = np.concatenate([np.full(len(x), i) for i, x in enumerate(frames)])
    keep = (T[:, 3] & 0xffff) >= 16
    T, F = T[keep], F[keep]
    y = np.log(((T[:, 2] - T[:, 1]) / 100.0) / (T[:, 3] & 0xffff))
    tile_id = np.unique(T[:, 5], return_inverse=True)[1]
    cu_id = np.unique((((T[:, 4] >> 32) & 0xf) << 8) | ((T[:, 4] >> 8) & 0xff), return_inverse=True)[1]
    a, b, g = np.zeros(tile_id.max() + 1), np.zeros(cu_id.max() + 1), np.zeros(len(frames))
    for _ in range(30):
        a = np.bincount(tile_id, y - b[cu_id] - g[F]) / np.maximum(np.bincount(tile_id), 1)
        b = np.bincount(cu_id, y - a[tile_id] - g[F]) / np.maximum(np.bincount(cu_id), 1); b -= b.mean()
        g = np.bincount(F, y - a[tile_id] - b[cu_id]) / np.maximum(np.bincount(F), 1); g -= g.mean()
    r = y - a[tile_id] - b[cu_id] - g[F]
    print(f"{len(frames)} frames, {len(y)} (tile, CU) observations of {len(a)} tiles on {len(b)} CUs: log(us per iteration) = tile + CU + frame + rest: sd of the tile term {a[tile_id].std():.3f}, "
          f"of the CU term {b[cu_id].std():.3f} (slowest CUs {np.sort(np.exp(b))[-4:].round(2).tolist()}, fastest {np.sort(np.exp(b))[:4].round(2).tolist()}), frame {g.std():.3f}, rest {r.std():.3f}")
    spans = [(x[:, 2].max() - x[:, 0].min()) / 100.0 for x in frames]
    print("  span of the recorded wavefronts by frame (rotation = frame mod 8):", [round(v, 1) for v in spans])
    # does the rest belong to the company a tile keeps?  mean tile term of the other recorded wavefronts on its CU in that frame
    key = F * 4096 + cu_id
    tot, cnt = np.bincount(key, a[tile_id]), np.bincount(key)
    others = (tot[key] - a[tile_id]) / np.maximum(cnt[key] - 1, 1)
    print(f"  correlation of the rest with the mean tile term of the wavefront's CU-mates {np.corrcoef(r, others)[0, 1]:.2f}; a CU's mates' mean tile term has sd {others.std():.3f}")
    # dispatch slot -> CU: is it a fixed pattern?  (the record's row IS the dispatch slot)
    raw = [view_np for view_np in raws]
    cu_of = [((((x[:4096, 4] >> 32) & 0xf) << 8) | ((x[:4096, 4] >> 8) & 0xff)) for x in raw]
    same = np.mean([np.mean(cu_of[i] == cu_of[0]) for i in range(1, len(cu_of))])
    periods = [P for P in (256, 512, 1024, 2048) if np.mean(cu_of[0][:-P] == cu_of[0][P:]) > 0.99]
    print(f"  slot -> CU: the same CU as in the first captured frame for {same:.3f} of the slots; periods found {periods}; slots 0..15 sit on XCD {((raw[0][:16, 4] >> 32) & 0xf).tolist()}, "
          f"CU word {((raw[0][:16, 4] >> 8) & 0xff).tolist()}; agreement at period 512: {np.mean(cu_of[0][:-512] == cu_of[0][512:]):.3f}")
    t = frames[-1]
    us = lambda x: x / 100.0
    t0 = t[:, 0].min()
    start, loop, end, tile = t[:, 0], t[:, 1], t[:, 2], t[:, 5]
    iters, bins, face, volume = t[:, 3] & 0xffff, (t[:, 3] >> 16) & 0xf, (t[:, 3] >> 20) & 0xf, (t[:, 3] >> 24) & 0xff
    hw, xcc = t[:, 4] & 0xffffffff, (t[:, 4] >> 32) & 0xf
    lit, samples, skipped = t[:, 6] & 0xffffffff, t[:, 6] >> 32, t[:, 7]
    cu = (xcc << 8) | ((hw >> 8) & 0xff)          # XCD | SE | SH | CU
    simd = (cu << 2) | ((hw >> 4) & 3)
    print(f"wavefronts recorded {len(t)} (first dispatch slots); span of their launch {us(end.max() - t0):.1f} us; starts spread over {us(start.max() - t0):.1f} us "
          f"(median {us(np.median(start) - t0):.1f}); set-up {us((loop - start).mean()):.2f} us mean, {us((loop - start).max()):.2f} max")
    per = us(end - loop) / np.maximum(iters, 1)
    for b in sorted(set(bins.tolist())):
        m = bins == b
        print(f"  bin {b}: {m.sum():5d} tiles, iterations {iters[m].mean():6.1f} (max {iters[m].max()}), us per iteration {per[m].mean():.3f} (p90 {np.percentile(per[m], 90):.3f}), "
              f"life {us((end - start)[m]).mean():6.1f} us, ends at {us(end[m] - t0).mean():6.1f} (max {us(end[m].max() - t0):.1f})")
    last = np.argsort(end)[-8:]
    print("  the eight that end last: " + ", ".join(f"[vol {volume[i]} face {face[i]} bin {bins[i]} it {iters[i]} end {us(end[i] - t0):.0f} us/it {per[i]:.2f} lit {lit[i] / max(samples[i], 1):.2f} free {skipped[i] / max(samples[i], 1):.2f}]" for i in last))
    span = us(end.max() - t0)
    edges = np.linspace(0, span, 11)[:-1] + span / 20
    print("  recorded wavefronts alive at 5 %, 15 % .. 95 % of the span:", [int(((us(start - t0) <= x) & (us(end - t0) > x)).sum()) for x in edges])
    # what makes a tile slow per iteration: its samples (lit = eight texels + two light quads fetched; free = no fetch at all), its face, or its neighbours on the CU?
    litf, freef, util = lit / np.maximum(samples, 1), skipped / np.maximum(samples, 1), samples / np.maximum(iters * 64, 1)
    A = np.stack([np.ones(len(t)), litf, freef, util], 1)
    coef, *_ = np.linalg.lstsq(A, per, rcond=None)
    res = per - A @ coef
    print(f"  us per iteration ~ {coef[0]:.2f} + {coef[1]:.2f} lit fraction + {coef[2]:.2f} fetch-free fraction + {coef[3]:.2f} lane utilisation; explains {1 - res.var() / per.var():.2f} of the variance "
          f"(lit {litf.mean():.2f}, free {freef.mean():.2f}, utilisation {util.mean():.2f})")
    print("  by face (7 = screen tile): " + "  ".join(f"{f}: n {int((face == f).sum())} it {iters[face == f].mean():.0f} us/it {per[face == f].mean():.2f}" for f in sorted(set(face.tolist()))))
    print("  by volume: " + " ".join(f"{v}:{per[volume == v].mean():.2f}/{int((volume == v).sum())}" for v in sorted(set(volume.tolist()))))
    cus = sorted(set(cu.tolist()))
    n_cu = np.array([(cu == c).sum() for c in cus]); per_cu = np.array([per[cu == c].mean() for c in cus]); res_cu = np.array([res[cu == c].mean() for c in cus]); end_cu = np.array([us(end[cu == c].max() - t0) for c in cus])
    print(f"  CUs seen {len(cus)}: recorded wavefronts per CU {n_cu.min()}-{n_cu.max()} (mean {n_cu.mean():.1f}); a CU's mean us per iteration {per_cu.min():.2f}-{per_cu.max():.2f} (sd {per_cu.std():.2f}; "
          f"of the residual after the fit above: sd {res_cu.std():.2f} -- against {res.std() / np.sqrt(n_cu.mean()):.2f} if wavefronts were independent); last end per CU {end_cu.min():.0f}-{end_cu.max():.0f} us (mean {end_cu.mean():.0f})")
    print(f"    correlation of a CU's wavefront count with its mean us per iteration {np.corrcoef(n_cu, per_cu)[0, 1]:.2f}; with its total iterations {np.corrcoef(np.array([iters[cu == c].sum() for c in cus]), end_cu)[0, 1]:.2f} (vs last end)")
    sim = sorted(set(simd.tolist())); n_s = np.array([(simd == x).sum() for x in sim])
    print(f"    SIMDs seen {len(sim)}: recorded wavefronts per SIMD {n_s.min()}-{n_s.max()}, histogram {np.bincount(n_s).tolist()}")
    print("  by XCD: ends at (mean) " + " ".join(f"{x}:{us(end[xcc == x] - t0).mean():.0f}" for x in sorted(set(xcc.tolist()))))
