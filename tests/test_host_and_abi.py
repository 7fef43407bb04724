"""Host-side logic (no GPU): DirectXMath-convention helpers against float64 numpy, the reference's SetVolumesWorld /
UpdateFrame byte layouts, and that libmultivolumes_amd.so exports every symbol the headers in include/ declare."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from multivolumes_amd import _lib, scene
from multivolumes_amd import raycaster as rc
from tests.helpers import host_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(mvlib):
    declared = set()
    for hdr in ("multivolumes.h", "multivolumes_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(mvh?_[a-z0-9_]+)\s*\(", text))
    declared -= {"mv_config", "mv_stats", "mv_ctx"}
    assert len(declared) >= 50
    for name in sorted(declared):
        assert hasattr(mvlib, name), f"{name} is declared in include/ but not exported"
        assert name in _lib.ABI, f"{name} has no ctypes signature in multivolumes_amd/_lib.py"
    hdr = open(os.path.join(ROOT, "include", "multivolumes.h")).read()
    assert mvlib.mv_abi_version() == int(re.search(r"#define MV_ABI_VERSION (\d+)u", hdr).group(1)) == 6
    assert C.sizeof(_lib.mv_config) == 64 and C.sizeof(_lib.mv_stats) == 128
    assert mvlib.mv_error_string(_lib.MV_ERR_NO_DEVICE).decode().startswith("no usable gfx950")


def test_the_headers_are_plain_c_and_link_against_the_library(tmp_path):
    """The drop-in boundary is a C ABI: both headers compile as strict C99 (a cgo / JNI / N-API binding includes them as they are), and a C program that
    uses nothing but them links against the library and reads the same struct sizes and ABI version as the ctypes mirror."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "multivolumes.h"\n#include "multivolumes_host.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%u %u %u %s\\n", (unsigned)sizeof(mv_config), (unsigned)sizeof(mv_stats), mv_abi_version(), mv_error_string(MV_ERR_TIMEOUT)); return 0; }\n')
    libdir = os.path.join(ROOT, "multivolumes_amd")
    exe = tmp_path / "abi.bin"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L" + libdir, "-lmultivolumes_amd", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(None, 3)
    assert out[:3] == ["64", "128", "6"] and "timeout" in out[3]


def test_no_gpu_means_failure_not_fallback(mvlib):
    """On a box without a GPU mv_create must fail loudly (the product has no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multivolumes_amd import MultiRayCaster, MVError
    with pytest.raises(MVError) as e:
        MultiRayCaster(8, 8, 1, 1, (8, 8), shadowMapSize=8)
    assert e.value.code == _lib.MV_ERR_NO_DEVICE


def np_look_at(eye, focus, up):
    eye, focus, up = (np.asarray(a, np.float64) for a in (eye, focus, up))
    z = focus - eye
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z
    m[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    return m


def test_directxmath_conventions():
    eye = (4.0, 16.0, -80.0)
    v = rc.look_at_lh(eye, (0, 0, 0))
    assert np.allclose(v, np_look_at(eye, (0, 0, 0), (0, 1, 0)), atol=1e-6)
    # the eye maps to the origin, the focus to +z (left-handed)
    assert np.allclose(np.array([*eye, 1.0]) @ v.astype(np.float64), [0, 0, 0, 1], atol=1e-4)
    f = np.array([0, 0, 0, 1.0]) @ v.astype(np.float64)
    assert abs(f[0]) < 1e-4 and abs(f[1]) < 1e-4 and f[2] > 0
    p = rc.perspective_fov_lh(math.pi / 4, 16 / 9, 1.0, 1000.0).astype(np.float64)
    near = np.array([0, 0, 1.0, 1]) @ p
    far = np.array([0, 0, 1000.0, 1]) @ p
    assert abs(near[2] / near[3]) < 1e-6 and abs(far[2] / far[3] - 1) < 1e-6
    top = np.array([0, math.tan(math.pi / 8) * 10, 10.0, 1]) @ p
    assert abs(top[1] / top[3] - 1) < 1e-6
    o = rc.orthographic_lh(20.0, 10.0, 1.0, 200.0).astype(np.float64)
    q = np.array([10.0, -5.0, 200.0, 1]) @ o
    assert np.allclose(q, [1, -1, 1, 1], atol=1e-6)
    assert np.allclose(rc.multiply(v, p), v.astype(np.float64) @ p, rtol=1e-5, atol=1e-5)


def test_set_volumes_world_matches_reference_grid():
    """MultiRayCaster.cpp:277-303: rowLength = ceil(sqrt(V)), colLength = ceil((float)(V / rowLength)) (integer division)."""
    lib = _lib.load()
    for V in (1, 2, 4, 5, 16, 64):
        w = np.zeros((V, 12), np.float32)
        lib.mvh_set_volumes_world(V, 20.0, rc._ptr(rc._f32((1, 2, 3), 3)), rc._ptr(w))
        row = math.ceil(math.sqrt(V))
        col = math.ceil(float(V // row))
        placed = min(V, row * col)
        for i in range(V):
            m = w[i].reshape(3, 4)  # XMFLOAT3X4: rows are the columns of the 4x4
            if i < placed:
                assert np.allclose(m[:, :3], 10 * np.eye(3))
                n, mm = i % row, i // row
                assert np.allclose(m[:, 3], [1 - (row / 2 - 0.5) * 30 + 30 * n, 2, 3 - (col / 2 - 0.5) * 30 + 30 * mm])
            else:
                assert not m.any()  # the reference leaves these value-initialised


def test_update_frame_layout():
    """CBPerFrame / PerObject bytes (MultiRayCaster.cpp:23-41, 316-353) decode the way Common.hlsli:28-49 reads them."""
    cfg = scene.CONFIGS["T1"]
    eye, vp, svp, cb, po = host_frame(cfg, frame_idx=7)
    f = cb.view(np.float32)
    assert np.allclose(f[0:4], [*eye, 1]) and np.allclose(f[4:8], [cfg["W"], cfg["H"], 0, 0])
    assert np.allclose(f[40:44], [*scene.LIGHT_PT, 1])
    assert cb.view(np.uint32)[52] == 7
    s2w = f[8:24].reshape(4, 4).T.astype(np.float64)  # stored transposed
    assert np.allclose(vp.astype(np.float64) @ s2w, np.eye(4), atol=1e-3)
    assert np.array_equal(f[24:40].reshape(4, 4), svp)
    p = po.view(np.float32).reshape(cfg["V"], 56)
    for i in range(cfg["V"]):
        wvp = p[i, 0:16].reshape(4, 4).T.astype(np.float64)
        wvpI = p[i, 16:32].reshape(4, 4).T.astype(np.float64)
        worldI = np.vstack([p[i, 32:44].reshape(3, 4), [0, 0, 0, 1]]).T.astype(np.float64)
        world = np.vstack([p[i, 44:56].reshape(3, 4), [0, 0, 0, 1]]).T.astype(np.float64)
        assert np.allclose(world @ worldI, np.eye(4), atol=1e-5)
        assert np.allclose(wvp @ wvpI, np.eye(4), atol=1e-3)
        assert np.allclose(world @ vp.astype(np.float64), wvp, rtol=1e-5, atol=1e-4)
        assert np.allclose(world[:3, :3], 10 * np.eye(3))


def test_synthetic_inputs_are_deterministic_and_sane(oracle):
    a, b = rc.synth_volume(16, 3), rc.synth_volume(16, 3)
    assert np.array_equal(a, b)
    assert np.array_equal(rc.synth_volume(16, 0), oracle.init_grid_data(16))
    base = rc.synth_volume(16, 0).view(np.float16).astype(np.float32)
    n = a.view(np.float16).astype(np.float32)
    assert np.array_equal(n[..., :3], base[..., :3])
    r = n[..., 3][base[..., 3] > 0.1] / base[..., 3][base[..., 3] > 0.1]
    assert r.min() >= 0.249 and r.max() <= 1.001 and 0.4 < r.mean() < 0.8
    cfg = scene.CONFIGS["T1"]
    eye, vp = scene.camera(cfg)
    d = rc.synth_sphere_depth(64, 64, vp, (0, 0, 0), 8.0)
    assert d[0, 0] == 1.0 and d[32, 32] < 1.0
    # centre pixel: distance from the eye to the sphere front along the view axis
    dist = np.linalg.norm(eye) - 8.0
    assert abs(d[32, 32] - 1000.0 * (dist - 1.0) / (dist * 999.0)) < 5e-4
    s = rc.synth_sphere_shadow(64, scene.shadow_vp(dict(cfg, mesh=True)), (0, 0, 0), 8.0)
    assert s[0, 0] == 65535 and s[32, 32] < 65535


def test_roofline_fraction_is_a_fraction_on_every_workload():
    """bench.py's roofline line: SURVEY 8(d)'s algorithmic bytes while they are a fraction of the peak; on workloads
    that are mostly served without HBM (C3-stress printed 1.40) the requested bytes, then the measured traffic, then NO fraction (not an invented one) --
    and the algorithmic figure stays beside it, uncapped."""
    import bench
    peak = bench.HBM_PEAK_GBS
    ms = 0.25
    at = lambda frac: frac * peak * 1e9 * ms * 1e-3   # bytes per launch that give that fraction
    r = bench.roofline_fraction(at(0.727), at(0.57), at(0.64), ms)
    assert abs(r["frac"] - 0.727) < 1e-9 and "algorithmic" in r["frac_basis"] and "frac_algorithmic_uncapped" not in r
    r = bench.roofline_fraction(at(1.40), at(0.80), at(0.40), ms)
    assert abs(r["frac"] - 0.80) < 1e-9 and abs(r["frac_algorithmic_uncapped"] - 1.40) < 1e-9 and "requested" in r["frac_basis"]
    r = bench.roofline_fraction(at(1.40), at(1.10), at(0.40), ms)
    assert abs(r["frac"] - 0.40) < 1e-9 and "traffic" in r["frac_basis"]
    r = bench.roofline_fraction(at(1.40), at(1.10), None, ms)
    assert r["frac"] is None and r["achieved"] is None and "none" in r["frac_basis"], "no measurement: no fraction (ADVICE r04: never an invented 1.0)"
    assert bench.roofline_fraction(1.0, 1.0, None, 0.0)["frac"] is None
    assert bench.LEVELS[0] == "peer" and bench.LEVELS.index("rccl-single-stream") < bench.LEVELS.index("rccl")
    # (the peer-store transport twice: four frames per launch first, then -- should that form fail on a fabric it has never met -- a frame per launch)
    assert bench.LEVELS[1] == "peer-one-frame-per-launch"
