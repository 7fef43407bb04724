"""DDS volume ingestion: the front end of LoadVolumeData(fileName) -- header parsing on CPU, the load through
CSR32FToRGBA16F on the GPU.  The files are written here from the public DDS layout (the reference's assets are not shipped)."""
import struct

import numpy as np
import pytest

from multivolumes_amd import raycaster as rc

DDSD_CAPS, DDSD_HEIGHT, DDSD_WIDTH, DDSD_PIXELFORMAT, DDSD_DEPTH = 0x1, 0x2, 0x4, 0x1000, 0x800000
DDPF_FOURCC, DDPF_RGB, DDPF_LUMINANCE = 0x4, 0x40, 0x20000
DDSCAPS2_VOLUME = 0x200000


def write_dds(path, payload, w, h, d, fourcc=0, pf_flags=DDPF_FOURCC, bits=0, masks=(0, 0, 0, 0), dx10=None, volume=True):
    hdr = [0] * 31
    hdr[0] = 124
    hdr[1] = DDSD_CAPS | DDSD_HEIGHT | DDSD_WIDTH | DDSD_PIXELFORMAT | (DDSD_DEPTH if volume else 0)
    hdr[2], hdr[3], hdr[5] = h, w, d if volume else 0
    hdr[18] = 32                      # DDS_PIXELFORMAT.dwSize
    hdr[19], hdr[20], hdr[21] = pf_flags, fourcc, bits
    hdr[22:26] = masks
    hdr[26] = 0x1000                  # DDSCAPS_TEXTURE
    hdr[27] = DDSCAPS2_VOLUME if volume else 0
    with open(path, "wb") as f:
        f.write(b"DDS " + struct.pack("<31I", *hdr))
        if dx10 is not None:
            f.write(struct.pack("<5I", dx10[0], dx10[1], 0, 1, 0))  # dxgiFormat, resourceDimension, misc, arraySize, misc2
        f.write(payload)


def test_dds_volume_formats(tmp_path):
    rng = np.random.default_rng(5)
    w, h, d = 7, 5, 3
    a = rng.random((d, h, w), dtype=np.float32)
    FOURCC_DX10 = struct.unpack("<I", b"DX10")[0]
    cases = {
        "dx10_r32f": (a.tobytes(), dict(fourcc=FOURCC_DX10, dx10=(41, 4)), a),
        "dx10_r16f": (a.astype(np.float16).tobytes(), dict(fourcc=FOURCC_DX10, dx10=(54, 4)), a.astype(np.float16).astype(np.float32)),
        "dx10_r8": ((a * 255).astype(np.uint8).tobytes(), dict(fourcc=FOURCC_DX10, dx10=(61, 4)), (a * 255).astype(np.uint8).astype(np.float32) / np.float32(255)),
        "dx10_r16": ((a * 65535).astype(np.uint16).tobytes(), dict(fourcc=FOURCC_DX10, dx10=(56, 4)), (a * 65535).astype(np.uint16).astype(np.float32) / np.float32(65535)),
        "legacy_r32f": (a.tobytes(), dict(fourcc=114), a),
        "legacy_r16f": (a.astype(np.float16).tobytes(), dict(fourcc=111), a.astype(np.float16).astype(np.float32)),
        "legacy_l8": ((a * 255).astype(np.uint8).tobytes(), dict(pf_flags=DDPF_LUMINANCE, bits=8, masks=(0xff, 0, 0, 0)), (a * 255).astype(np.uint8).astype(np.float32) / np.float32(255)),
    }
    rgba = np.zeros((d, h, w, 4), np.float32)
    rgba[..., 0] = a
    rgba[..., 1:] = 0.5
    cases["dx10_rgba32f"] = (rgba.tobytes(), dict(fourcc=FOURCC_DX10, dx10=(2, 4)), a)
    cases["legacy_rgba16f"] = (rgba.astype(np.float16).tobytes(), dict(fourcc=113), a.astype(np.float16).astype(np.float32))
    cases["legacy_a8b8g8r8"] = ((rgba * 255).astype(np.uint8).tobytes(), dict(pf_flags=DDPF_RGB, bits=32, masks=(0xff, 0xff00, 0xff0000, 0xff000000)),
                                (a * 255).astype(np.uint8).astype(np.float32) / np.float32(255))
    for name, (payload, kw, expect) in cases.items():
        p = tmp_path / f"{name}.dds"
        write_dds(p, payload, w, h, d, **kw)
        got = rc.load_dds_volume(p)
        assert got.shape == (d, h, w) and got.dtype == np.float32, name
        assert np.array_equal(got, expect), name


def test_dds_errors(tmp_path):
    a = np.zeros((2, 2, 2), np.float32)
    FOURCC_DX10 = struct.unpack("<I", b"DX10")[0]
    with pytest.raises(IOError, match="unreadable"):
        rc.load_dds_volume(tmp_path / "missing.dds")
    (tmp_path / "junk.dds").write_bytes(b"not a dds file at all" * 10)
    with pytest.raises(IOError, match="not a DDS"):
        rc.load_dds_volume(tmp_path / "junk.dds")
    write_dds(tmp_path / "flat.dds", a.tobytes(), 2, 2, 2, fourcc=114, volume=False)
    with pytest.raises(IOError, match="not a volume"):
        rc.load_dds_volume(tmp_path / "flat.dds")
    write_dds(tmp_path / "tex2d.dds", a.tobytes(), 2, 2, 2, fourcc=FOURCC_DX10, dx10=(41, 3))  # TEXTURE2D
    with pytest.raises(IOError, match="not a volume"):
        rc.load_dds_volume(tmp_path / "tex2d.dds")
    write_dds(tmp_path / "bc1.dds", a.tobytes(), 2, 2, 2, fourcc=struct.unpack("<I", b"DXT1")[0])
    with pytest.raises(IOError, match="unsupported"):
        rc.load_dds_volume(tmp_path / "bc1.dds")
    write_dds(tmp_path / "short.dds", a.tobytes()[:-4], 2, 2, 2, fourcc=114)
    with pytest.raises(IOError, match="truncated"):
        rc.load_dds_volume(tmp_path / "short.dds")


@pytest.mark.gpu
def test_volume_file_through_the_gpu_loader(tmp_path, oracle):
    """LoadVolumeFile = DDS -> CSR32FToRGBA16F on the GPU; equals the oracle's restatement of that shader, bit for bit."""
    from multivolumes_amd import MultiRayCaster
    rng = np.random.default_rng(9)
    N = 24
    src = rng.random((20, 28, 36), dtype=np.float32)  # (d, h, w) != N: the loader resamples with the linear sampler
    write_dds(tmp_path / "cloud.dds", src.tobytes(), 36, 28, 20, fourcc=struct.unpack("<I", b"DX10")[0], dx10=(41, 4))
    with MultiRayCaster(N, 8, 1, 1, (64, 64), shadowMapSize=16) as c:
        c.LoadVolumeFile(0, tmp_path / "cloud.dds")
        got = c.ReadBackVolume(0)
    exp = oracle.r32f_to_rgba16f(src, N)
    assert np.array_equal(np.asarray(got).reshape(-1), np.asarray(exp).reshape(-1))
