"""CPU suite: the host-side row permutation behind the trunk kernel's edge tiles (grok_alpha_zero_amd/csrc/tile_perm.hpp, through the test
hook of the emulation build).  The kernel variants with static sit-out masks (trunk.hpp conv_taps_static) TRUST these tables — the evaluator
checks them with tile_sitout before it selects such a variant — so: bijection, the masks the kernels assume, balance over the wave rows,
and the LDS bank-conflict condition of the fragment reads."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
    L = C.CDLL(os.path.join(EMU_DIR, "libgaz_emu.so"))
    L.gaz_test_tile_perm.restype = C.c_int
    return L


def _perm(L, H, W, boards, rows, wave_rows, per):
    p = np.zeros(rows, np.uint8); m = np.zeros(rows // 16, np.uint32)
    n = L.gaz_test_tile_perm(H, W, boards, rows, wave_rows, per, p.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
    return (p, m) if n == rows else (None, None)


def _sitout(rows16, H, W, boards):
    """taps on which all 16 rows read zero padding, computed independently of tile_sitout"""
    m = 0x1FF
    for r in rows16:
        if r >= boards * H * W:
            continue
        y, x = divmod(int(r) % (H * W), W)
        for q in range(9):
            dy, dx = q // 3 - 1, q % 3 - 1
            if 0 <= y + dy < H and 0 <= x + dx < W:
                m &= ~(1 << q)
    return m


@pytest.mark.parametrize("H,W,boards,rows,wave_rows,per,want", [
    (6, 7, 3, 128, 64, 2, {0: 0x007, 1: 0x049, 4: 0x1C0, 5: 0x124}),      # Connect4, 128-row tile: trunk.hpp SKIPSET 1
    (6, 7, 2, 96, 96, 3, {0: 0x007, 1: 0x1C0, 2: 0x049}),                 # Connect4, 96-row tile: SKIPSET 2
    (15, 15, 1, 256, 64, 2, {0: 0x007, 4: 0x1C0, 8: 0x049, 12: 0x124, 1: 0x1FF}),     # a Gomoku board (not used by a kernel: four wave-row roles)
    (3, 3, 14, 128, 64, 2, None),                                         # tiny boards: whatever comes out must still be a valid permutation
])
def test_tile_perm_is_a_bijection_with_the_assumed_sitout_masks(lib, H, W, boards, rows, wave_rows, per, want):
    p, m = _perm(lib, H, W, boards, rows, wave_rows, per)
    assert p is not None
    assert sorted(p.tolist()) == list(range(rows))                        # every image row computed exactly once
    for t in range(rows // 16):
        assert m[t] == _sitout(p[16 * t:16 * t + 16], H, W, boards)       # tile_sitout agrees with an independent restatement
    if want:
        for t, mask in want.items():
            assert m[t] & mask == mask, (t, hex(int(m[t])), hex(mask))
    # a wave row can only be as fast as its slowest wave: the edge tiles are dealt out evenly (all-padding tiles aside)
    per_row = [sum(bin(int(m[t]) & 0x1FF).count("1") for t in range(wr * wave_rows // 16, (wr + 1) * wave_rows // 16) if m[t] != 0x1FF)
               for wr in range(rows // wave_rows)]
    assert max(per_row) - min(per_row) <= 0 or want is None, per_row


def test_tile_perm_keeps_fragment_reads_nearly_conflict_free(lib):
    """ds_read_b128 serves lanes {0-3, 12-15} and {4-11} of a 16-row tile together; with the image swizzle they hit different banks iff
    their rows differ in row & 7.  The builder keeps every tile at two rows per residue wherever the edge sets allow it."""
    for (H, W, boards, rows, wave_rows, per) in [(6, 7, 3, 128, 64, 2), (6, 7, 2, 96, 96, 3)]:
        p, _ = _perm(lib, H, W, boards, rows, wave_rows, per)
        clashes = 0
        for t in range(rows // 16):
            tile = p[16 * t:16 * t + 16]
            for grp in ([0, 1, 2, 3, 12, 13, 14, 15], [4, 5, 6, 7, 8, 9, 10, 11]):
                res = [int(tile[i]) & 7 for i in grp]
                clashes += len(res) - len(set(res))
        assert clashes <= 2 * (rows // 16), clashes                       # natural order: 0; picking 16 of an edge's cells by hand costs a few


def test_tile_perm_refuses_shapes_it_cannot_serve(lib):
    assert _perm(lib, 6, 7, 4, 128, 64, 2)[0] is None                     # 168 cells do not fit 128 rows
    assert _perm(lib, 6, 7, 3, 120, 60, 2)[0] is None                     # rows must be whole MFMA tiles
