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


def _layout(L, H, W, boards, rows, wave_rows, per):
    p = np.zeros(rows, np.uint8); m = np.zeros(rows // 16, np.uint32); bo = np.zeros(max(boards, 1), np.int32); cl = C.c_int(-1)
    n = L.gaz_test_tile_perm(H, W, boards, rows, wave_rows, per, p.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p),
                             bo.ctypes.data_as(C.c_void_p), C.byref(cl))
    return (p, m, bo.tolist(), cl.value) if n == rows else (None, None, None, None)


def _perm(L, H, W, boards, rows, wave_rows, per):
    return _layout(L, H, W, boards, rows, wave_rows, per)[:2]


def _cell(r, boff, HW):
    for b, o in enumerate(boff):
        if o <= r < o + HW:
            return b, r - o
    return -1, 0


def _sitout(rows16, H, W, boff):
    """taps on which all 16 rows read zero padding, computed independently of tile_sitout"""
    m = 0x1FF
    for r in rows16:
        b, cell = _cell(int(r), boff, H * W)
        if b < 0:
            continue
        y, x = divmod(cell, W)
        for q in range(9):
            dy, dx = q // 3 - 1, q % 3 - 1
            if 0 <= y + dy < H and 0 <= x + dx < W:
                m &= ~(1 << q)
    return m


@pytest.mark.parametrize("H,W,boards,rows,wave_rows,per,want", [
    (6, 7, 3, 128, 64, 2, {0: 0x007, 1: 0x049, 4: 0x1C0, 5: 0x124}),      # Connect4, 128-row tile: trunk.hpp SKIPSET 1
    (6, 7, 2, 96, 96, 3, {0: 0x007, 1: 0x1C0, 2: 0x049}),                 # Connect4, 96-row tile: SKIPSET 2
    (15, 15, 1, 256, 64, 2, {0: 0x007, 4: 0x1C0, 8: 0x049, 12: 0x124}),      # a Gomoku board (not used by a kernel: four wave-row roles)
    (3, 3, 14, 128, 64, 2, None),                                         # tiny boards: whatever comes out must still be a valid permutation
])
def test_tile_perm_is_a_bijection_with_the_assumed_sitout_masks(lib, H, W, boards, rows, wave_rows, per, want):
    p, m, boff, _ = _layout(lib, H, W, boards, rows, wave_rows, per)
    assert p is not None
    assert sorted(p.tolist()) == list(range(rows))                        # every image row computed exactly once
    assert all(boff[b] + H * W <= (boff[b + 1] if b + 1 < boards else rows) for b in range(boards)) and boff[0] >= 0     # boards do not overlap
    for t in range(rows // 16):
        assert m[t] == _sitout(p[16 * t:16 * t + 16], H, W, boff)         # tile_sitout agrees with an independent restatement
    if want:
        for t, mask in want.items():
            assert m[t] & mask == mask, (t, hex(int(m[t])), hex(mask))
    # a wave row can only be as fast as its slowest wave: the edge tiles are dealt out evenly (all-padding tiles aside)
    per_row = [sum(bin(int(m[t]) & 0x1FF).count("1") for t in range(wr * wave_rows // 16, (wr + 1) * wave_rows // 16) if m[t] != 0x1FF)
               for wr in range(rows // wave_rows)]
    assert max(per_row) - min(per_row) <= 0 or want is None, per_row


def test_tile_layout_makes_every_fragment_read_conflict_free_on_every_tap(lib):
    """the round-2 permutation took SQ_LDS_BANK_CONFLICT from 9.6 M to 35.6 M cycles per launch.  A ds_read_b128 serves lanes
    {0-3, 12-15} and {4-11} of a 16-row tile together (per k-group); with the image swizzle they hit different banks iff their rows differ in
    row & 7.  Round 3: board offsets in the image + an exact assignment give every tile exactly two rows per residue — checked here ON EVERY TAP,
    with the addresses the kernel forms (tap-shifted rows crow + off, masked lanes on the zero rows ZROW + ((crow + off) & 15)), through the
    lane-group / bank rule of MI355X_MICROARCH.md (tools/lds_bank_sim.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_bank_sim as S
    for (H, W, boards, rows, wave_rows, per, n_ct, skip) in [(6, 7, 3, 128, 64, 2, 4, S.SKIP_BIG), (6, 7, 2, 96, 96, 3, 2, S.SKIP_SMALL)]:
        p, m, boff, clashes = _layout(lib, H, W, boards, rows, wave_rows, per)
        assert clashes == 0, (rows, boff, clashes)
        for t in range(rows // 16):
            tile = p[16 * t:16 * t + 16]
            for grp in ([0, 1, 2, 3, 12, 13, 14, 15], [4, 5, 6, 7, 8, 9, 10, 11]):
                assert sorted(int(tile[i]) & 7 for i in grp) == list(range(8)), (rows, t, tile)
        reads, writes, n_reads, _ = S.tile_conflicts(p.tolist(), H, W, boff, rows, wave_rows, n_ct, skip)
        nat_reads, nat_writes, _, _ = S.tile_conflicts(None, H, W, [b * H * W for b in range(boards)], rows, wave_rows, n_ct, None)
        assert reads == 0 and nat_reads == 0                              # no fragment read of any tap has a bank conflict
        assert writes <= nat_writes                                       # the 8-byte epilogue accesses: no worse than the natural order's inherent 2-way


def test_tile_perm_refuses_shapes_it_cannot_serve(lib):
    assert _perm(lib, 6, 7, 4, 128, 64, 2)[0] is None                     # 168 cells do not fit 128 rows
    assert _perm(lib, 6, 7, 3, 120, 60, 2)[0] is None                     # rows must be whole MFMA tiles
