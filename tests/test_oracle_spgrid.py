"""Pins the oracle's SPGrid addressing against the reference's own tests and constants
(/root/reference/src/geometry/sp_grid.rs:411-555).  This is the only part of the path the reference pins."""
import ctypes as C
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "spgrid_constants.json")))


def test_nbh_table_is_valid(oracle_lib):
    L = oracle_lib
    u32 = lambda v: v & 0xFFFFFFFF
    # sp_grid.rs:417-435: packi(-1,-1,-1), packi(-2,-2,-2), pack(5,5,5)
    assert int(L.oracle_pack(u32(-1), u32(-1), u32(-1))) == GOLD["PACKED_NEG_ONE"]
    assert int(L.oracle_pack(u32(-2), u32(-2), u32(-2))) == GOLD["PACKED_NEG_TWO"]
    assert int(L.oracle_pack(5, 5, 5)) == GOLD["PACKED_PLUS_FIVE"]
    # sp_grid.rs:444-455: PACKED_NBH_SHIFTS in sync with NBH_SHIFTS
    for k in range(27):
        s = (C.c_int * 3)()
        L.oracle_nbh_shift(k, s)
        assert list(s) == GOLD["NBH_SHIFTS"][k]
        assert int(L.oracle_packed_nbh_shift(k)) == GOLD["PACKED_NBH_SHIFTS"][k]
        assert int(L.oracle_pack(*GOLD["NBH_SHIFTS"][k])) == GOLD["PACKED_NBH_SHIFTS"][k]
    # sp_grid.rs:457-469: region shifts
    for k in range(7):
        assert int(L.oracle_packed_region_shift(k)) == GOLD["PACKED_NBH_REGION_SHIFTS"][k]
        assert int(L.oracle_pack(*GOLD["NBH_REGION_SHIFTS"][k])) == GOLD["PACKED_NBH_REGION_SHIFTS"][k]


def test_morton_roundtrip(oracle_lib):
    # sp_grid.rs:472-485: unpack(pack(c)) == c on [0,100) x [0,50)^2
    assert int(oracle_lib.oracle_check_pack_roundtrip(0, 100, 0, 50, 0, 50)) == 0


def test_morton_sum(oracle_lib):
    # sp_grid.rs:502-529: packed_add(pack(c), PACKED_NBH_SHIFTS[s]) == pack(c + NBH_SHIFTS[s]) for shifted ranges
    for k in range(10):
        sh = 2 ** k
        assert int(oracle_lib.oracle_check_packed_add(1 + sh, 50 + sh, 1 + sh, 25 + sh, 1 + sh, 25 + sh)) == 0


def test_region_colouring(oracle_lib):
    # sp_grid.rs:184-190: colour = low 3 Morton bits of the block; the 8 blocks of a 2x2x2 group get 8 colours
    cols = set()
    for bx in range(2):
        for by in range(2):
            for bz in range(2):
                rid = int(oracle_lib.oracle_pack(4 * bx + 8, 4 * by + 16, 4 * bz + 24)) & ~0xFFF
                cols.add(int(oracle_lib.oracle_region_color(rid)))
    assert cols == set(range(8))


def test_domain_is_256_cells(oracle_lib):
    # the intended 1 GiB map holds coordinates 0..255 per axis
    assert int(oracle_lib.oracle_pack(255, 255, 255)) + 48 <= 2 ** 30
    assert int(oracle_lib.oracle_pack(256, 0, 0)) + 48 > 2 ** 30
