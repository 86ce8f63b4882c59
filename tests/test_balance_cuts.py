"""The slab cuts every rank derives from the summed block-plane histogram (sparkl_hip.hip::balance_cuts): pure host arithmetic
behind sparkl_hip_debug_balance_cuts, so it is tested here on the CPU.  Planes are block planes; hist index = plane + 1."""
import ctypes as C

import numpy as np

from sparkl_amd import _lib

NP = 1024


def _cuts(per_plane, first_plane, cuts, max_shift=4):
    """per_plane: particles of consecutive block planes starting at `first_plane`; cuts: current cuts (first plane of ranks 1..)."""
    L = _lib.lib()
    hist = np.zeros(NP, dtype=np.int64)
    hist[first_plane + 1: first_plane + 1 + len(per_plane)] = per_plane
    world = len(cuts) + 1
    cur = np.zeros(world + 1, dtype=np.int64)
    cur[1:world] = cuts
    out = np.zeros(world + 1, dtype=np.int64)
    L.sparkl_hip_debug_balance_cuts.restype = C.c_int
    rc = L.sparkl_hip_debug_balance_cuts(hist.ctypes.data_as(C.c_void_p), C.c_int(NP), cur.ctypes.data_as(C.c_void_p), C.c_int(world), C.c_int(max_shift),
                                         out.ctypes.data_as(C.c_void_p))
    assert rc in (0, 1)
    return (list(out[1:world]) if rc == 1 else list(cuts)), rc


def _loads(per_plane, first_plane, cuts):
    edges = [first_plane] + list(cuts) + [first_plane + len(per_plane)]
    return [int(sum(per_plane[a - first_plane: b - first_plane])) for a, b in zip(edges[:-1], edges[1:])]


def test_balanced_cuts_stay():
    per = [1000] * 24
    cuts, rc = _cuts(per, 1, [7, 13, 19])
    assert cuts == [7, 13, 19]


def test_cuts_follow_a_drifting_histogram():
    # the soak of tools/probe.py slabs --drift: 30 planes, the bed has moved two planes up since the cuts were chosen
    per = [0, 0] + [58000] * 30
    cuts, rc = _cuts(per, 1, [8, 15, 23])
    assert rc == 1 and cuts == [10, 18, 25]
    loads = _loads(per, 1, cuts)
    assert max(loads) / (sum(loads) / 4) <= 1.08


def test_slivers_are_widened_one_rank_at_a_time():
    # three ranks start with one block plane each, the fourth with the rest (tests/test_gpu_slab.py's four-slab case): emigrants may only go
    # to an adjacent rank, so a cut never passes its neighbours' OLD positions; repeated re-bins converge to the quantiles
    per = [4096] * 9
    cuts = [1, 2, 3]
    seen = []
    for _ in range(6):
        new, rc = _cuts(per, 0, cuts)
        for k, c in enumerate(new):      # one rank far: between the old neighbouring cuts
            lo = cuts[k - 1] if k > 0 else 0
            hi = cuts[k + 1] if k + 1 < len(cuts) else 9
            assert lo <= c <= hi, (cuts, new)
        assert all(b > a for a, b in zip([0] + new, new + [9])), new
        cuts = new
        seen.append(list(new))
    loads = _loads(per, 0, cuts)
    assert seen[0] != [1, 2, 3] and max(loads) <= 3 * 4096, (seen, loads)    # 9 planes on 4 ranks: 3 + 2 + 2 + 2 at best


def test_eight_ranks_reach_plane_granularity():
    rng = np.random.default_rng(5)
    per = list((50000 + rng.integers(-5000, 5000, size=48)).astype(int))
    cuts = [2, 4, 6, 8, 10, 12, 14]          # everything piled on the last rank
    for _ in range(12):
        cuts, rc = _cuts(per, 3, cuts)
    loads = _loads(per, 3, cuts)
    assert max(loads) <= sum(loads) / 8 + max(per), (cuts, loads)


def test_empty_histogram_and_single_rank_leave_the_cuts():
    assert _cuts([0] * 8, 1, [3, 5])[1] == 0
    assert _cuts([10] * 8, 1, [])[1] == 0
