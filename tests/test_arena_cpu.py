"""CPU: the range bookkeeping of the context arena (csrc/tsq_arena.h, tsq_ctx_reserve) under random allocate / release
sequences, walked by tests/hostsim: live blocks never overlap, free neighbours always merge, everything released = one range."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("slab,steps", [(1 << 12, 2000), (1 << 20, 20000), (3 << 28, 5000)])
def test_arena_ranges_random_sequences(slab, steps):
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
    lib.sim_arena.restype = C.c_int32
    lib.sim_arena.argtypes = [C.c_uint64, C.c_int32, C.c_uint64]
    for seed in range(1, 6):
        assert lib.sim_arena(slab, steps, seed) == 0
