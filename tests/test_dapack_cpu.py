"""CPU: the arithmetic the packed-key routes rest on (csrc/tsq_dapack.h), walked by tests/hostsim: the mix of `key - kmin` is a
bijection of [0, 2^b) for every b the routes use (so equal entries ARE equal keys, util/codec/codec.go:363-382), and the composite of
several key columns is equal exactly when all cells are (codec.go:243-338)."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim():
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
    lib.sim_da_mix.restype = C.c_int32
    lib.sim_da_mix.argtypes = [C.c_int32, C.c_int64, C.c_uint64]
    lib.sim_da_compose.restype = C.c_int32
    lib.sim_da_compose.argtypes = [C.c_int32, C.c_int32, C.c_uint64]
    return lib


def test_mix_is_a_bijection_for_every_domain_width(sim):
    assert sim.sim_da_mix(22, 2_000_000, 7) == 0  # b <= 22 exhaustively, 23..31 on two million random values each


@pytest.mark.parametrize("n_keys", [1, 2, 3, 4])
def test_composite_keys_are_equal_exactly_when_all_cells_are(sim, n_keys):
    for seed in range(1, 9):
        assert sim.sim_da_compose(n_keys, 600, seed) == 0
