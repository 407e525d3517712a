import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure; built on demand with g++)."""
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def ctx():
    """one libtsq context on cuda:0 — GPU tests only.  No fallback: fails loudly without a GPU."""
    from tinysql_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _knobs_back_to_default(request):
    """a test may turn the context's test knobs (tsq_ctx_set_knob); the session-wide context gets its defaults back afterwards"""
    # TSQ_TEST_KNOBS="NAME=VALUE,..." (test runs only; the product reads no environment): the whole run with a knob off its default,
    # e.g. an A/B of a kernel variant over the full parity suite before its default is switched
    preset = [kv.split("=") for kv in os.environ.get("TSQ_TEST_KNOBS", "").split(",") if "=" in kv]
    if preset and "ctx" in request.fixturenames:
        from tinysql_amd import _abi as abi
        for k, v in preset:
            request.getfixturevalue("ctx").set_knob(getattr(abi, "KNOB_" + k), int(v))
    yield
    if "ctx" in request.fixturenames:
        request.getfixturevalue("ctx").reset_knobs()
