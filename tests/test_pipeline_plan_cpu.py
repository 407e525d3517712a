"""CPU: plan-time decisions of the device pipeline (tinysql_amd/gpu_pipeline.py) that need no GPU."""
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd import gpu_pipeline as G


def test_projection_of_a_string_valued_expression_is_refused_when_the_plan_is_built():
    # compile_expr accepts ETString roots since round 3 (VecEvalString), so the projection has to say Unsupported itself —
    # at construction, where the planner can still keep the Go ProjectionExec — instead of failing in Next
    s0, s1 = E.Column(0, abi.BYTES), E.Column(1, abi.BYTES)
    for e in (E.ScalarFunction("ifnull", s0, s1), E.ScalarFunction("if", E.ScalarFunction("isnull", s0), s1, E.Constant("x")), s0):
        assert e.eval_type == E.ETString
        with pytest.raises(E.Unsupported):
            G.GpuProjectionExec(None, None, [E.Column(2, abi.I64), e])
    # the same tree as a numeric root is fine to plan
    assert E.compile_expr(E.ScalarFunction("length", E.ScalarFunction("ifnull", s0, s1))).result_type == abi.I64
