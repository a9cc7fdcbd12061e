"""The tile plan (csrc/plan.cc, host only) on the RANDOM structures of the parity campaigns (tests/fuzz_cases.py): every row block in
exactly one slot, whole points per segment, the cooperative point-space ownership words — test_plan_cpu.py's invariants, on camera
counts around the wavefront / LDS limits and track lengths around the tile size instead of hand-picked shapes; and the generator itself
(deterministic in its seed, structures the oracle accepts)."""
import numpy as np
import pytest

import fuzz_cases
from conftest import pkg
from test_plan_cpu import check_plan_invariants, plan_of


def small_cases(first, count, max_obs=30000):
    for seed in range(first, first + count):
        case, k, _ = fuzz_cases.draw_case(seed)
        if case["n_obs"] <= max_obs:
            yield case, k


@pytest.mark.parametrize("first", [0, 40, 80])
def test_plan_invariants_on_random_structures(first):
    checked = 0
    for case, k in small_cases(first, 40):
        nr, ne, nf = case["shape"]
        if (nr, ne) != (2, 3) or case["prior_rows"] or case["locked"]:   # (the checker knows 3-wide points in 2-high rows, every row in a
            continue                                                     #  tile, and a camera cell in every row)
        p = fuzz_cases.build(pkg.problems, case, k)
        plan = plan_of(p)
        assert plan["eligible"], (case, plan.get("why"))
        check_plan_invariants(p, plan)
        checked += 1
    assert checked >= 8


def test_generator_is_deterministic_and_well_formed(oracle):
    for case, k in small_cases(200, 25, max_obs=5000):
        a, b = fuzz_cases.build(pkg.problems, case, k), fuzz_cases.build(pkg.problems, case, k)
        assert np.array_equal(a.values, b.values) and np.array_equal(a.bs.cell_value_pos, b.bs.cell_value_pos) and np.array_equal(a.D, b.D)
        assert a.bs.num_row_blocks == case["n_obs"] + case["prior_rows"] and (a.D >= 0.5).all()
        m = oracle.Matrix(a.bs, a.num_eliminate_blocks)   # the oracle's structure checks accept it
        x, summ = m.iterative_schur_solve(a.values, a.b, a.D, preconditioner=2, min_it=1, max_it=1, q_tol=-1.0, r_tol=-1.0)
        assert np.isfinite(x[a.bs.col_block_pos[a.num_eliminate_blocks]:]).all()
