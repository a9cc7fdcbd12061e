"""SCHUR_POWER_SERIES_EXPANSION — the inverse power-series operator, the
preconditioner built on it, and use_spse_initialization, against the oracle (which is pinned to
dense algebra in tests/test_oracle_dense.py::test_power_series_expansion_against_dense)."""
import numpy as np
import pytest

from test_gpu_operators import rel

pytestmark = pytest.mark.gpu


def problem(problems, kind):
    if kind == "bal":
        return problems.synthetic_bal(None, num_cameras=40, num_points=3000, num_observations=14000, seed=71, skew=0.5)
    if kind == "bal_long":
        return problems.synthetic_bal(None, num_cameras=150, num_points=200, num_observations=6000, seed=72)
    return problems.random_schur_problem(num_e_blocks=40, num_f_blocks=9, num_no_e_rows=0, seed=73)


@pytest.mark.parametrize("kind", ["bal", "bal_long", "general"])
def test_power_series_operator_and_apply(hip, oracle, problems, kind):
    p = problem(problems, kind)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    isc.compute_ftf_inverse()
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_POWER_SERIES_EXPANSION,
                                min_num_iterations=0, max_num_iterations=100, elimination_groups=[p.num_eliminate_blocks])
    s = hip.HipLinearSolver(o)
    s.set_structure(p.bs)
    s.load(p.values, p.b, p.D)
    s.schur_init()
    rng = np.random.default_rng(0)
    x, y0 = rng.standard_normal(m.num_cols_f), rng.standard_normal(m.num_cols_f)
    assert rel(s.power_series_operator(x, y0), isc.power_series_operator(x, y0)) <= 1e-12
    for iters, tol in ((1, 0.0), (5, 0.0), (8, 0.1), (50, 1e-3)):
        assert rel(s.spse_apply(x, iters, tol), isc.spse_apply(x, iters, tol)) <= 1e-11, (iters, tol)
    s.close()


@pytest.mark.parametrize("kind", ["bal", "general"])
@pytest.mark.parametrize("pre,init", [(3, False), (2, True), (3, True), (1, True)])
def test_spse_solver_variants(hip, oracle, problems, kind, pre, init):
    p = problem(problems, kind)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=300,
                                elimination_groups=[p.num_eliminate_blocks], use_spse_initialization=init,
                                max_num_spse_iterations=5, spse_tolerance=0.1)
    s = hip.HipLinearSolver(o)
    s.set_structure(p.bs)
    for q_tol, r_tol in ((0.1, -1.0), (-1.0, 1e-10)):
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
        xo, so = oracle.iterative_schur_solve_spse(m, p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=300, q_tol=q_tol,
                                                   r_tol=r_tol, use_spse_initialization=init, max_num_spse_iterations=5,
                                                   spse_tolerance=0.1)
        assert summ.termination_type == so.termination_type == hip.SUCCESS, (summ, so)
        assert abs(summ.num_iterations - so.num_iterations) <= 1, (summ, so)
        if summ.num_iterations == so.num_iterations:
            assert rel(x, xo) <= 1e-8
    s.close()
