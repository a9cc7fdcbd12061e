"""HIP path vs the committed fixtures (tests/golden/), through the C ABI, without the oracle in
the loop: what the GPU computes is compared with numbers recorded in the repository."""
import numpy as np
import pytest

import golden_util as G
from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_operators_match_fixture(hip, problems, name, force_generic):
    p, g = G.load_case(problems, name)
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, force_generic)
    s.load(p.values, p.b, p.D)
    assert rel(s.jtjx(g["x_probe"]), g["jtjx"]) <= 1e-12
    assert rel(s.jtb(), g["jtb"]) <= 1e-12
    assert rel(s.squared_column_norm(), g["squared_column_norm"]) <= 1e-12
    s.close()
    if "schur_sx" not in g:
        return
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, force_generic)
    s.load(p.values, p.b, p.D)
    s.schur_init()
    assert rel(s.schur_rhs(), g["schur_rhs"]) <= 1e-12
    assert rel(s.ete_inverse(), g["ete_inverse"]) <= 1e-12
    assert rel(s.schur_sx(g["xf_probe"]), g["schur_sx"]) <= 1e-12
    assert rel(s.back_substitute(g["xf_probe"]), g["back_substitute"]) <= 1e-12
    s.schur_jacobi_update()
    assert rel(s.preconditioner_blocks(), g["schur_jacobi_blocks"][0]) <= 1e-11
    s.close()


@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_solvers_match_fixture(hip, problems, name, force_generic):
    p, g = G.load_case(problems, name)
    K = G.K_FIXED
    cases = [(hip.CGNR, hip.JACOBI, "cgnr")]
    if "schur_fixed" in g:
        cases.append((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, "schur"))
    for st, pre, key in cases:
        s = make_solver(hip, p, st, pre, force_generic, min_it=K, max_it=K)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        assert summ.num_iterations == K
        assert rel(x, g[key + "_fixed"]) <= 1e-9, (key, rel(x, g[key + "_fixed"]))
        s = make_solver(hip, p, st, pre, force_generic, max_it=2000)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.0, r_tolerance=1e-13))
        s.close()
        assert summ.termination_type == hip.SUCCESS
        assert rel(x, g[key + "_converged"]) <= 1e-7, (key, rel(x, g[key + "_converged"]))  # CGNR: cond(J)^2


def test_known_answers_from_fixture_file(hip, problems):
    ka = G.known_answers()
    p = problems.linear_least_squares_problem(0)
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, max_it=100)
    x, _ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.0, r_tolerance=1e-15))
    np.testing.assert_allclose(x, ka["0"]["x_D"], atol=2e-8)
    x, _ = s.solve(p.values, p.b, hip.PerSolveOptions(D=None, q_tolerance=0.0, r_tolerance=1e-15))
    np.testing.assert_allclose(x, ka["0"]["x"], atol=1e-10)
    s.close()


def test_bal_evaluator_matches_fixture(hip):
    # the device Evaluator against recorded dual-number Jacobians, no oracle in the loop
    import os
    g = dict(np.load(os.path.join(G.GOLDEN, "bal_evaluator_small.npz")))
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, max_num_iterations=50)
    bp = hip.BalProblem(o, int(g["num_cameras"]), int(g["num_points"]), g["camera_index"], g["point_index"], g["observations"])
    cost, res, grad, vals = bp.evaluate(g["state"], residuals=True, gradient=True, jacobian=True)
    bp.close()
    assert abs(cost - float(g["cost"])) <= 1e-13 * float(g["cost"])
    assert rel(res, g["residuals"]) <= 1e-13
    assert rel(vals, g["jacobian_values"]) <= 1e-12
