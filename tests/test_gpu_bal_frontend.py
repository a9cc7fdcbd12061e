"""the device Evaluator for BAL problems and the trust-region loop around the
linear solvers, against the oracle's restatement (oracle/bal_harness.cc: dual-number Jacobians
of the Snavely residual, TrustRegionMinimizer + LevenbergMarquardtStrategy)."""
import numpy as np
import pytest

from test_gpu_operators import rel

pytestmark = pytest.mark.gpu


def make_pair(hip, oracle, nc, npts, nobs, seed, solver_type=5, pre=2, max_it=500, **gen):
    op = oracle.BalProblem.generate(nc, npts, nobs, seed=seed, **gen)
    bs, nelim = op.build_structure(True)
    cam, pt, obs = op.indices()
    o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=max_it)
    gp = hip.BalProblem(o, op.num_cameras, op.num_points, cam, pt, obs)
    return op, gp, bs, nelim


def test_sizes_and_row_order_match_the_reduced_program(hip, oracle):
    op, gp, bs, nelim = make_pair(hip, oracle, 6, 80, 400, seed=3)
    assert (gp.num_parameters, gp.num_residuals, gp.num_jacobian_values) == (bs.num_cols, bs.num_rows, bs.num_nonzeros)
    cam, pt, _ = op.indices()
    order = gp.row_order()
    assert sorted(order.tolist()) == list(range(op.num_observations))
    assert np.all(np.diff(pt[order]) >= 0)                      # grouped by point ...
    same = np.diff(pt[order]) == 0
    assert np.all(np.diff(order)[same] > 0)                     # ... stable in observation order
    # the structure the device built is the oracle's
    gp.close()


@pytest.mark.parametrize("seed,nc,npts,nobs", [(3, 6, 80, 400), (4, 20, 700, 4000), (5, 40, 3000, 14000)])
def test_evaluate_matches_oracle_dual_numbers(hip, oracle, seed, nc, npts, nobs):
    op, gp, bs, _ = make_pair(hip, oracle, nc, npts, nobs, seed)
    x = op.state()
    cost_o, res_o, vals_o = op.evaluate(x)
    cost, res, grad, vals = gp.evaluate(x, residuals=True, gradient=True, jacobian=True)
    assert abs(cost - cost_o) <= 1e-13 * cost_o
    assert rel(res, res_o) <= 1e-13
    # analytic Jacobian vs forward-mode duals of the same formula, entry by entry
    assert np.max(np.abs(vals - vals_o) / (np.abs(vals_o) + 1e-9 * np.abs(vals_o).max())) <= 1e-9
    assert rel(vals, vals_o) <= 1e-12
    m = oracle.Matrix(bs, 0)
    assert rel(grad, m.left_multiply(vals_o, res_o)) <= 1e-12
    # cost only
    c2, r2, g2, v2 = gp.evaluate(x)
    assert c2 == cost and r2 is None and g2 is None and v2 is None
    gp.close()


def test_small_angle_branch_of_the_rotation(hip, oracle):
    # angle-axis exactly zero: the first-order branch of AngleAxisRotatePoint (include/ceres/rotation.h:864-905)
    op, gp, bs, _ = make_pair(hip, oracle, 5, 60, 250, seed=9)
    x = op.state()
    x[3 * op.num_points:3 * op.num_points + 3] = 0.0          # camera 0
    x[3 * op.num_points + 18:3 * op.num_points + 21] = 0.0    # camera 2
    cost_o, res_o, vals_o = op.evaluate(x)
    cost, res, _, vals = gp.evaluate(x, residuals=True, jacobian=True)
    assert abs(cost - cost_o) <= 1e-13 * cost_o and rel(res, res_o) <= 1e-13 and rel(vals, vals_o) <= 1e-12
    gp.close()


def test_state_conversion_round_trip(hip, oracle):
    op, gp, bs, _ = make_pair(hip, oracle, 4, 30, 100, seed=1)
    x = op.state()
    np.testing.assert_array_equal(gp.state_from_bal(gp.state_to_bal(x)), x)
    gp.close()


def check_same_trajectory(Sa, Sb, cost_tol):
    assert Sa.num_iterations_logged == Sb.num_iterations_logged, (Sa.num_iterations_logged, Sb.num_iterations_logged)
    for i in range(Sa.num_iterations_logged):
        a, b = Sa.iterations[i], Sb.iterations[i]
        assert a.step_is_successful == b.step_is_successful and a.step_is_valid == b.step_is_valid, i
        assert abs(a.linear_solver_iterations - b.linear_solver_iterations) <= 1, i
        assert abs(a.cost - b.cost) <= cost_tol * abs(a.cost), (i, a.cost, b.cost)
        assert abs(a.radius - b.trust_region_radius) <= 1e-6 * a.radius, i
    assert abs(Sa.final_cost - Sb.final_cost) <= cost_tol * Sa.final_cost


@pytest.mark.parametrize("solver_type,pre", [(5, 2), (6, 1)])
def test_minimize_follows_the_oracle_trust_region_loop(hip, oracle, solver_type, pre):
    # same problem, same options: the oracle's loop with the oracle's linear solver vs the loop on the device.
    # Inexact Newton (eta = 0.1): the iterates agree as far as the CG termination decisions do; costs to 1e-6.
    op, gp, bs, nelim = make_pair(hip, oracle, 12, 800, 3600, seed=5, solver_type=solver_type, pre=pre)
    if solver_type == 6:
        op.build_structure(True)
    x0 = op.state()
    Sa = op.lm_solve(solver_type=solver_type, preconditioner=pre, max_it=500, max_num_iterations=12)
    x, Sb = gp.minimize(x0, max_num_iterations=12)
    assert Sb.initial_cost == pytest.approx(Sa.initial_cost, rel=1e-13)
    check_same_trajectory(Sa, Sb, 1e-6)
    assert Sb.final_cost < 0.5 * Sb.initial_cost
    assert Sb.termination_type == Sa.termination
    # the returned state is the one whose cost is reported
    assert gp.evaluate(x)[0] == pytest.approx(Sb.final_cost, rel=1e-12)
    assert rel(x, op.state()) <= 1e-5
    gp.close()


@pytest.mark.parametrize("solver_type,pre,shape", [(5, 2, (12, 800, 3600)), (6, 1, (12, 800, 3600)), (5, 2, (2600, 1500, 9000)), (5, 2, (100, 30, 2400))])
def test_evaluator_writing_the_tiles_is_the_two_pass_form(hip, oracle, monkeypatch, solver_type, pre, shape):
    """Round 4: inside ceres_hip_bal_minimize the evaluator writes the solver's tiles itself (bal_evaluate_tiles_kernel: tile order, no
    caller-layout E cells, no re-layout pass).  CERES_HIP_EVAL_TILES=0 is the earlier form — caller-layout values, gathered into the
    tiles by the gradient's pass: the same numbers in the same tiles, so the two loops must agree far below the inexact-Newton
    tolerance the oracle comparison has to allow.  Shapes: plain, more cameras than LDS rows (hybrid plan), points of more than
    64 observations (whole tiles, rounds)."""
    nc, npts, nobs = shape
    op, gp, bs, nelim = make_pair(hip, oracle, nc, npts, nobs, seed=5, solver_type=solver_type, pre=pre)
    assert gp.solver_info().kernel_path == hip.PATH_BAL
    x0 = op.state()
    runs = {}
    # "1" (the default): tiles from the evaluator; the camera-major preconditioner pass evaluates its F cells too where its items are long
    # (no caller-layout Jacobian exists then) and reads a caller-layout copy of them otherwise; "3" / "2": always the first / the second;
    # "0": the two-pass form
    blocks = {}
    for form in ("1", "2", "3", "0"):
        monkeypatch.setenv("CERES_HIP_EVAL_TILES", form)
        gp.minimize(x0, max_num_iterations=0)                        # evaluates at x0, solves nothing
        if solver_type == hip.ITERATIVE_SCHUR:
            blocks[form] = gp.preconditioner_blocks(not_inverted=True)   # camera-major pass over that Jacobian, as this form runs it
        elif form != "0":   # CGNR's uninverted blocks come from a generic kernel over the caller layout: refused, not served from stale memory
            with pytest.raises(hip.HipError, match="tiles only"):
                gp.preconditioner_blocks(not_inverted=True)
        runs[form] = gp.minimize(x0, max_num_iterations=6)
    xb, Sb = runs["0"]
    # the three forms evaluate the same formula in three kernels (other fused multiply-adds: the last bit of a cell may differ): the
    # preconditioner blocks agree to rounding, and what inexact solves (eta = 0.1) on top of them return agrees as far as the oracle
    # comparison's own tolerance — in practice far closer (1e-11 where the cameras are well determined)
    if blocks:
        print("blocks:", rel(blocks["1"], blocks["0"]), rel(blocks["2"], blocks["0"]), rel(blocks["3"], blocks["0"]))
        assert rel(blocks["2"], blocks["0"]) <= 1e-13 and rel(blocks["1"], blocks["0"]) <= 1e-13 and rel(blocks["3"], blocks["0"]) <= 1e-13
    loose = False
    for form, cost_tol, x_tol in ((f, 1e-5 if loose else 1e-9, 1e-3 if loose else 1e-7) for f in ("1", "2", "3")):
        xa, Sa = runs[form]
        assert Sa.num_iterations_logged == Sb.num_iterations_logged and Sa.num_iterations_logged >= 4
        for i in range(Sa.num_iterations_logged):
            a, b = Sa.iterations[i], Sb.iterations[i]
            assert (a.step_is_successful, a.step_is_valid) == (b.step_is_successful, b.step_is_valid), (form, i)
            assert abs(a.linear_solver_iterations - b.linear_solver_iterations) <= (1 if loose else 0), (form, i)
            assert abs(a.cost - b.cost) <= cost_tol * abs(a.cost) and abs(a.gradient_max_norm - b.gradient_max_norm) <= 100 * cost_tol * abs(a.gradient_max_norm), (form, i)
        assert rel(xa, xb) <= x_tol, form
        print("form", form, "cost diff", max(abs(Sa.iterations[i].cost - Sb.iterations[i].cost) / Sb.iterations[i].cost for i in range(Sa.num_iterations_logged)), "x", rel(xa, xb))
    # an evaluation through the API afterwards (caller-layout values, E cells included) still gives the whole Jacobian
    cost_o, res_o, vals_o = op.evaluate(xb)
    cost, res, grad, vals = gp.evaluate(xb, residuals=True, gradient=True, jacobian=True)
    assert rel(vals, vals_o) <= 1e-12 and rel(res, res_o) <= 1e-11   # (residuals near a minimum: differences of nearly equal pixels)
    gp.close()


def test_minimize_with_rejected_steps_and_without_jacobi_scaling(hip, oracle):
    # a huge initial radius makes the first steps overshoot: exercises the rejection path (reuse_diagonal,
    # radius /= decrease_factor) on both sides
    op, gp, bs, nelim = make_pair(hip, oracle, 10, 500, 2400, seed=7, param_noise=2.0)
    x0 = op.state()
    kw = dict(max_num_iterations=15, jacobi_scaling=0)
    Sa = op.lm_solve(solver_type=5, preconditioner=2, max_it=500, initial_radius=1e16, **kw)
    x, Sb = gp.minimize(x0, initial_trust_region_radius=1e16, **kw)
    assert Sa.num_unsuccessful_steps >= 3, "the scenario must contain rejected steps"
    check_same_trajectory(Sa, Sb, 1e-5)
    assert Sb.num_unsuccessful_steps == Sa.num_unsuccessful_steps
    gp.close()


def test_minimize_spse_preconditioner_with_rejected_steps(hip, oracle):
    # SCHUR_POWER_SERIES_EXPANSION needs blockdiag(F^T F + D_f^2)^-1, which contains D: a rejected step retries
    # with a smaller radius (= a new D) on the SAME Jacobian, and the cached inverse must be rebuilt
    # (ImplicitSchurComplement::Init recomputes it on every Solve).  With a stale inverse the CG iteration
    # counts of the retries drift away from the oracle's.
    op, gp, bs, nelim = make_pair(hip, oracle, 10, 500, 2400, seed=7, pre=hip.SCHUR_POWER_SERIES_EXPANSION, param_noise=2.0)
    x0 = op.state()
    m = oracle.Matrix(bs, nelim)

    def solve(values, b, D, q_tol, r_tol):
        x, summ = oracle.iterative_schur_solve_spse(m, values, b, D, preconditioner=3, min_it=0, max_it=500, q_tol=q_tol, r_tol=r_tol)
        return x, summ.termination_type, summ.num_iterations
    kw = dict(max_num_iterations=15, jacobi_scaling=0)
    Sa = op.lm_solve(solve_fn=solve, initial_radius=1e16, **kw)
    x, Sb = gp.minimize(x0, initial_trust_region_radius=1e16, **kw)
    assert Sa.num_unsuccessful_steps >= 3, "the scenario must contain rejected steps"
    check_same_trajectory(Sa, Sb, 1e-5)
    assert Sb.num_unsuccessful_steps == Sa.num_unsuccessful_steps
    gp.close()


def test_convergence_tests_terminate_like_the_oracle(hip, oracle):
    op, gp, bs, nelim = make_pair(hip, oracle, 8, 300, 1500, seed=21, pixel_noise=0.0, param_noise=0.01)
    x0 = op.state()
    Sa = op.lm_solve(solver_type=5, preconditioner=2, max_it=500, max_num_iterations=50)
    x, Sb = gp.minimize(x0, max_num_iterations=50)
    assert Sb.termination_type == hip.CONVERGENCE == Sa.termination
    assert Sb.message == Sa.message
    assert abs(Sb.num_iterations_logged - Sa.num_iterations_logged) <= 1
    gp.close()


def test_create_rejects_bad_indices(hip):
    o = hip.LinearSolverOptions(type=5, preconditioner_type=2, max_num_iterations=10)
    with pytest.raises(hip.HipError):
        hip.BalProblem(o, 2, 3, [0, 1, 2], [0, 1, 2], np.zeros(6))


def test_problem_from_bal_file(hip, oracle, tmp_path):
    op = oracle.BalProblem.generate(6, 90, 420, seed=13)
    op.build_structure(True)
    f = str(tmp_path / "problem.txt")
    assert op.write(f) == 0
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, max_num_iterations=100)
    gp, x0 = hip.BalProblem.from_file(o, f)
    np.testing.assert_allclose(x0, op.state(), rtol=1e-15)
    assert gp.evaluate(x0)[0] == pytest.approx(op.evaluate(op.state())[0], rel=1e-13)
    gp.close()


def test_cpp_host_mirror_of_the_bal_front_end(hip, oracle, tmp_path):
    # ceres-solver_amd/host/hip_bal_problem.h through host_driver: BAL file -> Evaluate -> Minimize, the same
    # numbers as the Python mirror produces over the same C ABI (also checks the struct layouts both bind)
    import os
    import re
    import subprocess
    from conftest import ROOT
    op = oracle.BalProblem.generate(8, 250, 1200, seed=17)
    op.build_structure(True)
    f = str(tmp_path / "problem.txt")
    assert op.write(f) == 0
    exe = os.path.join(ROOT, "ceres-solver_amd", "host", "host_driver")
    r = subprocess.run([exe, f, "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("bal ")][0]
    kv = dict(re.findall(r"(\w+)=([^ ]+)", line))
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, min_num_iterations=0, max_num_iterations=500)
    gp, x0 = hip.BalProblem.from_file(o, f)
    x, S = gp.minimize(x0, max_num_iterations=6)
    gp.close()
    assert int(kv["parameters"]) == gp.num_parameters and int(kv["residuals"]) == gp.num_residuals
    assert float(kv["initial_cost"]) == pytest.approx(S.initial_cost, rel=1e-13) == pytest.approx(float(kv["evaluated_initial"]), rel=1e-13)
    assert float(kv["final_cost"]) == pytest.approx(S.final_cost, rel=1e-9) and float(kv["evaluated_final"]) == pytest.approx(float(kv["final_cost"]), rel=1e-12)
    assert int(kv["successful"]) == S.num_successful_steps and int(kv["termination"]) == S.termination_type
    assert S.final_cost < 0.5 * S.initial_cost
