"""ITERATIVE_SCHUR with use_explicit_schur_complement — S formed by elimination
(dense storage), SCHUR_JACOBI from its diagonal blocks, CG on S, back-substitution on SUCCESS
(internal/ceres/schur_complement_solver.cc:100-158, 337-408).  Checked against the oracle's
SchurEliminator + its CG on the dense reduced system, and against the implicit solver."""
import numpy as np
import pytest

from test_gpu_operators import rel

pytestmark = pytest.mark.gpu


def explicit_options(hip, nelim, **kw):
    return hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[nelim],
                                   use_explicit_schur_complement=True, min_num_iterations=kw.pop("min_it", 0),
                                   max_num_iterations=kw.pop("max_it", 200), **kw)


def oracle_explicit(oracle, p, min_it, max_it, q_tol, r_tol):
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    lhs, rhs = m.schur_eliminate(p.values, p.b, p.D)
    nf = rhs.shape[0]
    S = np.triu(lhs.reshape(nf, nf))
    S = S + np.triu(S, 1).T
    sizes = p.bs.col_block_size[p.num_eliminate_blocks:]
    Minv = np.zeros_like(S)
    o = 0
    for n in sizes:
        Minv[o:o + n, o:o + n] = np.linalg.inv(S[o:o + n, o:o + n])
        o += n
    z, summ = oracle.cg_dense(S, rhs, Minv=Minv, min_it=min_it, max_it=max_it, q_tol=q_tol, r_tol=r_tol)
    x = m.schur_back_substitute(p.values, p.b, p.D, z)
    return x, summ.num_iterations, summ.termination_type


def solve(hip, p, o, q_tol, r_tol, **kw):
    s = hip.HipLinearSolver(o, **kw)
    s.set_structure(p.bs)
    if o.use_explicit_schur_complement:
        assert s.info().kernel_path == hip.PATH_GENERIC
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
    s.close()
    return x, summ


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_known_answer_problems(hip, oracle, problems, pid):
    p = problems.linear_least_squares_problem(pid)
    A = p.bs.to_dense(p.values)
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=p.num_cols), -1.0, 1e-12)
    assert s.termination_type == hip.SUCCESS
    assert np.linalg.norm(x - ref) < 1e-12 * max(1.0, np.linalg.norm(ref))


@pytest.mark.parametrize("kind", ["bal", "general"])
def test_fixed_iterations_match_oracle_explicit_solver(hip, oracle, problems, kind):
    if kind == "bal":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=14, num_points=500, num_observations=2400, seed=31)
    else:
        p = problems.random_schur_problem(num_e_blocks=60, num_f_blocks=9, seed=4)
    K = 5
    xo, its, _ = oracle_explicit(oracle, p, K, K, -1.0, -1.0)
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, min_it=K, max_it=K), -1.0, -1.0)
    assert s.num_iterations == K == its
    # NO_CONVERGENCE after max iterations: no back-substitution (schur_complement_solver.cc:150-154), x = [0; reduced solution]
    assert s.termination_type == hip.NO_CONVERGENCE
    ne = p.bs.col_block_pos[p.num_eliminate_blocks]
    assert np.all(x[:ne] == 0.0)
    assert rel(x[ne:], xo[ne:]) <= 1e-9


def test_converged_explicit_equals_implicit_and_oracle(hip, oracle, problems):
    p = problems.synthetic_bal(None, layout="schur", num_cameras=14, num_points=500, num_observations=2400, seed=32)
    xo, its, term = oracle_explicit(oracle, p, 0, 500, -1.0, 1e-13)
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=500), -1.0, 1e-13)
    assert s.termination_type == hip.SUCCESS and abs(s.num_iterations - its) <= 1
    assert rel(x, xo) <= 1e-9
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[p.num_eliminate_blocks],
                                min_num_iterations=0, max_num_iterations=500)
    xi, si = solve(hip, p, o, -1.0, 1e-13)
    assert rel(x, xi) <= 1e-9 and abs(si.num_iterations - s.num_iterations) <= 1


def test_sharded_elimination_in_loopback(hip, oracle, problems):
    # world > 1 branches (D_f^2 on one rank only, all-reduce of S) through the 1-rank communicator
    p = problems.synthetic_bal(None, layout="schur", num_cameras=10, num_points=300, num_observations=1400, seed=33)
    x1, s1 = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=300), -1.0, 1e-13)
    x2, s2 = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=300), -1.0, 1e-13, loopback_world=2)
    assert s1.termination_type == s2.termination_type == hip.SUCCESS
    assert rel(x2, x1) <= 1e-12


def test_option_validation(hip, problems):
    with pytest.raises(hip.HipError):   # "Only SCHUR_JACOBI is supported"
        hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.JACOBI, max_num_iterations=5,
                                                    use_explicit_schur_complement=True))
    with pytest.raises(hip.HipError):
        hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.CGNR, preconditioner_type=hip.JACOBI, max_num_iterations=5,
                                                    use_explicit_schur_complement=True))
    p = problems.synthetic_bal(None, layout="schur", num_cameras=1000, num_points=3000, num_observations=12000, seed=1, with_values=False)
    s = hip.HipLinearSolver(explicit_options(hip, p.num_eliminate_blocks))
    with pytest.raises(hip.HipError):   # 9000 reduced columns > 8192
        s.set_structure(p.bs)
    s.close()
