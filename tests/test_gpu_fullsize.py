"""BASELINE.json's sizes on the GPU through size-independent properties (the oracle is too slow
to be the checker at these sizes): linearity and symmetry of S and of J^T J + D^2, agreement of the
fused <2,3,9> kernels with the generic multi-pass kernels, and the normal-equation residual of
the returned step."""
import numpy as np
import pytest

from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ladybug(problems):
    return problems.synthetic_bal("ladybug1723", layout="schur", seed=38401, skew=0.6)


def test_ladybug_shape_fused_vs_generic(hip, ladybug):
    p = ladybug
    rng = np.random.default_rng(0)
    fused = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    generic = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, True)
    assert fused.info().kernel_path == hip.PATH_BAL and fused.info().camera_accum_in_lds == 1
    out = {}
    for name, s in (("fused", fused), ("generic", generic)):
        s.load(p.values, p.b, p.D)
        s.schur_init()
        x = rng.standard_normal(s.info().num_cols_f) if name == "fused" else x
        s.schur_jacobi_update()
        out[name] = dict(rhs=s.schur_rhs(), sx=s.schur_sx(x), bs=s.back_substitute(x), pre=s.preconditioner_blocks(),
                         ete=s.ete_inverse())
    for k in out["fused"]:
        assert rel(out["fused"][k], out["generic"][k]) <= 1e-11, (k, rel(out["fused"][k], out["generic"][k]))
    # linearity / symmetry / positive definiteness of S
    n = fused.info().num_cols_f
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Sx, Sy = fused.schur_sx(x), fused.schur_sx(y)
    assert rel(fused.schur_sx(2.0 * x - 3.0 * y), 2.0 * Sx - 3.0 * Sy) <= 1e-12
    assert abs(y @ Sx - x @ Sy) <= 1e-11 * abs(y @ Sx)
    assert x @ Sx > 0
    fused.close()
    generic.close()


def test_ladybug_shape_step_satisfies_normal_equations(hip, ladybug):
    p = ladybug
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.0, r_tolerance=1e-10))
    assert summ.termination_type == hip.SUCCESS, summ
    s.close()
    # (J^T J + D^2) x = J^T b, checked with the CGNR operators of a second instance
    q = type(p)(p.bs, p.values, p.b, p.D, p.num_eliminate_blocks)
    c = make_solver(hip, q, hip.CGNR, hip.JACOBI)
    c.load(p.values, p.b, p.D)
    g = c.jtb()
    assert rel(c.jtjx(x), g) <= 1e-7
    c.close()


def test_venice_shape_cgnr_properties_and_solve(hip, problems):
    p = problems.synthetic_bal("venice1778", layout="cgnr", seed=38401, skew=0.6)
    p.num_eliminate_blocks = 0
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, max_it=30, min_it=0)
    info = s.info()
    assert info.kernel_path == hip.PATH_BAL and info.num_observations == 5001946 and info.camera_accum_in_lds == 1
    s.load(p.values, p.b, p.D)
    rng = np.random.default_rng(1)
    n = info.num_cols
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = s.jtjx(x), s.jtjx(y)
    assert rel(s.jtjx(0.5 * x + 4.0 * y), 0.5 * Ax + 4.0 * Ay) <= 1e-12
    assert abs(y @ Ax - x @ Ay) <= 1e-11 * abs(y @ Ax)
    # x^T (J^T J + D^2) x = |J x|^2 + |D x|^2 with J x from the plain (generic) SpMV
    Jx = s.right_multiply(x)
    assert abs(x @ Ax - (Jx @ Jx + (p.D * x) @ (p.D * x))) <= 1e-11 * (x @ Ax)
    # LM-style solve: zeta termination, finite step, decreases the quadratic model
    step, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert summ.termination_type == hip.SUCCESS and np.isfinite(step).all(), summ
    g = s.jtb()
    model = 0.5 * step @ s.jtjx(step) - g @ step
    assert model < 0
    s.close()
