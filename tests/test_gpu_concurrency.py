"""Separate solver instances used from separate host threads at the same time (SURVEY §8b "Threading": a LinearSolver instance is
not thread-safe, but several Problems may solve concurrently with their own instances — internal/ceres/implicit_schur_complement.h:88-91,
solver.cc's per-Solve preprocessing — so the library may keep no unguarded process-global device state).  Every instance owns its
stream, buffers and status words; what is shared per process (the raised LDS ceilings, DENSE_SCHUR's look-ahead stream) sits behind
locks.  ctypes releases the GIL for the duration of a call, so the threads below really overlap inside libceres_hip.so."""
import threading

import numpy as np
import pytest

from step_check import assert_lm_style_step
from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


def _cases(hip, problems):
    """(name, problem, solver type, preconditioner, expected kernel path, make_solver keywords) — different shapes, both kernel paths, both solvers"""
    bal = problems.synthetic_bal(None, layout="schur", num_cameras=60, num_points=9000, num_observations=40000, seed=3, skew=0.5)
    quat = problems.synthetic_structured(45, 5000, 21000, seed=4, skew=0.4, camera_width=10)
    libmv = problems.synthetic_structured(40, 4000, 17000, seed=5, skew=0.5, camera_width=6, shared_widths=(8,), locked_cameras=(0,))
    homog = problems.synthetic_structured(40, 4000, 17000, seed=6, skew=0.5, camera_width=9, point_width=4)
    small = problems.synthetic_bal(None, layout="schur", num_cameras=16, num_points=3000, num_observations=11000, seed=8, skew=0.3)
    return [("bal_schur", bal, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, hip.PATH_BAL, {}),
            ("bal_cgnr", bal, hip.CGNR, hip.JACOBI, hip.PATH_BAL, {}),
            ("quaternion_cameras", quat, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, hip.PATH_BAL, {}),
            ("libmv_strip", libmv, hip.ITERATIVE_SCHUR, hip.JACOBI, hip.PATH_BAL, {}),
            ("homogeneous_points", homog, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, hip.PATH_BAL, {}),
            ("generic_kernels", small, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, hip.PATH_GENERIC, {"force_generic": True}),
            ("dense_schur", small, hip.DENSE_SCHUR, hip.IDENTITY, None, {"max_it": 1})]


def test_instances_on_separate_threads_do_not_disturb_each_other(hip, oracle, problems):
    cases = _cases(hip, problems)
    # the expected answers first, one after the other (the oracle's OpenMP team is not meant to be entered from several threads)
    want = []
    for name, p, typ, pre, path, kw in cases:
        if typ == hip.DENSE_SCHUR:
            # DENSE_SCHUR's exact solve = the converged iterative solve (rung 3 of the ladder, 1e-8)
            m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
            xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=0, max_it=500, q_tol=-1.0, r_tol=1e-14)
            assert so.termination_type == hip.SUCCESS
            want.append(("converged", xo))
        else:
            m = oracle.Matrix(p.bs, p.num_eliminate_blocks if typ == hip.ITERATIVE_SCHUR else 0)
            fn = m.iterative_schur_solve if typ == hip.ITERATIVE_SCHUR else m.cgnr_solve
            want.append(("lm_style", fn))
    results = [None] * len(cases)
    errors = []
    start = threading.Barrier(len(cases))
    ROUNDS = 12

    def work(k):
        name, p, typ, pre, path, kw = cases[k]
        try:
            start.wait(timeout=120)
            out = []
            for r in range(ROUNDS):
                # a fresh instance every few rounds: creation, structure analysis and destruction also overlap with the others' solves
                if r % 4 == 0:
                    if r:
                        s.close()
                    s = make_solver(hip, p, typ, pre, **dict({"max_it": 500}, **kw))
                    if path is not None:
                        assert s.info().kernel_path == path, (name, s.info().kernel_path)
                x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
                out.append((x.copy(), summ))
            s.close()
            results[k] = out
        except BaseException as ex:  # noqa: BLE001 - reported by the main thread
            errors.append((name, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,), name=cases[k][0]) for k in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a solver thread hangs"
    assert not errors, errors
    for k, (name, p, typ, pre, path, kw) in enumerate(cases):
        kind, ref = want[k]
        first = results[k][0][0]
        for x, summ in results[k]:
            assert summ.termination_type == hip.SUCCESS, (name, summ)
            # the same instance kind on the same inputs: the same step up to the order of the atomic camera sums (a zeta that sits
            # on the threshold may end a solve one iteration apart: assert_lm_style_step below is the unconditional check)
            if summ.num_iterations == results[k][0][1].num_iterations:
                assert rel(x, first) <= 1e-9, (name, rel(x, first))
            if kind == "converged":
                assert rel(x, ref) <= 1e-8, (name, rel(x, ref))
            else:
                assert_lm_style_step(x, summ, lambda lo, hi, q, r: ref(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r),
                                     0.1, hip.SUCCESS)


def test_device_lm_steps_on_separate_threads(hip, oracle, problems):
    """ceres_hip_lm_compute_step (the whole LevenbergMarquardtStrategy::ComputeStep neighbourhood) from four threads at once, two of
    them on instances of the SAME structure: per-instance status words, scalar slots and pinned step buffers must not be shared."""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=50, num_points=8000, num_observations=36000, seed=9, skew=0.5)
    q = problems.synthetic_structured(45, 5000, 21000, seed=10, skew=0.4, camera_width=6)
    plan = [(p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (p, hip.CGNR, hip.JACOBI), (q, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)]
    radius = 1e4
    sols, errors = [None] * len(plan), []
    start = threading.Barrier(len(plan))

    def work(k):
        prob, typ, pre = plan[k]
        try:
            s = make_solver(hip, prob, typ, pre, max_it=500)
            start.wait(timeout=120)
            out = []
            for r in range(10):
                step, summ, mcc = s.lm_compute_step(prob.values, prob.b, radius, 0.1)
                out.append((step.copy(), summ, mcc))
                step, summ, mcc = s.lm_compute_step(None, None, radius / 2, 0.1, reuse_diagonal=True, values_unchanged=True)
                out.append((step.copy(), summ, mcc))
            s.close()
            sols[k] = out
        except BaseException as ex:  # noqa: BLE001
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(plan))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a solver thread hangs"
    assert not errors, errors
    from test_gpu_lm_step import check_step
    for k, (prob, typ, pre) in enumerate(plan):
        diag = np.clip(oracle.Matrix(prob.bs, 0).squared_column_norm(prob.values), 1e-6, 1e32)
        for idx, (step, summ, mcc) in enumerate(sols[k]):
            rad = radius if idx % 2 == 0 else radius / 2
            if idx < 2 or summ.num_iterations != sols[k][idx % 2][1].num_iterations:   # the oracle once per (instance, radius) ...
                check_step(oracle, hip, prob, typ, pre, np.sqrt(diag / rad), step, summ, mcc, 0.1)
            else:   # ... the repeats against that step
                assert rel(step, sols[k][idx % 2][0]) <= 1e-9, (k, idx)
