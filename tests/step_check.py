"""Rung (4) of the parity ladder without an escape hatch: the solve LevenbergMarquardtStrategy issues
(q_tolerance = eta, r_tolerance = -1) compared UNCONDITIONALLY with the oracle's.

CG stops on zeta = i (Q1 - Q0) / Q1 < eta (internal/ceres/conjugate_gradients_solver.h:273-284); a re-associated sum can move
that test across the threshold, so the two solves may stop one iteration apart — both steps are then valid inexact-Newton
steps, but they are different vectors.  What must hold whatever the counts are:

  * both terminate with SUCCESS through the zeta test, and the zeta each one reports is below eta;
  * the counts differ by at most one;
  * the product's step equals the ORACLE'S CG ITERATE OF THE SAME ITERATION NUMBER to 1e-9 (the oracle re-run with
    min = max = k iterations): the product computes the same sequence, it only leaves it at a neighbouring index;
  * hence |x_hip - x_oracle| <= the norm of one CG update, |x_{k+1} - x_k| of the oracle's own sequence.
"""
import re

import numpy as np

STEP_TOL = 1e-9


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def reported_zeta(message):
    num = r"([-+]?(?:[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?|nan|inf))"
    m = re.search(rf"zeta = {num} < {num}", message)
    assert m, f"not a zeta termination: {message!r}"
    return float(m.group(1)), float(m.group(2))


def assert_lm_style_step(x, summ, oracle_solve, eta, success, tol=STEP_TOL):
    """x, summ: the product's step and summary.  oracle_solve(min_it, max_it, q_tol, r_tol) -> (x, summary) runs the oracle's
    solver on the same inputs.  Returns the oracle's eta-terminated (x, summary)."""
    xo, so = oracle_solve(0, 500, eta, -1.0)
    assert summ.termination_type == so.termination_type == success, (summ, so)
    for msg in (summ.message, so.message):
        z, q = reported_zeta(msg)
        assert z < q and abs(q - eta) <= 1e-6 * abs(eta), msg   # %e prints seven digits
    k_hip, k_or = summ.num_iterations, so.num_iterations
    assert abs(k_hip - k_or) <= 1, (summ, so)
    assert np.isfinite(x).all()
    if k_hip == k_or:
        assert rel(x, xo) <= tol, rel(x, xo)
        return xo, so
    # one iteration apart: same CG sequence, left at a neighbouring index
    xk, sk = oracle_solve(k_hip, k_hip, -1.0, -1.0)
    assert sk.num_iterations == k_hip
    assert rel(x, xk) <= tol, (rel(x, xk), k_hip, k_or)
    one_update = np.linalg.norm(xk - xo)   # |x_{k+1} - x_k| of the oracle's sequence
    assert np.linalg.norm(x - xo) <= one_update * (1.0 + 1e-6) + tol * np.linalg.norm(xo)
    return xo, so
