"""Regenerates the fixtures in this directory.  Run from the repository root:

    python tests/golden/make_golden.py

Two kinds of fixture:

* known_answers.json — the reference's OWN known-answer vectors for this path, transcribed from
  its test-problem factory (internal/ceres/linear_least_squares_problems.cc:85-160 problem 0,
  :253-303 problem 2, :387-434 problem 3, and the same matrices in problems 4-6), i.e. the
  numbers the reference's unit tests (iterative_schur_complement_solver_test.cc:75-117,
  schur_eliminator_test.cc, implicit_schur_complement_test.cc) check its solvers against.
  These pin the oracle ("parity pinned", oracle/ceres_oracle.h header).
* bal_evaluator_small.npz — a small BAL problem (indices, observations, state) with the oracle's
  Snavely residuals and dual-number Jacobian at that state.
* bal_*.npz — outputs of the pinned oracle on small seeded <2,3,9> problems (inputs are
  regenerated from the seed by ceres-solver_amd/problems.py; the file also stores a checksum of
  the inputs so that a generator change cannot silently re-define the fixture).  The -m gpu
  tests compare the HIP path against these WITHOUT calling the oracle, the CPU tests check that
  the oracle still reproduces them.

The reference cannot be built here (Eigen3 + abseil are absent), so there is no
fixture produced by reference binaries; the known answers above are the reference-authored
anchors.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))   # tests/: reference_fixtures.py (the reference-held known-answer problems)

import __graft_entry__ as entry  # noqa: E402

CASES = {
    # name: (layout, cameras, points, observations, seed, skew)
    "bal_schur_small": ("schur", 12, 300, 1500, 101, 0.0),
    "bal_cgnr_small": ("cgnr", 12, 300, 1500, 102, 0.0),
    "bal_schur_skewed": ("schur", 30, 900, 6000, 103, 0.7),
}
K_FIXED = 6  # fixed CG iteration count for the solver vectors (min = max = K, q_tol = -1)


def input_checksum(p):
    h = np.zeros(3)
    for i, a in enumerate((p.values, p.b, p.D)):
        w = np.cos(np.arange(a.size) * 0.61803398875)
        h[i] = float(a @ w)
    return h


def make_case(pkg, oracle, name):
    layout, nc, npts, nobs, seed, skew = CASES[name]
    p = pkg.problems.synthetic_bal(None, layout=layout, num_cameras=nc, num_points=npts, num_observations=nobs,
                                   seed=seed, skew=skew)
    out = {"input_checksum": input_checksum(p)}
    rng = np.random.default_rng(seed + 7)
    x = rng.standard_normal(p.num_cols)
    m0 = oracle.Matrix(p.bs, 0)
    # (JtJ + D^2) x and J^T b: cgnr_solver.cc:59-83
    jx = m0.right_multiply(p.values, x)
    out["x_probe"] = x
    out["jtjx"] = m0.left_multiply(p.values, jx) + p.D * p.D * x
    out["jtb"] = m0.left_multiply(p.values, p.b)
    out["squared_column_norm"] = m0.squared_column_norm(p.values)
    out["cgnr_fixed"] = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=K_FIXED, max_it=K_FIXED, q_tol=-1.0,
                                      r_tol=-1.0)[0]
    out["cgnr_converged"] = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, max_it=2000, q_tol=0.0, r_tol=1e-13)[0]
    if layout == "schur":
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
        isc = oracle.ImplicitSchurComplement(m)
        isc.init(p.values, p.D, p.b)
        xf = rng.standard_normal(m.num_cols_f)
        out["xf_probe"] = xf
        out["schur_rhs"] = isc.rhs()
        out["schur_sx"] = isc.sx(xf)
        out["ete_inverse"] = isc.ete_inverse()
        out["back_substitute"] = isc.back_substitute(xf)
        out["schur_jacobi_blocks"] = m.schur_jacobi(p.values, p.D)
        out["schur_fixed"] = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=K_FIXED, max_it=K_FIXED,
                                                     q_tol=-1.0, r_tol=-1.0)[0]
        out["schur_converged"] = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, max_it=2000, q_tol=0.0,
                                                         r_tol=1e-13)[0]
    return p, out


def make_evaluator_case(oracle):
    """Snavely residuals and dual-number Jacobian (oracle/bal_harness.cc) of a small BAL problem at its
    perturbed start, two cameras with zero rotation (first-order branch of AngleAxisRotatePoint)."""
    op = oracle.BalProblem.generate(7, 120, 600, seed=77, skew=0.3)
    bs, nelim = op.build_structure(True)
    cam, pt, obs = op.indices()
    x = op.state()
    x[3 * op.num_points:3 * op.num_points + 3] = 0.0
    x[3 * op.num_points + 27:3 * op.num_points + 30] = 0.0
    cost, res, vals = op.evaluate(x)
    return {"num_cameras": np.int64(op.num_cameras), "num_points": np.int64(op.num_points), "camera_index": cam,
            "point_index": pt, "observations": obs, "state": x, "cost": np.float64(cost), "residuals": res,
            "jacobian_values": vals}


def main():
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    ev = make_evaluator_case(oracle)
    np.savez_compressed(os.path.join(HERE, "bal_evaluator_small.npz"), **ev)
    print("bal_evaluator_small", {k: np.asarray(v).shape for k, v in ev.items()})
    ka = {}
    for pid in (0, 2, 3, 4, 5, 6):
        import reference_fixtures
        p = reference_fixtures.linear_least_squares_problem(pid)
        ka[str(pid)] = {k: (None if v is None else np.asarray(v).tolist()) for k, v in p.known.items()}
        ka[str(pid)]["num_cols"] = int(p.num_cols)
        ka[str(pid)]["num_eliminate_blocks"] = int(p.num_eliminate_blocks)
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump({"source": "internal/ceres/linear_least_squares_problems.cc (reference-authored known answers)",
                   "problems": ka}, f, indent=1, sort_keys=True)
    for name in CASES:
        _, out = make_case(pkg, oracle, name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np.asarray(v) for k, v in out.items()})
        print(name, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
