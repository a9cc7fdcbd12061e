"""The oracle against the committed fixtures: the reference-authored known answers
(tests/golden/known_answers.json) and the oracle's own recorded outputs on seeded <2,3,9>
problems (tests/golden/bal_*.npz) — the latter guard the checker against drift, the -m gpu
tests in test_gpu_golden.py compare the HIP path with the same files."""
import numpy as np
import pytest

import golden_util as G


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def test_known_answer_file_matches_the_problem_factory(problems):
    ka = G.known_answers()
    assert sorted(ka) == ["0", "2", "3", "4", "5", "6"]
    for pid, rec in ka.items():
        p = problems.linear_least_squares_problem(int(pid))
        assert rec["num_cols"] == p.num_cols and rec["num_eliminate_blocks"] == p.num_eliminate_blocks
        for k, v in p.known.items():
            if v is None:
                assert rec[k] is None
            else:
                np.testing.assert_array_equal(np.asarray(rec[k]), np.asarray(v))


@pytest.mark.parametrize("pid", ["0", "2", "5"])
def test_oracle_solvers_reach_the_reference_known_answers(oracle, problems, pid):
    # linear_least_squares_problems.cc:85-160 (problem 0: x = [2, 3], x_D), :253-303 (problem 2)
    rec = G.known_answers()[pid]
    p = problems.linear_least_squares_problem(int(pid))
    m = oracle.Matrix(p.bs, 0)
    if rec.get("x_D") is not None:
        x, s = m.cgnr_solve(p.values, p.b, p.D, preconditioner=1, max_it=200, r_tol=1e-15)
        np.testing.assert_allclose(x, rec["x_D"], atol=2e-8)  # printed with 8 digits in the reference
    if rec.get("x") is not None:
        x, s = m.cgnr_solve(p.values, p.b, None, preconditioner=1, max_it=200, r_tol=1e-15)
        np.testing.assert_allclose(x, rec["x"], atol=1.1e-4 if pid != "0" else 1e-10)


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_reproduces_recorded_outputs(oracle, problems, name):
    p, g = G.load_case(problems, name)
    _, fresh = G.MAKER.make_case(type("Pkg", (), {"problems": problems}), oracle, name)
    for k, v in g.items():
        if k == "input_checksum":
            continue
        tol = 1e-9 if k.endswith("_converged") else 1e-12
        assert rel(fresh[k], v) <= tol, (name, k, rel(fresh[k], v))


def test_oracle_reproduces_recorded_evaluator_case(oracle):
    import os
    g = dict(np.load(os.path.join(G.GOLDEN, "bal_evaluator_small.npz")))
    fresh = G.MAKER.make_evaluator_case(oracle)
    np.testing.assert_array_equal(fresh["camera_index"], g["camera_index"])
    np.testing.assert_array_equal(fresh["observations"], g["observations"])
    assert abs(float(fresh["cost"]) - float(g["cost"])) <= 1e-14 * float(g["cost"])
    assert rel(fresh["residuals"], g["residuals"]) <= 1e-14 and rel(fresh["jacobian_values"], g["jacobian_values"]) <= 1e-14
