"""Loading of the committed fixtures under tests/golden/ (see tests/golden/make_golden.py)."""
import importlib.util
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MAKER = _maker()
CASES = MAKER.CASES
K_FIXED = MAKER.K_FIXED


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)["problems"]


def load_case(problems, name):
    """Regenerates the inputs from the seed, checks them against the stored input checksum and
    returns (problem, fixture dict)."""
    layout, nc, npts, nobs, seed, skew = CASES[name]
    p = problems.synthetic_bal(None, layout=layout, num_cameras=nc, num_points=npts, num_observations=nobs, seed=seed,
                               skew=skew)
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    np.testing.assert_allclose(MAKER.input_checksum(p), g["input_checksum"], rtol=1e-12,
                               err_msg="the seeded generator no longer reproduces this fixture's inputs")
    return p, g
