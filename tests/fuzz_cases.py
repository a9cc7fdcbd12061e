"""The random structures of the parity campaigns (tools/fuzz_parity.py, tools/fuzz_multirank.py, tests/test_gpu_fuzz.py): pure numpy on
the package's generators (P = ceres_solver_amd.problems), no device, no oracle — also imported by the ranks of a sharded campaign."""
import numpy as np

SHAPES = [(2, 3, w) for w in range(2, 11)] + [(2, 2, w) for w in (2, 3, 4, 6, 9)] + [(2, 4, w) for w in range(2, 11)] + \
         [(3, 3, 3), (4, 4, 2), (4, 4, 3), (4, 4, 4)]


def draw_case(seed, big=False):
    """big: half a million to three million observations — enough tiles per workgroup (40, 100) for the software-pipelined kernels and the
    staged x; the small cases (the default) never reach them."""
    rng = np.random.default_rng(1000003 * seed + 17)
    n_cams = int(rng.choice([2, 3, 7, 16, 63, 64, 65, 130, 500, 1800, 2261, 2262, 2300, 2600, 5000, 30000]))
    n_points = int(rng.choice([1, 2, 5, 63, 64, 65, 200, 1500, 6000, 25000]))
    mean = float(rng.choice([1.2, 2.5, 4.0, 7.0]))
    if big:
        n_cams = int(rng.choice([16, 130, 500, 1800, 2262, 5000, 30000]))
        n_points = int(rng.choice([150000, 400000]))
        mean = float(rng.choice([2.5, 4.0, 7.0]))
    k = np.clip(rng.geometric(1.0 / mean, size=n_points), 1, n_cams)
    if rng.random() < 0.35:   # some long tracks: whole tiles, rounds, more than eight tiles
        longs = [v for v in (63, 64, 65, 127, 128, 129, 200, 511, 512, 513, 700) if v <= n_cams]
        if longs:
            idx = rng.choice(n_points, size=min(n_points, int(rng.integers(1, 6))), replace=False)
            k[idx] = rng.choice(longs, size=idx.shape[0])
    if rng.random() < 0.1:
        k[:] = 1   # every point seen once
    nr, ne, nf = SHAPES[int(rng.integers(len(SHAPES)))] if rng.random() < 0.5 else (2, 3, 9)
    shared = ()
    if (nr, ne) == (2, 3) and nf in (6, 9) and rng.random() < 0.3:
        shared = [(8,), (3,), (5, 3), (4,)][int(rng.integers(4))]
    locked = ()
    if rng.random() < 0.25:
        locked = tuple(int(c) for c in rng.choice(n_cams, size=min(n_cams - 1, int(rng.integers(1, 4))), replace=False))
    prior_rows = int(rng.choice([0, 0, 0, 1, 7, 40])) if not shared else 0
    skew = float(rng.choice([0.0, 0.5, 1.0]))
    return dict(seed=seed, n_cams=n_cams, n_points=n_points, n_obs=int(k.sum()), max_track=int(k.max()), shape=[nr, ne, nf], shared=list(shared),
                locked=list(locked), prior_rows=prior_rows, skew=skew), k, rng


def build(P, case, k, layout="schur"):
    n_cams, n_points = case["n_cams"], case["n_points"]
    point_of_obs = np.repeat(np.arange(n_points, dtype=np.int64), k)
    weights = None
    if case["skew"] > 0:
        weights = np.arange(1, n_cams + 1, dtype=np.float64) ** (-case["skew"])
        weights /= weights.sum()
    cam = P._distinct_cameras(np.random.default_rng(case["seed"] + 5), n_cams, point_of_obs, weights)
    order = np.lexsort((cam, point_of_obs))
    nr, ne, nf = case["shape"]
    p = P.structured_bal(n_cams, n_points, point_of_obs, cam[order], nf, tuple(case["shared"]), True, tuple(case["locked"]), None, layout,
                         case["seed"] + 1, True, ne, nr)
    if case["prior_rows"]:
        p = P.add_camera_rows(p, case["prior_rows"], seed=case["seed"], row_size=nf, camera_width=nf)
    # a regulariser of the size of the entries: a point seen once has a singular E^T E (2 x 3 cell), and with the LM diagonal at radius
    # 1e4 the 3 x 3 block's condition number is 1e4 — the oracle and the product then differ by 1e-9 .. 1e-8 in the blocks that go
    # through its inverse (first run of this campaign, profiles/r06y_*), which says nothing about either.  Conditioning is not what
    # this campaign looks for; tests/test_gpu_edge_cases.py does the same for its runs of one-observation points.
    p.D = 0.5 + np.random.default_rng(case["seed"] + 9).random(p.bs.num_cols)
    return p


