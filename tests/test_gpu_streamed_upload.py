"""The upload hidden behind the evaluator (include/ceres_hip.h: ceres_hip_values_begin / _ready / _end): rows go up
from several threads while later rows are still being written, unscaled, Jacobi scaling on the device — and the step must be the
one ceres_hip_lm_compute_step computes from the same values — the same kernels on the same bytes, so equal up to the order of the
camera-space additions in LDS (1e-12) — which in turn is checked against the oracle.

The reference's side of this boundary: ProgramEvaluator::Evaluate's parallel loop (internal/ceres/program_evaluator.h:168-300) writing
BlockSparseMatrix::values() (pinned: internal/ceres/block_jacobian_writer.cc:261-262), then ScaleColumns
(internal/ceres/trust_region_minimizer.cc:263-279)."""
import threading

import numpy as np
import pytest

from test_gpu_lm_step import check_step, reference_step

pytestmark = pytest.mark.gpu


def same(a, b, tol=1e-12):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b))) <= tol * float(np.linalg.norm(np.asarray(b)))


def make(hip, p, solver_type, pre):
    o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                elimination_groups=[p.num_eliminate_blocks if solver_type != hip.CGNR else 0])
    s = hip.HipLinearSolver(o)
    s.set_structure(p.bs)
    return s


def stream_up(s, p, values_src, b_src, n_threads=8, run=97, skip_every=0, scale=None, order="shuffled"):
    """Eight 'evaluator' threads fill the host arrays run by run (runs in a shuffled order) and announce them."""
    bs = p.bs
    hv = np.full(values_src.shape[0], np.nan)
    hb = np.full(b_src.shape[0], np.nan)
    ptr = bs.row_cell_ptr.astype(np.int64)
    rsz = bs.row_block_size.astype(np.int64)
    csz = bs.col_block_size.astype(np.int64)
    runs = [(r0, min(bs.num_row_blocks, r0 + run)) for r0 in range(0, bs.num_row_blocks, run)]
    if order == "shuffled":
        np.random.default_rng(5).shuffle(runs)
    s.values_begin(hv, hb)
    lock, errors = threading.Lock(), []

    def worker(t):
        try:
            while True:
                with lock:
                    if not runs:
                        return
                    k = len(runs)
                    r0, r1 = runs.pop()
                for r in range(r0, r1):
                    for c in range(ptr[r], ptr[r + 1]):
                        a = int(bs.cell_value_pos[c])
                        n = int(rsz[r] * csz[bs.cell_col_block[c]])
                        hv[a:a + n] = values_src[a:a + n]
                    b0 = int(bs.row_block_pos[r])
                    hb[b0:b0 + int(rsz[r])] = b_src[b0:b0 + int(rsz[r])]
                if skip_every and k % skip_every == 0:
                    continue   # never announced: values_end sends it
                s.values_ready(r0, r1 - r0)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    s.values_end(scale)
    return s.stream_stats()


@pytest.mark.parametrize("layout,solver_type,pre,streams", [("schur", 5, 2, 2), ("schur", 6, 1, 2), ("cgnr", 6, 1, 1)])
def test_streamed_step_equals_the_plain_step(hip, oracle, problems, layout, solver_type, pre, streams):
    p = problems.synthetic_bal(None, layout=layout, num_cameras=31, num_points=4000, num_observations=17000, seed=11, skew=0.5)
    ref = make(hip, p, solver_type, pre)
    step0, summ0, mcc0 = ref.lm_compute_step(p.values, p.b, 1e4, 0.1)
    D = ref.lm_diagonal()
    check_step(oracle, hip, p, solver_type, pre, D, step0, summ0, mcc0, 0.1)
    s = make(hip, p, solver_type, pre)
    early, late, k = stream_up(s, p, p.values, p.b, skip_every=7)
    assert k == streams and early > 0 and late > 0 and early + late == 8 * (p.values.shape[0] + p.b.shape[0])
    step, summ, mcc = s.lm_compute_step(None, None, 1e4, 0.1, values_unchanged=True)
    assert summ.num_iterations == summ0.num_iterations and same(step, step0) and same(mcc, mcc0)
    # a second evaluation on the same handle (the next LM iteration): other values, everything announced, in row order
    v2 = p.values * (1.0 + 0.1 * np.cos(np.arange(p.values.shape[0])))
    early, late, _ = stream_up(s, p, v2, p.b, run=1000, order="rows")
    assert late == 0
    step2, summ2, mcc2 = s.lm_compute_step(None, None, 1e4, 0.1, values_unchanged=True)
    step2r, summ2r, mcc2r = ref.lm_compute_step(v2, p.b, 1e4, 0.1)
    assert same(step2, step2r) and same(mcc2, mcc2r)


def test_jacobi_scaling_on_the_device(hip, oracle, problems):
    p = problems.synthetic_bal(None, layout="schur", num_cameras=19, num_points=2500, num_observations=11000, seed=3, skew=0.4)
    m = oracle.Matrix(p.bs, 0)
    scale = 1.0 / (1.0 + np.sqrt(m.squared_column_norm(p.values)))   # TrustRegionMinimizer's jacobian_scaling_ (:263-279)
    scaled = ref_scaled = None
    ref = make(hip, p, 5, 2)
    ref.load(p.values, p.b)
    scaled = ref.scale_columns(scale)                                  # BlockSparseMatrix::ScaleColumns, already covered against the oracle
    step0, summ0, mcc0 = ref.lm_compute_step(scaled, p.b, 1e4, 0.1)
    s = make(hip, p, 5, 2)
    stream_up(s, p, p.values, p.b, scale=scale)                        # the UNSCALED values go up
    step, summ, mcc = s.lm_compute_step(None, None, 1e4, 0.1, values_unchanged=True)
    assert summ.num_iterations == summ0.num_iterations and same(step, step0) and same(mcc, mcc0)
    ps = type(p)(p.bs, scaled, p.b, None, p.num_eliminate_blocks)
    check_step(oracle, hip, ps, 5, 2, s.lm_diagonal(), step, summ, mcc, 0.1)
    del ref_scaled


def test_a_layout_without_monotone_streams_goes_up_at_the_end(hip, problems):
    # the cells of the rows in reverse order in the value array: not one or two monotone streams
    p = problems.synthetic_bal(None, layout="schur", num_cameras=9, num_points=300, num_observations=1200, seed=2)
    bs = p.bs
    n_cells = bs.cell_value_pos.shape[0]
    rows_of_cell = np.repeat(np.arange(bs.num_row_blocks), np.diff(bs.row_cell_ptr))
    size = bs.row_block_size[rows_of_cell].astype(np.int64) * bs.col_block_size[bs.cell_col_block].astype(np.int64)
    new_pos = np.zeros(n_cells, dtype=np.int64)
    new_pos[::-1] = np.concatenate([[0], np.cumsum(size[::-1])[:-1]])
    values = np.empty_like(p.values)
    for c in range(n_cells):
        values[new_pos[c]:new_pos[c] + size[c]] = p.values[bs.cell_value_pos[c]:bs.cell_value_pos[c] + size[c]]
    bs2 = type(bs)(bs.row_block_size, bs.row_block_pos, bs.col_block_size, bs.col_block_pos, bs.row_cell_ptr, bs.cell_col_block, new_pos.astype(np.int32))
    p2 = type(p)(bs2, values, p.b, None, p.num_eliminate_blocks)
    ref = make(hip, p2, 5, 2)
    step0, summ0, mcc0 = ref.lm_compute_step(p2.values, p2.b, 1e4, 0.1)
    s = make(hip, p2, 5, 2)
    early, late, k = stream_up(s, p2, p2.values, p2.b)
    assert k == 0 and early == 0 and late == 8 * (values.shape[0] + p.b.shape[0])
    step, summ, mcc = s.lm_compute_step(None, None, 1e4, 0.1, values_unchanged=True)
    assert same(step, step0) and same(mcc, mcc0)


def test_misuse_is_an_error(hip, problems):
    p = problems.synthetic_bal(None, layout="schur", num_cameras=7, num_points=200, num_observations=800, seed=4)
    s = make(hip, p, 5, 2)
    with pytest.raises(hip.HipError):
        s.values_ready(0, 1)                      # outside begin / end
    with pytest.raises(hip.HipError):
        s.values_end()
    hv, hb = p.values.copy(), p.b.copy()
    s.values_begin(hv, hb)
    s.values_ready(0, 10)
    with pytest.raises(hip.HipError):
        s.values_ready(5, 1)                      # announced twice
    with pytest.raises(hip.HipError):
        s.values_ready(p.bs.num_row_blocks - 1, 2)  # out of range
    s.values_end()
    step, summ, mcc = s.lm_compute_step(None, None, 1e4, 0.1, values_unchanged=True)
    assert np.isfinite(step).all() and mcc > 0


@pytest.mark.parametrize("shuffle,streams", [(False, 1), (True, 0)])
def test_streaming_on_the_generic_kernels(hip, oracle, problems, shuffle, streams):
    """Any block sizes, rows with several F cells and rows without an E block (the generic path): a row-sequential value layout is one
    stream, a shuffled one goes up at the end; the solve on the streamed values equals the plain solve and the oracle's."""
    p = problems.random_schur_problem(num_e_blocks=40, num_f_blocks=9, seed=8, shuffle_values=shuffle)
    ref = make(hip, p, 5, 2)
    assert ref.info().kernel_path == hip.PATH_GENERIC
    step0, summ0, mcc0 = ref.lm_compute_step(p.values, p.b, 1e4, 1e-12)
    s = make(hip, p, 5, 2)
    early, late, k = stream_up(s, p, p.values, p.b, run=3)
    assert k == streams and (early > 0) == (streams > 0)
    step, summ, mcc = s.lm_compute_step(None, None, 1e4, 1e-12, values_unchanged=True)
    assert summ.termination_type == hip.SUCCESS and same(step, step0, 1e-9) and same(mcc, mcc0, 1e-9)
    check_step(oracle, hip, p, 5, 2, s.lm_diagonal(), step, summ, mcc, 1e-12, tol=1e-8)
