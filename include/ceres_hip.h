/*
 * ceres_hip.h — C ABI of the MI355X-native Levenberg–Marquardt linear-solve path.
 *
 * This is the drop-in boundary.  Everything behind it is hand-written HIP for
 * gfx950; everything in front of it is host code (the Ceres-side adapter shown
 * in INTEGRATION.md, the C++ mirror in ceres-solver_amd/host/, or the ctypes
 * binding in ceres-solver_amd/__init__.py).  Plain C types only: pointers,
 * sizes, POD structs.  No torch types, no C++ types, no exceptions.
 *
 * Reference interfaces each entry point replaces (paths relative to the
 * ceres-solver tree, "I/" = internal/ceres/):
 *
 *   ceres_hip_create / _destroy      LinearSolver::Create(options)              I/linear_solver.cc:75-126
 *                                    (cases CGNR :80-88, ITERATIVE_SCHUR :111-116)
 *   ceres_hip_set_structure          the CompressedRowBlockStructure a solver   I/block_structure.h:52-182
 *                                    reads through A->block_structure(); one
 *                                    instance sees constant sparsity            I/linear_solver.h:137-142
 *   ceres_hip_solve                  LinearSolver::Solve(A, b, per_solve, x)    I/linear_solver.h:339-342
 *                                    = CgnrSolver::SolveImpl                    I/cgnr_solver.cc:146-207
 *                                    = IterativeSchurComplementSolver::SolveImpl I/iterative_schur_complement_solver.cc:64-157
 *   ceres_hip_lm_compute_step*       LevenbergMarquardtStrategy::ComputeStep + the model cost change of
 *                                    TrustRegionMinimizer::ComputeTrustRegionStep
 *                                                                               I/levenberg_marquardt_strategy.cc:69-157
 *   options.use_explicit_schur_complement   SparseSchurComplementSolver's CG path (f2)  I/schur_complement_solver.cc:337-408
 *   SCHUR_POWER_SERIES_EXPANSION     PowerSeriesExpansionPreconditioner (f3)    I/power_series_expansion_preconditioner.cc:57-89
 *   ceres_hip_bal_*                  Evaluator::Evaluate and TrustRegionMinimizer::Minimize for bundle
 *                                    adjustment in BAL form (f4)                I/evaluator.h:116-124, I/trust_region_minimizer.cc:68-845
 *   ceres_hip_op_*                   the individual operators of SURVEY.md §8(a), exported so
 *                                    that parity tests and roofline runs can address them one
 *                                    at a time (file:line given next to each declaration).
 *
 * Conventions
 *   - every function returning int returns CERES_HIP_OK (0) or a negative
 *     CERES_HIP_E_* code; on error ceres_hip_last_error() holds a message.  A
 *     non-zero return from ceres_hip_solve maps to
 *     LinearSolverTerminationType::FATAL_ERROR on the Ceres side.
 *   - "host" pointers are ordinary (or pinned) host memory owned by the caller
 *     and not retained past the call; "dev" pointers are HIP device pointers on
 *     the solver's device, also caller-owned.
 *   - all scalars on this path are IEEE fp64, all indices int32, exactly as in
 *     the reference (SURVEY.md §8 header).
 *   - a solver handle is not thread-safe (same contract as the reference,
 *     I/implicit_schur_complement.h:88-91); distinct handles are independent.
 */
#ifndef CERES_HIP_H_
#define CERES_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): ceres_hip_info grew by collectives_last_step (128 -> 136 bytes); ceres_hip_lm_options.reserved became values_unchanged */
#define CERES_HIP_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------- */
#define CERES_HIP_OK 0
#define CERES_HIP_E_INVALID (-1)     /* bad argument / call order               */
#define CERES_HIP_E_UNSUPPORTED (-2) /* structure or option not implemented     */
#define CERES_HIP_E_HIP (-3)         /* a HIP runtime call failed               */
#define CERES_HIP_E_COMM (-4)        /* an RCCL call failed, or a peer never arrived (p2p timeout) */
#define CERES_HIP_E_NODEVICE (-5)    /* no usable gfx950 device                 */

/* ---- enums: numeric values equal the reference's -------------------------
 * solver_type         ceres::LinearSolverType          include/ceres/types.h:57-91
 * preconditioner_type ceres::PreconditionerType        include/ceres/types.h:93-141
 * termination_type    LinearSolverTerminationType      I/linear_solver.h:57-74     */
#define CERES_HIP_DENSE_SCHUR 3 /* DenseSchurComplementSolver: dense S, Cholesky; num_cols_f <= CERES_HIP_MAX_EXPLICIT_SCHUR_COLS */
#define CERES_HIP_ITERATIVE_SCHUR 5
#define CERES_HIP_CGNR 6

#define CERES_HIP_IDENTITY 0
#define CERES_HIP_JACOBI 1
#define CERES_HIP_SCHUR_JACOBI 2
#define CERES_HIP_SCHUR_POWER_SERIES_EXPANSION 3 /* ITERATIVE_SCHUR only */

#define CERES_HIP_SUCCESS 0
#define CERES_HIP_NO_CONVERGENCE 1
#define CERES_HIP_FAILURE 2
#define CERES_HIP_FATAL_ERROR 3

/* Which device code path set_structure selected (ceres_hip_info.kernel_path). */
#define CERES_HIP_PATH_GENERIC 0 /* any block sizes, multi-pass kernels          */
#define CERES_HIP_PATH_BAL 1     /* fused single-pass kernels: bundle-adjustment structures — one point cell (2, 3 or 4 wide), at most one
                                  * camera cell and a few shared blocks per row, rows 2, 3 or 4 high; every static specialisation of
                                  * internal/ceres/generate_template_specializations.py:55-75 (csrc/common.h: BalShapeCompiled) */

/* ---- flattened CompressedRowBlockStructure (I/block_structure.h:52-182) ---
 * cols[j] = {col_block_size[j], col_block_pos[j]}
 * rows[i].block = {row_block_size[i], row_block_pos[i]}
 * rows[i].cells = cells[row_cell_ptr[i] .. row_cell_ptr[i+1])
 * cells[k] = {cell_col_block[k] (Cell::block_id), cell_value_pos[k] (Cell::position)}
 * Cell values are row-major row_size x col_size at values + position
 * (I/block_sparse_matrix.cc:239-274).  Nothing about the value layout is
 * assumed: E|F-split (I/block_jacobian_writer.cc:68-167), row-sequential and
 * interleaved layouts are all accepted because every access goes through
 * cell_value_pos.                                                           */
typedef struct ceres_hip_block_structure {
  int32_t num_row_blocks;
  int32_t num_col_blocks;
  const int32_t* row_block_size; /* [num_row_blocks]   */
  const int32_t* row_block_pos;  /* [num_row_blocks]   */
  const int32_t* col_block_size; /* [num_col_blocks]   */
  const int32_t* col_block_pos;  /* [num_col_blocks]   */
  const int32_t* row_cell_ptr;   /* [num_row_blocks+1] */
  const int32_t* cell_col_block; /* [num_cells]        */
  const int32_t* cell_value_pos; /* [num_cells]        */
} ceres_hip_block_structure;

/* ---- LinearSolver::Options subset that this path reads --------------------
 * I/linear_solver.h:148-230; defaults as there (max_num_iterations = 1!).   */
typedef struct ceres_hip_options {
  int32_t solver_type;           /* CERES_HIP_CGNR | CERES_HIP_ITERATIVE_SCHUR             */
  int32_t preconditioner_type;   /* IDENTITY | JACOBI | SCHUR_JACOBI                       */
  int32_t min_num_iterations;    /* LinearSolver::Options::min_num_iterations              */
  int32_t max_num_iterations;    /* LinearSolver::Options::max_num_iterations              */
  int32_t residual_reset_period; /* LinearSolver::Options::residual_reset_period (10)      */
  int32_t num_eliminate_blocks;  /* elimination_groups[0]; 0 for CGNR                      */
  int32_t device;                /* HIP device ordinal                                     */
  int32_t force_generic_path;    /* 1: never select the fused BAL kernels (testing)        */
  int32_t cg_check_interval;     /* CG iterations enqueued between host polls of the
                                    device-side termination flag; <=0 -> adaptive 2,4,8,16 */
  int32_t jacobian_storage;      /* <2,3,9> path: 0 = fp64 tiles (parity mode); 1 = Jacobian rounded
                                    to fp32 in the tiles, fp64 arithmetic (NOT parity: accuracy mode,
                                    SURVEY.md §7 item 6; halves the HBM traffic of the hot kernels)  */
  /* SCHUR_POWER_SERIES_EXPANSION (LinearSolver::Options, I/linear_solver.h:168-185) */
  int32_t max_num_spse_iterations; /* 0 -> 5 (the reference's default)                       */
  int32_t use_spse_initialization; /* start CG from the power-series estimate of S^-1 rhs    */
  double spse_tolerance;           /* for the initialisation; the preconditioner uses 0     */
  /* ITERATIVE_SCHUR with LinearSolver::Options::use_explicit_schur_complement:
   * LinearSolver::Create then returns SparseSchurComplementSolver (I/linear_solver.cc:104-109), which
   * forms S with SchurEliminator::Eliminate and runs SCHUR_JACOBI-preconditioned CG on it
   * (SolveReducedLinearSystemUsingConjugateGradients, I/schur_complement_solver.cc:337-408).
   * Here: on one rank S is stored block-sparse like the reference's BlockRandomAccessSparseMatrix (any size that fits in
   * HBM); a sharded run stores it DENSE (num_cols_f <= CERES_HIP_MAX_EXPLICIT_SCHUR_COLS) and all-reduces it.
   * preconditioner_type must be SCHUR_JACOBI (the reference CHECKs the same).                */
  int32_t use_explicit_schur_complement;
  int32_t reserved;
} ceres_hip_options;
#define CERES_HIP_MAX_EXPLICIT_SCHUR_COLS 8192

/* ---- LinearSolver::Summary (I/linear_solver.h:320-326) ------------------- */
typedef struct ceres_hip_summary {
  double residual_norm;     /* -1, as the reference's iterative solvers leave it */
  int32_t num_iterations;
  int32_t termination_type; /* CERES_HIP_SUCCESS ... CERES_HIP_FATAL_ERROR      */
  char message[256];
} ceres_hip_summary;

/* What set_structure derived; all counts in scalars unless noted. */
typedef struct ceres_hip_info {
  int32_t kernel_path;      /* CERES_HIP_PATH_*                                        */
  int32_t num_rows, num_cols;
  int32_t num_cols_e, num_cols_f;
  int32_t num_row_blocks_e; /* rows whose first cell is an E block (PMV ctor :60-66)   */
  int32_t num_e_blocks, num_f_blocks;
  int32_t row_block_size, e_block_size, f_block_size; /* DetectStructure; -1 = dynamic */
  int64_t num_nonzeros;
  int64_t num_observations; /* BAL path: rows with one E and one F cell; else 0        */
  int64_t num_tiles;        /* BAL path: 64-slot tiles after packing                   */
  int64_t device_bytes;     /* HBM held by this handle                                 */
  int32_t camera_accum_in_lds; /* BAL path: 1 if the F-space accumulators fit in LDS   */
  int32_t world_size, rank;
  int32_t p2p_enabled;      /* the one-shot peer-to-peer all-reduce is connected AND passed its self-test */
  int32_t p2p_fine_grained; /* its receive buffer is a fine-grained allocation (0: the runtime could only export a
                               coarse-grained one — fine between ranks that share a device, self-tested otherwise)  */
  /* BAL path with more cameras than LDS rows (camera_accum_in_lds == 0): hybrid accumulation (csrc/plan.cc) — the popular cameras
   * in every workgroup's LDS, every other camera in the windows of a few workgroups, points grouped accordingly                  */
  int32_t camera_accum_hybrid;        /* 1: hybrid plan; 0: every observation's F^T z is spilled (or camera_accum_in_lds)          */
  int32_t hybrid_popular_rows;        /* accumulator rows every workgroup gives to the popular cameras                             */
  int64_t num_observations_in_lds;    /* observations summed in LDS by the tile pass (the others are spilled to the ring)          */
  int32_t points_renumbered;          /* 1: the tiles hold the points in an internal order (fuller tiles, hybrid groups)           */
  int32_t cg_iteration_in_operator; /* 1: camera space of at most 512 scalars — the S.x pass finishes the CG iteration itself (one launch) */
  int32_t collectives_last_step;    /* sharded instances: all-reduces this rank issued since its last ceres_hip_lm_compute_step began      */
} ceres_hip_info;

typedef struct ceres_hip_solver ceres_hip_solver; /* opaque */

/* ---- lifetime ------------------------------------------------------------ */
int ceres_hip_abi_version(void);
/* Number of visible HIP devices whose arch is gfx950 (0 => nothing will run). */
int ceres_hip_device_count(void);
/* Create a solver; returns NULL on failure (see ceres_hip_last_error(NULL)). */
ceres_hip_solver* ceres_hip_create(const ceres_hip_options* options);
void ceres_hip_destroy(ceres_hip_solver* s);
/* Message for the most recent error on s (or, with s == NULL, of create). */
const char* ceres_hip_last_error(const ceres_hip_solver* s);

/* Upload the (constant) sparsity once per solver instance.  For a sharded run
 * each rank passes the structure of ITS rows only: E (point) column blocks are
 * disjoint across ranks, F (camera) column blocks are the same on every rank
 * (SURVEY.md §8e).  On a sharded instance (a communicator is connected) the call
 * is COLLECTIVE: the ranks agree on the kernel path and the fused shape — a shard
 * can look like bundle adjustment to one rank alone — and on where the exchanges
 * run; every rank must make it, in the same order among its collective calls.  */
int ceres_hip_set_structure(ceres_hip_solver* s, const ceres_hip_block_structure* bs);
int ceres_hip_get_info(const ceres_hip_solver* s, ceres_hip_info* info);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ----------------------
 * Rank 0 obtains an id, the host distributes the 128 bytes by any means
 * (torch.distributed broadcast in bench.py), every rank calls comm_init.
 * After that ceres_hip_solve / the op_* entry points sum F-space (camera)
 * quantities over ranks with ncclAllReduce on the solver's stream, and the
 * CGNR inner products over the E-space (point) part likewise.               */
#define CERES_HIP_UNIQUE_ID_BYTES 128
int ceres_hip_comm_get_unique_id(uint8_t id[CERES_HIP_UNIQUE_ID_BYTES]);
int ceres_hip_comm_init(ceres_hip_solver* s, const uint8_t id[CERES_HIP_UNIQUE_ID_BYTES],
                        int32_t rank, int32_t world_size);

/* Alternative (or additional) communicator: a one-shot peer-to-peer all-reduce over buffers mapped with hipIpc
 * — xGMI between the GPUs of a node, plain device memory between two ranks that share one GPU.  The sums a
 * sharded solve exchanges are 9 or 81 doubles per camera: latency-bound, several per step, so the collective is a
 * single kernel (push to every peer, flag, wait, sum in rank order = identical bits on every rank) instead of an
 * RCCL call.  Step 1: every rank calls _prepare (before set_structure) with the largest vector it will sum
 * (99 * num_f_blocks covers a step: blocks, rhs and column norms are summed in ONE all-reduce) and receives a 64-byte hipIpc handle; the host gathers the handles of all
 * ranks in rank order by any means; step 2: every rank calls _connect.  With both communicators present, messages
 * longer than max_elements go to RCCL; with only this one they are cut into pieces.  world_size <= 8.       */
#define CERES_HIP_IPC_HANDLE_BYTES 64
int ceres_hip_comm_p2p_prepare(ceres_hip_solver* s, int32_t rank, int32_t world_size, int64_t max_elements,
                               uint8_t handle_out[CERES_HIP_IPC_HANDLE_BYTES]);
/* _connect is collective and ENDS WITH THE SELF-TEST below: it returns non-zero, with the path disabled on this rank, unless known
 * values made it through every peer's buffer.  Callers never get an untested peer-to-peer path.                         */
int ceres_hip_comm_p2p_connect(ceres_hip_solver* s, const uint8_t* all_handles /* world_size x 64 bytes */);
/* Collective self-test (six all-reduces of multi-chunk vectors with values that change every round — a receive slot is
 * re-used every second epoch, so stale cached lines show from the third round on; 5 s timeout): non-zero leaves the
 * path disabled on this rank; ranks then agree on the verdict out of band and call _disable everywhere if any failed
 * (RCCL takes over).                                                                                               */
int ceres_hip_comm_p2p_selftest(ceres_hip_solver* s);
int ceres_hip_comm_p2p_disable(ceres_hip_solver* s);
/* Debug / measurement (collective): average microseconds of `iters` back-to-back all-reduces of n doubles. */
int ceres_hip_debug_allreduce_timing(ceres_hip_solver* s, int64_t n, int32_t iters, double* avg_us);

/* ---- the boundary call ----------------------------------------------------
 * LinearSolver::Solve (I/linear_solver.h:339-342).  values: num_nonzeros
 * doubles in the caller's layout; b: num_rows; D: num_cols or NULL
 * (PerSolveOptions::D, :237-257); x: num_cols, fully overwritten unless the
 * termination type is FAILURE/FATAL_ERROR
 * (I/iterative_schur_complement_solver.cc:150-154).                          */
int ceres_hip_solve(ceres_hip_solver* s, const double* host_values, const double* host_b,
                    const double* host_D, double q_tolerance, double r_tolerance,
                    double* host_x, ceres_hip_summary* summary);
/* LinearSolver::Solve again on the values and b of the previous ceres_hip_solve / ceres_hip_load on this handle, with a new D (the
 * retry after a rejected trust-region step: I/levenberg_marquardt_strategy.cc:164-176 shrinks the radius, the minimizer calls
 * ComputeStep on the same Jacobian, I/trust_region_minimizer.cc:381-461).  Nothing but D goes up, the tiles are not rebuilt.      */
int ceres_hip_solve_unchanged_values(ceres_hip_solver* s, const double* host_D, double q_tolerance, double r_tolerance,
                                     double* host_x, ceres_hip_summary* summary);
/* Same, all four arrays already resident in HBM (used by bench.py so that the
 * timed region holds no PCIe traffic; also the form a device evaluator uses).
 * STREAM ORDER of every *_device entry point: the work runs on the handle's own stream, created hipStreamNonBlocking — it does NOT
 * wait for the legacy default stream or for any stream of the caller.  Whatever produced dev_values / dev_b / dev_D, and whatever
 * last wrote the output array (a fill, say), must have COMPLETED before the call (hipStreamSynchronize or an event the host waited
 * for); when the call returns, its device work has completed and the output may be read from any stream.  (Found by
 * tools/fuzz_sequence.py --threads: a NaN fill of the output queued on a busy default stream landed after the step.)           */
int ceres_hip_solve_device(ceres_hip_solver* s, const double* dev_values, const double* dev_b,
                           const double* dev_D, double q_tolerance, double r_tolerance,
                           double* dev_x, ceres_hip_summary* summary);

/* ---- operator-level entry points (parity tests, roofline runs) ------------
 * All of them act on the state loaded by ceres_hip_load.  Vector arguments are
 * HOST pointers; results are copied back synchronously.                      */

/* Upload values/b/D (b, D may be NULL) and run the per-step re-layout kernel. */
int ceres_hip_load(ceres_hip_solver* s, const double* host_values, const double* host_b,
                   const double* host_D);
int ceres_hip_load_device(ceres_hip_solver* s, const double* dev_values, const double* dev_b,
                          const double* dev_D);

/* y += A x      BlockSparseMatrix::RightMultiplyAndAccumulate  I/block_sparse_matrix.cc:239-274 */
int ceres_hip_op_right_multiply(ceres_hip_solver* s, const double* x, double* y);
/* y += A^T x    BlockSparseMatrix::LeftMultiplyAndAccumulate   I/block_sparse_matrix.cc:278-349 */
int ceres_hip_op_left_multiply(ceres_hip_solver* s, const double* x, double* y);
/* PartitionedMatrixView<kRow,kE,kF> products: E = the first cell of each of the first
 * num_row_blocks_e rows, F = all other cells.  x / y live in the part's own column space
 * (E: num_cols_e doubles, F: num_cols_f doubles indexed at col_block_pos - num_cols_e), rows in the
 * full row space; all four accumulate into y.
 *   y += E x    RightMultiplyAndAccumulateE        I/partitioned_matrix_view_impl.h:112-139
 *   y += F x    RightMultiplyAndAccumulateF        :141-190
 *   y += E^T x  LeftMultiplyAndAccumulateE(Single|Multi)Threaded  :192-250
 *   y += F^T x  LeftMultiplyAndAccumulateF(Single|Multi)Threaded  :252-375                       */
int ceres_hip_op_right_multiply_e(ceres_hip_solver* s, const double* x, double* y);
int ceres_hip_op_right_multiply_f(ceres_hip_solver* s, const double* x, double* y);
int ceres_hip_op_left_multiply_e(ceres_hip_solver* s, const double* x, double* y);
int ceres_hip_op_left_multiply_f(ceres_hip_solver* s, const double* x, double* y);
/* blockdiag(E^T E) / blockdiag(F^T F), dense row-major blocks in column-block order (no D).
 * UpdateBlockDiagonalEtE / UpdateBlockDiagonalFtF                  :446-658                       */
int ceres_hip_op_block_diagonal_ete(ceres_hip_solver* s, double* blocks, int64_t capacity);
int ceres_hip_op_block_diagonal_ftf(ceres_hip_solver* s, double* blocks, int64_t capacity);
/* x[j] = |A_j|^2  BlockSparseMatrix::SquaredColumnNorm           I/block_sparse_matrix.cc:351-401 */
int ceres_hip_op_squared_column_norm(ceres_hip_solver* s, double* x);
/* y = (A^T A + D^2) x, one fused pass.  CgnrLinearOperator::RightMultiplyAndAccumulate
 * with y zeroed first, as CG calls it                           I/cgnr_solver.cc:98-114        */
int ceres_hip_op_jtjx(ceres_hip_solver* s, const double* x, double* y);
/* y = A^T b     CgnrSolver::SolveImpl rhs                        I/cgnr_solver.cc:188-191       */
int ceres_hip_op_jtb(ceres_hip_solver* s, double* y);

/* ImplicitSchurComplement::Init: block_diagonal (E^T E + D_e^2)^-1 and
 * rhs = F^T (b - E (E^T E)^-1 E^T b)                             I/implicit_schur_complement.cc:49-97,179-204,251-276 */
int ceres_hip_op_schur_init(ceres_hip_solver* s);
/* rhs of the reduced system: num_cols_f doubles. */
int ceres_hip_get_schur_rhs(ceres_hip_solver* s, double* rhs);
/* Inverse E^T E blocks, dense row-major e x e each, in E-block order. */
int ceres_hip_get_ete_inverse(ceres_hip_solver* s, double* blocks, int64_t capacity);
/* y = S x       ImplicitSchurComplement::RightMultiplyAndAccumulate (assigns y)
 *                                                                I/implicit_schur_complement.cc:106-144 */
int ceres_hip_op_schur_sx(ceres_hip_solver* s, const double* x, double* y);
/* ImplicitSchurComplement::BackSubstitute: z (num_cols_f) -> x (num_cols)
 *                                                                I/implicit_schur_complement.cc:208-243 */
int ceres_hip_op_back_substitute(ceres_hip_solver* s, const double* z, double* x);

/* y += (F^T F + D_f^2)^-1 F^T E (E^T E + D_e^2)^-1 E^T F x
 * ImplicitSchurComplement::InversePowerSeriesOperatorRightMultiplyAccumulate    I/implicit_schur_complement.cc:146-174
 * (call after ceres_hip_op_schur_init; computes the F^T F inverse blocks on first use) */
int ceres_hip_op_power_series_operator(ceres_hip_solver* s, const double* x, double* y);
/* y = sum_{k=0..n} Z^k (F^T F)^-1 x, stopping early when |term| < tolerance * |first term|
 * PowerSeriesExpansionPreconditioner::RightMultiplyAndAccumulate  I/power_series_expansion_preconditioner.cc:57-84 */
int ceres_hip_op_spse_apply(ceres_hip_solver* s, const double* x, double* y, int32_t max_num_spse_iterations,
                            double spse_tolerance);

/* BlockSparseJacobiPreconditioner::UpdateImpl                    I/block_jacobi_preconditioner.cc:59-115 */
int ceres_hip_op_block_jacobi_update(ceres_hip_solver* s);
/* SchurJacobiPreconditioner::UpdateImpl = SchurEliminator::Eliminate into a
 * block-diagonal lhs + Invert     I/schur_jacobi_preconditioner.cc:87-97, I/schur_eliminator_impl.h:184-311 */
int ceres_hip_op_schur_jacobi_update(ceres_hip_solver* s);
/* The preconditioner's inverted diagonal blocks, dense row-major, block order.
 * If not_inverted != 0 the blocks as they were BEFORE Invert() are returned
 * (i.e. the diagonal blocks of J^T J + D^2, or of S).                        */
int ceres_hip_get_preconditioner_blocks(ceres_hip_solver* s, int32_t not_inverted,
                                        double* blocks, int64_t capacity);
/* y += M^-1 x   BlockRandomAccessDiagonalMatrix::RightMultiplyAndAccumulate
 *                                                                I/block_random_access_diagonal_matrix.cc:102-116 */
int ceres_hip_op_precond_apply(ceres_hip_solver* s, const double* x, double* y);

/* SchurEliminator::Eliminate with a DENSE lhs (all S(i,j) cells present) and
 * rhs; lhs is num_cols_f x num_cols_f row-major, upper block triangle filled
 * like the reference (I/schur_eliminator_impl.h:184-311,548-565).  Generic
 * kernels; meant for the small/explicit-S callers.         */
int ceres_hip_op_schur_eliminate_dense(ceres_hip_solver* s, double* lhs, double* rhs);
/* The explicit Schur complement of a solver created with use_explicit_schur_complement = 1 (one rank): the lhs
 * SparseSchurComplementSolver::InitStorage allocates (I/schur_complement_solver.cc:224-290) in the storage of
 * BlockRandomAccessSparseMatrix (I/block_random_access_sparse_matrix.cc:51-110) — block pairs (i <= j, F-relative ids,
 * sorted; every (i, i) present), each a row-major n_i x n_j cell at pair_offset.
 *   _storage_info            counts
 *   _op_schur_eliminate_sparse  SchurEliminator::Eliminate into that lhs (incl. D_f^2); returns pairs, offsets, values
 *   _op_schur_symmetric_multiply  y += S x from the stored upper block triangle
 *                            BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate  :125-163            */
int ceres_hip_schur_storage_info(const ceres_hip_solver* s, int64_t* num_block_pairs, int64_t* num_values);
int ceres_hip_op_schur_eliminate_sparse(ceres_hip_solver* s, int32_t* pair_i, int32_t* pair_j, int64_t* pair_offset,
                                        double* values, int64_t pair_capacity, int64_t value_capacity);
int ceres_hip_op_schur_symmetric_multiply(ceres_hip_solver* s, const double* x, double* y);
/* SchurEliminator::BackSubstitute                                I/schur_eliminator_impl.h:314-380 */
int ceres_hip_op_eliminator_back_substitute(ceres_hip_solver* s, const double* z, double* x);

/* BLAS-1 used by CG: Dot / Norm / Axpby   I/eigen_vector_ops.h:47-101.  n <= num_cols. */
int ceres_hip_op_dot(ceres_hip_solver* s, const double* x, const double* y, int64_t n,
                     double* result);
int ceres_hip_op_axpby(ceres_hip_solver* s, double a, const double* x, double b, const double* y,
                       int64_t n, double* z);

/* ---- the neighbours of Solve inside one trust-region step, on the device ----
 * LevenbergMarquardtStrategy::ComputeStep (I/levenberg_marquardt_strategy.cc:69-157):
 *   diag = clamp(SquaredColumnNorm(J), min, max) unless reuse_diagonal; D = sqrt(diag / radius);
 *   Solve(J, residuals, {D, eta, -1}, step); finite check; step = -step
 * and the model-cost bookkeeping of TrustRegionMinimizer::ComputeTrustRegionStep
 * (I/trust_region_minimizer.cc:420-438): model_cost_change = -(J step)'(f + J step / 2).
 * J is read in place; nothing but the step and a few scalars returns to the host.            */
typedef struct ceres_hip_lm_options {
  double radius;          /* trust-region radius                                   */
  double min_diagonal;    /* Solver::Options::min_lm_diagonal (1e-6; >= 0, I/solver.cc:414) */
  double max_diagonal;    /* Solver::Options::max_lm_diagonal (1e32)               */
  double eta;             /* q_tolerance of the linear solve                       */
  int32_t reuse_diagonal; /* 1 after a rejected step: keep the previous diag(J'J)  */
  /* 1: `values` and `residuals` are what the PREVIOUS ceres_hip_lm_compute_step* / ceres_hip_load* on this handle received — the
   * Jacobian is re-evaluated only in HandleSuccessfulStep (I/trust_region_minimizer.cc:832-837), so the ComputeStep that follows a
   * rejected or invalid step (I/levenberg_marquardt_strategy.cc:134,164,170: the calls that also set reuse_diagonal_) solves on the
   * SAME matrix with a smaller radius.  The solver then keeps what it holds: no host-to-device copy (the host pointers may be NULL),
   * no re-layout into tiles; the step's first pass reads the resident tiles.  A caller that re-evaluated says 0 (separate from
   * reuse_diagonal: the two are independent statements).  Device entry point: the same device pointers, contents untouched.        */
  int32_t values_unchanged;
} ceres_hip_lm_options;
typedef struct ceres_hip_lm_result {
  ceres_hip_summary linear_solver; /* FAILURE also when the step is not finite      */
  double model_cost_change;
  int32_t step_is_finite;
  int32_t reserved;
} ceres_hip_lm_result;
int ceres_hip_lm_compute_step(ceres_hip_solver* s, const double* host_values, const double* host_residuals,
                              const ceres_hip_lm_options* options, double* host_step, ceres_hip_lm_result* result);
int ceres_hip_lm_compute_step_device(ceres_hip_solver* s, const double* dev_values, const double* dev_residuals,
                                     const ceres_hip_lm_options* options, double* dev_step,
                                     ceres_hip_lm_result* result);
/* ---- the upload hidden behind the evaluator ---------------------------------------------------------------
 * Through ceres_hip_lm_compute_step a step on a Venice-sized problem costs 21 ms, 18.4 of them the host-to-device copy of the
 * Jacobian the CPU Evaluator has just written (1.04 GB at 56 GB/s).  The evaluator writes it row block by row block, over hundreds of
 * milliseconds and from several threads (ProgramEvaluator::Evaluate's parallel loop, I/program_evaluator.h:168-300, into the pinned
 * values_ of BlockSparseMatrix(use_page_locked_memory), I/block_jacobian_writer.cc:261-262): finished rows can go up while later
 * rows are still being evaluated.
 *   _begin(values, residuals)      before the evaluator starts: the host arrays it will fill (pinned for the copies to be asynchronous;
 *                                  they must stay valid until _end returns)
 *   _ready(first_row_block, n)     THREAD-SAFE; row blocks [first, first + n) are complete in both arrays: their value ranges and
 *                                  residuals are enqueued on a copy stream.  Every row block at most once; any order; any granularity.
 *                                  (Layouts whose rows are not one or two monotone value streams — Ceres' own are: all E cells then
 *                                  all F cells in row order, or row-sequential — are remembered and sent by _end.)
 *   _end(column_scale)             rows nobody announced go up now; the solver's stream waits for the copies; column_scale != NULL:
 *                                  BlockSparseMatrix::ScaleColumns on the copy in HBM (I/block_sparse_matrix.cc:403-450) — the evaluator
 *                                  hands over the UNSCALED Jacobian, Jacobi scaling (I/trust_region_minimizer.cc:263-279) happens here.
 * The values then count as loaded by ceres_hip_load: ceres_hip_lm_compute_step(s, NULL, NULL, {.., values_unchanged = 1}, ..) computes the
 * step without another copy (the first pass re-lays the tiles out as after any load).                                              */
int ceres_hip_values_begin(ceres_hip_solver* s, const double* host_values, const double* host_residuals);
int ceres_hip_values_ready(ceres_hip_solver* s, int32_t first_row_block, int32_t num_row_blocks);
int ceres_hip_values_end(ceres_hip_solver* s, const double* host_column_scale);
/* bytes that went up before / inside _end of the last streamed upload; value streams of the layout (0: not streamable, 1, 2) */
int ceres_hip_get_stream_stats(const ceres_hip_solver* s, int64_t* bytes_early, int64_t* bytes_late, int32_t* value_streams);

/* The D the last ceres_hip_lm_compute_step* used (num_cols doubles). */
int ceres_hip_get_lm_diagonal(ceres_hip_solver* s, double* host_D);
/* values[.., col] *= scale[col] on the loaded (device) copy; returns the scaled values if
 * host_values_out != NULL.  BlockSparseMatrix::ScaleColumns       I/block_sparse_matrix.cc:403-450 */
int ceres_hip_op_scale_columns(ceres_hip_solver* s, const double* host_scale, double* host_values_out);

/* ---- the Evaluator side of the boundary for bundle adjustment in BAL form ----
 * For the one cost function of bundle_adjuster (examples/snavely_reprojection_error.h:53-105,
 * camera = angle-axis(3) translation(3) focal k1 k2, 2 residuals per observation) this replaces
 * ceres::internal::Evaluator (I/evaluator.h:98-158: Evaluate(state, cost, residuals, gradient,
 * jacobian), Plus) — ProgramEvaluator + autodiff Jets + BlockJacobianWriter in the reference —
 * and TrustRegionMinimizer::Minimize with the Levenberg-Marquardt strategy
 * (I/trust_region_minimizer.cc:68-845, I/levenberg_marquardt_strategy.cc:69-157) around the
 * linear solvers above, with the Jacobian, the residuals and every vector of the loop resident
 * in HBM: per iteration only a few scalars cross PCIe.
 *
 * The reduced program is Schur-ordered (I/reorder_program.cc:278-360): state = [3 doubles per
 * point, points 0..n_p-1 | 9 doubles per camera], residual rows grouped by point, stable in
 * observation order; the Jacobian has the BlockSparseMatrix layout BlockJacobianWriter produces
 * for it (all E cells, 6 doubles per row, then all F cells, 18 per row), which
 * is fully determined by ceres_hip_bal_get_row_order.                                                   */
typedef struct ceres_hip_bal ceres_hip_bal;
/* `options` configures the linear solver (solver_type CGNR or ITERATIVE_SCHUR, preconditioner,
 * iteration limits); num_eliminate_blocks is set to num_points.  BAL reader: examples/bal_problem.cc:75-135. */
ceres_hip_bal* ceres_hip_bal_create(const ceres_hip_options* options, int32_t num_cameras, int32_t num_points,
                                    int64_t num_observations, const int32_t* camera_index,
                                    const int32_t* point_index, const double* observations);
void ceres_hip_bal_destroy(ceres_hip_bal* p);
const char* ceres_hip_bal_last_error(const ceres_hip_bal* p);
/* The linear solver this problem drives (borrowed; operators, timing, info). */
ceres_hip_solver* ceres_hip_bal_linear_solver(ceres_hip_bal* p);
/* Evaluator::NumParameters / NumResiduals, and the number of Jacobian values (24 per observation). */
int ceres_hip_bal_sizes(const ceres_hip_bal* p, int64_t* num_parameters, int64_t* num_residuals, int64_t* num_jacobian_values);
/* row_observation[r] = index of the observation residual row block r belongs to. */
int ceres_hip_bal_get_row_order(const ceres_hip_bal* p, int32_t* row_observation);
/* Evaluator::Evaluate.  Host pointers; cost is required, the others may be NULL.  jacobian_values
 * is UNSCALED; gradient = J^T residuals.  Leaves the evaluated point loaded in the linear solver
 * (as ceres_hip_load_device would), so the ceres_hip_op_* entry points can be applied to it. */
int ceres_hip_bal_evaluate(ceres_hip_bal* p, const double* state, double* cost, double* residuals, double* gradient,
                           double* jacobian_values);
/* Solver::Options of the trust-region loop (include/ceres/solver.h defaults in comments). */
typedef struct ceres_hip_minimizer_options {
  int32_t max_num_iterations;            /* 50 */
  int32_t jacobi_scaling;                /* 1 */
  int32_t max_consecutive_invalid_steps; /* max_num_consecutive_invalid_steps, 5 */
  int32_t reserved;
  double initial_trust_region_radius;    /* 1e4 */
  double max_trust_region_radius;        /* 1e16 */
  double min_trust_region_radius;        /* 1e-32 */
  double min_lm_diagonal;                /* 1e-6 */
  double max_lm_diagonal;                /* 1e32 */
  double min_relative_decrease;          /* 1e-3 */
  double eta;                            /* 1e-1 */
  double function_tolerance;             /* 1e-6 */
  double gradient_tolerance;             /* 1e-10 */
  double parameter_tolerance;            /* 1e-8 */
} ceres_hip_minimizer_options;
void ceres_hip_minimizer_default_options(ceres_hip_minimizer_options* o);
/* IterationSummary (include/ceres/iteration_callback.h), the fields the loop itself produces. */
typedef struct ceres_hip_iteration_summary {
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int32_t step_is_successful, step_is_valid, linear_solver_iterations, linear_solver_termination;
} ceres_hip_iteration_summary;
#define CERES_HIP_MAX_LOGGED_ITERATIONS 256
#define CERES_HIP_CONVERGENCE 0    /* TerminationType CONVERGENCE    */
#define CERES_HIP_NO_CONVERGENCE_T 1 /* NO_CONVERGENCE */
#define CERES_HIP_MINIMIZER_FAILURE 2 /* FAILURE */
typedef struct ceres_hip_minimizer_summary {
  double initial_cost, final_cost;
  int32_t num_successful_steps, num_unsuccessful_steps, num_linear_solves, termination_type;
  double linear_solver_seconds, evaluation_seconds, total_seconds;
  int32_t num_iterations_logged, reserved;
  ceres_hip_iteration_summary iterations[CERES_HIP_MAX_LOGGED_ITERATIONS];
  char message[256];
} ceres_hip_minimizer_summary;
/* TrustRegionMinimizer::Minimize with LEVENBERG_MARQUARDT, monotonic steps.  state: host, in/out. */
int ceres_hip_bal_minimize(ceres_hip_bal* p, const ceres_hip_minimizer_options* options, double* state,
                           ceres_hip_minimizer_summary* summary);
/* Timing probe of the evaluator that writes the solver's tiles (what ceres_hip_bal_minimize runs per Jacobian evaluation on the
 * fused <2,3,9> path): average microseconds of `iters` back-to-back launches at `state`; flags != 0 switch groups of its stores
 * off (experiments; the handle's Jacobian is unusable afterwards until the next evaluation).                                  */
int ceres_hip_debug_bal_evaluate_tiles_timing(ceres_hip_bal* p, const double* state, int32_t flags, int32_t iters, double* avg_us);

/* ---- timing for the roofline numbers --------------------------------------
 * Runs `iters` back-to-back launches of one operator on the solver's stream
 * with device-resident operands and brackets them with HIP events recorded on
 * that same stream; returns the average milliseconds per application.        */
#define CERES_HIP_TIMED_JTJX 1
#define CERES_HIP_TIMED_SX 2
#define CERES_HIP_TIMED_SCHUR_INIT 3 /* incl. the fused re-layout, as a solve runs it */
#define CERES_HIP_TIMED_SCHUR_JACOBI 4
#define CERES_HIP_TIMED_BACK_SUBSTITUTE 5
#define CERES_HIP_TIMED_PACK 6
#define CERES_HIP_TIMED_BLOCK_JACOBI 7
#define CERES_HIP_TIMED_COPY 8 /* plain device copy of the values array: HBM ceiling probe */
#define CERES_HIP_TIMED_CGNR_SETUP 10 /* CGNR per-step set-up on the <2,3,9> path (re-layout + J^T b + JACOBI blocks) */
#define CERES_HIP_TIMED_READ_STREAM 9 /* read-only pass over the packed tiles, same loads as the fused kernels */
int ceres_hip_time_op(ceres_hip_solver* s, int32_t op, int32_t iters, double* avg_ms);
/* Per-phase event timings (ms) of the most recent ceres_hip_solve* / ceres_hip_lm_compute_step*.  The phases are bracketed by HIP events
 * on the solver's stream, and an event record between two kernels idles the device for about 6 us (a barrier packet with a completion
 * signal; seven of them in a step): they are recorded only after ceres_hip_set_phase_timing(s, 1) (or with CERES_HIP_TIMING=1 in the
 * environment).  Without it only total_ms is filled, from the host's clock around the call, and operator_applications.               */
int ceres_hip_set_phase_timing(ceres_hip_solver* s, int32_t enable);
typedef struct ceres_hip_solve_timing {
  double upload_ms, pack_ms, setup_ms, preconditioner_ms, cg_ms, back_substitute_ms,
      download_ms, total_ms;
  int32_t operator_applications; /* lhs applications inside CG (incl. residual resets) */
  int32_t reserved;
} ceres_hip_solve_timing;
int ceres_hip_get_last_timing(const ceres_hip_solver* s, ceres_hip_solve_timing* t);

/* ---- debug: the host-side tile packing plan of the <2,3,9> path -------------
 * Pure host code (no device needed): lets the CPU test-suite check the plan the
 * fused kernels rely on.  Arrays are sized n_tiles*64 (slots) / n_tiles; call once
 * with slot_capacity = 0 to obtain n_tiles.  slot_row = row block of the slot or -1. */
int ceres_hip_debug_plan(const ceres_hip_block_structure* bs, int32_t num_eliminate_blocks, int32_t* eligible,
                         int64_t* n_tiles, int32_t* slot_row, int32_t* slot_cam, int32_t* slot_pt,
                         uint32_t* slot_seg, int32_t* tile_kind, int32_t* tile_aux, int64_t slot_capacity,
                         char* why_not, int32_t why_capacity);

/* DenseCholesky::FactorAndSolve on a caller-supplied matrix (I/dense_cholesky.cc: what DENSE_SCHUR runs on its reduced system,
 * I/schur_complement_solver.cc:163-222): A is n x n row-major with its UPPER triangle authoritative, x = A^-1 b.  The blocked
 * factorisation (128-wide panels, trailing update on v_mfma_f64_16x16x4_f64) runs `repeats` times from fresh copies; *factor_ms =
 * its average duration (HIP events), *failed = 1 if a pivot was not positive (x is then b).  Needs only a created handle.          */
int ceres_hip_op_dense_cholesky_solve(ceres_hip_solver* s, int32_t n, const double* A, const double* b, double* x, int32_t repeats,
                                      double* factor_ms, int32_t* failed);

/* Debug: the camera-accumulation plan for more cameras than LDS rows (csrc/plan.cc; pure host code).  With groups >= 2 and
 * num_eliminate_blocks > 0 this is the HYBRID plan a Schur solver builds for `groups` workgroups of `rows` LDS accumulator rows;
 * with groups = 0 the spill-everything plan.  counts[8] = {n_tiles, hybrid (0/1), rows per workgroup, rows shared by every
 * workgroup (the popular cameras), first flush row of the ring, ring rows, entries of the second pass, units of the second pass}.
 * slot_word = camera | accumulator row << 20 (row 0xFFF: the slot is spilled), -1 = padding slot; tile_zbase = ring row of a tile's
 * first spilled slot; group g walks the tiles [grp_tile_ptr[g], grp_tile_ptr[g+1]) and flushes accumulator row r to ring row
 * counts[4] + g rows + r; unit u sums the ring rows entry_row[unit_begin[u] .. unit_end[u]) into camera unit_cam[u].  Call once with
 * capacities 0 for the counts.  Returns CERES_HIP_E_UNSUPPORTED if the structure is not <2,3,9>-shaped or its cameras fit in LDS. */
int ceres_hip_debug_hybrid_plan(const ceres_hip_block_structure* bs, int32_t num_eliminate_blocks, int32_t groups, int32_t rows,
                                int64_t counts[8], int32_t* slot_word, int32_t* slot_row, int32_t* tile_zbase, int32_t* grp_tile_ptr,
                                int32_t* entry_row, int32_t* unit_cam, int32_t* unit_begin, int32_t* unit_end,
                                int64_t slot_capacity, int64_t entry_capacity, int64_t unit_capacity);

/* Debug: where the plan (csrc/plan.cc; pure host code) puts the points of more than 64 observations, and the ROUNDS the streaming
 * kernels take them in.  renumber != 0: the plan of a Schur solver (points renumbered); groups >= 2: the hybrid plan.  The tiles of
 * range g (a hybrid group, or range 0 = everything) are [range_tile_ptr[g], range_tile_ptr[g+1]); from long_ptr[g] on they belong to
 * long points (tile_kind 3 = head of a point that is in the rounds, 1 = of one that is not, tile_aux = its number of tiles;
 * 2 = continuation).  A workgroup takes whole SEQUENCES of rounds: those of range g are [round_ptr[g], round_ptr[g+1]), sequence q is
 * the rounds [seq_ptr[q], seq_ptr[q+1]).  Round r gives wave w the tile round_word[8 r + w] & 0x3FFFFFF (0xFFFFFFFF: none); bits
 * 26..28 = first wave of the point, bits 29..31 = its number of waves - 1; round_flag[r]: 0 = a round of whole points, else 1 (sum)
 * or 2 (apply), + 4 on the last round of the phase: the rounds of ONE point of more than 8 tiles.  counts[5] = {n_tiles, ranges,
 * rounds, long points behind the others (0/1), sequences}.  seq_ptr holds sequences + 1 entries within round_capacity + 1.  Call
 * once with capacities 0 for the counts.                                                                                          */
int ceres_hip_debug_long_rounds(const ceres_hip_block_structure* bs, int32_t num_eliminate_blocks, int32_t renumber, int32_t groups,
                                int32_t rows, int64_t counts[5], int32_t* tile_kind, int32_t* tile_aux, int32_t* range_tile_ptr,
                                int32_t* long_ptr, int32_t* round_ptr, int32_t* seq_ptr, int32_t* round_flag, uint32_t* round_word,
                                int64_t tile_capacity, int64_t range_capacity, int64_t round_capacity);

/* Debug: which cameras' part of x the streaming kernels keep in LDS beside the accumulators (csrc/plan.cc, BalPlan::xhot_cam; pure
 * host code; the plan of a Schur solver when num_eliminate_blocks > 0).  counts[4] = {n_tiles, cameras, staged cameras, LDS bytes the
 * accumulators take}; staged_cam[r] = the camera of LDS row r (most observed first); slot_word = camera | (row + 1) << 20 for a staged
 * camera's slot, camera alone otherwise, -1 = padding slot.  counts[2] = 0 when the accumulators do not all fit in LDS (the hybrid plan's
 * regime) or CERES_HIP_XHOT=0.  Call once with capacities 0 for the counts.  Returns CERES_HIP_E_UNSUPPORTED for structures off the fused path. */
int ceres_hip_debug_staged_x_plan(const ceres_hip_block_structure* bs, int32_t num_eliminate_blocks, int64_t counts[4], int32_t* staged_cam,
                                  int32_t* slot_word, int64_t staged_capacity, int64_t slot_capacity);

/* Debug: exercise the sharded (world > 1) code paths on one GPU through a 1-rank RCCL
 * communicator; the instance must then be given the WHOLE problem.  Call before set_structure. */
int ceres_hip_debug_comm_loopback(ceres_hip_solver* s, int32_t logical_world);
/* Measurement (bench.py: extra.shard_ceiling): this instance becomes rank 0 of `logical_world` ranks whose PEERS ARE GHOSTS.  The
 * peer-to-peer all-reduce kernel does all its work — pushes into every peer's slot, sets and awaits every peer's flags, adds the
 * `world` slots in rank order — against local dummy buffers that read "arrived" with zero contributions.  Given ONE RANK'S SHARD of
 * a problem the instance runs the sharded code path alone on the device: the time a perfect interconnect would give that rank
 * (its sums are the shard's own, so its results are not the whole problem's).  Call before set_structure; max_elements as for
 * ceres_hip_comm_p2p_prepare.                                                                                                  */
int ceres_hip_debug_comm_ghost_peers(ceres_hip_solver* s, int32_t logical_world, int64_t max_elements);

#ifdef __cplusplus
}
#endif
#endif /* CERES_HIP_H_ */
