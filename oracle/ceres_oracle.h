/*
 * ceres_oracle.h — CPU restatement of the reference's LM linear-solve path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the
 * checker / the reported CPU baseline — never as part of the product path
 * (the product is include/ceres_hip.h + ceres-solver_amd/csrc, which fails
 * loudly when no gfx950 device is present).
 *
 * Parity status: PINNED by the reference's own known-answer problems
 * (internal/ceres/linear_least_squares_problems.cc:78-185,657-675, and the
 * conjugate-gradients tests, internal/ceres/conjugate_gradients_solver_test.cc:57-140)
 * and by the dense-algebra constructions of the reference's unit tests
 * re-expressed with numpy (tests/test_oracle_*.py).  The reference itself
 * cannot be compiled in this image (Eigen3 and abseil are REQUIRED and
 * absent) so there is no oracle/_ref; CGNR at the LinearSolver
 * boundary has no reference test at all and is pinned only through its parts
 * and against dense normal-equation solves.
 *
 * Every function cites the reference file:line it restates ("I/" =
 * internal/ceres/).  The code is written from the algorithms, not from the
 * source text: generic loops instead of the templated small-BLAS, std::map
 * instead of absl::btree_map, OpenMP instead of ParallelFor.
 */
#ifndef CERES_ORACLE_H_
#define CERES_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field layout as ceres_hip_block_structure (include/ceres_hip.h);
 * restates CompressedRowBlockStructure, I/block_structure.h:52-182.        */
typedef struct oracle_block_structure {
  int32_t num_row_blocks;
  int32_t num_col_blocks;
  const int32_t* row_block_size;
  const int32_t* row_block_pos;
  const int32_t* col_block_size;
  const int32_t* col_block_pos;
  const int32_t* row_cell_ptr;
  const int32_t* cell_col_block;
  const int32_t* cell_value_pos;
} oracle_block_structure;

typedef struct oracle_summary {
  double residual_norm;
  int32_t num_iterations;
  int32_t termination_type; /* 0 SUCCESS 1 NO_CONVERGENCE 2 FAILURE 3 FATAL_ERROR */
  char message[256];
} oracle_summary;

typedef struct oracle_matrix oracle_matrix; /* structure + transpose + chunks */

/* Sum `n` doubles in place over all ranks (sharded solves). NULL = 1 rank. */
typedef void (*oracle_allreduce_fn)(void* ctx, double* buf, int64_t n);

void oracle_set_num_threads(int n);
int oracle_get_num_threads(void);

oracle_matrix* oracle_matrix_create(const oracle_block_structure* bs, int num_eliminate_blocks);
void oracle_matrix_destroy(oracle_matrix* m);
int oracle_matrix_num_rows(const oracle_matrix* m);
int oracle_matrix_num_cols(const oracle_matrix* m);
int oracle_matrix_num_cols_e(const oracle_matrix* m);
int oracle_matrix_num_cols_f(const oracle_matrix* m);
int oracle_matrix_num_row_blocks_e(const oracle_matrix* m);
int64_t oracle_matrix_num_nonzeros(const oracle_matrix* m);
/* DetectStructure, I/detect_structure.cc:39-121; -1 = Eigen::Dynamic. */
void oracle_detect_structure(const oracle_matrix* m, int* row, int* e, int* f);

/* BlockSparseMatrix, I/block_sparse_matrix.cc */
void oracle_right_multiply(const oracle_matrix* m, const double* values, const double* x, double* y);
void oracle_left_multiply(const oracle_matrix* m, const double* values, const double* x, double* y);
void oracle_squared_column_norm(const oracle_matrix* m, const double* values, double* x);
void oracle_scale_columns(const oracle_matrix* m, double* values, const double* scale);
void oracle_to_dense(const oracle_matrix* m, const double* values, double* dense_row_major);

/* PartitionedMatrixView, I/partitioned_matrix_view_impl.h */
void oracle_right_multiply_e(const oracle_matrix* m, const double* values, const double* x, double* y);
void oracle_right_multiply_f(const oracle_matrix* m, const double* values, const double* x, double* y);
void oracle_left_multiply_e(const oracle_matrix* m, const double* values, const double* x, double* y);
void oracle_left_multiply_f(const oracle_matrix* m, const double* values, const double* x, double* y);
/* Concatenated dense row-major diagonal blocks. */
void oracle_block_diagonal_ete(const oracle_matrix* m, const double* values, double* blocks);
void oracle_block_diagonal_ftf(const oracle_matrix* m, const double* values, double* blocks);

/* InvertPSDMatrix / BlockRandomAccessDiagonalMatrix::Invert: in-place inverse of
 * an SPD n x n row-major matrix reading only the upper triangle.           */
int oracle_invert_psd(int n, double* a);

/* ImplicitSchurComplement, I/implicit_schur_complement.cc */
typedef struct oracle_isc oracle_isc;
oracle_isc* oracle_isc_create(const oracle_matrix* m);
void oracle_isc_destroy(oracle_isc* s);
void oracle_isc_init(oracle_isc* s, const double* values, const double* D, const double* b);
void oracle_isc_sx(oracle_isc* s, const double* x, double* y);
void oracle_isc_rhs(const oracle_isc* s, double* rhs);
void oracle_isc_ete_inverse(const oracle_isc* s, double* blocks);
void oracle_isc_back_substitute(oracle_isc* s, const double* z, double* x);

/* SchurEliminator, I/schur_eliminator_impl.h.  diagonal_only: lhs is the
 * concatenated diagonal blocks (BlockRandomAccessDiagonalMatrix); otherwise a
 * dense num_cols_f x num_cols_f row-major matrix with every cell present.
 * b, D, rhs may be NULL exactly as in the reference.                       */
void oracle_schur_eliminate(const oracle_matrix* m, const double* values, const double* b,
                            const double* D, int diagonal_only, double* lhs, double* rhs);
void oracle_schur_back_substitute(const oracle_matrix* m, const double* values, const double* b,
                                  const double* D, const double* z, double* y);

/* Preconditioners: concatenated dense diagonal blocks, inverted and (if
 * raw != NULL) as they were before inversion.                              */
void oracle_block_jacobi(const oracle_matrix* m, const double* values, const double* D,
                         double* inverted, double* raw);
void oracle_schur_jacobi(const oracle_matrix* m, const double* values, const double* D,
                         double* inverted, double* raw);
/* y += blockdiag(blocks) x over the given block sizes. */
void oracle_block_diagonal_apply(int num_blocks, const int32_t* block_size, const double* blocks,
                                 const double* x, double* y);

/* ConjugateGradientsSolver on a dense SPD system (known-answer tests),
 * I/conjugate_gradients_solver.h:108-306.  Minv may be NULL (identity).    */
void oracle_cg_dense(int n, const double* A, const double* b, const double* Minv, int min_it,
                     int max_it, int reset_period, double q_tol, double r_tol, double* x,
                     oracle_summary* summary);

/* CgnrSolver::SolveImpl, I/cgnr_solver.cc:146-207.  preconditioner: 0 IDENTITY, 1 JACOBI. */
void oracle_cgnr_solve(const oracle_matrix* m, const double* values, const double* b,
                       const double* D, int preconditioner, int min_it, int max_it,
                       int reset_period, double q_tol, double r_tol, double* x,
                       oracle_summary* summary);
/* IterativeSchurComplementSolver::SolveImpl, I/iterative_schur_complement_solver.cc:64-157.
 * preconditioner: 0 IDENTITY, 1 JACOBI (blockdiag(F^T F)^-1), 2 SCHUR_JACOBI. */
void oracle_iterative_schur_solve(const oracle_matrix* m, const double* values, const double* b,
                                  const double* D, int preconditioner, int min_it, int max_it,
                                  int reset_period, double q_tol, double r_tol, double* x,
                                  oracle_summary* summary);
/* SCHUR_POWER_SERIES_EXPANSION (I/power_series_expansion_preconditioner.cc:57-84,
 * I/implicit_schur_complement.cc:146-174): preconditioner = 3 and/or use_spse_initialization. */
void oracle_iterative_schur_solve_spse(const oracle_matrix* m, const double* values, const double* b,
                                       const double* D, int preconditioner, int min_it, int max_it,
                                       int reset_period, double q_tol, double r_tol, int use_spse_initialization,
                                       int max_num_spse_iterations, double spse_tolerance, double* x,
                                       oracle_summary* summary);
void oracle_isc_compute_ftf_inverse(oracle_isc* s);
void oracle_isc_power_series_operator(oracle_isc* s, const double* x, double* y);
void oracle_isc_spse_apply(oracle_isc* s, const double* x, double* y, int max_num_spse_iterations, double spse_tolerance);
/* The same two solvers on one rank's shard of the rows (E blocks disjoint
 * across ranks, F blocks replicated).  F-space vectors and
 * diagonal blocks are summed with `allreduce`; for CGNR the E-space parts of
 * the inner products are summed too.                                       */
void oracle_cgnr_solve_sharded(const oracle_matrix* m, const double* values, const double* b,
                               const double* D, int preconditioner, int min_it, int max_it,
                               int reset_period, double q_tol, double r_tol, double* x,
                               oracle_summary* summary, oracle_allreduce_fn allreduce, void* ctx);
void oracle_iterative_schur_solve_sharded(const oracle_matrix* m, const double* values,
                                          const double* b, const double* D, int preconditioner,
                                          int min_it, int max_it, int reset_period, double q_tol,
                                          double r_tol, double* x, oracle_summary* summary,
                                          oracle_allreduce_fn allreduce, void* ctx);

/* ---- BAL harness (bal_harness.cc) ----------------------------------------
 * Caller of the boundary restated so that "LM steps" exist:
 * examples/bal_problem.cc:75-135 (file format), examples/snavely_reprojection_error.h:53-105
 * (residual), I/block_jacobian_writer.cc:68-263 + I/reorder_program.cc:278-360
 * (Jacobian layout), I/levenberg_marquardt_strategy.cc:69-177 and
 * I/trust_region_minimizer.cc:68-137,246-461,781-847 (LM loop).            */
typedef struct oracle_bal oracle_bal;
/* Synthetic scene: cameras on a ring looking at a point cloud, every point
 * seen by >= 2 distinct cameras.  num_observations is approximate on input and
 * exact on the returned problem.  skew > 0 makes camera popularity power-law. */
oracle_bal* oracle_bal_generate(int num_cameras, int num_points, int64_t num_observations,
                                double skew, double pixel_noise, double param_noise,
                                uint64_t seed);
oracle_bal* oracle_bal_read(const char* filename);
int oracle_bal_write(const oracle_bal* p, const char* filename);
void oracle_bal_destroy(oracle_bal* p);
int oracle_bal_num_cameras(const oracle_bal* p);
int oracle_bal_num_points(const oracle_bal* p);
int64_t oracle_bal_num_observations(const oracle_bal* p);
/* parameters: 9*num_cameras camera doubles then 3*num_points point doubles (BAL file order). */
double* oracle_bal_parameters(oracle_bal* p);
const int32_t* oracle_bal_camera_index(const oracle_bal* p);
const int32_t* oracle_bal_point_index(const oracle_bal* p);
const double* oracle_bal_observations(const oracle_bal* p);

/* Jacobian structure in one of the two layouts the reference produces:
 * schur_ordering=1: column blocks = points (3) then cameras (9), rows grouped by
 *   point, values E|F-split   (ITERATIVE_SCHUR; I/block_jacobian_writer.cc:68-167);
 * schur_ordering=0: column blocks in order of first use camera,point,..., rows in
 *   observation order, cells sorted by column block, values row-sequential (CGNR).
 * Arrays are owned by the problem object and valid until it is destroyed or the
 * layout is rebuilt.  Returns the number of eliminate blocks (points or 0).   */
int oracle_bal_build_structure(oracle_bal* p, int schur_ordering, oracle_block_structure* out);
/* state: num_cols doubles in the structure's column order.                   */
void oracle_bal_get_state(const oracle_bal* p, double* state);
void oracle_bal_set_state(oracle_bal* p, const double* state);
/* AngleAxisRotatePoint (include/ceres/rotation.h:864-905) on n (angle_axis, point) pairs; and the Snavely residual with
 * its 2x9 | 2x3 Jacobian by dual numbers (examples/snavely_reprojection_error.h:53-105).  For pinning tests. */
void oracle_angle_axis_rotate_points(int64_t n, const double* angle_axis, const double* pts, double* out);
void oracle_snavely_batch(int64_t n, const double* cams, const double* pts, const double* obs, double* residuals,
                          double* jac_cam, double* jac_pt);

/* residuals (2/obs, row order) and, if values != NULL, the Jacobian values in the
 * structure's layout; returns cost = 1/2 |r|^2.                              */
double oracle_bal_evaluate(const oracle_bal* p, const double* state, double* residuals,
                           double* values);

typedef int (*oracle_linear_solve_fn)(void* ctx, const double* values, const double* b,
                                      const double* D, double q_tolerance, double r_tolerance,
                                      double* x, oracle_summary* summary);
typedef struct oracle_lm_options {
  int max_num_iterations;       /* 50  solver.h */
  double initial_radius;        /* 1e4 */
  double max_radius;            /* 1e16 */
  double min_radius;            /* 1e-32 */
  double min_lm_diagonal;       /* 1e-6 */
  double max_lm_diagonal;       /* 1e32 */
  double min_relative_decrease; /* 1e-3 */
  double eta;                   /* 0.1 */
  double function_tolerance;    /* 1e-6 */
  double gradient_tolerance;    /* 1e-10 */
  double parameter_tolerance;   /* 1e-8 */
  int jacobi_scaling;           /* 1 */
  int max_consecutive_invalid_steps; /* 5 */
} oracle_lm_options;
typedef struct oracle_lm_iteration {
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius;
  int32_t step_is_successful, step_is_valid, linear_solver_iterations, linear_termination;
} oracle_lm_iteration;
typedef struct oracle_lm_summary {
  double initial_cost, final_cost;
  int32_t num_successful_steps, num_unsuccessful_steps, num_linear_solves, termination;
  double linear_solver_seconds, total_seconds;
  int32_t num_iterations_logged;
  oracle_lm_iteration iterations[256];
  char message[256];
} oracle_lm_summary;
void oracle_lm_default_options(oracle_lm_options* o);
/* Minimise; the linear solve at each step goes through `solve` (the oracle's
 * own solvers or, in the GPU tests, ceres_hip_solve).                        */
void oracle_lm_solve(oracle_bal* p, const oracle_block_structure* bs, const oracle_lm_options* o,
                     oracle_linear_solve_fn solve, void* ctx, oracle_lm_summary* summary);
/* Ready-made solve callbacks on the oracle's own solvers; ctx = oracle_solver_ctx*. */
typedef struct oracle_solver_ctx {
  const oracle_matrix* m;
  int solver_type;    /* 5 ITERATIVE_SCHUR, 6 CGNR */
  int preconditioner; /* 0,1,2 */
  int min_it, max_it, reset_period;
} oracle_solver_ctx;
int oracle_solver_callback(void* ctx, const double* values, const double* b, const double* D,
                           double q_tolerance, double r_tolerance, double* x,
                           oracle_summary* summary);

#ifdef __cplusplus
}
#endif
#endif
