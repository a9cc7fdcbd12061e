// hip_linear_solver.h — C++ host side above the C ABI, mirroring the reference's plugin
// interface for this path so that a Ceres maintainer recognises every name:
//
//   ceres_hip::Block / Cell / CompressedRow / CompressedRowBlockStructure
//                                  ~ internal/ceres/block_structure.h:52-182
//   ceres_hip::BlockSparseMatrix   ~ internal/ceres/block_sparse_matrix.h (storage + structure only)
//   ceres_hip::LinearSolver::{Options, PerSolveOptions, Summary, Create, Solve}
//                                  ~ internal/ceres/linear_solver.h:148-354, linear_solver.cc:51-126
//   ceres_hip::HipLinearSolver     = what INTEGRATION.md's HipCgnrSolver /
//                                    HipIterativeSchurComplementSolver do inside Ceres
//
// Header-only and dependency-free (no Eigen, no abseil): the reference's own headers
// cannot be compiled in this environment, so this mirror is what the host driver and the
// tests build against.  The version that derives from the real
// ceres::internal::BlockSparseMatrixSolver is spelled out in INTEGRATION.md.
#ifndef CERES_HIP_HOST_HIP_LINEAR_SOLVER_H_
#define CERES_HIP_HOST_HIP_LINEAR_SOLVER_H_

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ceres_hip.h"
#include "flatten_block_structure.h"

namespace ceres_hip {

struct Block {
  int size = -1, position = -1;
  Block() = default;
  Block(int s, int p) : size(s), position(p) {}
};
struct Cell {
  int block_id = -1, position = -1;
  Cell() = default;
  Cell(int b, int p) : block_id(b), position(p) {}
};
struct CompressedRow {
  Block block;
  std::vector<Cell> cells;
};
struct CompressedRowBlockStructure {
  std::vector<Block> cols;
  std::vector<CompressedRow> rows;
};

class BlockSparseMatrix {
 public:
  explicit BlockSparseMatrix(CompressedRowBlockStructure* bs) : bs_(bs) {
    for (const auto& c : bs_->cols) num_cols_ += c.size;
    for (const auto& r : bs_->rows) {
      num_rows_ += r.block.size;
      for (const auto& cell : r.cells) {
        const int n = r.block.size * bs_->cols[cell.block_id].size;
        num_nonzeros_ += n;
        if (cell.position + n > extent_) extent_ = cell.position + n;
      }
    }
    values_.assign(extent_, 0.0);
  }
  int num_rows() const { return num_rows_; }
  int num_cols() const { return num_cols_; }
  int num_nonzeros() const { return num_nonzeros_; }
  const double* values() const { return values_.data(); }
  double* mutable_values() { return values_.data(); }
  const CompressedRowBlockStructure* block_structure() const { return bs_.get(); }

 private:
  std::unique_ptr<CompressedRowBlockStructure> bs_;
  std::vector<double> values_;
  int num_rows_ = 0, num_cols_ = 0, num_nonzeros_ = 0, extent_ = 0;
};

// Values equal ceres::LinearSolverType / PreconditionerType / LinearSolverTerminationType.
enum LinearSolverType { DENSE_SCHUR = CERES_HIP_DENSE_SCHUR, ITERATIVE_SCHUR = CERES_HIP_ITERATIVE_SCHUR, CGNR = CERES_HIP_CGNR };
enum PreconditionerType { IDENTITY = CERES_HIP_IDENTITY, JACOBI = CERES_HIP_JACOBI, SCHUR_JACOBI = CERES_HIP_SCHUR_JACOBI,
                          SCHUR_POWER_SERIES_EXPANSION = CERES_HIP_SCHUR_POWER_SERIES_EXPANSION };
enum class LinearSolverTerminationType { SUCCESS = 0, NO_CONVERGENCE = 1, FAILURE = 2, FATAL_ERROR = 3 };

class LinearSolver {
 public:
  struct Options {
    LinearSolverType type = ITERATIVE_SCHUR;
    PreconditionerType preconditioner_type = JACOBI;
    int min_num_iterations = 1;
    int max_num_iterations = 1;  // the reference's default (internal/ceres/linear_solver.h:165-166)
    int residual_reset_period = 10;
    std::vector<int> elimination_groups;
    bool use_explicit_schur_complement = false;  // internal/ceres/linear_solver.h:187-190
    int max_num_spse_iterations = 5;
    bool use_spse_initialization = false;
    double spse_tolerance = 0.1;
    int device = 0;  // not in the reference: HIP device ordinal
  };
  struct PerSolveOptions {
    const double* D = nullptr;
    double r_tolerance = 0.0;
    double q_tolerance = 0.0;
  };
  struct Summary {
    double residual_norm = -1.0;
    int num_iterations = -1;
    LinearSolverTerminationType termination_type = LinearSolverTerminationType::FAILURE;
    std::string message;
  };
  virtual ~LinearSolver() = default;
  virtual Summary Solve(BlockSparseMatrix* A, const double* b, const PerSolveOptions& per_solve_options, double* x) = 0;
  // LinearSolver::Create with its fall-backs (internal/ceres/linear_solver.cc:51-73):
  // ITERATIVE_SCHUR with no eliminated blocks becomes CGNR, SCHUR_JACOBI becomes JACOBI.
  static std::unique_ptr<LinearSolver> Create(const Options& options);
};

// The drop-in: flattens the structure on the first Solve (one instance sees one sparsity,
// internal/ceres/linear_solver.h:137-142), forwards to the C ABI, translates the summary.
class HipLinearSolver final : public LinearSolver {
 public:
  explicit HipLinearSolver(const Options& options) : options_(options) {
    ceres_hip_options o{};
    o.solver_type = options.type;
    o.preconditioner_type = options.preconditioner_type;
    o.min_num_iterations = options.min_num_iterations;
    o.max_num_iterations = options.max_num_iterations;
    o.residual_reset_period = options.residual_reset_period;
    o.num_eliminate_blocks = options.elimination_groups.empty() ? 0 : options.elimination_groups[0];
    o.device = options.device;
    o.use_explicit_schur_complement = options.use_explicit_schur_complement ? 1 : 0;
    o.max_num_spse_iterations = options.max_num_spse_iterations;
    o.use_spse_initialization = options.use_spse_initialization ? 1 : 0;
    o.spse_tolerance = options.spse_tolerance;
    handle_ = ceres_hip_create(&o);
    if (!handle_) throw std::runtime_error(std::string("ceres_hip_create: ") + ceres_hip_last_error(nullptr));
  }
  ~HipLinearSolver() override { ceres_hip_destroy(handle_); }
  HipLinearSolver(const HipLinearSolver&) = delete;
  HipLinearSolver& operator=(const HipLinearSolver&) = delete;

  // LevenbergMarquardtStrategy::ComputeStep (internal/ceres/levenberg_marquardt_strategy.cc:69-157) and the model cost change of
  // TrustRegionMinimizer::ComputeTrustRegionStep (internal/ceres/trust_region_minimizer.cc:425-437) in one call:
  // diag = clamp(SquaredColumnNorm(J)) unless reuse_diagonal, D = sqrt(diag / radius), Solve at q_tolerance = eta,
  // finite check, step = -x, model_cost_change = -(J step)'(f + J step / 2).
  struct LmStep {
    Summary summary;                 // FAILURE also when the step is not finite
    double model_cost_change = 0.0;
    bool step_is_finite = false;
  };
  LmStep ComputeLmStep(BlockSparseMatrix* jacobian, const double* residuals, double radius, double eta, double* step,
                       bool reuse_diagonal = false, double min_diagonal = 1e-6, double max_diagonal = 1e32,
                       bool values_unchanged = false) {   // the retry after a rejected step: J and f are what the previous call received
    LmStep out;
    if (!EnsureStructure(jacobian, &out.summary)) return out;
    ceres_hip_lm_options o{radius, min_diagonal, max_diagonal, eta, reuse_diagonal ? 1 : 0, values_unchanged ? 1 : 0};
    ceres_hip_lm_result r{};
    const int rc = ceres_hip_lm_compute_step(handle_, jacobian->values(), residuals, &o, step, &r);
    out.summary.residual_norm = r.linear_solver.residual_norm;
    out.summary.num_iterations = r.linear_solver.num_iterations;
    out.summary.termination_type = rc == CERES_HIP_OK ? static_cast<LinearSolverTerminationType>(r.linear_solver.termination_type)
                                                      : LinearSolverTerminationType::FATAL_ERROR;
    out.summary.message = rc == CERES_HIP_OK ? r.linear_solver.message : ceres_hip_last_error(handle_);
    out.model_cost_change = r.model_cost_change;
    out.step_is_finite = r.step_is_finite != 0;
    return out;
  }

  // The upload hidden behind the evaluator (include/ceres_hip.h: ceres_hip_values_begin / _ready / _end): BeginValues before
  // ProgramEvaluator::Evaluate's parallel loop starts writing jacobian->values() and residuals; ValuesReady from any evaluator thread
  // when a run of row blocks is complete; EndValues(jacobian_scaling) after the loop (the UNSCALED values went up: Jacobi scaling happens
  // on the device); then ComputeLmStepOnStreamedValues.
  bool BeginValues(BlockSparseMatrix* jacobian, const double* residuals, Summary* summary = nullptr) {
    Summary local;
    if (!EnsureStructure(jacobian, summary ? summary : &local)) return false;
    return ceres_hip_values_begin(handle_, jacobian->values(), residuals) == CERES_HIP_OK;
  }
  bool ValuesReady(int first_row_block, int num_row_blocks) { return ceres_hip_values_ready(handle_, first_row_block, num_row_blocks) == CERES_HIP_OK; }
  bool EndValues(const double* column_scale = nullptr) { return ceres_hip_values_end(handle_, column_scale) == CERES_HIP_OK; }
  LmStep ComputeLmStepOnStreamedValues(double radius, double eta, double* step, bool reuse_diagonal = false, double min_diagonal = 1e-6,
                                       double max_diagonal = 1e32) {
    LmStep out;
    ceres_hip_lm_options o{radius, min_diagonal, max_diagonal, eta, reuse_diagonal ? 1 : 0, 1};
    ceres_hip_lm_result r{};
    const int rc = ceres_hip_lm_compute_step(handle_, nullptr, nullptr, &o, step, &r);
    out.summary.residual_norm = r.linear_solver.residual_norm;
    out.summary.num_iterations = r.linear_solver.num_iterations;
    out.summary.termination_type = rc == CERES_HIP_OK ? static_cast<LinearSolverTerminationType>(r.linear_solver.termination_type)
                                                      : LinearSolverTerminationType::FATAL_ERROR;
    out.summary.message = rc == CERES_HIP_OK ? r.linear_solver.message : ceres_hip_last_error(handle_);
    out.model_cost_change = r.model_cost_change;
    out.step_is_finite = r.step_is_finite != 0;
    return out;
  }

  Summary Solve(BlockSparseMatrix* A, const double* b, const PerSolveOptions& per_solve_options, double* x) override {
    Summary summary;
    if (!EnsureStructure(A, &summary)) return summary;
    ceres_hip_summary s{};
    const int rc = ceres_hip_solve(handle_, A->values(), b, per_solve_options.D, per_solve_options.q_tolerance,
                                   per_solve_options.r_tolerance, x, &s);
    summary.residual_norm = s.residual_norm;
    summary.num_iterations = s.num_iterations;
    summary.termination_type = rc == CERES_HIP_OK ? static_cast<LinearSolverTerminationType>(s.termination_type)
                                                  : LinearSolverTerminationType::FATAL_ERROR;
    summary.message = rc == CERES_HIP_OK ? s.message : ceres_hip_last_error(handle_);
    return summary;
  }
  ceres_hip_info info() const { ceres_hip_info i{}; ceres_hip_get_info(handle_, &i); return i; }
  ceres_hip_solver* handle() const { return handle_; }

 private:
  // Flattens the structure on first use: one instance sees one sparsity (internal/ceres/linear_solver.h:137-142).
  bool EnsureStructure(BlockSparseMatrix* A, Summary* summary) {
    if (!structure_set_) {
      const FlatBlockStructure flat_arrays = FlattenBlockStructure(*A->block_structure());   // (flatten_block_structure.h: the code INTEGRATION.md's adapter runs)
      const ceres_hip_block_structure flat = flat_arrays.view();
      if (ceres_hip_set_structure(handle_, &flat) != CERES_HIP_OK) {
        summary->termination_type = LinearSolverTerminationType::FATAL_ERROR;
        summary->message = ceres_hip_last_error(handle_);
        return false;
      }
      structure_set_ = true;
    }
    return true;
  }
  Options options_;
  ceres_hip_solver* handle_ = nullptr;
  bool structure_set_ = false;
};

inline std::unique_ptr<LinearSolver> LinearSolver::Create(const Options& options) {
  Options o = options;
  const int nelim = o.elimination_groups.empty() ? 0 : o.elimination_groups[0];
  if (o.type == ITERATIVE_SCHUR && nelim == 0) {
    o.type = CGNR;
    if (o.preconditioner_type == SCHUR_JACOBI) o.preconditioner_type = JACOBI;
  }
  return std::make_unique<HipLinearSolver>(o);
}

}  // namespace ceres_hip
#endif
