// host_driver.cc — exercises the C++ host mirror (hip_linear_solver.h) end to end on a GPU:
// the reference's LinearLeastSquaresProblem2 (internal/ceres/linear_least_squares_problems.cc:301-439)
// and a small BAL-shaped <2,3,9> problem, both solved with ITERATIVE_SCHUR and CGNR and
// checked against a dense solve of the regularised normal equations (the pattern of
// internal/ceres/iterative_schur_complement_solver_test.cc:75-117).  Prints one line per
// case and exits non-zero on any mismatch; then the whole Levenberg-Marquardt step (ComputeLmStep) on the BAL-shaped problem
// against dense algebra.  Built by build.py with g++ against the C ABI.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "hip_linear_solver.h"
#include "hip_bal_problem.h"

using namespace ceres_hip;

namespace {

// Row-major dense image of A.
std::vector<double> Dense(const BlockSparseMatrix& A) {
  const int m = A.num_rows(), n = A.num_cols();
  std::vector<double> dense(size_t(m) * n, 0.0);
  const auto* bs = A.block_structure();
  for (const auto& r : bs->rows)
    for (const auto& c : r.cells) {
      const Block& cb = bs->cols[c.block_id];
      for (int i = 0; i < r.block.size; ++i)
        for (int j = 0; j < cb.size; ++j)
          dense[size_t(r.block.position + i) * n + cb.position + j] = A.values()[c.position + i * cb.size + j];
    }
  return dense;
}

// Dense reference: solve (A^T A + D^2) x = A^T b by Gaussian elimination with pivoting.
std::vector<double> DenseSolve(const BlockSparseMatrix& A, const std::vector<double>& b, const std::vector<double>& D) {
  const int m = A.num_rows(), n = A.num_cols();
  const std::vector<double> dense = Dense(A);
  std::vector<double> H(size_t(n) * n, 0.0), g(n, 0.0);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < m; ++k) s += dense[size_t(k) * n + i] * dense[size_t(k) * n + j];
      H[size_t(i) * n + j] = s;
    }
    H[size_t(i) * n + i] += D[i] * D[i];
    for (int k = 0; k < m; ++k) g[i] += dense[size_t(k) * n + i] * b[k];
  }
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (std::fabs(H[size_t(r) * n + c]) > std::fabs(H[size_t(p) * n + c])) p = r;
    for (int j = 0; j < n; ++j) std::swap(H[size_t(c) * n + j], H[size_t(p) * n + j]);
    std::swap(g[c], g[p]);
    for (int r = c + 1; r < n; ++r) {
      const double f = H[size_t(r) * n + c] / H[size_t(c) * n + c];
      for (int j = c; j < n; ++j) H[size_t(r) * n + j] -= f * H[size_t(c) * n + j];
      g[r] -= f * g[c];
    }
  }
  std::vector<double> x(n);
  for (int i = n - 1; i >= 0; --i) {
    double s = g[i];
    for (int j = i + 1; j < n; ++j) s -= H[size_t(i) * n + j] * x[j];
    x[i] = s / H[size_t(i) * n + i];
  }
  return x;
}

std::unique_ptr<BlockSparseMatrix> Problem2(std::vector<double>* b, std::vector<double>* D, int* nelim) {
  auto* bs = new CompressedRowBlockStructure;
  for (int c = 0; c < 5; ++c) bs->cols.emplace_back(1, c);
  const int cells[6][3] = {{0, 2, -1}, {0, 3, -1}, {1, 4, -1}, {1, 2, -1}, {1, 2, -1}, {2, 3, 4}};
  const double vals[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 1, 1, 1, 1};
  int nnz = 0;
  for (int r = 0; r < 6; ++r) {
    bs->rows.emplace_back();
    bs->rows.back().block = Block(1, r);
    for (int k = 0; k < 3 && cells[r][k] >= 0; ++k) bs->rows.back().cells.emplace_back(cells[r][k], nnz++);
  }
  auto A = std::make_unique<BlockSparseMatrix>(bs);
  for (int i = 0; i < nnz; ++i) A->mutable_values()[i] = vals[i];
  b->assign({0, 1, 2, 3, 4, 5});
  D->assign(5, 1.0);
  *nelim = 2;
  return A;
}

std::unique_ptr<BlockSparseMatrix> SmallBal(int n_cams, int n_pts, std::vector<double>* b, std::vector<double>* D, int* nelim) {
  std::mt19937_64 rng(38401);
  std::normal_distribution<double> gauss;
  auto* bs = new CompressedRowBlockStructure;
  for (int p = 0; p < n_pts; ++p) bs->cols.emplace_back(3, 3 * p);
  for (int c = 0; c < n_cams; ++c) bs->cols.emplace_back(9, 3 * n_pts + 9 * c);
  std::vector<std::pair<int, int>> obs;
  for (int p = 0; p < n_pts; ++p) {
    const int k = 2 + int(rng() % 3), start = int(rng() % n_cams);
    for (int j = 0; j < k; ++j) obs.emplace_back(p, (start + j) % n_cams);
  }
  const int n_obs = int(obs.size());
  for (int r = 0; r < n_obs; ++r) {
    bs->rows.emplace_back();
    bs->rows.back().block = Block(2, 2 * r);
    bs->rows.back().cells.emplace_back(obs[r].first, 6 * r);                       // E cells first
    bs->rows.back().cells.emplace_back(n_pts + obs[r].second, 6 * n_obs + 18 * r);  // then F cells
  }
  auto A = std::make_unique<BlockSparseMatrix>(bs);
  for (int i = 0; i < A->num_nonzeros(); ++i) A->mutable_values()[i] = gauss(rng);
  b->resize(A->num_rows());
  for (auto& v : *b) v = gauss(rng);
  D->assign(A->num_cols(), 0.3);
  *nelim = n_pts;
  return A;
}

int RunCase(const char* name, BlockSparseMatrix* A, const std::vector<double>& b, const std::vector<double>& D, int nelim,
            LinearSolverType type, PreconditionerType pre, double tol) {
  LinearSolver::Options o;
  o.type = type;
  o.preconditioner_type = pre;
  o.min_num_iterations = 0;
  o.max_num_iterations = 4 * A->num_cols();
  o.elimination_groups = {type == CGNR ? 0 : nelim};
  auto solver = LinearSolver::Create(o);
  LinearSolver::PerSolveOptions ps;
  ps.D = D.data();
  ps.r_tolerance = 1e-13;
  ps.q_tolerance = -1.0;  // the zeta test at q_tolerance = 0 fires on rounding noise near convergence: terminate on |r| only
  std::vector<double> x(A->num_cols(), std::nan(""));
  const auto s = solver->Solve(A, b.data(), ps, x.data());
  const auto ref = DenseSolve(*A, b, D);
  double num = 0, den = 0;
  for (size_t i = 0; i < x.size(); ++i) { num += (x[i] - ref[i]) * (x[i] - ref[i]); den += ref[i] * ref[i]; }
  const double err = std::sqrt(num / den);
  const bool ok = s.termination_type == LinearSolverTerminationType::SUCCESS && err <= tol;
  std::printf("%s %-28s type=%d pre=%d iterations=%d rel_err=%.2e (%s)\n", ok ? "PASS" : "FAIL", name, int(type), int(pre),
              s.num_iterations, err, s.message.c_str());
  return ok ? 0 : 1;
}

// LevenbergMarquardtStrategy::ComputeStep + the model cost change through the C++ mirror, against dense algebra:
// D = sqrt(clamp(diag(J'J)) / radius), step = -(J'J + D^2)^-1 J'f, model cost change = -(J step)'(f + J step / 2).
int RunLmStep(const char* name, BlockSparseMatrix* A, const std::vector<double>& f, int nelim, LinearSolverType type,
              PreconditionerType pre, double tol) {
  LinearSolver::Options o;
  o.type = type;
  o.preconditioner_type = pre;
  o.min_num_iterations = 0;
  o.max_num_iterations = 4 * A->num_cols();
  o.elimination_groups = {type == CGNR ? 0 : nelim};
  HipLinearSolver solver(o);
  const double radius = 1e4;
  const int n = A->num_cols(), m = A->num_rows();
  const std::vector<double> dense = Dense(*A);
  std::vector<double> D(n);
  for (int j = 0; j < n; ++j) {
    double d = 0;
    for (int i = 0; i < m; ++i) d += dense[size_t(i) * n + j] * dense[size_t(i) * n + j];
    D[j] = std::sqrt(std::min(std::max(d, 1e-6), 1e32) / radius);
  }
  std::vector<double> ref = DenseSolve(*A, f, D);
  for (auto& v : ref) v = -v;
  double ref_cost = 0;
  for (int i = 0; i < m; ++i) {
    double mi = 0;
    for (int j = 0; j < n; ++j) mi += dense[size_t(i) * n + j] * ref[j];
    ref_cost -= mi * (f[i] + mi / 2);
  }
  std::vector<double> step(n, std::nan(""));
  const auto r = solver.ComputeLmStep(A, f.data(), radius, /*eta=*/1e-13, step.data());
  double num = 0, den = 0;
  for (int j = 0; j < n; ++j) { num += (step[j] - ref[j]) * (step[j] - ref[j]); den += ref[j] * ref[j]; }
  const double err = std::sqrt(num / den), cost_err = std::fabs(r.model_cost_change - ref_cost) / std::fabs(ref_cost);
  const bool ok = r.summary.termination_type == LinearSolverTerminationType::SUCCESS && r.step_is_finite && err <= tol && cost_err <= tol;
  std::printf("%s %-28s type=%d pre=%d iterations=%d step_rel_err=%.2e model_cost_rel_err=%.2e (%s)\n", ok ? "PASS" : "FAIL", name,
              int(type), int(pre), r.summary.num_iterations, err, cost_err, r.summary.message.c_str());
  return ok ? 0 : 1;
}

// The same step with the Jacobian STREAMED up by eight "evaluator" threads (ceres_hip_values_begin / _ready / _end): every thread
// writes its runs of row blocks into the host arrays the solver was given — the E cells and the F cells of the rows, and their
// residuals, as BlockJacobianWriter lays them out — and announces each run; the values go up UNSCALED and EndValues applies a column
// scaling on the device.  Checked against ComputeLmStep on the host-scaled Jacobian (same solver type): the two must agree to rounding.
int RunStreamedLmStep(const char* name, BlockSparseMatrix* A, const std::vector<double>& f, int nelim, LinearSolverType type,
                      PreconditionerType pre, bool with_scale) {
  LinearSolver::Options o;
  o.type = type;
  o.preconditioner_type = pre;
  o.min_num_iterations = 0;
  o.max_num_iterations = 4 * A->num_cols();
  o.elimination_groups = {type == CGNR ? 0 : nelim};
  const int n = A->num_cols();
  const auto* bs = A->block_structure();
  const int nrb = int(bs->rows.size());
  std::vector<double> scale(n, 1.0);
  if (with_scale) for (int j = 0; j < n; ++j) scale[j] = 1.0 / (1.0 + 0.01 * (j % 17));
  // reference: the scaled Jacobian through the plain entry point
  const std::vector<double> unscaled(A->values(), A->values() + A->num_nonzeros());
  std::vector<double> scaled = unscaled;
  for (const auto& r : bs->rows)
    for (const auto& c : r.cells) {
      const Block& cb = bs->cols[c.block_id];
      for (int i = 0; i < r.block.size; ++i)
        for (int j = 0; j < cb.size; ++j) scaled[c.position + i * cb.size + j] *= scale[cb.position + j];
    }
  std::vector<double> ref(n, std::nan("")), step(n, std::nan(""));
  double ref_cost = 0;
  {
    for (int i = 0; i < A->num_nonzeros(); ++i) A->mutable_values()[i] = scaled[i];
    HipLinearSolver solver(o);
    const auto r = solver.ComputeLmStep(A, f.data(), 1e4, 1e-13, ref.data());
    ref_cost = r.model_cost_change;
    if (r.summary.termination_type != LinearSolverTerminationType::SUCCESS) { std::printf("FAIL %s: reference step: %s\n", name, r.summary.message.c_str()); return 1; }
  }
  // streamed: the host arrays start as garbage and are filled by the threads
  for (int i = 0; i < A->num_nonzeros(); ++i) A->mutable_values()[i] = std::nan("");
  std::vector<double> fh(f.size(), std::nan(""));
  HipLinearSolver solver(o);
  if (!solver.BeginValues(A, fh.data())) { std::printf("FAIL %s: BeginValues: %s\n", name, ceres_hip_last_error(solver.handle())); return 1; }
  const int kThreads = 8, kRun = 5;
  std::atomic<int> next{0}, failed{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < kThreads; ++t)
    pool.emplace_back([&, t] {
      for (;;) {
        const int r0 = next.fetch_add(kRun);
        if (r0 >= nrb) break;
        const int r1 = std::min(nrb, r0 + kRun);
        for (int r = r0; r < r1; ++r) {
          const auto& row = bs->rows[r];
          for (const auto& c : row.cells) {
            const int len = row.block.size * bs->cols[c.block_id].size;
            for (int i = 0; i < len; ++i) A->mutable_values()[c.position + i] = unscaled[c.position + i];
          }
          for (int i = 0; i < row.block.size; ++i) fh[row.block.position + i] = f[row.block.position + i];
        }
        if (t % 3 == 2 && r1 - r0 > 1) {   // some threads announce row by row, the last row first
          for (int r = r1 - 1; r >= r0; --r) if (!solver.ValuesReady(r, 1)) failed = 1;
        } else if (t % 3 == 1 && r0 % (2 * kRun) == 0) {
          // ... and some runs are never announced: EndValues must send them
        } else if (!solver.ValuesReady(r0, r1 - r0)) failed = 1;
      }
    });
  for (auto& th : pool) th.join();
  if (failed || !solver.EndValues(with_scale ? scale.data() : nullptr)) { std::printf("FAIL %s: streaming: %s\n", name, ceres_hip_last_error(solver.handle())); return 1; }
  const auto r = solver.ComputeLmStepOnStreamedValues(1e4, 1e-13, step.data());
  int64_t early = 0, late = 0; int32_t streams = 0;
  ceres_hip_get_stream_stats(solver.handle(), &early, &late, &streams);
  for (int i = 0; i < A->num_nonzeros(); ++i) A->mutable_values()[i] = unscaled[i];   // leave A as it was
  double num = 0, den = 0;
  for (int j = 0; j < n; ++j) { num += (step[j] - ref[j]) * (step[j] - ref[j]); den += ref[j] * ref[j]; }
  const double err = std::sqrt(num / den), cost_err = std::fabs(r.model_cost_change - ref_cost) / std::fabs(ref_cost);
  const bool ok = r.summary.termination_type == LinearSolverTerminationType::SUCCESS && r.step_is_finite && err <= 1e-12 && cost_err <= 1e-12 && streams > 0 && early > 0;
  std::printf("%s %-28s type=%d scale=%d value_streams=%d bytes_early=%lld bytes_late=%lld step_rel_diff=%.2e model_cost_rel_diff=%.2e (%s)\n", ok ? "PASS" : "FAIL", name,
              int(type), int(with_scale), int(streams), (long long)early, (long long)late, err, cost_err, r.summary.message.c_str());
  return ok ? 0 : 1;
}

}  // namespace

// host_driver <problem.txt> [max_num_iterations]: BALProblem + Evaluator + TrustRegionMinimizer through the C++ mirror
int RunBalFile(const char* filename, int max_it) {
  const BalData data = BalData::Read(filename);
  LinearSolver::Options o;
  o.type = ITERATIVE_SCHUR;
  o.preconditioner_type = SCHUR_JACOBI;
  o.min_num_iterations = 0;
  o.max_num_iterations = 500;
  HipBalProblem problem(o, data);
  std::vector<double> x = data.State();
  double cost0 = 0;
  if (!problem.Evaluate(x.data(), &cost0, nullptr, nullptr, nullptr)) return 1;
  ceres_hip_minimizer_options mo;
  ceres_hip_minimizer_default_options(&mo);
  mo.max_num_iterations = max_it;
  const ceres_hip_minimizer_summary s = problem.Minimize(mo, x.data());
  double cost1 = 0;
  if (!problem.Evaluate(x.data(), &cost1, nullptr, nullptr, nullptr)) return 1;
  std::printf("bal parameters=%d residuals=%d initial_cost=%.17g evaluated_initial=%.17g final_cost=%.17g evaluated_final=%.17g "
              "successful=%d unsuccessful=%d termination=%d message=%s\n",
              problem.NumParameters(), problem.NumResiduals(), s.initial_cost, cost0, s.final_cost, cost1, s.num_successful_steps,
              s.num_unsuccessful_steps, s.termination_type, s.message);
  return 0;
}

int main(int argc, char** argv) {
  if (ceres_hip_device_count() < 1) {
    std::printf("FAIL no gfx950 device visible (the library has no CPU path)\n");
    return 2;
  }
  if (argc >= 2) return RunBalFile(argv[1], argc >= 3 ? std::atoi(argv[2]) : 10);
  int bad = 0, nelim = 0;
  std::vector<double> b, D;
  auto p2 = Problem2(&b, &D, &nelim);
  bad += RunCase("problem2", p2.get(), b, D, nelim, ITERATIVE_SCHUR, SCHUR_JACOBI, 1e-11);
  bad += RunCase("problem2", p2.get(), b, D, nelim, ITERATIVE_SCHUR, JACOBI, 1e-11);
  bad += RunCase("problem2", p2.get(), b, D, nelim, CGNR, JACOBI, 1e-9);
  auto bal = SmallBal(7, 60, &b, &D, &nelim);
  bad += RunCase("bal<2,3,9>", bal.get(), b, D, nelim, ITERATIVE_SCHUR, SCHUR_JACOBI, 1e-9);
  bad += RunCase("bal<2,3,9>", bal.get(), b, D, nelim, CGNR, JACOBI, 1e-7);
  // the step ends on the zeta test (q_tolerance = eta), not on a residual: 1e-6 on the step, the model cost is second order in it
  bad += RunLmStep("lm step bal<2,3,9>", bal.get(), b, nelim, ITERATIVE_SCHUR, SCHUR_JACOBI, 1e-6);
  bad += RunLmStep("lm step bal<2,3,9>", bal.get(), b, nelim, CGNR, JACOBI, 1e-6);
  // the Jacobian streamed up by eight evaluator threads while "evaluation" is still going on, unscaled, Jacobi scaling on the device
  auto big = SmallBal(23, 900, &b, &D, &nelim);
  bad += RunStreamedLmStep("streamed lm step bal<2,3,9>", big.get(), b, nelim, ITERATIVE_SCHUR, SCHUR_JACOBI, false);
  bad += RunStreamedLmStep("streamed lm step bal<2,3,9>", big.get(), b, nelim, ITERATIVE_SCHUR, SCHUR_JACOBI, true);
  bad += RunStreamedLmStep("streamed lm step bal<2,3,9>", big.get(), b, nelim, CGNR, JACOBI, true);
  std::printf(bad ? "host_driver: %d case(s) FAILED\n" : "host_driver: all cases passed\n", bad);
  return bad ? 1 : 0;
}
