// C++ host-side mirror of the BAL front end of include/ceres_hip.h: what
// examples/bal_problem.{h,cc} (the reader), ceres::internal::Evaluator (Evaluate) and
// TrustRegionMinimizer::Minimize are to bundle_adjuster, over the C ABI.  Header-only, no Ceres.
#ifndef CERES_HIP_HOST_BAL_PROBLEM_H_
#define CERES_HIP_HOST_BAL_PROBLEM_H_

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "ceres_hip.h"
#include "hip_linear_solver.h"

namespace ceres_hip {

// examples/bal_problem.cc:75-135 — "cameras points observations", one "camera point x y" line per
// observation, then 9 doubles per camera and 3 per point.
struct BalData {
  int num_cameras = 0, num_points = 0;
  std::vector<int32_t> camera_index, point_index;
  std::vector<double> observations;  // 2 per observation
  std::vector<double> parameters;    // file order: cameras, then points

  static BalData Read(const std::string& filename) {
    BalData d;
    FILE* f = std::fopen(filename.c_str(), "r");
    if (!f) throw std::runtime_error("cannot open " + filename);
    long long n_obs = 0;
    bool ok = std::fscanf(f, "%d %d %lld", &d.num_cameras, &d.num_points, &n_obs) == 3 && n_obs >= 0;
    if (ok) {
      d.camera_index.resize(n_obs); d.point_index.resize(n_obs); d.observations.resize(2 * n_obs);
      for (long long i = 0; ok && i < n_obs; ++i)
        ok = std::fscanf(f, "%d %d %lf %lf", &d.camera_index[i], &d.point_index[i], &d.observations[2 * i],
                         &d.observations[2 * i + 1]) == 4;
      d.parameters.resize(9 * size_t(d.num_cameras) + 3 * size_t(d.num_points));
      for (size_t i = 0; ok && i < d.parameters.size(); ++i) ok = std::fscanf(f, "%lf", &d.parameters[i]) == 1;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error(filename + " is not a BAL file");
    return d;
  }
  // state of the reduced, Schur-ordered program: points first, then cameras
  std::vector<double> State() const {
    std::vector<double> x(parameters.begin() + 9 * size_t(num_cameras), parameters.end());
    x.insert(x.end(), parameters.begin(), parameters.begin() + 9 * size_t(num_cameras));
    return x;
  }
};

class HipBalProblem {
 public:
  HipBalProblem(const LinearSolver::Options& options, const BalData& d) {
    ceres_hip_options o{};
    o.solver_type = options.type;
    o.preconditioner_type = options.preconditioner_type;
    o.min_num_iterations = options.min_num_iterations;
    o.max_num_iterations = options.max_num_iterations;
    o.residual_reset_period = options.residual_reset_period;
    o.device = options.device;
    handle_ = ceres_hip_bal_create(&o, d.num_cameras, d.num_points, int64_t(d.camera_index.size()), d.camera_index.data(),
                                   d.point_index.data(), d.observations.data());
    if (!handle_) throw std::runtime_error(std::string("ceres_hip_bal_create: ") + ceres_hip_bal_last_error(nullptr));
    ceres_hip_bal_sizes(handle_, &num_parameters_, &num_residuals_, &num_jacobian_values_);
  }
  ~HipBalProblem() { ceres_hip_bal_destroy(handle_); }
  HipBalProblem(const HipBalProblem&) = delete;
  HipBalProblem& operator=(const HipBalProblem&) = delete;

  int NumParameters() const { return int(num_parameters_); }   // Evaluator::NumParameters, I/evaluator.h:151
  int NumResiduals() const { return int(num_residuals_); }     // Evaluator::NumResiduals, :158
  // Evaluator::Evaluate (I/evaluator.h:116-124); residuals / gradient / jacobian values may be null
  bool Evaluate(const double* state, double* cost, double* residuals, double* gradient, double* jacobian_values) {
    return ceres_hip_bal_evaluate(handle_, state, cost, residuals, gradient, jacobian_values) == CERES_HIP_OK;
  }
  // TrustRegionMinimizer::Minimize; state in/out
  ceres_hip_minimizer_summary Minimize(const ceres_hip_minimizer_options& options, double* state) {
    ceres_hip_minimizer_summary s{};
    if (ceres_hip_bal_minimize(handle_, &options, state, &s) != CERES_HIP_OK)
      throw std::runtime_error(std::string("ceres_hip_bal_minimize: ") + ceres_hip_bal_last_error(handle_));
    return s;
  }

 private:
  ceres_hip_bal* handle_ = nullptr;
  int64_t num_parameters_ = 0, num_residuals_ = 0, num_jacobian_values_ = 0;
};

}  // namespace ceres_hip
#endif
