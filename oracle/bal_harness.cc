// bal_harness.cc — the CALLER of the boundary, restated so that "LM steps" exist.
//
// TEST INFRASTRUCTURE ONLY (see ceres_oracle.h).  Restates, from the algorithms:
//   BAL text format                 examples/bal_problem.cc:75-135
//   Snavely reprojection residual   examples/snavely_reprojection_error.h:53-105
//   AngleAxisRotatePoint            include/ceres/rotation.h (Rodrigues + first-order branch at 0)
//   Jacobian layouts                I/block_jacobian_writer.cc:59-64,68-167, I/reorder_program.cc:278-360
//   LM strategy                     I/levenberg_marquardt_strategy.cc:69-177
//   trust-region loop               I/trust_region_minimizer.cc:68-137,246-461,781-847
// Derivatives come from forward-mode dual numbers with 12 partials (9 camera + 3
// point), which is what AutoDiffCostFunction<.., 2, 9, 3> evaluates with Jets.

#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

#include "ceres_oracle.h"

namespace {

struct Dual {
  double v;
  double d[12];
  Dual() : v(0) { std::memset(d, 0, sizeof(d)); }
  explicit Dual(double x) : v(x) { std::memset(d, 0, sizeof(d)); }
  Dual(double x, int k) : v(x) { std::memset(d, 0, sizeof(d)); d[k] = 1.0; }
};
inline Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
inline Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
inline Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < 12; ++i) r.d[i] = -a.d[i]; return r; }
inline Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
inline Dual operator/(const Dual& a, const Dual& b) {
  Dual r; const double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int i = 0; i < 12; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
inline Dual operator+(const Dual& a, double b) { Dual r = a; r.v += b; return r; }
inline Dual operator+(double a, const Dual& b) { return b + a; }
inline Dual operator-(double a, const Dual& b) { Dual r = -b; r.v += a; return r; }
inline Dual operator*(const Dual& a, double b) { Dual r; r.v = a.v * b; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * b; return r; }
inline Dual dsin(const Dual& a) { Dual r; r.v = std::sin(a.v); const double c = std::cos(a.v); for (int i = 0; i < 12; ++i) r.d[i] = c * a.d[i]; return r; }
inline Dual dcos(const Dual& a) { Dual r; r.v = std::cos(a.v); const double s = -std::sin(a.v); for (int i = 0; i < 12; ++i) r.d[i] = s * a.d[i]; return r; }
inline Dual dsqrt(const Dual& a) { Dual r; r.v = std::sqrt(a.v); const double h = 0.5 / r.v; for (int i = 0; i < 12; ++i) r.d[i] = h * a.d[i]; return r; }

inline double val(double x) { return x; }
inline double val(const Dual& x) { return x.v; }
inline double tsin(double x) { return std::sin(x); }
inline double tcos(double x) { return std::cos(x); }
inline double tsqrt(double x) { return std::sqrt(x); }
inline Dual tsin(const Dual& x) { return dsin(x); }
inline Dual tcos(const Dual& x) { return dcos(x); }
inline Dual tsqrt(const Dual& x) { return dsqrt(x); }

template <typename T>
void angle_axis_rotate(const T aa[3], const T pt[3], T out[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (val(theta2) != 0.0) {
    const T theta = tsqrt(theta2);
    const T ct = tcos(theta), st = tsin(theta);
    const T inv = T(1.0) / theta;
    const T w[3] = {aa[0] * inv, aa[1] * inv, aa[2] * inv};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - ct);
    for (int i = 0; i < 3; ++i) out[i] = pt[i] * ct + wxp[i] * st + w[i] * tmp;
  } else {
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    for (int i = 0; i < 3; ++i) out[i] = pt[i] + wxp[i];
  }
}

template <typename T>
void snavely(const T* cam, const T* pt, double ox, double oy, T res[2]) {
  T p[3];
  angle_axis_rotate(cam, pt, p);
  p[0] = p[0] + cam[3]; p[1] = p[1] + cam[4]; p[2] = p[2] + cam[5];
  const T xp = -p[0] / p[2];
  const T yp = -p[1] / p[2];
  const T r2 = xp * xp + yp * yp;
  const T distortion = T(1.0) + r2 * (cam[7] + cam[8] * r2);
  res[0] = cam[6] * distortion * xp - T(ox);
  res[1] = cam[6] * distortion * yp - T(oy);
}

}  // namespace

struct oracle_bal {
  int nc = 0, np = 0;
  int64_t no = 0;
  std::vector<int32_t> cam_idx, pt_idx;
  std::vector<double> obs, params;  // params: 9*nc then 3*np (file order)
  // current layout
  int schur = -1;
  std::vector<int32_t> row_obs;      // row block -> observation
  std::vector<int32_t> cam_block, pt_block;  // column block ids
  std::vector<int32_t> rsz, rpos, csz, cpos, rptr, ccol, cval;
  int nelim = 0;
};

extern "C" {

// The evaluator's building blocks, exported so that tests can pin them to what the reference's own tests expect of
// the functions they restate (internal/ceres/rotation_test.cc:1809-1900: AngleAxisRotatePoint == the product with
// AngleAxisToRotationMatrix, also for |angle_axis| ~ 1e-16 and exactly 0).
void oracle_angle_axis_rotate_points(int64_t n, const double* angle_axis, const double* pts, double* out) {
  for (int64_t i = 0; i < n; ++i) angle_axis_rotate<double>(angle_axis + 3 * i, pts + 3 * i, out + 3 * i);
}
// residual (2) and its Jacobian (2 x 9 camera | 2 x 3 point, row-major) of SnavelyReprojectionError for n (camera, point,
// observation) triples, by forward-mode duals of the same formula (examples/snavely_reprojection_error.h:53-105)
void oracle_snavely_batch(int64_t n, const double* cams, const double* pts, const double* obs, double* residuals, double* jac_cam,
                          double* jac_pt) {
  for (int64_t i = 0; i < n; ++i) {
    Dual c[9], q[3], r[2];
    for (int k = 0; k < 9; ++k) c[k] = Dual(cams[9 * i + k], k);
    for (int k = 0; k < 3; ++k) q[k] = Dual(pts[3 * i + k], 9 + k);
    snavely<Dual>(c, q, obs[2 * i], obs[2 * i + 1], r);
    for (int a = 0; a < 2; ++a) {
      residuals[2 * i + a] = r[a].v;
      if (jac_cam) for (int k = 0; k < 9; ++k) jac_cam[18 * i + 9 * a + k] = r[a].d[k];
      if (jac_pt) for (int k = 0; k < 3; ++k) jac_pt[6 * i + 3 * a + k] = r[a].d[9 + k];
    }
  }
}

int oracle_bal_num_cameras(const oracle_bal* p) { return p->nc; }
int oracle_bal_num_points(const oracle_bal* p) { return p->np; }
int64_t oracle_bal_num_observations(const oracle_bal* p) { return p->no; }
double* oracle_bal_parameters(oracle_bal* p) { return p->params.data(); }
const int32_t* oracle_bal_camera_index(const oracle_bal* p) { return p->cam_idx.data(); }
const int32_t* oracle_bal_point_index(const oracle_bal* p) { return p->pt_idx.data(); }
const double* oracle_bal_observations(const oracle_bal* p) { return p->obs.data(); }
void oracle_bal_destroy(oracle_bal* p) { delete p; }

oracle_bal* oracle_bal_read(const char* filename) {
  FILE* f = std::fopen(filename, "r");
  if (!f) return nullptr;
  auto* p = new oracle_bal;
  long long no = 0;
  if (std::fscanf(f, "%d %d %lld", &p->nc, &p->np, &no) != 3) { std::fclose(f); delete p; return nullptr; }
  p->no = no;
  p->cam_idx.resize(no); p->pt_idx.resize(no); p->obs.resize(2 * no);
  for (int64_t i = 0; i < no; ++i)
    if (std::fscanf(f, "%d %d %lf %lf", &p->cam_idx[i], &p->pt_idx[i], &p->obs[2 * i], &p->obs[2 * i + 1]) != 4) { std::fclose(f); delete p; return nullptr; }
  p->params.resize(9 * size_t(p->nc) + 3 * size_t(p->np));
  for (double& v : p->params)
    if (std::fscanf(f, "%lf", &v) != 1) { std::fclose(f); delete p; return nullptr; }
  std::fclose(f);
  return p;
}

int oracle_bal_write(const oracle_bal* p, const char* filename) {
  FILE* f = std::fopen(filename, "w");
  if (!f) return 1;
  std::fprintf(f, "%d %d %lld\n", p->nc, p->np, (long long)p->no);
  for (int64_t i = 0; i < p->no; ++i)
    std::fprintf(f, "%d %d %.16e %.16e\n", p->cam_idx[i], p->pt_idx[i], p->obs[2 * i], p->obs[2 * i + 1]);
  for (double v : p->params) std::fprintf(f, "%.16e\n", v);
  std::fclose(f);
  return 0;
}

// Synthetic BAL-shaped scene (SURVEY.md §8d "Synthetic inputs"): same block
// counts as a named dataset, realistic conditioning because the Jacobian comes
// from real projective geometry rather than N(0,1) noise.
oracle_bal* oracle_bal_generate(int nc, int np, int64_t no_target, double skew, double pixel_noise,
                                double param_noise, uint64_t seed) {
  auto* p = new oracle_bal;
  p->nc = nc; p->np = np;
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> gauss(0.0, 1.0);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::vector<double> truth(9 * size_t(nc) + 3 * size_t(np));
  const double kPi = 3.14159265358979323846;
  for (int c = 0; c < nc; ++c) {
    const double ang = 2 * kPi * (c + 0.25 * uni(rng)) / nc, rad = 8.0 + 4.0 * uni(rng);
    const double C[3] = {rad * std::cos(ang), rad * std::sin(ang), 1.5 * gauss(rng)};
    double zc[3], n = std::sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
    for (int i = 0; i < 3; ++i) zc[i] = C[i] / n;  // camera looks down -z
    const double up[3] = {0.05 * gauss(rng), 0.05 * gauss(rng), 1.0};
    double xc[3] = {up[1] * zc[2] - up[2] * zc[1], up[2] * zc[0] - up[0] * zc[2], up[0] * zc[1] - up[1] * zc[0]};
    n = std::sqrt(xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2]);
    for (int i = 0; i < 3; ++i) xc[i] /= n;
    const double yc[3] = {zc[1] * xc[2] - zc[2] * xc[1], zc[2] * xc[0] - zc[0] * xc[2], zc[0] * xc[1] - zc[1] * xc[0]};
    const double R[9] = {xc[0], xc[1], xc[2], yc[0], yc[1], yc[2], zc[0], zc[1], zc[2]};
    // rotation matrix -> angle axis
    const double tr = R[0] + R[4] + R[8];
    double ct = std::max(-1.0, std::min(1.0, (tr - 1.0) / 2.0));
    const double theta = std::acos(ct);
    double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double* cam = &truth[9 * size_t(c)];
    for (int i = 0; i < 3; ++i) cam[i] = an > 1e-12 ? ax[i] / an * theta : 0.0;
    for (int i = 0; i < 3; ++i) cam[3 + i] = -(R[3 * i] * C[0] + R[3 * i + 1] * C[1] + R[3 * i + 2] * C[2]);
    cam[6] = 800.0 + 400.0 * uni(rng);
    cam[7] = 2e-2 * gauss(rng);
    cam[8] = 1e-3 * gauss(rng);
  }
  for (int q = 0; q < np; ++q) {
    double* pt = &truth[9 * size_t(nc) + 3 * size_t(q)];
    pt[0] = 1.5 * gauss(rng); pt[1] = 1.5 * gauss(rng); pt[2] = 0.8 * gauss(rng);
  }
  // Track lengths: >= 2, mean no_target/np, adjusted to hit no_target exactly.
  const int kmax = nc;
  std::vector<int> k(np, std::min(2, kmax));
  int64_t total = int64_t(np) * std::min(2, kmax);
  const double mean_extra = std::max(0.0, double(no_target) / np - 2.0);
  if (mean_extra > 0) {
    std::geometric_distribution<int> geo(1.0 / (1.0 + mean_extra));
    for (int q = 0; q < np; ++q) { const int e = std::min(geo(rng), kmax - k[q]); k[q] += e; total += e; }
  }
  std::uniform_int_distribution<int> pick(0, np - 1);
  while (total < no_target) { const int q = pick(rng); if (k[q] < kmax) { ++k[q]; ++total; } }
  while (total > no_target) { const int q = pick(rng); if (k[q] > 2) { --k[q]; --total; } else if (int64_t(np) * 2 >= no_target) break; }
  p->no = total;
  // Camera popularity.
  std::vector<double> cdf(nc);
  double acc = 0;
  for (int c = 0; c < nc; ++c) { acc += std::pow(double(c + 1), -skew); cdf[c] = acc; }
  p->cam_idx.reserve(total); p->pt_idx.reserve(total); p->obs.reserve(2 * total);
  std::vector<char> used(nc, 0);
  std::vector<int> chosen;
  for (int q = 0; q < np; ++q) {
    chosen.clear();
    int tries = 0;
    while (int(chosen.size()) < k[q]) {
      int c;
      if (tries++ < 64 * k[q]) {
        c = int(std::lower_bound(cdf.begin(), cdf.end(), uni(rng) * acc) - cdf.begin());
        c = std::min(c, nc - 1);
      } else {  // dense fallback: first unused camera after a random start
        c = int(uni(rng) * nc) % nc;
        while (used[c]) c = (c + 1) % nc;
      }
      if (used[c]) continue;
      used[c] = 1;
      chosen.push_back(c);
    }
    std::sort(chosen.begin(), chosen.end());
    for (int c : chosen) {
      used[c] = 0;
      double r[2];
      snavely<double>(&truth[9 * size_t(c)], &truth[9 * size_t(nc) + 3 * size_t(q)], 0.0, 0.0, r);
      p->cam_idx.push_back(c); p->pt_idx.push_back(q);
      p->obs.push_back(r[0] + pixel_noise * gauss(rng));
      p->obs.push_back(r[1] + pixel_noise * gauss(rng));
    }
  }
  p->params = truth;
  for (int c = 0; c < nc; ++c) {
    double* cam = &p->params[9 * size_t(c)];
    for (int i = 0; i < 3; ++i) cam[i] += 0.02 * param_noise * gauss(rng);
    for (int i = 3; i < 6; ++i) cam[i] += 0.2 * param_noise * gauss(rng);
    cam[6] *= 1.0 + 0.02 * param_noise * gauss(rng);
  }
  for (size_t i = 9 * size_t(nc); i < p->params.size(); ++i) p->params[i] += param_noise * gauss(rng);
  return p;
}

int oracle_bal_build_structure(oracle_bal* p, int schur, oracle_block_structure* out) {
  const int64_t no = p->no;
  p->schur = schur;
  p->row_obs.resize(no);
  std::iota(p->row_obs.begin(), p->row_obs.end(), 0);
  p->cam_block.assign(p->nc, -1);
  p->pt_block.assign(p->np, -1);
  const int ncb = p->nc + p->np;
  p->csz.assign(ncb, 0); p->cpos.assign(ncb, 0);
  if (schur) {
    // Points are elimination group 0 and come first; rows are grouped by point
    // (stable in observation order).  I/reorder_program.cc:278-360.
    std::stable_sort(p->row_obs.begin(), p->row_obs.end(),
                     [&](int32_t a, int32_t b) { return p->pt_idx[a] < p->pt_idx[b]; });
    for (int q = 0; q < p->np; ++q) { p->pt_block[q] = q; p->csz[q] = 3; }
    for (int c = 0; c < p->nc; ++c) { p->cam_block[c] = p->np + c; p->csz[p->np + c] = 9; }
    p->nelim = p->np;
  } else {
    // Program order = order of first use in AddResidualBlock(cost, loss, camera, point).
    int next = 0;
    for (int64_t i = 0; i < no; ++i) {
      if (p->cam_block[p->cam_idx[i]] < 0) { p->cam_block[p->cam_idx[i]] = next; p->csz[next++] = 9; }
      if (p->pt_block[p->pt_idx[i]] < 0) { p->pt_block[p->pt_idx[i]] = next; p->csz[next++] = 3; }
    }
    for (int c = 0; c < p->nc; ++c) if (p->cam_block[c] < 0) { p->cam_block[c] = next; p->csz[next++] = 9; }
    for (int q = 0; q < p->np; ++q) if (p->pt_block[q] < 0) { p->pt_block[q] = next; p->csz[next++] = 3; }
    p->nelim = 0;
  }
  for (int j = 1; j < ncb; ++j) p->cpos[j] = p->cpos[j - 1] + p->csz[j - 1];
  p->rsz.assign(no, 2); p->rpos.resize(no); p->rptr.resize(no + 1);
  p->ccol.resize(2 * no); p->cval.resize(2 * no);
  for (int64_t r = 0; r < no; ++r) {
    const int o = p->row_obs[r];
    p->rpos[r] = int32_t(2 * r);
    p->rptr[r] = int32_t(2 * r);
    const int cb = p->cam_block[p->cam_idx[o]], pb = p->pt_block[p->pt_idx[o]];
    if (schur) {  // E cells first, then F cells: I/block_jacobian_writer.cc:141-162
      p->ccol[2 * r] = pb; p->cval[2 * r] = int32_t(6 * r);
      p->ccol[2 * r + 1] = cb; p->cval[2 * r + 1] = int32_t(6 * no + 18 * r);
    } else {  // cells sorted by column block, 24 values per row stored row-sequentially: :59-64
      const int64_t base = 24 * r;
      if (cb < pb) { p->ccol[2 * r] = cb; p->cval[2 * r] = int32_t(base); p->ccol[2 * r + 1] = pb; p->cval[2 * r + 1] = int32_t(base + 18); }
      else { p->ccol[2 * r] = pb; p->cval[2 * r] = int32_t(base); p->ccol[2 * r + 1] = cb; p->cval[2 * r + 1] = int32_t(base + 6); }
    }
  }
  p->rptr[no] = int32_t(2 * no);
  out->num_row_blocks = int32_t(no);
  out->num_col_blocks = ncb;
  out->row_block_size = p->rsz.data(); out->row_block_pos = p->rpos.data();
  out->col_block_size = p->csz.data(); out->col_block_pos = p->cpos.data();
  out->row_cell_ptr = p->rptr.data(); out->cell_col_block = p->ccol.data(); out->cell_value_pos = p->cval.data();
  return p->nelim;
}

void oracle_bal_get_state(const oracle_bal* p, double* state) {
  for (int c = 0; c < p->nc; ++c) std::memcpy(state + p->cpos[p->cam_block[c]], &p->params[9 * size_t(c)], 72);
  for (int q = 0; q < p->np; ++q) std::memcpy(state + p->cpos[p->pt_block[q]], &p->params[9 * size_t(p->nc) + 3 * size_t(q)], 24);
}
void oracle_bal_set_state(oracle_bal* p, const double* state) {
  for (int c = 0; c < p->nc; ++c) std::memcpy(&p->params[9 * size_t(c)], state + p->cpos[p->cam_block[c]], 72);
  for (int q = 0; q < p->np; ++q) std::memcpy(&p->params[9 * size_t(p->nc) + 3 * size_t(q)], state + p->cpos[p->pt_block[q]], 24);
}

double oracle_bal_evaluate(const oracle_bal* p, const double* state, double* residuals, double* values) {
  double cost = 0;
  const int64_t no = p->no;
#pragma omp parallel for num_threads(oracle_get_num_threads()) reduction(+ : cost) schedule(static)
  for (int64_t r = 0; r < no; ++r) {
    const int o = p->row_obs[r];
    const int cb = p->cam_block[p->cam_idx[o]], pb = p->pt_block[p->pt_idx[o]];
    const double* cam = state + p->cpos[cb];
    const double* pt = state + p->cpos[pb];
    double res[2];
    if (values) {
      Dual c[9], q[3], rr[2];
      for (int i = 0; i < 9; ++i) c[i] = Dual(cam[i], i);
      for (int i = 0; i < 3; ++i) q[i] = Dual(pt[i], 9 + i);
      snavely<Dual>(c, q, p->obs[2 * o], p->obs[2 * o + 1], rr);
      res[0] = rr[0].v; res[1] = rr[1].v;
      const bool cam_first = p->ccol[2 * r] == cb;
      double* fv = values + p->cval[2 * r + (cam_first ? 0 : 1)];
      double* ev = values + p->cval[2 * r + (cam_first ? 1 : 0)];
      for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 9; ++j) fv[i * 9 + j] = rr[i].d[j];
        for (int j = 0; j < 3; ++j) ev[i * 3 + j] = rr[i].d[9 + j];
      }
    } else {
      snavely<double>(cam, pt, p->obs[2 * o], p->obs[2 * o + 1], res);
    }
    if (residuals) { residuals[2 * r] = res[0]; residuals[2 * r + 1] = res[1]; }
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
  }
  return cost;
}

void oracle_lm_default_options(oracle_lm_options* o) {
  // include/ceres/solver.h defaults
  o->max_num_iterations = 50;
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->min_relative_decrease = 1e-3;
  o->eta = 1e-1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->jacobi_scaling = 1;
  o->max_consecutive_invalid_steps = 5;
}

void oracle_lm_solve(oracle_bal* p, const oracle_block_structure* bs, const oracle_lm_options* o,
                     oracle_linear_solve_fn solve, void* ctx, oracle_lm_summary* S) {
  using clock = std::chrono::steady_clock;
  const auto t_start = clock::now();
  std::memset(S, 0, sizeof(*S));
  oracle_matrix* m = oracle_matrix_create(bs, p->nelim);
  const int n = oracle_matrix_num_cols(m), nr = oracle_matrix_num_rows(m);
  const int64_t nnz = oracle_matrix_num_nonzeros(m);
  std::vector<double> x(n), cand(n), res(nr), J(nnz), grad(n), scale(n, 1.0), diag(n), lmd(n), step(n), delta(n),
      model(nr);
  oracle_bal_get_state(p, x.data());
  double radius = o->initial_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int invalid_run = 0;
  bool one_success = false;

  auto log_iter = [&](const oracle_lm_iteration& it) { if (S->num_iterations_logged < 256) S->iterations[S->num_iterations_logged++] = it; };
  double x_cost = 0, grad_max = 0;
  int iteration = 0;
  // EvaluateGradientAndJacobian, I/trust_region_minimizer.cc:246-314
  auto eval_jacobian = [&]() {
    x_cost = oracle_bal_evaluate(p, x.data(), res.data(), J.data());
    std::fill(grad.begin(), grad.end(), 0.0);
    oracle_left_multiply(m, J.data(), res.data(), grad.data());
    if (o->jacobi_scaling) {
      if (iteration == 0) {
        oracle_squared_column_norm(m, J.data(), scale.data());
        for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i]));
      }
      oracle_scale_columns(m, J.data(), scale.data());
    }
    grad_max = 0;
    for (int i = 0; i < n; ++i) grad_max = std::max(grad_max, std::fabs(grad[i]));
  };
  eval_jacobian();
  S->initial_cost = x_cost;
  S->termination = 1;  // NO_CONVERGENCE
  oracle_lm_iteration it0{};
  it0.cost = x_cost; it0.gradient_max_norm = grad_max; it0.radius = radius; it0.step_is_valid = 1; it0.step_is_successful = 1;
  log_iter(it0);

  auto finish = [&](int term, const char* msg) {
    S->termination = term;
    std::snprintf(S->message, sizeof(S->message), "%s", msg);
  };
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue, :320-362
    if (iteration >= o->max_num_iterations) { finish(1, "Maximum number of iterations reached."); break; }
    if (grad_max <= o->gradient_tolerance) { finish(0, "Gradient tolerance reached."); break; }
    if (radius <= o->min_radius) { finish(0, "Minimum trust region radius reached."); break; }
    ++iteration;
    oracle_lm_iteration it{};
    // LevenbergMarquardtStrategy::ComputeStep, I/levenberg_marquardt_strategy.cc:69-157
    if (!reuse_diagonal) {
      oracle_squared_column_norm(m, J.data(), diag.data());
      for (int i = 0; i < n; ++i) diag[i] = std::min(std::max(diag[i], o->min_lm_diagonal), o->max_lm_diagonal);
    }
    for (int i = 0; i < n; ++i) lmd[i] = std::sqrt(diag[i] / radius);
    std::fill(step.begin(), step.end(), std::numeric_limits<double>::quiet_NaN());  // InvalidateArray
    oracle_summary ls{};
    const auto t0 = clock::now();
    const int rc = solve(ctx, J.data(), res.data(), lmd.data(), o->eta, -1.0, step.data(), &ls);
    S->linear_solver_seconds += std::chrono::duration<double>(clock::now() - t0).count();
    ++S->num_linear_solves;
    if (rc != 0) ls.termination_type = 3;
    if (ls.termination_type != 3 && ls.termination_type != 2) {
      bool finite = true;
      for (int i = 0; i < n; ++i) if (!std::isfinite(step[i])) { finite = false; break; }
      if (!finite) ls.termination_type = 2; else for (int i = 0; i < n; ++i) step[i] = -step[i];
    }
    reuse_diagonal = true;
    it.linear_solver_iterations = ls.num_iterations;
    it.linear_termination = ls.termination_type;
    if (ls.termination_type == 3) { finish(2, "Linear solver failed due to unrecoverable non-numeric causes."); break; }
    // ComputeTrustRegionStep, I/trust_region_minimizer.cc:381-461
    double model_cost_change = 0;
    bool valid = false;
    if (ls.termination_type != 2) {
      std::fill(model.begin(), model.end(), 0.0);
      oracle_right_multiply(m, J.data(), step.data(), model.data());
      for (int i = 0; i < nr; ++i) model_cost_change -= model[i] * (res[i] + model[i] / 2.0);
      valid = model_cost_change > 0.0;
    }
    it.step_is_valid = valid;
    if (!valid) {  // HandleInvalidStep :466-497
      if (++invalid_run >= o->max_consecutive_invalid_steps) { finish(2, "Too many consecutive invalid steps."); break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;  // StepIsInvalid == StepRejected(0)
      it.cost = x_cost; it.gradient_max_norm = grad_max; it.radius = radius;
      ++S->num_unsuccessful_steps;
      log_iter(it);
      continue;
    }
    invalid_run = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    for (int i = 0; i < n; ++i) cand[i] = x[i] + delta[i];
    const double cand_cost = oracle_bal_evaluate(p, cand.data(), nullptr, nullptr);
    double xn = 0, dn = 0;
    for (int i = 0; i < n; ++i) { xn += x[i] * x[i]; dn += delta[i] * delta[i]; }
    it.step_norm = std::sqrt(dn);
    if (one_success && it.step_norm <= o->parameter_tolerance * (std::sqrt(xn) + o->parameter_tolerance)) {
      finish(0, "Parameter tolerance reached."); it.cost = x_cost; it.radius = radius; log_iter(it); break;
    }
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= o->function_tolerance * x_cost) {
      finish(0, "Function tolerance reached."); it.cost = x_cost; it.radius = radius; log_iter(it); break;
    }
    it.relative_decrease = (x_cost - cand_cost) / model_cost_change;  // TrustRegionStepEvaluator, monotonic
    if (it.relative_decrease > o->min_relative_decrease) {  // HandleSuccessfulStep :829-845
      x = cand;
      one_success = true;
      eval_jacobian();
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(o->max_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      it.step_is_successful = 1;
      ++S->num_successful_steps;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      ++S->num_unsuccessful_steps;
    }
    it.cost = it.step_is_successful ? x_cost : cand_cost;
    it.gradient_max_norm = grad_max;
    it.radius = radius;
    log_iter(it);
  }
  S->final_cost = x_cost;
  oracle_bal_set_state(p, x.data());
  oracle_matrix_destroy(m);
  S->total_seconds = std::chrono::duration<double>(clock::now() - t_start).count();
}

}  // extern "C"
