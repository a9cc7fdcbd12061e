// the evaluator side of the boundary for bundle adjustment in BAL form.
//
// What the reference does per LM iteration outside LinearSolver::Solve (ProgramEvaluator,
// I/program_evaluator.h:137-300, with autodiff Jets of examples/snavely_reprojection_error.h:
// 53-105 and BlockJacobianWriter, I/block_jacobian_writer.cc:68-167) is, for this one cost
// function, an embarrassingly parallel map over the observations: one thread per observation
// evaluates the Snavely reprojection residual and its ANALYTIC 2x9 / 2x3 Jacobian and writes
// both straight into the BlockSparseMatrix value layout the solver reads (E cells, then F cells:
// I/block_jacobian_writer.cc:141-162), already Jacobi-scaled (TrustRegionMinimizer,
// I/trust_region_minimizer.cc:263-279) so that no separate ScaleColumns pass exists.
// HBM-bound by its 208 B/observation of output; the arithmetic (~400 flop) is register resident.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device.h"
#include "snavely.h"

namespace chip {

namespace {

__device__ __forceinline__ double wave_sum_e(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_max_e(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}

template <bool JAC>
__global__ __launch_bounds__(kVecBlock) void bal_evaluate_kernel(BalEvalArgs A) {
  __shared__ double sh[4];
  double cost = 0.0;
  for (int64_t r = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; r < A.n_rows; r += int64_t(gridDim.x) * kVecBlock) {
    const int c = A.row_cam[r], p = A.row_pt[r];
    const double2 o = A.row_obs[r];
    double cam[9], X[3], res[2], jc[18], jp[6];
    const double* cs = A.state + A.cam_base + 9 * int64_t(c);
    const double* ps = A.state + 3 * int64_t(p);
#pragma unroll
    for (int i = 0; i < 9; ++i) cam[i] = cs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) X[i] = ps[i];
    snavely<JAC>(cam, X, o.x, o.y, res, jc, jp);
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    if (A.residuals) reinterpret_cast<double2*>(A.residuals)[r] = make_double2(res[0], res[1]);
    if constexpr (JAC) {
      if (A.scale) {
        const double* sc = A.scale + A.cam_base + 9 * int64_t(c);
        const double* sp = A.scale + 3 * int64_t(p);
#pragma unroll
        for (int j = 0; j < 9; ++j) { const double v = sc[j]; jc[j] *= v; jc[9 + j] *= v; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { const double v = sp[j]; jp[j] *= v; jp[3 + j] *= v; }
      }
      double2* e = reinterpret_cast<double2*>(A.values + 6 * r);                // 48 B per row, 16-byte aligned
      double2* fo = reinterpret_cast<double2*>(A.values + 6 * A.n_rows + 18 * r);  // 144 B per row
#pragma unroll
      for (int j = 0; j < 3; ++j) e[j] = make_double2(jp[2 * j], jp[2 * j + 1]);
#pragma unroll
      for (int j = 0; j < 9; ++j) fo[j] = make_double2(jc[2 * j], jc[2 * j + 1]);
    }
  }
  cost = wave_sum_e(cost);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cost;
  __syncthreads();
  if (threadIdx.x == 0) A.partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// The same evaluation in TILE order (round 4): one wavefront per tile of 64 observation slots, the Jacobian goes straight
// into the layout the solver's passes read — J[tile][12][64] double2 (E pairs 0-2, F pairs 3-11), b[tile][64] double2, zeros in
// the padding slots — with 1 KiB-contiguous stores per wave instruction, so the step has no re-layout pass and no caller-layout
// E cells at all.  The F cells ALSO go to the caller layout (the camera-major preconditioner pass reads 144-byte cells there, where a
// cell is contiguous): the wave's 64 cells are staged in LDS and written 16 bytes per lane in cell order, so that a wave
// instruction covers whole cells of consecutive rows (a point's rows are consecutive: runs of a few hundred bytes) instead of 64
// separate 16-byte pieces 144 bytes apart — the pattern that holds bal_evaluate_kernel<true> at 2.5 TB/s.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(double2* p, double x, double y) {
  const double2 v = make_double2(x, y);
  v4i raw;
  __builtin_memcpy(&raw, &v, 16);
  __builtin_nontemporal_store(raw, reinterpret_cast<v4i*>(p));
}

// {state, scale} of every parameter block as ONE 16-byte-aligned record — [3 + 3] doubles per point, [9 + 9] per camera — so that a lane
// of the tile-order evaluator fetches its camera with 9 and its point with 3 16-byte loads instead of 18 + 6 8-byte ones: lanes of a
// tile see 64 different cameras, every load instruction costs the texture path one line per lane, and the instruction count is what
// that kernel's time follows (bal_evaluate_kernel<false>: 12 such gathers per observation, 87 us on the Venice shape = 64 lines x 12 per
// 64 observations at one line per clock and CU).
__global__ __launch_bounds__(kVecBlock) void bal_pack_state_kernel(const double* state, const double* scale, int64_t n_points, int64_t n_cameras,
                                                                   double* pt_pack, double* cam_pack) {
  // one thread per 16-byte PAIR of the output (3 per point, 9 per camera): coalesced stores, reads of neighbouring words
  const int64_t e = int64_t(blockIdx.x) * kVecBlock + threadIdx.x, n_pt_pairs = 3 * n_points;
  if (e < n_pt_pairs) {
    const int64_t i = e / 3;
    const int k = int(e - 3 * i);   // pairs of [x0 x1 x2 s0 s1 s2]
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int j = 2 * k + h; v[h] = j < 3 ? state[3 * i + j] : (scale ? scale[3 * i + j - 3] : 1.0); }
    reinterpret_cast<double2*>(pt_pack)[e] = make_double2(v[0], v[1]);
  } else if (e < n_pt_pairs + 9 * n_cameras) {
    const int64_t q = e - n_pt_pairs, c = q / 9, o = 3 * n_points + 9 * c;
    const int k = int(q - 9 * c);   // pairs of [state 9 | scale 9]
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int j = 2 * k + h; v[h] = j < 9 ? state[o + j] : (scale ? scale[o + j - 9] : 1.0); }
    reinterpret_cast<double2*>(cam_pack)[q] = make_double2(v[0], v[1]);
  }
}

// Software pipeline, two stages deep: a tile's index words are loaded two iterations ahead, its parameter records one iteration ahead,
// and nothing a wave waits for is ever queued BEHIND a store — the vector memory counter retires in order, so a load issued after tile
// N's 23 store instructions cannot be consumed before every one of them is acknowledged (the first version did exactly that: its
// time was compute + tile stores + F copy + residuals, 135 + 190 + 230 + 40 us on the Venice shape, nothing overlapped;
// profiles/r04w_eval_tiles_store_groups_unpipelined.jsonl).  The loads are unconditional (clamped tile index, padding slots read record 0): the
// wait counts stay static.
struct TileIdx { int bp, fp, cam, pt; double2 obs; };
struct TileRec { double2 c[9]; double2 p[3]; };
__device__ __forceinline__ void issue_tile_idx(const BalEvalTilesArgs& T, int64_t tile, int lane, TileIdx& x) {
  const int64_t sl = tile * 64 + lane;
  x.bp = T.slot_bpos[sl]; x.fp = T.slot_fpos[sl]; x.cam = T.slot_cam[sl]; x.pt = T.slot_pt[sl]; x.obs = T.slot_obs[sl];
}
__device__ __forceinline__ void issue_tile_rec(const BalEvalTilesArgs& T, const TileIdx& x, TileRec& r) {
  const double2* cp = reinterpret_cast<const double2*>(T.cam_pack + 18 * int64_t(x.cam));
  const double2* pp = reinterpret_cast<const double2*>(T.pt_pack + 6 * int64_t(x.pt));
#pragma unroll
  for (int i = 0; i < 9; ++i) r.c[i] = cp[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) r.p[i] = pp[i];
}

template <int DBG, bool F_COPY>   // DBG: BalEvalTilesArgs::debug_flags, compile-time (a run-time flag's branches cost the static wait counts)
__global__ __launch_bounds__(kVecBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void bal_evaluate_tiles_kernel(BalEvalTilesArgs T) {
  static_assert(kVecBlock == 256, "four waves per workgroup");
  __shared__ double sh[4];
  __shared__ double2 cells[4][64 * 9];   // a wave's 64 F cells, cell-major (9 pairs each)
  const BalEvalArgs& A = T.e;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = int64_t(gridDim.x) * 4, last = T.n_tiles - 1;
  double cost = 0.0;
  int64_t tile = int64_t(blockIdx.x) * 4 + wave;
  TileIdx ix, ix1;
  TileRec rec;
  if (tile < T.n_tiles) {
    issue_tile_idx(T, tile, lane, ix);
    issue_tile_idx(T, tile + stride < last ? tile + stride : last, lane, ix1);
    issue_tile_rec(T, ix, rec);
    // everything of the prologue has arrived before the loop is entered: the loop header then inherits the back edge's state (23 stores
    // in flight, no load) and not the prologue's — the compiler merges the two to the smaller count, which made every iteration wait
    // for ten of the previous tile's stores
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  }
  for (; tile < T.n_tiles; tile += stride) {
    TileRec rec1;
    TileIdx ix2;
    issue_tile_rec(T, ix1, rec1);                                                           // tile + stride: its index words arrived an iteration ago
    issue_tile_idx(T, tile + 2 * stride < last ? tile + 2 * stride : last, lane, ix2);
    const int64_t sl = tile * 64 + lane;
    const int bp = ix.bp;                      // 2 x row, -1: padding
    const int fp = bp >= 0 ? ix.fp : -1;
    double res[2] = {0.0, 0.0}, jc[18], jp[6];
    {
      const double cam9[9] = {rec.c[0].x, rec.c[0].y, rec.c[1].x, rec.c[1].y, rec.c[2].x, rec.c[2].y, rec.c[3].x, rec.c[3].y, rec.c[4].x};
      const double X3[3] = {rec.p[0].x, rec.p[0].y, rec.p[1].x};
      snavely<true>(cam9, X3, ix.obs.x, ix.obs.y, res, jc, jp);
      const double sc[9] = {rec.c[4].y, rec.c[5].x, rec.c[5].y, rec.c[6].x, rec.c[6].y, rec.c[7].x, rec.c[7].y, rec.c[8].x, rec.c[8].y};
#pragma unroll
      for (int j = 0; j < 9; ++j) { jc[j] *= sc[j]; jc[9 + j] *= sc[j]; }
      jp[0] *= rec.p[1].y; jp[3] *= rec.p[1].y; jp[1] *= rec.p[2].x; jp[4] *= rec.p[2].x; jp[2] *= rec.p[2].y; jp[5] *= rec.p[2].y;
    }
    if (bp < 0) {   // padding: zeros in the tiles (the record it evaluated is somebody's, the values are dropped)
      res[0] = res[1] = 0.0;
#pragma unroll
      for (int j = 0; j < 18; ++j) jc[j] = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) jp[j] = 0.0;
    }
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    // (padding lanes store their zeros into their own slot of the b tile — which holds zeros — instead of skipping the store: a store
    // under a lane predicate may be branched over, and the compiler then counts on fewer stores in flight, so that its wait for the
    // NEXT tile's index words would cover ten of this tile's stores.  Not one shared dummy line: non-temporal stores of thousands of
    // waves to the same kilobyte serialise — 1.07 ms instead of 0.5.)
    if constexpr (!(DBG & 4)) *(bp >= 0 ? reinterpret_cast<double2*>(A.residuals) + (bp >> 1) : T.b_out + sl) = make_double2(res[0], res[1]);
    double2* o = T.J_out + tile * T.tile_pitch + lane;
    if constexpr (!(DBG & 2)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) nt_store(o + j * 64, jp[2 * j], jp[2 * j + 1]);
#pragma unroll
      for (int j = 0; j < 9; ++j) nt_store(o + (3 + j) * 64, jc[2 * j], jc[2 * j + 1]);
    }
    if constexpr (!(DBG & 4)) nt_store(T.b_out + sl, res[0], res[1]);
    if constexpr (F_COPY && !(DBG & 1)) {
#pragma unroll
      for (int j = 0; j < 9; ++j) cells[wave][lane * 9 + j] = make_double2(jc[2 * j], jc[2 * j + 1]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int q = k * 64 + lane, slot = q / 9, piece = q - 9 * slot;
        const int f = __shfl(fp, slot, 64);
        const double2 v = cells[wave][q];
        // (non-temporal: the camera-major pass reads these 720 MB three kernels later; 609 -> 485 us in the unpipelined kernel)
        double2* const dst = f >= 0 ? reinterpret_cast<double2*>(A.values + f) + piece : T.b_out + (tile * 64 + slot);
        if constexpr ((DBG & 16) != 0) *dst = v;
        else nt_store(dst, v.x, v.y);
      }
      __builtin_amdgcn_wave_barrier();
    }
    ix = ix1; ix1 = ix2; rec = rec1;
  }
  cost = wave_sum_e(cost);
  if (lane == 0) sh[wave] = cost;
  __syncthreads();
  if (threadIdx.x == 0) A.partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// delta = step .* scale ; candidate = x + delta ; partials: |x|^2 at [b], |delta|^2 at [grid + b]
__global__ __launch_bounds__(kVecBlock) void bal_candidate_kernel(const double* x, const double* step, const double* scale,
                                                                  double* delta, double* cand, int64_t n, double* partials) {
  __shared__ double sh[8];
  double xn = 0, dn = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock) {
    const double d = scale ? step[i] * scale[i] : step[i];
    const double xi = x[i];
    delta[i] = d;
    cand[i] = xi + d;
    xn += xi * xi;
    dn += d * d;
  }
  xn = wave_sum_e(xn); dn = wave_sum_e(dn);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = xn; sh[4 + (threadIdx.x >> 6)] = dn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    partials[gridDim.x + blockIdx.x] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

// max_i |g_i / scale_i|: the gradient of the UNSCALED problem from J^T f of the scaled Jacobian
__global__ __launch_bounds__(kVecBlock) void bal_gradient_max_kernel(const double* g, const double* scale, int64_t n, double* partials) {
  __shared__ double sh[4];
  double m = 0;
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock)
    m = fmax(m, fabs(scale ? g[i] / scale[i] : g[i]));
  m = wave_max_e(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// scale = 1 / (1 + sqrt(squared column norm))      I/trust_region_minimizer.cc:263-270
__global__ __launch_bounds__(kVecBlock) void bal_jacobi_scale_kernel(const double* colnorm2, double* scale, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * kVecBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVecBlock)
    scale[i] = 1.0 / (1.0 + sqrt(colnorm2[i]));
}

int grid_for(int64_t n) {
  int64_t g = (n + kVecBlock - 1) / kVecBlock;
  return int(g < 1 ? 1 : (g > kMaxVecGrid ? kMaxVecGrid : g));
}

}  // namespace

hipError_t LaunchBalEvaluate(const BalEvalArgs& A, bool jacobian, int* nparts, hipStream_t stream) {
  // one observation per thread and several per thread at scale: 2048 workgroups keep 256 CUs busy
  int64_t g = (A.n_rows + kVecBlock - 1) / kVecBlock;
  const int grid = int(g < 1 ? 1 : (g > 2048 ? 2048 : g));
  *nparts = grid;
  if (jacobian) hipLaunchKernelGGL(bal_evaluate_kernel<true>, dim3(grid), dim3(kVecBlock), 0, stream, A);
  else hipLaunchKernelGGL(bal_evaluate_kernel<false>, dim3(grid), dim3(kVecBlock), 0, stream, A);
  return hipGetLastError();
}

hipError_t LaunchBalEvaluateTiles(const BalEvalTilesArgs& T, int64_t n_points, int64_t n_cameras, int* nparts, hipStream_t stream) {
  hipLaunchKernelGGL(bal_pack_state_kernel, dim3(unsigned((3 * n_points + 9 * n_cameras + kVecBlock - 1) / kVecBlock)), dim3(kVecBlock), 0, stream,
                     T.e.state, T.e.scale, n_points, n_cameras, T.pt_pack, T.cam_pack);
  const int64_t g = (T.n_tiles + 3) / 4;
  const int grid = int(g < 1 ? 1 : (g > 2048 ? 2048 : g));
  *nparts = grid;
#define EVAL_TILES_CASE(D) case D: hipLaunchKernelGGL((bal_evaluate_tiles_kernel<D, true>), dim3(grid), dim3(kVecBlock), 0, stream, T); break;
  if (!T.e.values) {
    hipLaunchKernelGGL((bal_evaluate_tiles_kernel<0, false>), dim3(grid), dim3(kVecBlock), 0, stream, T);
  } else {
    switch (T.debug_flags) {
      EVAL_TILES_CASE(0) EVAL_TILES_CASE(1) EVAL_TILES_CASE(2) EVAL_TILES_CASE(3) EVAL_TILES_CASE(4) EVAL_TILES_CASE(7) EVAL_TILES_CASE(16)
      default: return hipErrorInvalidValue;
    }
  }
#undef EVAL_TILES_CASE
  return hipGetLastError();
}

hipError_t LaunchBalCandidate(const double* x, const double* step, const double* scale, double* delta, double* cand, int64_t n,
                              double* partials, int* nparts, hipStream_t stream) {
  const int grid = grid_for(n);
  *nparts = grid;
  hipLaunchKernelGGL(bal_candidate_kernel, dim3(grid), dim3(kVecBlock), 0, stream, x, step, scale, delta, cand, n, partials);
  return hipGetLastError();
}

hipError_t LaunchBalGradientMax(const double* g, const double* scale, int64_t n, double* partials, int* nparts, hipStream_t stream) {
  const int grid = grid_for(n);
  *nparts = grid;
  hipLaunchKernelGGL(bal_gradient_max_kernel, dim3(grid), dim3(kVecBlock), 0, stream, g, scale, n, partials);
  return hipGetLastError();
}

hipError_t LaunchBalJacobiScale(const double* colnorm2, double* scale, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(bal_jacobi_scale_kernel, dim3(grid_for(n)), dim3(kVecBlock), 0, stream, colnorm2, scale, n);
  return hipGetLastError();
}

}  // namespace chip
