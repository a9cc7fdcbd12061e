// kernels_schur.hip — explicit Schur complement solvers, any block sizes up to kMaxGenericBlock.
//
//   SparseSchurComplementSolver with use_explicit_schur_complement (ITERATIVE_SCHUR on a stored S):
//     S lives in the block-sparse storage of BlockRandomAccessSparseMatrix (I/block_random_access_sparse_matrix.cc:51-110),
//     the block pairs SparseSchurComplementSolver::InitStorage derives from the chunks (I/schur_complement_solver.cc:224-290);
//     SchurEliminator::Eliminate (I/schur_eliminator_impl.h:184-311) fills it as a GATHER: one wavefront per stored block walks the
//     (chunk, cell, cell) contributions plan.cc listed for it — no mutex per cell as in the reference, no atomics, bit-reproducible;
//     CG applies it with SymmetricRightMultiplyAndAccumulate (:125-163): each stored (upper-triangle) block serves y_i += M x_j and
//     y_j += M^T x_i, here one thread per output scalar through the row list and the transposed list of its block.
//   DenseSchurComplementSolver (DENSE_SCHUR, I/schur_complement_solver.cc:163-222): dense S, Cholesky factorisation of its
//     upper triangle (DenseCholesky, I/dense_cholesky.cc), two triangular solves — a right-looking blocked factorisation with
//     32-wide panels (diagonal block in LDS, panel solve one thread per row, LDS-tiled symmetric rank-32 update).
#include <hip/hip_runtime.h>

#include <mutex>

#include "device.h"

namespace chip {
namespace {

constexpr int kB = 256;

__device__ __forceinline__ int find_pair(const int64_t* off, int n, int64_t e) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- Eliminate into the block-sparse lhs -------------------------------------------------------------------------
// One wavefront per work ITEM — at most kSchurItem consecutive triples of a stored block (i, j), i <= j (common.h) —, lane l owns
// entries l, l + 64, ... (a, b) = (entry / n_j, entry % n_j) of the block and leaves the item's partial block in `scratch`;
// schur_sparse_combine_kernel adds a block's items up in list order (bit-reproducible) and D_f^2 onto the diagonal of (i, i).
// (One wavefront per BLOCK, as this kernel first was, lasts as long as the longest list: the diagonal block of the most popular
// camera of a 456-camera / 500 k-observation problem sums 11 700 triples — 94 ms, against 9 ms for factoring the result.)
// Contribution of a triple (chunk e, cell k1 of block i in row r1, cell k2 of block j in row r2):
//     [r1 == r2] F1^T F2  -  (E_r1^T F1)^T (E^T E + D_e^2)^-1 (E_r2^T F2)
// and for an E-free row just F1^T F2.
__global__ __launch_bounds__(kB) void schur_sparse_eliminate_kernel(GenStructure G, SchurPairs P, const double* __restrict__ v,
                                                                    const double* __restrict__ ete_inv) {
  const int item = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
  if (item >= P.n_items) return;
  const int pair = P.item_pair[item];
  const int lane = threadIdx.x & 63;
  const int bi = P.pair_i[pair], bj = P.pair_j[pair];
  const int ji = G.nelim + bi, jj = G.nelim + bj;
  const int n1 = G.csz[ji], n2 = G.csz[jj];
  const int nent = n1 * n2;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // nent <= 256
  for (int64_t t = P.item_t0[item]; t < P.item_t1[item]; ++t) {
    const int e = P.trip_e[t], k1 = P.trip_k1[t], k2 = P.trip_k2[t];
    const int r1 = P.cell_row[k1], r2 = P.cell_row[k2];
    const int rs1 = G.rsz[r1], rs2 = G.rsz[r2];
    const double* f1 = v + G.cval[k1];
    const double* f2 = v + G.cval[k2];
    const double* E1 = nullptr;
    const double* E2 = nullptr;
    const double* inv = nullptr;
    int es = 0;
    if (e >= 0) {
      es = G.csz[e];
      E1 = v + G.cval[G.rptr[r1]];
      E2 = v + G.cval[G.rptr[r2]];
      inv = ete_inv + G.diag_off_e[e];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ent = lane + 64 * q;
      if (ent >= nent) break;
      const int a = ent / n2, b = ent - a * n2;
      double s = 0.0;
      if (r1 == r2)
        for (int r = 0; r < rs1; ++r) s += f1[r * n1 + a] * f2[r * n2 + b];
      if (e >= 0) {
        // u1 = E_r1^T F1[:, a], u2 = E_r2^T F2[:, b]; es <= 16
        double w = 0.0;
        for (int p = 0; p < es; ++p) {
          double u1 = 0.0;
          for (int r = 0; r < rs1; ++r) u1 += E1[r * es + p] * f1[r * n1 + a];
          double iv = 0.0;
          for (int c = 0; c < es; ++c) {
            double u2 = 0.0;
            for (int r = 0; r < rs2; ++r) u2 += E2[r * es + c] * f2[r * n2 + b];
            iv += inv[p * es + c] * u2;
          }
          w += u1 * iv;
        }
        s -= w;
      }
      acc[q] += s;
    }
  }
  double* out = P.scratch + P.item_off[item];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ent = lane + 64 * q;
    if (ent >= nent) break;
    out[ent] = acc[q];
  }
}
// S[value e] = sum of the items of its block, in list order (+ D_f^2 on the diagonal of a diagonal block).  One thread per value.
__global__ __launch_bounds__(kB) void schur_sparse_combine_kernel(GenStructure G, SchurPairs P, const double* __restrict__ D, int64_t total,
                                                                  double* __restrict__ S) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= total) return;
  const int pair = find_pair(P.pair_off, P.npairs, e);
  const int64_t ent = e - P.pair_off[pair];
  double s = 0.0;
  for (int it = P.pair_item_ptr[pair]; it < P.pair_item_ptr[pair + 1]; ++it) s += P.scratch[P.item_off[it] + ent];
  const int bi = P.pair_i[pair];
  if (D && bi == P.pair_j[pair]) {
    const int ji = G.nelim + bi, n = G.csz[ji];
    const int a = int(ent / n), b = int(ent - int64_t(a) * n);
    if (a == b) { const double d = D[G.cpos[ji] + a]; s += d * d; }
  }
  S[e] = s;
}

// y = S x with only the upper block triangle stored: one thread per scalar of y.
__global__ __launch_bounds__(kB) void schur_sparse_symv_kernel(GenStructure G, SchurPairs P, const double* __restrict__ S,
                                                               const double* __restrict__ x, double* __restrict__ y,
                                                               const int* __restrict__ status, int accumulate) {
  if (status && *status != 0) return;
  const int g = blockIdx.x * kB + threadIdx.x;
  if (g >= G.ncf) return;
  const int jb = G.col_block_of[G.nce + g];
  const int i = jb - G.nelim;
  const int a = G.nce + g - G.cpos[jb];
  const int ni = G.csz[jb];
  double s = 0.0;
  for (int p = P.row_ptr[i]; p < P.row_ptr[i + 1]; ++p) {  // cells (i, j >= i): row a of M
    const int jc = G.nelim + P.pair_j[p];
    const int nj = G.csz[jc];
    const double* M = S + P.pair_off[p] + int64_t(a) * nj;
    const double* xj = x + (G.cpos[jc] - G.nce);
    for (int b = 0; b < nj; ++b) s += M[b] * xj[b];
  }
  for (int q = P.col_ptr[i]; q < P.col_ptr[i + 1]; ++q) {  // cells (j < i, i): column a of M, i.e. row a of M^T
    const int p = P.col_pair[q];
    const int jc = G.nelim + P.pair_i[p];
    const int nj = G.csz[jc];
    const double* M = S + P.pair_off[p] + a;
    const double* xj = x + (G.cpos[jc] - G.nce);
    for (int b = 0; b < nj; ++b) s += M[int64_t(b) * ni] * xj[b];
  }
  if (accumulate) y[g] += s; else y[g] = s;
}

// blocks[i] = S(i, i): the first cell of block row i
__global__ __launch_bounds__(kB) void schur_sparse_diag_kernel(GenStructure G, SchurPairs P, const double* __restrict__ S,
                                                               const int64_t* __restrict__ diag_off_f, double* __restrict__ blocks) {
  const int nf = G.ncb - G.nelim;
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= diag_off_f[nf]) return;
  const int q = find_pair(diag_off_f, nf, e);
  blocks[e] = S[P.pair_off[P.row_ptr[q]] + (e - diag_off_f[q])];
}

// lhs (dense n x n row-major, zeroed) <- the stored blocks (i <= j) of the block-sparse S: what DENSE_SCHUR factors.  One thread per value.
__global__ __launch_bounds__(kB) void schur_blocks_to_dense_kernel(GenStructure G, SchurPairs P, const double* __restrict__ S, int64_t total,
                                                                   double* __restrict__ lhs) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= total) return;
  const int p = find_pair(P.pair_off, P.npairs, e);
  const int ji = G.nelim + P.pair_i[p], jj = G.nelim + P.pair_j[p];
  const int n2 = G.csz[jj];
  const int64_t r = e - P.pair_off[p];
  const int a = int(r / n2), b = int(r - int64_t(a) * n2);
  lhs[int64_t(G.cpos[ji] - G.nce + a) * G.ncf + (G.cpos[jj] - G.nce + b)] = S[e];
}

// ---- dense Cholesky (upper triangle authoritative, result L in the LOWER triangle, row-major n x n) --------------
constexpr int kNb = 32;

// Mirror the stored upper triangle into the lower one (the factorisation below works on the lower triangle).
__global__ __launch_bounds__(kB) void dense_mirror_upper_kernel(double* __restrict__ A, int n) {
  const int64_t e = int64_t(blockIdx.x) * kB + threadIdx.x;
  if (e >= int64_t(n) * n) return;
  const int r = int(e / n), c = int(e % n);
  if (r > c) A[e] = A[int64_t(c) * n + r];
}

// Right-looking blocked Cholesky with kPanel-wide panels (DenseCholesky::FactorAndSolve for DENSE_SCHUR's reduced system,
// I/schur_complement_solver.cc:163-222, I/dense_cholesky.cc).  Per panel three kernels:
//   dense_potrf_panel_kernel   the kw x kw diagonal block, in LDS, one workgroup: 16-wide steps — a 16 x 16 factorisation AND inversion
//                              by one wavefront on registers and shuffles, the rows below as X^T = L_ss^-1 A^T and the block's own
//                              trailing update as 16 x 16 MFMA tiles on LDS operands;
//   dense_trsm_mfma_kernel     rows below the block: X^T = L_kk^-1 A^T, one wavefront per 16 rows, a chain of MFMAs whose accumulators
//                              are the next products' operands (blocked substitution; only the 16 x 16 diagonal sub-blocks are inverted);
//   dense_syrk_mfma_kernel     the trailing update C -= X X^T, the n^3 / 3 of the factorisation: v_mfma_f64_16x16x4_f64, one wavefront
//                              per 64 x 64 output tile (16 accumulator tiles), K = the panel width.
// A rank-32 update (the first version) moves 16 B per 64 flops of every trailing entry: HBM-bound at a quarter of the fp64 rate; at
// K = 128 the update is compute-bound and sits on the matrix pipe.  gfx950's fp64 MFMA peak equals its fp64 vector peak (78.6 TFLOP/s:
// 32 flop / clk / SIMD); what MFMA buys is the operand traffic — 8 eight-byte loads per lane feed 16 instructions of 2048 flops.
constexpr int kPanel = 128;
constexpr int kPanelPitch = kPanel + 1;   // LDS row pitch (doubles): column accesses hit distinct banks
constexpr int kSub = 16;                  // step width inside the diagonal block
typedef double v4f64 __attribute__((ext_vector_type(4)));

// Where the inverse of the 16 x 16 diagonal sub-block s of a panel's diagonal block is parked for the kernels that multiply by it:
// a 16 x 16 sub-block strictly ABOVE the diagonal of the same kPanel x kPanel block (free: the factorisation lives in the lower triangle).
__device__ __forceinline__ int linv_block_row(int s) { return s < kPanel / kSub - 1 ? s : kPanel / kSub - 3; }
__device__ __forceinline__ int linv_block_col(int s) { return s < kPanel / kSub - 1 ? s + 1 : kPanel / kSub - 1; }

// One MFMA tile product on LDS / register operands, all in the layouts of v_mfma_f64_16x16x4_f64 (lane l: li = l & 15, lq = l >> 4):
// an A operand is A[li][4 kk + lq], a B operand B[4 kk + lq][li], an accumulator register r holds C[lq + 4 r][li] — so an accumulator IS
// the B operand of a following product (register r <-> k-step r): triangular solves chain without moving data.
__global__ __launch_bounds__(256) void dense_potrf_panel_kernel(double* __restrict__ A, int n, int k0, int kw, int* fail_flag) {
  extern __shared__ double T[];   // [kwp][kPanelPitch], lower triangle; kwp = kw rounded up to the step width, padded with the identity
  __shared__ double Li[kSub][kSub + 1];   // inverse of the current diagonal sub-block
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int kwp = (kw + kSub - 1) / kSub * kSub;
  {   // thread = column, two rows per sweep (coalesced, no division), 16 loads in flight per thread before the first is consumed
    const int c = tid & 127;
    for (int rb = tid >> 7; rb < kwp; rb += 32) {
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = rb + 2 * t;
        v[t] = (c <= r && r < kw) ? A[int64_t(k0 + r) * n + (k0 + c)] : (r == c ? 1.0 : 0.0);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = rb + 2 * t;
        if (c <= r && r < kwp) T[r * kPanelPitch + c] = v[t];
      }
    }
  }
  __syncthreads();
  for (int s0 = 0; s0 < kwp; s0 += kSub) {
    if (wv == 0) {
      // lane r < 16 owns row r of the 16 x 16 diagonal sub-block; column by column, the pivot row's entries travel by shuffle
      double row[kSub];
#pragma unroll
      for (int c = 0; c < kSub; ++c) row[c] = (lane < kSub && c <= lane) ? T[(s0 + lane) * kPanelPitch + s0 + c] : (c == lane ? 1.0 : 0.0);
      // The pivot chain is 16 dependent steps per sub-block, 128 per panel: sqrt and divide (long software sequences in fp64) are
      // replaced by ONE reciprocal square root per pivot — v_rsq_f64 refined by two Newton steps — and multiplications by it
      // (L_jj = d y, column j *= y, and the inverse below multiplies by the same y = 1 / L_jj): a few ulps from the divided form.
      bool ok = true;
      double ri[kSub];   // 1 / L[j][j]
#pragma unroll
      for (int j = 0; j < kSub; ++j) {
        double d = __shfl(row[j], j, 64);   // T[j][j]
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        double y = __builtin_amdgcn_rsq(d);
        y = y * (1.5 - 0.5 * d * y * y);
        y = y * (1.5 - 0.5 * d * y * y);
        ri[j] = y;
        if (lane == j) row[j] = d * y;
        else if (lane > j) row[j] *= y;
#pragma unroll
        for (int c = j + 1; c < kSub; ++c) {
          const double tcj = __shfl(row[j], c, 64);   // T[c][j]
          if (lane >= c) row[c] -= row[j] * tcj;
        }
      }
      if (!ok && lane == 0) atomicExch(fail_flag, 1);
#pragma unroll
      for (int c = 0; c < kSub; ++c) if (lane < kSub && c <= lane) T[(s0 + lane) * kPanelPitch + s0 + c] = row[c];
      // its inverse: lane j < 16 solves L x = e_j (column j), the entries of L travel by shuffle from the lanes that own their rows
      double x[kSub];
#pragma unroll
      for (int r = 0; r < kSub; ++r) {
        double v = (r == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) v -= __shfl(row[k], r, 64) * x[k];   // L[r][k]
        x[r] = v * ri[r];
      }
      if (lane < kSub) {
#pragma unroll
        for (int r = 0; r < kSub; ++r) Li[r][lane] = x[r];
        if (kw == kPanel) {   // the rows under this block are solved against it by dense_trsm_mfma_kernel
          const int s = s0 / kSub;
          double* dst = A + int64_t(k0 + kSub * linv_block_row(s)) * n + (k0 + kSub * linv_block_col(s)) + lane;
#pragma unroll
          for (int r = 0; r < kSub; ++r) dst[int64_t(r) * n] = x[r];
        }
      }
    }
    __syncthreads();
    const int below = kwp - s0 - kSub;   // rows of the block under the sub-block (a multiple of 16)
    const int nstrips = below / kSub;
    // strips of 16 rows under the sub-block: X^T = L_ss^-1 A^T, one MFMA chain per strip (4 waves take them in turn)
    for (int st = wv; st < nstrips; st += 4) {
      double* base = T + (s0 + kSub + kSub * st + li) * kPanelPitch + s0;   // row (strip row li), column s0 + ..
      v4f64 t = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[li][4 * kk + lq], base[4 * kk + lq], t, 0, 0, 0);
      // (B operand of k-step kk = A^T[4 kk + lq][li] = the strip's row li, column 4 kk + lq: read straight from its place)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) base[lq + 4 * r] = t[r];   // X^T[lq + 4 r][li] back to row li, column lq + 4 r
    }
    __syncthreads();
    // the block's own trailing update, lower triangle, in 16 x 16 tiles (tr, tc), tc <= tr: C -= X_tr X_tc^T
    const int n_tiles = nstrips * (nstrips + 1) / 2;
    for (int e = wv; e < n_tiles; e += 4) {
      int tr = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
      while (tr * (tr + 1) / 2 > e) --tr;
      while ((tr + 1) * (tr + 2) / 2 <= e) ++tr;
      const int tc = e - tr * (tr + 1) / 2;
      const double* xa = T + (s0 + kSub + kSub * tr + li) * kPanelPitch + s0;
      const double* xb = T + (s0 + kSub + kSub * tc + li) * kPanelPitch + s0;
      double* cp = T + (s0 + kSub + kSub * tr + lq) * kPanelPitch + s0 + kSub + kSub * tc + li;
      v4f64 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cp[4 * r * kPanelPitch];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * kk + lq], xb[4 * kk + lq], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) cp[4 * r * kPanelPitch] = c[r];   // (entries above the diagonal inside a diagonal tile are never read)
    }
    __syncthreads();
  }
  for (int r = tid >> 7; r < kw; r += 2) {
    const int c = tid & 127;
    if (c <= r) A[int64_t(k0 + r) * n + (k0 + c)] = T[r * kPanelPitch + c];
  }
}

// Rows i >= k0 + kPanel: A[i, k0 : k0 + kPanel] <- A[i, k0 : k0 + kPanel] L_kk^-T, in the transposed form X^T = L_kk^-1 A^T: one
// wavefront per strip of 16 rows, 16 columns at a time — R_s = A_s^T - sum_{s' < s} L[s][s'] X_s'^T (MFMA, the X_s'^T are this wave's
// own accumulators), X_s^T = L_ss^-1 R_s (MFMA, the inverted diagonal sub-blocks left by dense_potrf_panel_kernel).  144 MFMAs per strip.
__global__ __launch_bounds__(256) void dense_trsm_mfma_kernel(double* __restrict__ A, int n, int k0) {
  constexpr int kS = kPanel / kSub;
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int i0 = k0 + kPanel + kSub * (blockIdx.x * 4 + (threadIdx.x >> 6));
  if (i0 >= n) return;
  const bool live = i0 + li < n;
  double* arow = A + int64_t(min(i0 + li, n - 1)) * n + k0;
  const double* Lb = A + int64_t(k0) * n + k0;
  v4f64 X[kS];
#pragma unroll
  for (int s = 0; s < kS; ++s) {
    v4f64 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = arow[kSub * s + lq + 4 * r];
#pragma unroll
    for (int sp = 0; sp < s; ++sp) {
      const double* lp = Lb + int64_t(kSub * s + li) * n + kSub * sp + lq;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[4 * kk], X[sp][kk], acc, 0, 0, 0);
    }
    const double* ip = Lb + int64_t(kSub * linv_block_row(s) + li) * n + kSub * linv_block_col(s) + lq;
    v4f64 t = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ip[4 * kk], acc[kk], t, 0, 0, 0);
    X[s] = t;
  }
  if (live) {
#pragma unroll
    for (int s = 0; s < kS; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) arow[kSub * s + lq + 4 * r] = X[s][r];
  }
}

// The same substitution with L_kk (and the parked inverses of its diagonal sub-blocks) staged in LDS once per workgroup (round 4): in the
// kernel above every one of a strip's 144 MFMAs takes its A operand from memory — 144 dependent-looking 8-byte gathers with a row
// pitch of n doubles per wavefront, and 50 us per panel at n = 8190 although the MFMA chain itself is 2 us long; the panel chain
// (potrf -> trsm -> potrf -> trsm) is what the whole factorisation waits for (the bulk updates run beside it: LaunchDenseCholesky).
// Here a workgroup copies the 128 x 128 block (coalesced rows, 132 KB of LDS) and its wavefronts take strips in turn.
__global__ __launch_bounds__(256) void dense_trsm_lds_kernel(double* __restrict__ A, int n, int k0) {
  extern __shared__ double Ls[];   // [kPanel][kPanelPitch]: lower triangle = L_kk, the parked 16 x 16 inverses above the diagonal
  constexpr int kS = kPanel / kSub;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  {
    const double* Lb = A + int64_t(k0) * n + k0;
    const int c = tid & 127;
    for (int rb = tid >> 7; rb < kPanel; rb += 32) {   // thread = column, two rows per sweep, 16 loads in flight
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = Lb[int64_t(rb + 2 * t) * n + c];
#pragma unroll
      for (int t = 0; t < 16; ++t) Ls[(rb + 2 * t) * kPanelPitch + c] = v[t];
    }
  }
  __syncthreads();
  const int n_strips = (n - k0 - kPanel + kSub - 1) / kSub;
  for (int strip = blockIdx.x * 4 + (tid >> 6); strip < n_strips; strip += gridDim.x * 4) {
    const int i0 = k0 + kPanel + kSub * strip;
    const bool live = i0 + li < n;
    double* arow = A + int64_t(min(i0 + li, n - 1)) * n + k0;
    v4f64 X[kS], R[kS];
#pragma unroll
    for (int s = 0; s < kS; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) R[s][r] = arow[kSub * s + lq + 4 * r];
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      v4f64 acc = R[s];
#pragma unroll
      for (int sp = 0; sp < s; ++sp) {
        const double* lp = Ls + (kSub * s + li) * kPanelPitch + kSub * sp + lq;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lp[4 * kk], X[sp][kk], acc, 0, 0, 0);
      }
      const double* ip = Ls + (kSub * linv_block_row(s) + li) * kPanelPitch + kSub * linv_block_col(s) + lq;
      v4f64 t = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ip[4 * kk], acc[kk], t, 0, 0, 0);
      X[s] = t;
    }
    if (live) {
#pragma unroll
      for (int s = 0; s < kS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) arow[kSub * s + lq + 4 * r] = X[s][r];
    }
  }
}

// Trailing update of the LOWER triangle: A[i, j] -= sum_p X[i, p] X[j, p], X = A[:, k0 : k0 + kPanel], i >= j >= k0 + kPanel.
// Workgroup = 128 x 128 outputs, wavefront = 64 x 64 = 4 x 4 MFMA tiles (16 accumulators in AGPRs).  The two 128-row operand strips
// go through LDS 32 panel columns at a time — coalesced loads into registers while the previous chunk multiplies, two LDS buffers,
// one barrier per chunk — and every k-step takes its eight operands (A[li][4 ks + lq] / B[4 ks + lq][li]: both "row li of the strip,
// column 4 ks + lq") from there: row pitch 36 doubles = the 64 lanes of a read fall two to a bank, the floor for 8-byte reads.
// (The first version read its operands straight from memory, one wavefront per SIMD: 4 x the MFMA-bound time, waiting for loads.)
constexpr int kSyrkChunk = 32;
constexpr int kSyrkPitch = 36;
constexpr int kSyrkOperand = 128 * kSyrkPitch;   // doubles of one operand strip in one buffer
// kw = 128 or 256 panel columns [k0, k0 + kw) at once (full panels only: a narrower last panel has nothing below it); rows from
// `first` on, 128-wide block columns [cb0, ncb) counted from `first` (ncb = 0: to the end).  Two panels per update halve the traffic on C — at K = 128 the
// 16 bytes read and written per entry per 256 flops put the update on the HBM roof (16 flop / B x 5 TB/s = the MFMA peak), at 256 under it.
__global__ __launch_bounds__(256) void dense_syrk_mfma_kernel(double* __restrict__ A, int n, int k0, int kw, int first, int cb0, int ncb) {
  extern __shared__ double sh[];   // [2 buffers][A strip | B strip][128][kSyrkPitch]
  const int bi = blockIdx.y, bj = blockIdx.x + cb0;   // block columns [cb0, ncb) of the trailing matrix (ncb = 0: to the end)
  if (bj > bi || (ncb > 0 && bj >= ncb)) return;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int ib = first + 128 * bi, jb = first + 128 * bj;
  const int i0 = ib + 64 * (wv >> 1), j0 = jb + 64 * (wv & 1);
  const bool active = i0 < n && j0 < n && j0 <= i0 + 63;   // inside the matrix and not entirely above the diagonal
  const int li = lane & 15, lq = lane >> 4;
  // staging: element e = tid + 256 j of a 128 x 32 chunk, row e >> 5, column e & 31 (a wavefront = two 256-byte row segments)
  const int srow = tid >> 5, scol = tid & 31;
  const double* ga[16];
  const double* gb[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {   // rows past the end are clamped: their results are never stored
    ga[j] = A + int64_t(min(ib + srow + 8 * j, n - 1)) * n + k0 + scol;
    gb[j] = A + int64_t(min(jb + srow + 8 * j, n - 1)) * n + k0 + scol;
  }
  double ra[16], rb[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { ra[j] = ga[j][0]; rb[j] = gb[j][0]; }
  // The wave's 64 x 64 outputs as 16 x 4 accumulators of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction,
  // one double per lane): acc[p][c] of lane (q = lane & 3, blk = (lane >> 2) & 3, h = lane >> 4) is C[4 p + h][16 c + 4 blk + q].
  // Operand layout of the instruction (tools/probes/mfma4_layout_probe.hip, profiles/r04n_*): A(block, row, k) sits in lane
  // row + 4 block + 16 k, B(block, k, col) in col + 4 block + 16 k, D(block, row, col) in col + 4 block + 16 row — so the B operand of
  // column group c is "row lane & 15 of the strip's 16-row group c, column 4 ks + (lane >> 4)", exactly the 16x16x4 form's operand, and
  // the A operand of row strip p is row 4 p + (lane & 3), the same for the four blocks (an LDS broadcast).  WHY: on gfx950 the
  // 16x16x4 f64 instruction issues every 143 cycles per SIMD (35 TFLOP/s on the chip), the 4x4x4_4b form every 16.7 (77 TFLOP/s):
  // tools/probes/mfma_f64_probe.hip, profiles/r04m_mfma_f64_probe.txt.  CERES_HIP_AB_SYRK_16X16=1 builds the round-3 form (A/B).
#ifndef CERES_HIP_AB_SYRK_16X16
#define CERES_HIP_AB_SYRK_16X16 0
#endif
#if CERES_HIP_AB_SYRK_16X16
  v4f64 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
#else
  double acc4[16][4];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc4[p][c] = 0.0;
#endif
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sh[(srow + 8 * j) * kSyrkPitch + scol] = ra[j];
    sh[kSyrkOperand + (srow + 8 * j) * kSyrkPitch + scol] = rb[j];
  }
  __syncthreads();
  const int nch = kw / kSyrkChunk;
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    const double* cur = sh + (ch & 1) * 2 * kSyrkOperand;
    double* nxt = sh + ((ch + 1) & 1) * 2 * kSyrkOperand;
    const bool more = ch + 1 < nch;
    if (more) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { ra[j] = ga[j][kSyrkChunk * (ch + 1)]; rb[j] = gb[j][kSyrkChunk * (ch + 1)]; }
    }
    if (active) {
      const double* pb = cur + kSyrkOperand + (64 * (wv & 1) + li) * kSyrkPitch + lq;
#if CERES_HIP_AB_SYRK_16X16
      const double* pa = cur + (64 * (wv >> 1) + li) * kSyrkPitch + lq;
#pragma unroll
      for (int ks = 0; ks < kSyrkChunk / 4; ++ks) {
        double av[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { av[t] = pa[16 * t * kSyrkPitch + 4 * ks]; bv[t] = pb[16 * t * kSyrkPitch + 4 * ks]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
#else
      const double* pa4 = cur + (64 * (wv >> 1) + (lane & 3)) * kSyrkPitch + lq;
#pragma unroll
      for (int ks = 0; ks < kSyrkChunk / 4; ++ks) {
        double ap[16], bv[4];
#pragma unroll
        for (int p = 0; p < 16; ++p) ap[p] = pa4[4 * p * kSyrkPitch + 4 * ks];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = pb[16 * t * kSyrkPitch + 4 * ks];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc4[p][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[p], bv[c], acc4[p][c], 0, 0, 0);
      }
#endif
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        nxt[(srow + 8 * j) * kSyrkPitch + scol] = ra[j];
        nxt[kSyrkOperand + (srow + 8 * j) * kSyrkPitch + scol] = rb[j];
      }
    }
    __syncthreads();   // the next buffer is complete, and nobody still reads the one the round after next overwrites
  }
  if (!active) return;
  // C -= acc, one row of tiles at a time: 16 loads in flight, then 16 predicated stores (addresses clamped into the matrix)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double c[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = min(i0 + 16 * a + lq + 4 * r, n - 1), j = min(j0 + 16 * b + li, n - 1);
        c[b][r] = A[int64_t(i) * n + j];
      }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * a + lq + 4 * r, j = j0 + 16 * b + li;
#if CERES_HIP_AB_SYRK_16X16
        if (i < n && j <= i) A[int64_t(i) * n + j] = c[b][r] - acc[a][b][r];
#else
        if (i < n && j <= i) A[int64_t(i) * n + j] = c[b][r] - acc4[4 * a + r][b];   // row 16 a + 4 r + lq = 4 p + h with p = 4 a + r
#endif
      }
  }
}

// Triangular solves with the factor, one right-hand side, blocked by kNb: x <- L^-1 x then x <- L^-T x.
__global__ __launch_bounds__(64) void dense_trsv_diag_kernel(const double* __restrict__ A, int n, int k0, int nb, double* __restrict__ x, int transposed) {
  // the nb x nb diagonal block through LDS (coalesced rows), lane r owns x[k0 + r]; column by column, the solved entry travels by shuffle
  // (a single thread walking the block in global memory, as this kernel first did, took 35 us per call: 18 ms of a solve at n = 8192)
  __shared__ double Ls[kNb][kNb + 1];
  const int t = threadIdx.x;
  for (int e = t; e < kNb * kNb; e += 64) {
    const int r = e / kNb, c = e % kNb;
    Ls[r][c] = (r < nb && c <= r) ? A[int64_t(k0 + r) * n + (k0 + c)] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  double v = t < nb ? x[k0 + t] : 0.0;
  if (!transposed) {
#pragma unroll
    for (int c = 0; c < kNb; ++c) {
      if (t == c) v /= Ls[c][c];
      const double xc = __shfl(v, c, 64);
      if (t > c && t < kNb) v -= Ls[t][c] * xc;
    }
  } else {
#pragma unroll
    for (int c = kNb - 1; c >= 0; --c) {
      if (t == c) v /= Ls[c][c];
      const double xc = __shfl(v, c, 64);
      if (t < c) v -= Ls[c][t] * xc;
    }
  }
  if (t < nb) x[k0 + t] = v;
}
// forward: x[i] -= sum_c L[i, k0 + c] x[k0 + c] for i >= k0 + nb;  backward (transposed): x[i] -= sum_c L[k0 + c, i] x[k0 + c] for i < k0
__global__ __launch_bounds__(kB) void dense_trsv_update_kernel(const double* __restrict__ A, int n, int k0, int nb, double* __restrict__ x, int transposed) {
  __shared__ double xs[kNb];
  if (threadIdx.x < kNb) xs[threadIdx.x] = int(threadIdx.x) < nb ? x[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int i = (transposed ? 0 : k0 + nb) + blockIdx.x * kB + threadIdx.x;
  if (transposed ? i >= k0 : i >= n) return;
  double s = 0.0;
  if (!transposed) { for (int c = 0; c < nb; ++c) s += A[int64_t(i) * n + (k0 + c)] * xs[c]; }
  else { for (int c = 0; c < nb; ++c) s += A[int64_t(k0 + c) * n + i] * xs[c]; }
  x[i] -= s;
}

inline unsigned blocks_for(int64_t n) { return unsigned((n + kB - 1) / kB); }

}  // namespace

hipError_t LaunchSchurSparseEliminate(const GenStructure& G, const SchurPairs& P, const double* values, const double* ete_inv,
                                      const double* D, double* S, hipStream_t s) {
  if (P.n_items > 0) hipLaunchKernelGGL(schur_sparse_eliminate_kernel, dim3((P.n_items + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, s, G, P, values, ete_inv);
  if (P.total_values > 0) hipLaunchKernelGGL(schur_sparse_combine_kernel, dim3(blocks_for(P.total_values)), dim3(kB), 0, s, G, P, D, P.total_values, S);
  return hipGetLastError();
}
hipError_t LaunchSchurBlocksToDense(const GenStructure& G, const SchurPairs& P, const double* S, int64_t total, double* lhs, hipStream_t s) {
  if (total > 0) hipLaunchKernelGGL(schur_blocks_to_dense_kernel, dim3(blocks_for(total)), dim3(kB), 0, s, G, P, S, total, lhs);
  return hipGetLastError();
}
hipError_t LaunchSchurSparseSymv(const GenStructure& G, const SchurPairs& P, const double* S, const double* x, double* y,
                                 const int* status, int accumulate, hipStream_t s) {
  if (G.ncf > 0) hipLaunchKernelGGL(schur_sparse_symv_kernel, dim3(blocks_for(G.ncf)), dim3(kB), 0, s, G, P, S, x, y, status, accumulate);
  return hipGetLastError();
}
hipError_t LaunchSchurSparseDiag(const GenStructure& G, const SchurPairs& P, const double* S, const int64_t* diag_off_f, double* blocks,
                                 int64_t total, hipStream_t s) {
  if (total > 0) hipLaunchKernelGGL(schur_sparse_diag_kernel, dim3(blocks_for(total)), dim3(kB), 0, s, G, P, S, diag_off_f, blocks);
  return hipGetLastError();
}

// A side stream and two events per device for the look-ahead below (created once, shared by every factorisation on the device:
// calls that overlap in time still order correctly through their event waits, they just share the side stream).
namespace {
struct LookAhead { hipStream_t side = nullptr; hipEvent_t panels = nullptr, bulk = nullptr; bool ok = false; };
LookAhead* look_ahead_for_current_device() {
  static std::mutex mu;
  static LookAhead per_device[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  LookAhead& L = per_device[dev];
  if (!L.ok) {
    if (hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L.panels, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L.bulk, hipEventDisableTiming) != hipSuccess) return nullptr;
    L.ok = true;
  }
  return &L;
}
}  // namespace

// In-place Cholesky of the symmetric n x n matrix whose UPPER triangle is authoritative; L is left in the lower triangle.
// Panels go in PAIRS (P0, P1 = 256 columns).  After P0 only P1's 128 columns are updated; everything behind the pair takes both panels
// in ONE update with K = 256.  That update is split: the 256 columns of the NEXT pair first, on the caller's stream, which then goes
// straight on to factor them; the bulk behind them on a side stream, concurrently (look-ahead) — the chain potrf / trsm / potrf / trsm
// of a pair is latency-bound on one or a few CUs and as long as the bulk update of the pair before it.
hipError_t LaunchDenseCholesky(double* A, int n, int* fail_flag, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  // The dynamic-LDS ceiling is an attribute of a (kernel, DEVICE) pair: raised once per device, under a lock (a process-wide "once" left
  // the kernels of a second device — two handles with different options.device — at the default ceiling and their launches failed).
  // The panel kernel keeps a kPanel x kPanel block in LDS (132 KB) next to its static arrays, the update two operand strips twice (147 KB).
  {
    static std::mutex lds_mu;
    static unsigned long long lds_done = 0ull;   // mask of devices 0..63
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(lds_mu);
    if (!(lds_done & bit)) {
      const void* kernels[3] = {reinterpret_cast<const void*>(dense_potrf_panel_kernel), reinterpret_cast<const void*>(dense_syrk_mfma_kernel),
                                reinterpret_cast<const void*>(dense_trsm_lds_kernel)};
      for (const void* k : kernels) {
        hipFuncAttributes fa;
        if (hipError_t e = hipFuncGetAttributes(&fa, k); e != hipSuccess) return e;
        const int dyn = int(kMaxLdsBytes) - int(fa.sharedSizeBytes);   // static + dynamic LDS must fit the CU's 160 KB
        if (hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); e != hipSuccess) return e;
      }
      lds_done |= bit;
    }
  }
  hipLaunchKernelGGL(dense_mirror_upper_kernel, dim3(blocks_for(int64_t(n) * n)), dim3(kB), 0, s, A, n);
  const size_t syrk_lds = 4 * kSyrkOperand * sizeof(double);
  LookAhead* la = n >= 24 * kPanel ? look_ahead_for_current_device() : nullptr;   // below ~3000 columns the event traffic costs more than it hides (n = 2052: 2.7 vs 3.9 ms)
  // (the side stream and its events are shared per device: one factorisation ENQUEUES at a time — a wait captures the record that
  // precedes it in enqueue order, so whole sequences must not interleave)
  static std::mutex enqueue_mu;
  std::unique_lock<std::mutex> enqueue_lock(enqueue_mu, std::defer_lock);
  if (la) enqueue_lock.lock();
  bool bulk_pending = false;
  auto panel = [&](int k0, int kw) {
    const size_t lds = size_t((kw + kSub - 1) / kSub * kSub) * kPanelPitch * sizeof(double);
    hipLaunchKernelGGL(dense_potrf_panel_kernel, dim3(1), dim3(256), lds, s, A, n, k0, kw, fail_flag);
    const int rest = n - k0 - kw;
    if (rest > 0) {   // (rest > 0: a full panel)
      static const bool from_memory = [] { const char* e = getenv("CERES_HIP_AB_TRSM_GLOBAL"); return e && atoi(e) != 0; }();   // (A/B: the round-3 kernel)
      if (from_memory) hipLaunchKernelGGL(dense_trsm_mfma_kernel, dim3((rest + 63) / 64), dim3(256), 0, s, A, n, k0);
      else hipLaunchKernelGGL(dense_trsm_lds_kernel, dim3(std::min((rest + 63) / 64, 256)), dim3(256), size_t(kPanel) * kPanelPitch * sizeof(double), s, A, n, k0);
    }
    return rest;
  };
  for (int k0 = 0; k0 < n; k0 += 2 * kPanel) {
    // ---- the pair's two panels, on the caller's stream
    const int kw0 = n - k0 < kPanel ? n - k0 : kPanel;
    const int rest0 = panel(k0, kw0);
    if (rest0 <= 0) break;
    const unsigned blocks0 = unsigned((rest0 + 127) / 128);
    // P1's columns (block column 0 behind P0) take P0 now; if nothing lies behind P1 this is the whole trailing matrix
    hipLaunchKernelGGL(dense_syrk_mfma_kernel, dim3(1, blocks0), dim3(256), syrk_lds, s, A, n, k0, kPanel, k0 + kPanel, 0, 1);
    const int k1 = k0 + kPanel;
    const int kw1 = n - k1 < kPanel ? n - k1 : kPanel;
    const int rest1 = panel(k1, kw1);
    if (rest1 <= 0) break;
    // ---- everything behind the pair takes both panels at once: K = 256 from k0, rows and columns from k1 + 128 on
    const int first = k1 + kPanel;
    const unsigned blocks = unsigned((rest1 + 127) / 128);
    if (la && blocks > 2) {
      if (hipError_t e = hipEventRecord(la->panels, s); e != hipSuccess) return e;
      // the next pair's columns (they were part of the previous pair's bulk update: wait for it), then straight on to factor them
      if (bulk_pending) if (hipError_t e = hipStreamWaitEvent(s, la->bulk, 0); e != hipSuccess) return e;
      hipLaunchKernelGGL(dense_syrk_mfma_kernel, dim3(2, blocks), dim3(256), syrk_lds, s, A, n, k0, 2 * kPanel, first, 0, 2);
      // the bulk behind them, concurrently (the side stream is in order: this pair's bulk follows the previous pair's)
      if (hipError_t e = hipStreamWaitEvent(la->side, la->panels, 0); e != hipSuccess) return e;
      hipLaunchKernelGGL(dense_syrk_mfma_kernel, dim3(blocks - 2, blocks), dim3(256), syrk_lds, la->side, A, n, k0, 2 * kPanel, first, 2, 0);
      if (hipError_t e = hipEventRecord(la->bulk, la->side); e != hipSuccess) return e;
      bulk_pending = true;
    } else {
      if (bulk_pending) { if (hipError_t e = hipStreamWaitEvent(s, la->bulk, 0); e != hipSuccess) return e; bulk_pending = false; }
      hipLaunchKernelGGL(dense_syrk_mfma_kernel, dim3(blocks, blocks), dim3(256), syrk_lds, s, A, n, k0, 2 * kPanel, first, 0, 0);
    }
  }
  if (bulk_pending) if (hipError_t e = hipStreamWaitEvent(s, la->bulk, 0); e != hipSuccess) return e;   // join
  return hipGetLastError();
}
// x <- (L L^T)^-1 x
hipError_t LaunchDenseCholeskySolve(const double* A, int n, double* x, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  for (int k0 = 0; k0 < n; k0 += kNb) {
    const int nb = n - k0 < kNb ? n - k0 : kNb;
    hipLaunchKernelGGL(dense_trsv_diag_kernel, dim3(1), dim3(64), 0, s, A, n, k0, nb, x, 0);
    const int rest = n - k0 - nb;
    if (rest > 0) hipLaunchKernelGGL(dense_trsv_update_kernel, dim3(blocks_for(rest)), dim3(kB), 0, s, A, n, k0, nb, x, 0);
  }
  const int last = ((n - 1) / kNb) * kNb;
  for (int k0 = last; k0 >= 0; k0 -= kNb) {
    const int nb = n - k0 < kNb ? n - k0 : kNb;
    hipLaunchKernelGGL(dense_trsv_diag_kernel, dim3(1), dim3(64), 0, s, A, n, k0, nb, x, 1);
    if (k0 > 0) hipLaunchKernelGGL(dense_trsv_update_kernel, dim3(blocks_for(k0)), dim3(kB), 0, s, A, n, k0, nb, x, 1);
  }
  return hipGetLastError();
}

}  // namespace chip
