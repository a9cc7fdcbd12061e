// kernels_bal.hip — fused single-pass kernels for the static <2,3,9> (BAL) structure.
//
// Data layout in HBM (built once per LM step from the caller's values, whatever their layout — by the step's first pass over
// J, which gathers through cell.position and writes the tiles on the way; bal_pack_kernel is the stand-alone form):
//   tiles of 64 observation slots, one wavefront per tile;
//   J   [tile][12][64] double2   pair j of slot l holds Jacobian doubles (2j, 2j+1) of the
//                                24 = 6 (E, 2x3 row-major) + 18 (F, 2x9 row-major)
//   b   [tile][64]     double2   the two residuals of the slot
//   cam [tile][64] int32, seg [tile][64] uint32 (first | last<<8 | valid<<16 | store lanes of the tile's dense point range),
//   pt0 [tile] int32             points of a tile are consecutive ids: a slot's point = pt0 + (segment heads at or below it) - 1
// so that every global load of the hot kernels is a 16-byte-per-lane, 1 KiB-per-wave
// contiguous access and the algorithmic traffic is 192 + 8 bytes per observation.  Streamed-once data is read with
// non-temporal loads (stream_load<>), the tiles are written with non-temporal stores (tile_store<>).
//
// All per-point reductions (E^T·, (E^T E)^-1) are segmented wavefront scans over
// __shfl_up — observations of a point are adjacent lanes by construction of the
// plan (plan.cc); points longer than 64 observations own whole tiles, which sit behind the
// normal ones and are taken in ROUNDS: one tile per wave of a workgroup, the tile sums exchanged
// through LDS, every tile finished from registers (compute_long_round, fused_long_rounds).  Per-camera sums (F^T·) go to a
// workgroup-private accumulator in LDS (9 doubles per camera: 128 KB for Venice's
// 1778 cameras, LDS is 160 KB) with ds_add_f64, flushed once per workgroup and
// combined by bal_reduce_partials_kernel; when the cameras do not fit in LDS the tile pass leaves F_o^T z_o per slot
// (72-byte rows, transposed through LDS into coalesced stores) and bal_camera_chunk_kernel sums them camera by camera.
// Camera-major work on the Jacobian itself (the 9x9 preconditioner blocks) is bal_camera_items_kernel / bal_camera_items_mfma_kernel
// (per-item partial sums: lane per observation, or — short items — the matrix pipe) + bal_invert9_kernel (sums a camera's items,
// adds D^2 or the fused LM diagonal, inverts).  For camera spaces of a few hundred scalars the S.x pass also finishes the CG
// iteration (cg_iteration_tail).
//
// Reference operators restated by each MODE (file:line in include/ceres_hip.h):
//   kSx        ImplicitSchurComplement::RightMultiplyAndAccumulate (4 passes there, 1 here)
//   kJtJx      CgnrLinearOperator::RightMultiplyAndAccumulate      (2 passes there, 1 here); also leaves partial sums of x . y
//   kJtb       A^T b
//   kInit      ImplicitSchurComplement::Init: (E^T E + D^2)^-1, rhs; also the 2x2 blocks
//              M_o = I - E_o (E^T E)^-1 E_o^T that SCHUR_JACOBI needs, and the fused LM diagonal of the point columns
//   kCgnrInit  J^T b, the point blocks of JACOBI and the fused LM diagonal, in one pass
//   kEte       block diagonal (E^T E + D^2)^-1 only (CGNR JACOBI point blocks)
//   kColNorm   BlockSparseMatrix::SquaredColumnNorm
//   kBackSub   ImplicitSchurComplement::BackSubstitute; as the last kernel of an LM step also the negation, the finite check
//              and the model-cost partial sums (gated on the CG status word when enqueued speculatively)
//   kSpseZ     the inverse power-series operator of SCHUR_POWER_SERIES_EXPANSION (S.x without the F^T F term)
//   kJx        J x (model cost where nothing else has it in hand)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "device.h"

namespace chip {

namespace {

__device__ __forceinline__ double shfl_up(double v, int d) { return __shfl_up(v, (unsigned)d, 64); }
__device__ __forceinline__ double shfl_idx(double v, int l) { return __shfl(v, l, 64); }
__device__ __forceinline__ double shfl_xor(double v, int m) { return __shfl_xor(v, m, 64); }

// Inclusive segmented scan (segments = runs of lanes [first,last]); afterwards the
// last lane of each segment holds the segment sum.  `span` bounds the longest
// segment in the tile (wave-uniform) so that short tracks take fewer steps.
template <int N>
__device__ __forceinline__ void seg_scan(double (&v)[N], int lane, int first, int span) {
  for (int d = 1; d < span; d <<= 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double t = shfl_up(v[i], d);
      if (lane - d >= first) v[i] += t;
    }
  }
}
template <int N>
__device__ __forceinline__ void seg_allreduce(double (&v)[N], int lane, int first, int last, int span) {
  seg_scan<N>(v, lane, first, span);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = shfl_idx(v[i], last);
}
template <int N>
__device__ __forceinline__ void wave_allreduce(double (&v)[N]) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += shfl_xor(v[i], m);
  }
}

// Inverse of the SPD 3x3 matrix [a0 a1 a2; a1 a3 a4; a2 a4 a5] through its Cholesky
// factor (the reference: selfadjointView<Upper>().llt().solve(I),
// I/implicit_schur_complement.cc:179-204).  Output in the same packed order.
__device__ __forceinline__ void invert_spd3(const double (&a)[6], double (&o)[6]) {
  const double l00 = sqrt(a[0]);
  const double i00 = 1.0 / l00;
  const double l10 = a[1] * i00, l20 = a[2] * i00;
  const double l11 = sqrt(a[3] - l10 * l10);
  const double i11 = 1.0 / l11;
  const double l21 = (a[4] - l20 * l10) * i11;
  const double l22 = sqrt(a[5] - l20 * l20 - l21 * l21);
  const double i22 = 1.0 / l22;
  // W = L^-1 (lower): w00 w10 w11 w20 w21 w22
  const double w10 = -l10 * i00 * i11;
  const double w21 = -l21 * i11 * i22;
  const double w20 = -(l20 * i00 + l21 * w10) * i22;
  // A^-1 = W^T W
  o[0] = i00 * i00 + w10 * w10 + w20 * w20;
  o[1] = w10 * i11 + w20 * w21;
  o[2] = w20 * i22;
  o[3] = i11 * i11 + w21 * w21;
  o[4] = w21 * i22;
  o[5] = i22 * i22;
}


// Streamed-once data is read with non-temporal loads: the L2 then evicts it first, and what is re-used (camera vectors, dirty
// point-space output lines, the partial sums) stays.  CERES_HIP_AB_NT_LOADS selects how far this goes in A/B builds
// (tools/build_variant.sh): 0 = nowhere, 1 = the 12 pairs of a packed tile in the pipelined kernels, 2 = + packed tiles and b in the
// unpipelined kernels (the product), 3 = + the caller-layout gathers of a step's first pass, 4 = + index words and per-point
// inverses.  Measured on the Venice shape (profiles/r02t_nt_load_levels_venice.txt): level 1 S.x 0.2138 -> 0.2000 ms and JtJx
// -7 %, level 2 back-substitution 0.252 -> 0.233 ms; level 3 is a disaster (non-temporal loads bypass the L1 that the per-lane
// 8-byte gathers of a 192-byte record live on: kInit 0.50 -> 0.75 ms) and level 4 costs S.x 2 %.
// The tiles a step's first pass writes on the way (fused re-layout) go out with non-temporal stores: 1 GB that the L2 cannot keep
// until its next reader anyway (kInit 0.484 -> 0.467 ms, CGNR set-up 0.734 -> 0.719 ms, stand-alone pack 0.376 -> 0.360 ms;
// profiles/r02v_nt_tile_stores_venice.txt).  CERES_HIP_AB_NT_TILE_STORES: 0 = plain stores, 1 = tiles, 2 = + M_o and (E^T E)^-1.
#ifndef CERES_HIP_AB_NT_TILE_STORES
#define CERES_HIP_AB_NT_TILE_STORES 1
#endif
template <int LEVEL = 1, typename T>
__device__ __forceinline__ void tile_store(T* p, const T& v) {
#if CERES_HIP_AB_NT_TILE_STORES
  if constexpr (CERES_HIP_AB_NT_TILE_STORES < LEVEL) { *p = v; return; }
  typedef int v4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(T) == 16, "16-byte tile elements");
  v4 raw;
  __builtin_memcpy(&raw, &v, 16);
  __builtin_nontemporal_store(raw, reinterpret_cast<v4*>(p));
#else
  *p = v;
#endif
}
#ifndef CERES_HIP_AB_NT_LOADS
#define CERES_HIP_AB_NT_LOADS 2
#endif
template <int LEVEL, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (CERES_HIP_AB_NT_LOADS >= LEVEL) {
    if constexpr (sizeof(T) == 16) {
      typedef int v4 __attribute__((ext_vector_type(4)));
      const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
      T out;
      __builtin_memcpy(&out, &v, 16);
      return out;
    } else {
      return __builtin_nontemporal_load(p);
    }
  } else {
    return *p;
  }
}

struct Slot {
  double e[6], f[18];
  double b0, b1;
  int64_t slot;
  int64_t zbase;  // cameras not in LDS: ring row of the tile's first spilled slot (wave-uniform)
  uint32_t seg;
  int cam, pt, first, last;
  int acc;        // LDS accumulator row of the slot's camera (kSlotSpill: none, the slot's F^T z is spilled); plan.cc, slot_word
  bool valid;
};

// Derives the fields of a slot from its segment word.  The point id is not stored per slot (it was:
// 4 of 204 bytes per observation): points of a tile are consecutive ids starting at tile_pt0, so a
// slot's point is pt0 + (number of segment heads at or below its lane) - 1.  Tiles of a long point
// have one segment: every lane gets pt0.
__device__ __forceinline__ void finish_slot(Slot& s, int lane, int pt0) {
  const uint32_t sg = s.seg;
  s.first = sg & 0xff;
  s.last = (sg >> 8) & 0xff;
  s.valid = (sg >> 16) & 1;
  const unsigned long long heads = __ballot(s.valid && lane == s.first);
  const int rank = __popcll(heads & ((2ull << lane) - 1ull)) - 1;
  s.pt = pt0 + rank;
  s.acc = int(uint32_t(s.cam) >> kSlotCamBits);   // the index word: camera | accumulator row << kSlotCamBits (0 above the camera when every camera has its LDS row)
  s.cam &= (1 << kSlotCamBits) - 1;
  if (!s.valid) { s.cam = 0; s.pt = 0; s.acc = kSlotSpill; }
}

// Loads one slot.  Normally from the packed tiles; on the FIRST pass over a step's Jacobian
// (A.src_values != nullptr and may_gather) the 24 doubles are gathered from the caller's
// layout through slot_epos / slot_fpos and the tile is written on the way, which fuses the
// re-layout (bal_pack_kernel) into the first kernel that needs the data anyway.
// CAN_GATHER is a compile-time property of the mode (only kInit / kCgnrInit are ever a step's
// first pass): the streaming kernels do not carry the gather code or its registers.
// F32: the tiles hold the Jacobian rounded to fp32 ([tile][6][64] float4, 96 B per observation instead
// of 192); all arithmetic stays fp64.  Not a parity mode: accuracy is reported.
template <bool CAN_GATHER, bool F32>
__device__ __forceinline__ void load_slot(const BalArgs& A, int64_t tile, int lane, Slot& s, bool want_b,
                                          bool may_gather = true) {
  const int64_t sl = tile * kTile + lane;
  s.slot = sl;
  s.zbase = A.tile_zbase ? A.tile_zbase[tile] : 0;
  s.b0 = 0.0; s.b1 = 0.0;
  if (CAN_GATHER && A.src_values && may_gather) {
    const int ep = A.slot_epos[sl], fp = A.slot_fpos[sl];
    double v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = 0.0;
    if (ep >= 0) {
      const double* e = A.src_values + ep;
      const double* f = A.src_values + fp;
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = stream_load<3>(e + i);
#pragma unroll
      for (int i = 0; i < 18; ++i) v[6 + i] = stream_load<3>(f + i);
      if (A.src_b) { const int bp = A.slot_bpos[sl]; s.b0 = stream_load<3>(A.src_b + bp); s.b1 = stream_load<3>(A.src_b + bp + 1); }
    }
    if constexpr (F32) {
      float4* o = A.Jf_out + tile * (6 * kTile) + lane;
#pragma unroll
      for (int q = 0; q < 6; ++q) tile_store(o + q * kTile, make_float4(float(v[4 * q]), float(v[4 * q + 1]), float(v[4 * q + 2]), float(v[4 * q + 3])));
#pragma unroll
      for (int i = 0; i < 24; ++i) v[i] = double(float(v[i]));  // this pass computes with what later passes will read
    } else {
      double2* o = A.J_out + tile * kTilePitch + lane;
#pragma unroll
      for (int j = 0; j < kPairsPerSlot; ++j) tile_store(o + j * kTile, make_double2(v[2 * j], v[2 * j + 1]));
    }
    if (A.src_b) tile_store(A.b_out + sl, make_double2(s.b0, s.b1));
#pragma unroll
    for (int i = 0; i < 6; ++i) s.e[i] = v[i];
#pragma unroll
    for (int i = 0; i < 18; ++i) s.f[i] = v[6 + i];
  } else if constexpr (F32) {
    const float4* J = A.Jf + tile * (6 * kTile) + lane;
    float4 p[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) p[q] = stream_load<2>(J + q * kTile);
    if (want_b && A.have_b) { const double2 bb = stream_load<2>(A.b + sl); s.b0 = bb.x; s.b1 = bb.y; }
    double v[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) { v[4 * q] = p[q].x; v[4 * q + 1] = p[q].y; v[4 * q + 2] = p[q].z; v[4 * q + 3] = p[q].w; }
#pragma unroll
    for (int i = 0; i < 6; ++i) s.e[i] = v[i];
#pragma unroll
    for (int i = 0; i < 18; ++i) s.f[i] = v[6 + i];
  } else {
    const double2* J = A.J + tile * kTilePitch + lane;
    double2 p[kPairsPerSlot];
#pragma unroll
    for (int j = 0; j < kPairsPerSlot; ++j) p[j] = stream_load<2>(J + j * kTile);
    if (want_b && A.have_b) { const double2 bb = stream_load<2>(A.b + sl); s.b0 = bb.x; s.b1 = bb.y; }
    s.e[0] = p[0].x; s.e[1] = p[0].y; s.e[2] = p[1].x; s.e[3] = p[1].y; s.e[4] = p[2].x; s.e[5] = p[2].y;
#pragma unroll
    for (int j = 0; j < 9; ++j) { s.f[2 * j] = p[3 + j].x; s.f[2 * j + 1] = p[3 + j].y; }
  }
  s.cam = A.slot_cam[sl];
  s.seg = A.slot_seg[sl];
  finish_slot(s, lane, A.tile_pt0[tile]);
}

// The software-pipelined streaming kernel splits a slot load in three, each of which only ISSUES
// loads and consumes nothing: the index words (SlotIdx), the 12 pairs of a packed fp64 tile
// (issue_pairs), and what is addressed THROUGH the index words (issue_aux, further down).
struct SlotIdx { int cam; uint32_t seg; };
__device__ __forceinline__ void issue_idx(const BalArgs& A, int64_t tile, int lane, SlotIdx& i) {
  const int64_t sl = tile * kTile + lane;
  i.cam = stream_load<4>(A.slot_cam + sl);
  i.seg = stream_load<4>(A.slot_seg + sl);
}
// NT: the tile stream is larger than the Infinity Cache (non-temporal loads, see stream_load); a problem whose tiles fit there — and in
// the L2s, whose workgroup -> tile mapping is the same in every pass — is better off with plain loads: the next pass finds them on die
// (Ladybug shape: S.x 49 -> 41 us, Dubrovnik 13.0 -> 11.7; profiles/r03n_small_shapes_nt_loads_ab.txt).
template <bool NT = true>
__device__ __forceinline__ void issue_pairs(const BalArgs& A, int64_t tile, int lane, Slot& s) {
  s.slot = tile * kTile + lane;
  s.zbase = A.tile_zbase ? A.tile_zbase[tile] : 0;
  s.b0 = 0.0; s.b1 = 0.0;
  const double2* J = A.J + tile * kTilePitch + lane;
  double2 p[kPairsPerSlot];
#pragma unroll
  for (int j = 0; j < kPairsPerSlot; ++j) p[j] = NT ? stream_load<1>(J + j * kTile) : J[j * kTile];
  s.e[0] = p[0].x; s.e[1] = p[0].y; s.e[2] = p[1].x; s.e[3] = p[1].y; s.e[4] = p[2].x; s.e[5] = p[2].y;
#pragma unroll
  for (int j = 0; j < 9; ++j) { s.f[2 * j] = p[3 + j].x; s.f[2 * j + 1] = p[3 + j].y; }
}

__device__ __forceinline__ int pt_off(const BalArgs& A, int p) { return A.pt_pos ? A.pt_pos[p] : 3 * p; }
__device__ __forceinline__ int cam_off(const BalArgs& A, int c) { return A.cam_pos ? A.cam_pos[c] : 9 * c; }

__device__ __forceinline__ void load_ete_inverse(const BalArgs& A, int pt, double (&ei)[6]) {
  const double2* q = reinterpret_cast<const double2*>(A.etei + int64_t(pt) * 6);
  const double2 a = stream_load<4>(q), b = stream_load<4>(q + 1), c = stream_load<4>(q + 2);
  ei[0] = a.x; ei[1] = a.y; ei[2] = b.x; ei[3] = b.y; ei[4] = c.x; ei[5] = c.y;
}
__device__ __forceinline__ void sym3_mul(const double (&m)[6], const double (&u)[3], double (&v)[3]) {
  v[0] = m[0] * u[0] + m[1] * u[1] + m[2] * u[2];
  v[1] = m[1] * u[0] + m[3] * u[1] + m[4] * u[2];
  v[2] = m[2] * u[0] + m[4] * u[1] + m[5] * u[2];
}

// The 9 camera scalars of a slot (72 contiguous bytes; L2-resident for the camera counts that fit LDS).
__device__ __forceinline__ void load_xc(const BalArgs& A, int cam, double (&xc)[9]) {
  const int co = cam_off(A, cam);
#pragma unroll
  for (int k = 0; k < 9; ++k) xc[k] = A.x_f[co + k];
}

// t = F * xc
__device__ __forceinline__ void f_times(const Slot& s, const double (&xc)[9], double& t0, double& t1) {
  t0 = 0; t1 = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { t0 += s.f[k] * xc[k]; t1 += s.f[9 + k] * xc[k]; }
}
// Camera-space contribution F^T z of one observation.  LDS: nine ds_add_f64 into the workgroup's accumulator, one row per camera.
// Otherwise (more cameras than LDS rows) the HYBRID form: a slot whose camera has a row in THIS workgroup's accumulator (one of
// the popular cameras every workgroup holds, or a camera of the workgroup's own window: Slot::acc, plan.cc) adds into it the same
// way; the others are SPILLED — the nine products are stored in a ring, the spilled slots of a tile back to back from the tile's
// ring row s.zbase on, and bal_camera_chunk_kernel sums them camera by camera in a second pass (global fp64 atomics on a few
// thousand hot addresses are an order of magnitude slower).  A lane storing its own 72-byte row would issue nine 8-byte stores
// at a 72-byte stride, and it is the L2's request rate, not its bytes, that such stores exhaust (the per-slot output cost as much
// as reading the 200 B / slot tile stream): the wave packs the rows through a private LDS strip (behind the accumulator rows) and
// stores one contiguous range.  Without a hybrid plan (CGNR on caller-ordered vectors, chunked rings) every slot spills.
template <bool LDS>
__device__ __forceinline__ void scatter_ft(const BalArgs& A, const Slot& s, double* acc, double z0, double z1) {
  if constexpr (LDS) {
    if (!s.valid) return;
    const int base = 9 * s.cam;
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicAdd(&acc[base + k], s.f[k] * z0 + s.f[9 + k] * z1);  // ds_add_f64
  } else {
    const bool local = s.valid && s.acc != kSlotSpill;
    if (local) {
      const int base = 9 * s.acc;
#pragma unroll
      for (int k = 0; k < 9; ++k) atomicAdd(&acc[base + k], s.f[k] * z0 + s.f[9 + k] * z1);
    }
    const bool spill = s.valid && !local;
    // (nothing to store for a tile without spilled slots: that includes the tiles of long points the pipelined kernel runs through
    // here with every lane masked — their real values are written by whichever wave owns the point's head tile)
    const unsigned long long mask = __ballot(spill);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int rank = __popcll(mask & ((1ull << lane) - 1ull)), n9 = 9 * __popcll(mask);
    double* st = acc + 9 * A.hyb_rows + (threadIdx.x >> 6) * (kTile * 9);  // this wave's strip
    if (spill) {
#pragma unroll
      for (int k = 0; k < 9; ++k) st[rank * 9 + k] = s.f[k] * z0 + s.f[9 + k] * z1;
    }
    __builtin_amdgcn_wave_barrier();
    double* w = A.zbuf + 9 * s.zbase;
#pragma unroll
    for (int j = 0; j < 9; ++j) if (kTile * j + lane < n9) w[kTile * j + lane] = st[kTile * j + lane];
    __builtin_amdgcn_wave_barrier();
  }
}

// Camera-column squared norms: sum over the observation of F[:,k]^2 (LDS accumulators only).
__device__ __forceinline__ void scatter_f_squares(const Slot& s, double* acc) {
  if (!s.valid) return;
  const int base = 9 * s.cam;
#pragma unroll
  for (int k = 0; k < 9; ++k) atomicAdd(&acc[base + k], s.f[k] * s.f[k] + s.f[9 + k] * s.f[9 + k]);
}

// A/B builds only (tools/build_variant.sh, loaded through CERES_HIP_LIBRARY): how the cooperative JtJx path stores y_e.
// 0 = plain stores (the product), 1 = non-temporal stores, 2 = no stores at all (timing ablation: results are wrong).
#ifndef CERES_HIP_AB_YE_STORE
#define CERES_HIP_AB_YE_STORE 0
#endif
// 3 (timing ablation, needs CERES_HIP_AB_TILE_ROWS=13): the output goes into the 1 KiB gap right behind the tile it was computed from
// (`alt`), i.e. into the DRAM neighbourhood the wave is reading anyway, instead of the point-space vector.
__device__ __forceinline__ void store_ye(double* p, double v, double* alt = nullptr) {
#if CERES_HIP_AB_YE_STORE == 1
  __builtin_nontemporal_store(v, p);
#elif CERES_HIP_AB_YE_STORE == 2
  (void)p; (void)v;
#elif CERES_HIP_AB_YE_STORE == 3
  (void)p; *alt = v;
#elif CERES_HIP_AB_YE_STORE == 4   // sc1: system-scope (write-through) store
  (void)alt; asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#elif CERES_HIP_AB_YE_STORE == 5   // sc0 sc1
  (void)alt; asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#elif CERES_HIP_AB_YE_STORE == 6   // sc0 sc1 nt
  (void)alt; asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
#else
  (void)alt;
  *p = v;
#endif
}
// A/B builds only: non-temporal loads for the point-space x / D of the pipelined JtJx (each scalar is read by exactly one tile)
#ifndef CERES_HIP_AB_XE_NT
#define CERES_HIP_AB_XE_NT 0
#endif
__device__ __forceinline__ double load_xe(const double* p) {
#if CERES_HIP_AB_XE_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
// XCD-aware tile walk: workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and neighbouring tiles share lines of
// everything indexed by point (x_e, D_e, y_e, (E^T E)^-1: a tile's point range starts wherever the previous one ended).  Workgroup b
// therefore walks the tile groups of "logical workgroup" (b % 8) * (grid / 8) + b / 8: XCD x gets the logical workgroups
// [x grid / 8, (x + 1) grid / 8), whose tiles are neighbours at every step of the grid-strided loop, so a shared line is fetched
// into ONE L2 instead of two.  Measured on the Venice shape (profiles/r03a_ab_*): S.x 0.2010 -> 0.1933 ms, JtJx 0.2374 -> 0.2334 ms.
// CERES_HIP_AB_XCD_GROUP=0 in an A/B build restores the identity mapping.
#ifndef CERES_HIP_AB_XCD_GROUP
#define CERES_HIP_AB_XCD_GROUP 1
#endif
__device__ __forceinline__ int64_t logical_workgroup() {
#if CERES_HIP_AB_XCD_GROUP
  return (gridDim.x % 8 == 0) ? int64_t(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : int64_t(blockIdx.x);
#else
  return blockIdx.x;
#endif
}

__device__ __forceinline__ double* ye_alt(const BalArgs& A, const Slot& s) {
#if CERES_HIP_AB_YE_STORE == 3
  static_assert(CERES_HIP_AB_TILE_ROWS == 13, "the gap behind the tile");
  return reinterpret_cast<double*>(const_cast<double2*>(A.J) + (s.slot / kTile) * kTilePitch + kPairsPerSlot * kTile);
#else
  (void)A; (void)s;
  return nullptr;
#endif
}

enum Mode { kSx = 0, kJtJx = 1, kJtb = 2, kInit = 3, kEte = 4, kBackSub = 5, kCgnrInit = 6, kColNorm = 7, kJx = 8, kSpseZ = 9 };

template <int MODE>
constexpr bool kWantsB = (MODE == kJtb || MODE == kInit || MODE == kBackSub || MODE == kCgnrInit || MODE == kJx);
template <int MODE>
constexpr bool kCanGather = (MODE == kInit || MODE == kCgnrInit || MODE == kColNorm || MODE == kJtb);

// E^T E (packed symmetric) of one slot.
__device__ __forceinline__ void ete_of(const Slot& s, double (&a)[6]) {
  a[0] = s.e[0] * s.e[0] + s.e[3] * s.e[3];
  a[1] = s.e[0] * s.e[1] + s.e[3] * s.e[4];
  a[2] = s.e[0] * s.e[2] + s.e[3] * s.e[5];
  a[3] = s.e[1] * s.e[1] + s.e[4] * s.e[4];
  a[4] = s.e[1] * s.e[2] + s.e[4] * s.e[5];
  a[5] = s.e[2] * s.e[2] + s.e[5] * s.e[5];
}

// Regularisation of the point block.  Either D is given (A.D_e), or — fused LM diagonal,
// A.lm_radius > 0 — it is formed right here from the block's own diagonal, which IS
// diag(J^T J) over the point's columns: d = clamp(a_jj, min, max), D^2 = d / radius
// (LevenbergMarquardtStrategy::ComputeStep, I/levenberg_marquardt_strategy.cc:84-96); `writer`
// lanes record d (for a later reuse_diagonal step) and D.
// pt / po: the point and its offset in x_e / y_e; the caller's D (and the LM diagonal) may live at another offset (A.d_pos: CGNR on
// internally numbered points), and then the writer also leaves D in the internal order (A.D_int_out) for the operator.
__device__ __forceinline__ void add_e_diagonal(const BalArgs& A, int pt, int po, double (&a)[6], bool writer) {
  const int pd = A.d_pos ? A.d_pos[pt] : po;
  if (A.lm_radius > 0.0) {
    const double d0 = fmin(fmax(a[0], A.lm_min), A.lm_max), d1 = fmin(fmax(a[3], A.lm_min), A.lm_max),
                 d2 = fmin(fmax(a[5], A.lm_min), A.lm_max);
    const double q0 = d0 / A.lm_radius, q1 = d1 / A.lm_radius, q2 = d2 / A.lm_radius;
    if (writer) {
      const double s0 = sqrt(q0), s1 = sqrt(q1), s2 = sqrt(q2);
      A.lm_diag_e[pd] = d0; A.lm_diag_e[pd + 1] = d1; A.lm_diag_e[pd + 2] = d2;
      A.lm_D_e[pd] = s0; A.lm_D_e[pd + 1] = s1; A.lm_D_e[pd + 2] = s2;
      if (A.D_int_out) { A.D_int_out[po] = s0; A.D_int_out[po + 1] = s1; A.D_int_out[po + 2] = s2; }
    }
    a[0] += q0; a[3] += q1; a[5] += q2;
  } else if (A.D_e) {
    const double d0 = A.D_e[pd], d1 = A.D_e[pd + 1], d2 = A.D_e[pd + 2];
    if (writer && A.D_int_out) { A.D_int_out[po] = d0; A.D_int_out[po + 1] = d1; A.D_int_out[po + 2] = d2; }
    a[0] += d0 * d0; a[3] += d1 * d1; a[5] += d2 * d2;
  }
}

__device__ __forceinline__ void store_ete_inverse(const BalArgs& A, int pt, const double (&ei)[6]) {
  if (A.etei) {
    double2* q = reinterpret_cast<double2*>(A.etei + int64_t(pt) * 6);
    tile_store<2>(q, make_double2(ei[0], ei[1])); tile_store<2>(q + 1, make_double2(ei[2], ei[3])); tile_store<2>(q + 2, make_double2(ei[4], ei[5]));
  }
  if (A.point_blocks) {  // dense 3x3 in the all-blocks diagonal store (CGNR JACOBI)
    double* o = A.point_blocks + (A.pt_diag_off ? A.pt_diag_off[pt] : int64_t(9) * pt);
    o[0] = ei[0]; o[1] = ei[1]; o[2] = ei[2]; o[3] = ei[1]; o[4] = ei[3]; o[5] = ei[4]; o[6] = ei[2]; o[7] = ei[4]; o[8] = ei[5];
  }
}

// Everything after the per-point quantities are known, for kInit.
template <bool LDS>
__device__ __forceinline__ void init_apply(const BalArgs& A, const Slot& s, int64_t sl, double b0, double b1,
                                           const double (&ei)[6], const double (&g)[3], double* acc) {
  double h[3];
  sym3_mul(ei, g, h);
  const double w0 = b0 - (s.e[0] * h[0] + s.e[1] * h[1] + s.e[2] * h[2]);
  const double w1 = b1 - (s.e[3] * h[0] + s.e[4] * h[1] + s.e[5] * h[2]);
  if (A.have_b) scatter_ft<LDS>(A, s, acc, w0, w1);
  if (A.Mo && s.valid) {  // M = I - E Ei E^T, symmetric 2x2
    const double r0[3] = {s.e[0], s.e[1], s.e[2]}, r1[3] = {s.e[3], s.e[4], s.e[5]};
    double q0[3], q1[3];
    sym3_mul(ei, r0, q0);
    sym3_mul(ei, r1, q1);
    double2* mo = reinterpret_cast<double2*>(A.Mo + 4 * (A.mo_index ? int64_t(A.mo_index[sl]) : sl));  // [record][4]: m00 m01 m11 m01, 32 B
    tile_store<2>(mo, make_double2(1.0 - (r0[0] * q0[0] + r0[1] * q0[1] + r0[2] * q0[2]), -(r0[0] * q1[0] + r0[1] * q1[1] + r0[2] * q1[2])));

    const double m01 = -(r0[0] * q1[0] + r0[1] * q1[1] + r0[2] * q1[2]);
    tile_store<2>(mo + 1, make_double2(1.0 - (r1[0] * q1[0] + r1[1] * q1[1] + r1[2] * q1[2]), m01));   // (m01 again: bal_camera_items_mfma_kernel)
  }
}

// What a streaming mode (kSx, kSpseZ, kJtJx) needs from global memory besides the slot itself.
struct StreamAux {
  double xc[9];            // the slot's camera part of x
  double ei[6];            // kSx / kSpseZ: (E^T E)^-1 of the slot's point
  double xa, xb, da, db;   // kJtJx, cooperative: scalars `lane` and `64 + lane` of the tile's point range of x, D
  double xp[3], dd[3];     // kJtJx, otherwise: the slot's own point part of x, D
  int64_t base;
  bool coop;
};

// Issues the loads of `aux`; consumes only the slot's index words (s.cam, s.pt).
template <int MODE>
__device__ __forceinline__ void load_aux(const BalArgs& A, const Slot& s, int lane, int npts, StreamAux& x) {
  load_xc(A, s.cam, x.xc);
  x.xa = x.xb = x.da = x.db = 0.0;
  x.base = 0;
  x.coop = false;
#pragma unroll
  for (int j = 0; j < 3; ++j) { x.xp[j] = 0.0; x.dd[j] = 0.0; }
  if constexpr (MODE == kSx || MODE == kSpseZ) {
    load_ete_inverse(A, s.pt, x.ei);
  } else if constexpr (MODE == kJtJx) {
    // Point-space x / D / y of a tile are ONE contiguous range when the layout is
    // points-then-cameras (pt_pos == nullptr): [3 p0, 3 p0 + 3 npts).  Then lane L loads and
    // stores scalars L (and 64 + L) of that range — dense 8-byte-per-lane accesses — and the
    // per-observation copies travel by wavefront shuffle, instead of three strided loads per
    // lane and three partial-wave scattered stores per point.
    x.coop = A.pt_pos == nullptr && !(A.flags & 1);
    if (x.coop) {
      const int n3 = 3 * npts;
      x.base = 3 * int64_t(__builtin_amdgcn_readfirstlane(s.pt));  // lane 0 of a normal tile is valid
      if (lane < n3) { x.xa = A.x_e[x.base + lane]; if (A.D_e) x.da = A.D_e[x.base + lane]; }
      if (n3 > 64 && lane + 64 < n3) { x.xb = A.x_e[x.base + 64 + lane]; if (A.D_e) x.db = A.D_e[x.base + 64 + lane]; }
    } else {
      const int po = pt_off(A, s.pt);
      x.xp[0] = A.x_e[po]; x.xp[1] = A.x_e[po + 1]; x.xp[2] = A.x_e[po + 2];
      if (A.D_e) { x.dd[0] = A.D_e[po]; x.dd[1] = A.D_e[po + 1]; x.dd[2] = A.D_e[po + 2]; }
    }
  }
}

// The arithmetic of a streaming mode on one normal tile; issues no global loads.
template <int MODE, bool LDS>
__device__ __forceinline__ void compute_stream(const BalArgs& A, const Slot& s, int lane, int span, int npts,
                                               const StreamAux& x, double* acc, double& dot) {
  if constexpr (MODE == kSx || MODE == kSpseZ) {
    double v[3];
    double t0, t1;
    f_times(s, x.xc, t0, t1);
    double u[3] = {s.e[0] * t0 + s.e[3] * t1, s.e[1] * t0 + s.e[4] * t1, s.e[2] * t0 + s.e[5] * t1};
    if (!s.valid) { u[0] = u[1] = u[2] = 0; }
    seg_allreduce<3>(u, lane, s.first, s.last, span);
    sym3_mul(x.ei, u, v);
    // kSx: F^T (F x - E (E^T E)^-1 E^T F x);  kSpseZ: only the second term, F^T E (E^T E)^-1 E^T F x
    // (ImplicitSchurComplement::InversePowerSeriesOperatorRightMultiplyAccumulate, :146-174)
    const double ev0 = s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
    const double ev1 = s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
    const double z0 = MODE == kSx ? t0 - ev0 : ev0;
    const double z1 = MODE == kSx ? t1 - ev1 : ev1;
    scatter_ft<LDS>(A, s, acc, z0, z1);
  } else {
    static_assert(MODE == kJtJx, "streaming modes");
    const int n3 = 3 * npts;
    double xp[3] = {x.xp[0], x.xp[1], x.xp[2]};
    if (x.coop) {
      const int li = s.valid ? 3 * s.pt - int(x.base) : 0;
      if (n3 <= 64) {
#pragma unroll
        for (int j = 0; j < 3; ++j) xp[j] = shfl_idx(x.xa, li + j);
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double va = shfl_idx(x.xa, (li + j) & 63), vb = shfl_idx(x.xb, (li + j) & 63);
          xp[j] = (li + j) < 64 ? va : vb;
        }
      }
    }
    double z0, z1;
    f_times(s, x.xc, z0, z1);
    z0 += s.e[0] * xp[0] + s.e[1] * xp[1] + s.e[2] * xp[2];
    z1 += s.e[3] * xp[0] + s.e[4] * xp[1] + s.e[5] * xp[2];
    scatter_ft<LDS>(A, s, acc, z0, z1);
    double w[3] = {s.e[0] * z0 + s.e[3] * z1, s.e[1] * z0 + s.e[4] * z1, s.e[2] * z0 + s.e[5] * z1};
    if (!s.valid) { w[0] = w[1] = w[2] = 0; }
    seg_scan<3>(w, lane, s.first, span);
    if (x.coop) {
      // lane L gathers component L % 3 of point L / 3 from that point's last lane and stores scalar L
      const int ta = (s.seg >> 17) & 63, tb = (s.seg >> 24) & 63;
      {
        const double v0 = shfl_idx(w[0], ta), v1 = shfl_idx(w[1], ta), v2 = shfl_idx(w[2], ta);
        const int c = lane % 3;
        const double v = c == 0 ? v0 : (c == 1 ? v1 : v2);
        if ((s.seg >> 23) & 1) { const double yv = v + x.da * x.da * x.xa; store_ye(A.y_e + x.base + lane, yv, ye_alt(A, s) + lane); dot += x.xa * yv; }
      }
      if (n3 > 64) {
        const double v0 = shfl_idx(w[0], tb), v1 = shfl_idx(w[1], tb), v2 = shfl_idx(w[2], tb);
        const int c = (lane + 1) % 3;  // (64 + lane) % 3
        const double v = c == 0 ? v0 : (c == 1 ? v1 : v2);
        if ((s.seg >> 30) & 1) { const double yv = v + x.db * x.db * x.xb; store_ye(A.y_e + x.base + 64 + lane, yv, ye_alt(A, s) + 64 + lane); dot += x.xb * yv; }
      }
    } else if (s.valid && lane == s.last) {
      const int po = pt_off(A, s.pt);
#pragma unroll
      for (int j = 0; j < 3; ++j) { const double yv = w[j] + x.dd[j] * x.dd[j] * xp[j]; A.y_e[po + j] = yv; dot += xp[j] * yv; }
    }
  }
}

template <int MODE, bool LDS, bool F32>
__device__ __forceinline__ void process_tile(const BalArgs& A, int64_t tile, int lane, int span, int npts, double* acc,
                                             double& lane_acc) {
  Slot s;
  load_slot<kCanGather<MODE>, F32>(A, tile, lane, s, kWantsB<MODE>);
  const int64_t sl = tile * kTile + lane;
  const int po = pt_off(A, s.pt);
  if constexpr (MODE == kSx || MODE == kSpseZ || MODE == kJtJx) {
    StreamAux x;
    load_aux<MODE>(A, s, lane, npts, x);
    compute_stream<MODE, LDS>(A, s, lane, span, npts, x, acc, lane_acc);
  } else if constexpr (MODE == kJtb) {
    scatter_ft<LDS>(A, s, acc, s.b0, s.b1);
    double w[3] = {s.e[0] * s.b0 + s.e[3] * s.b1, s.e[1] * s.b0 + s.e[4] * s.b1, s.e[2] * s.b0 + s.e[5] * s.b1};
    if (!s.valid) { w[0] = w[1] = w[2] = 0; }
    seg_scan<3>(w, lane, s.first, span);
    if (s.valid && lane == s.last) {
#pragma unroll
      for (int j = 0; j < 3; ++j) A.y_e[po + j] = w[j];
    }
  } else if constexpr (MODE == kInit || MODE == kEte) {
    double r[9];
    {
      double a[6];
      ete_of(s, a);
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] = s.valid ? a[i] : 0.0;
    }
    const double b0 = s.b0, b1 = s.b1;
    r[6] = s.valid ? s.e[0] * b0 + s.e[3] * b1 : 0.0;
    r[7] = s.valid ? s.e[1] * b0 + s.e[4] * b1 : 0.0;
    r[8] = s.valid ? s.e[2] * b0 + s.e[5] * b1 : 0.0;
    if constexpr (MODE == kInit) {
      seg_allreduce<9>(r, lane, s.first, s.last, span);
    } else {
      double a6[6] = {r[0], r[1], r[2], r[3], r[4], r[5]};
      seg_allreduce<6>(a6, lane, s.first, s.last, span);
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] = a6[i];
    }
    double a[6] = {r[0], r[1], r[2], r[3], r[4], r[5]}, ei[6];
    add_e_diagonal(A, s.pt, po, a, s.valid && lane == s.last);
    if (!s.valid) { a[0] = a[3] = a[5] = 1.0; a[1] = a[2] = a[4] = 0.0; }
    invert_spd3(a, ei);
    if (s.valid && lane == s.last) store_ete_inverse(A, s.pt, ei);
    if constexpr (MODE == kInit) {
      const double g[3] = {r[6], r[7], r[8]};
      init_apply<LDS>(A, s, sl, b0, b1, ei, g, acc);
    }
  } else if constexpr (MODE == kColNorm) {
    // diag(J^T J): BlockSparseMatrix::SquaredColumnNorm (I/block_sparse_matrix.cc:351-401), one pass
    double w[3] = {s.e[0] * s.e[0] + s.e[3] * s.e[3], s.e[1] * s.e[1] + s.e[4] * s.e[4], s.e[2] * s.e[2] + s.e[5] * s.e[5]};
    if (!s.valid) { w[0] = w[1] = w[2] = 0; }
    scatter_f_squares(s, acc);
    seg_scan<3>(w, lane, s.first, span);
    if (s.valid && lane == s.last) { A.y_e[po] = w[0]; A.y_e[po + 1] = w[1]; A.y_e[po + 2] = w[2]; }
  } else if constexpr (MODE == kJx) {
    // model cost change of a trust-region step: -(J x)'(f + J x / 2), I/trust_region_minimizer.cc:420-438;
    // no per-point or per-camera reduction, every tile (also those of long points) is independent
    double xc[9];
    load_xc(A, s.cam, xc);
    const double x0 = A.x_e[po], x1 = A.x_e[po + 1], x2 = A.x_e[po + 2];
    double m0, m1;
    f_times(s, xc, m0, m1);
    m0 += s.e[0] * x0 + s.e[1] * x1 + s.e[2] * x2;
    m1 += s.e[3] * x0 + s.e[4] * x1 + s.e[5] * x2;
    if (s.valid) lane_acc -= m0 * (s.b0 + 0.5 * m0) + m1 * (s.b1 + 0.5 * m1);
  } else if constexpr (MODE == kCgnrInit) {
    // CGNR set-up in one pass: rhs = J^T b (point part by segment sums, camera part scattered)
    // and, if requested, the JACOBI point blocks (E^T E + D^2)^-1.
    double r[9];
    {
      double a[6];
      ete_of(s, a);
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] = s.valid ? a[i] : 0.0;
    }
    r[6] = s.valid ? s.e[0] * s.b0 + s.e[3] * s.b1 : 0.0;
    r[7] = s.valid ? s.e[1] * s.b0 + s.e[4] * s.b1 : 0.0;
    r[8] = s.valid ? s.e[2] * s.b0 + s.e[5] * s.b1 : 0.0;
    scatter_ft<LDS>(A, s, acc, s.b0, s.b1);
    seg_scan<9>(r, lane, s.first, span);
    if (s.valid && lane == s.last) {
      A.y_e[po] = r[6]; A.y_e[po + 1] = r[7]; A.y_e[po + 2] = r[8];
      if (A.point_blocks) {
        double a[6] = {r[0], r[1], r[2], r[3], r[4], r[5]}, ei[6];
        add_e_diagonal(A, s.pt, po, a, true);
        invert_spd3(a, ei);
        store_ete_inverse(A, s.pt, ei);
      }
    }
  } else if constexpr (MODE == kBackSub) {
    double zc[9], ei[6];
    load_xc(A, s.cam, zc);
    load_ete_inverse(A, s.pt, ei);  // issued with the other loads, by all lanes, not behind the scan
    double t0, t1;
    f_times(s, zc, t0, t1);
    t0 = s.b0 - t0; t1 = s.b1 - t1;
    double u[3] = {s.e[0] * t0 + s.e[3] * t1, s.e[1] * t0 + s.e[4] * t1, s.e[2] * t0 + s.e[5] * t1};
    if (!s.valid) { u[0] = u[1] = u[2] = 0; }
    seg_scan<3>(u, lane, s.first, span);
    double v[3];
    sym3_mul(ei, u, v);  // the point's solution on the last lane of its segment
    if (s.valid && lane == s.last) {
      const double sg = A.negate_out ? -1.0 : 1.0;  // the LM step is -x
      A.y_e[po] = sg * v[0]; A.y_e[po + 1] = sg * v[1]; A.y_e[po + 2] = sg * v[2];
      if (A.negate_out && !isfinite(v[0] + v[1] + v[2])) atomicAdd(A.nonfinite, 1);
    }
    if (A.scalar_out) {
      // Model cost change of the trust-region step -x, fused: with m = J x of this observation
      // (F z + E y, y broadcast from the segment's last lane), -(J step)'(f + J step / 2) = m'(f - m / 2).
      // I/trust_region_minimizer.cc:420-438; saves the separate pass over J (kJx) per LM step.
#pragma unroll
      for (int j = 0; j < 3; ++j) v[j] = shfl_idx(v[j], s.last);
      const double m0 = (s.b0 - t0) + s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
      const double m1 = (s.b1 - t1) + s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
      if (s.valid) lane_acc += m0 * (s.b0 - 0.5 * m0) + m1 * (s.b1 - 0.5 * m1);
    }
  }
}

// A point with more than 64 observations: tiles [tile, tile+nt) belong to it alone.
// Sweep 1 accumulates the per-point sums over all its tiles, sweep 2 (only where the
// per-observation result depends on them) re-reads the tiles, which are L2-warm.
template <int MODE, bool LDS, bool F32>
__device__ __forceinline__ void process_long_point(const BalArgs& A, int64_t tile, int nt, int lane, double* acc, double& lane_acc) {
  Slot s;
  if constexpr (MODE == kColNorm) {
    double w[3] = {0, 0, 0};
    int pt = 0;
    for (int t = 0; t < nt; ++t) {
      load_slot<true, F32>(A, tile + t, lane, s, false);
      if (t == 0) pt = __shfl(s.pt, 0, 64);
      scatter_f_squares(s, acc);
      if (s.valid) { w[0] += s.e[0] * s.e[0] + s.e[3] * s.e[3]; w[1] += s.e[1] * s.e[1] + s.e[4] * s.e[4]; w[2] += s.e[2] * s.e[2] + s.e[5] * s.e[5]; }
    }
    wave_allreduce<3>(w);
    if (lane == 0) { const int po = pt_off(A, pt); A.y_e[po] = w[0]; A.y_e[po + 1] = w[1]; A.y_e[po + 2] = w[2]; }
    return;
  }
  if constexpr (MODE == kSx || MODE == kBackSub || MODE == kSpseZ) {
    double u[3] = {0, 0, 0};
    int pt = 0;
    for (int t = 0; t < nt; ++t) {
      load_slot<kCanGather<MODE>, F32>(A, tile + t, lane, s, kWantsB<MODE>);
      if (t == 0) pt = __shfl(s.pt, 0, 64);
      double xc[9];
      load_xc(A, s.cam, xc);
      double t0, t1;
      f_times(s, xc, t0, t1);
      if constexpr (MODE == kBackSub) { t0 = s.b0 - t0; t1 = s.b1 - t1; }
      if (s.valid) { u[0] += s.e[0] * t0 + s.e[3] * t1; u[1] += s.e[1] * t0 + s.e[4] * t1; u[2] += s.e[2] * t0 + s.e[5] * t1; }
    }
    wave_allreduce<3>(u);
    double ei[6], v[3];
    load_ete_inverse(A, pt, ei);
    sym3_mul(ei, u, v);
    if constexpr (MODE == kBackSub) {
      if (lane == 0) {
        const int po = pt_off(A, pt);
        const double sg = A.negate_out ? -1.0 : 1.0;
        A.y_e[po] = sg * v[0]; A.y_e[po + 1] = sg * v[1]; A.y_e[po + 2] = sg * v[2];
        if (A.negate_out && !isfinite(v[0] + v[1] + v[2])) atomicAdd(A.nonfinite, 1);
      }
      if (A.scalar_out) {  // fused model cost change: second sweep over the point's (L2-warm) tiles
        for (int t = 0; t < nt; ++t) {
          load_slot<false, F32>(A, tile + t, lane, s, true, false);
          double xc[9];
          load_xc(A, s.cam, xc);
          double t0, t1;
          f_times(s, xc, t0, t1);
          const double m0 = t0 + s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
          const double m1 = t1 + s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
          if (s.valid) lane_acc += m0 * (s.b0 - 0.5 * m0) + m1 * (s.b1 - 0.5 * m1);
        }
      }
    } else {
      for (int t = 0; t < nt; ++t) {
        load_slot<false, F32>(A, tile + t, lane, s, false, false);
        double xc[9];
        load_xc(A, s.cam, xc);
        double t0, t1;
        f_times(s, xc, t0, t1);
        const double ev0 = s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
        const double ev1 = s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
        scatter_ft<LDS>(A, s, acc, MODE == kSx ? t0 - ev0 : ev0, MODE == kSx ? t1 - ev1 : ev1);
      }
    }
  } else if constexpr (MODE == kJtJx || MODE == kJtb) {
    double w[3] = {0, 0, 0}, xp[3] = {0, 0, 0};
    int pt = 0, po = 0;
    for (int t = 0; t < nt; ++t) {
      load_slot<kCanGather<MODE>, F32>(A, tile + t, lane, s, kWantsB<MODE>);
      if (t == 0) {
        pt = __shfl(s.pt, 0, 64);
        po = pt_off(A, pt);
        if constexpr (MODE == kJtJx) { xp[0] = A.x_e[po]; xp[1] = A.x_e[po + 1]; xp[2] = A.x_e[po + 2]; }
      }
      double z0, z1;
      if constexpr (MODE == kJtJx) {
        double xc[9];
        load_xc(A, s.cam, xc);
        f_times(s, xc, z0, z1);
        z0 += s.e[0] * xp[0] + s.e[1] * xp[1] + s.e[2] * xp[2];
        z1 += s.e[3] * xp[0] + s.e[4] * xp[1] + s.e[5] * xp[2];
      } else {
        z0 = s.b0; z1 = s.b1;
      }
      scatter_ft<LDS>(A, s, acc, z0, z1);
      if (s.valid) { w[0] += s.e[0] * z0 + s.e[3] * z1; w[1] += s.e[1] * z0 + s.e[4] * z1; w[2] += s.e[2] * z0 + s.e[5] * z1; }
    }
    wave_allreduce<3>(w);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double d = 0;
        if (MODE == kJtJx && A.D_e) { d = A.D_e[po + j]; d = d * d * xp[j]; }
        A.y_e[po + j] = w[j] + d;
        if (MODE == kJtJx) lane_acc += xp[j] * (w[j] + d);  // x_e . y_e share of CG's p.q
      }
    }
  } else if constexpr (MODE == kInit || MODE == kEte || MODE == kCgnrInit) {
    double r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int pt = 0;
    for (int t = 0; t < nt; ++t) {
      load_slot<kCanGather<MODE>, F32>(A, tile + t, lane, s, kWantsB<MODE>);
      if (t == 0) pt = __shfl(s.pt, 0, 64);
      if constexpr (MODE == kCgnrInit) scatter_ft<LDS>(A, s, acc, s.b0, s.b1);
      if (s.valid) {
        double a[6];
        ete_of(s, a);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] += a[i];
        r[6] += s.e[0] * s.b0 + s.e[3] * s.b1; r[7] += s.e[1] * s.b0 + s.e[4] * s.b1; r[8] += s.e[2] * s.b0 + s.e[5] * s.b1;
      }
    }
    wave_allreduce<9>(r);
    const int po = pt_off(A, pt);
    double a[6] = {r[0], r[1], r[2], r[3], r[4], r[5]}, ei[6];
    add_e_diagonal(A, pt, po, a, lane == 0);
    invert_spd3(a, ei);
    if constexpr (MODE == kCgnrInit) {
      if (lane == 0) {
        A.y_e[po] = r[6]; A.y_e[po + 1] = r[7]; A.y_e[po + 2] = r[8];
        if (A.point_blocks) store_ete_inverse(A, pt, ei);
      }
    } else {
      if (lane == 0) store_ete_inverse(A, pt, ei);
    }
    if constexpr (MODE == kInit) {
      const double g[3] = {r[6], r[7], r[8]};
      for (int t = 0; t < nt; ++t) {
        load_slot<false, F32>(A, tile + t, lane, s, true, false);  // second sweep: from the tiles just written
        const int64_t sl = (tile + t) * kTile + lane;
        init_apply<LDS>(A, s, sl, s.b0, s.b1, ei, g, acc);
      }
    }
  }
}

// The long points of the unpipelined kernels, in the same ROUNDS as the streaming kernels' (plan.cc, compute_long_round): the waves of
// a workgroup — in sub-workgroups of kRoundWaves — take one tile each, exchange the tile's sums (3 or 9 doubles) through LDS and finish
// the tile from registers; a point of more than kRoundWaves tiles runs sum rounds, then (where the per-observation result needs the
// sums: S.x, Init's M_o and rhs, the model cost of the back-substitution) apply rounds that re-read its tiles.  Every wave of the
// workgroup runs the same number of rounds (the barriers are the workgroup's): a sub-workgroup out of rounds idles.
template <int MODE, bool LDS, int BLOCK, bool F32>
__device__ __forceinline__ void fused_long_rounds(const BalArgs& A, int lane, bool grouped, int range, double* acc, double& lane_acc) {
  constexpr int NX = (MODE == kInit || MODE == kEte || MODE == kCgnrInit) ? 9 : 3;
  constexpr int SUBS = BLOCK / 64 / kRoundWaves;
  constexpr bool kTwoPhase = (MODE == kSx || MODE == kSpseZ || MODE == kInit || MODE == kBackSub);
  static_assert(SUBS >= 1, "a workgroup holds at least one round");
  __shared__ double xr[SUBS][kRoundWaves][NX];
  const int wave_all = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int sub = wave_all / kRoundWaves, wave = wave_all % kRoundWaves;
  // a workgroup that owns a hybrid group takes the group's sequences; otherwise the workgroups share ALL sequences (also those of a
  // hybrid plan: the passes that scatter nothing walk every tile)
  const int64_t qe = grouped ? int64_t(A.round_ptr[range + 1]) : int64_t(A.n_seq);
  const int64_t qstride = grouped ? 1 : int64_t(gridDim.x) * SUBS;
  const int64_t q_first = grouped ? int64_t(A.round_ptr[range]) : logical_workgroup() * SUBS;
  // rounds of the busiest sub-workgroup (a hybrid group's sequences all go to sub-workgroup 0: scattering modes run 8 waves there)
  int64_t n_iter = 0;
  for (int j = 0; j < (grouped ? 1 : SUBS); ++j) {
    int64_t n = 0;
    if (grouped) { if (q_first < qe) n = A.seq_ptr[qe] - A.seq_ptr[q_first]; }
    else for (int64_t q = q_first + j; q < qe; q += qstride) n += A.seq_ptr[q + 1] - A.seq_ptr[q];
    n_iter = max(n_iter, n);
  }
  int64_t it_q = q_first + (grouped ? 0 : sub);
  bool it_real = grouped ? (sub == 0 && it_q < qe) : it_q < qe;
  int64_t it_r = 0, it_end = 0;
  if (it_real) { it_r = A.seq_ptr[it_q]; it_end = A.seq_ptr[grouped ? qe : it_q + 1]; if (grouped) it_q = qe - 1; }
  double carry[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) carry[i] = 0.0;
  for (int64_t iter = 0; iter < n_iter; ++iter) {
    uint32_t word = kRoundIdle, word0 = kRoundIdle;
    int flag = 0;
    if (it_real) {
      word = A.round_word[it_r * kRoundWaves + wave];
      word0 = A.round_word[it_r * kRoundWaves];
      flag = A.round_flag[it_r];
      if (it_r + 1 < it_end) ++it_r;
      else if (it_q + qstride < qe) { it_q += qstride; it_r = A.seq_ptr[it_q]; it_end = A.seq_ptr[it_q + 1]; }
      else it_real = false;
    }
    const bool active = word != kRoundIdle;
    const int64_t tile = int64_t(word & 0x3FFFFFFu);
    const int w0 = flag ? 0 : (active ? int((word >> 26) & 7u) : wave);
    const int cnt = flag ? (word0 != kRoundIdle ? int((word0 >> 29) & 7u) + 1 : 1) : (active ? int((word >> 29) & 7u) + 1 : 1);
    const bool apply_round = (flag & kRoundApply) != 0;
    Slot s;
    s.valid = false; s.pt = 0; s.cam = 0; s.acc = kSlotSpill; s.b0 = s.b1 = 0.0; s.zbase = 0; s.slot = 0;
    double part[NX], t0 = 0.0, t1 = 0.0, xp[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < NX; ++i) part[i] = 0.0;
    int pt = 0, po = 0;
    if (active && (!apply_round || kTwoPhase)) {
      // (an apply round re-reads the tiles the sum rounds wrote: never a gather)
      load_slot<kCanGather<MODE>, F32>(A, tile, lane, s, kWantsB<MODE>, !apply_round);
      pt = __builtin_amdgcn_readfirstlane(s.pt);   // one segment: every valid lane holds the point
      po = pt_off(A, pt);
    }
    // ---- this tile's share of the point's sums
    if (active && !apply_round) {
      if constexpr (MODE == kColNorm) {
        scatter_f_squares(s, acc);
        if (s.valid) { part[0] = s.e[0] * s.e[0] + s.e[3] * s.e[3]; part[1] = s.e[1] * s.e[1] + s.e[4] * s.e[4]; part[2] = s.e[2] * s.e[2] + s.e[5] * s.e[5]; }
      } else if constexpr (MODE == kSx || MODE == kSpseZ || MODE == kBackSub) {
        double xc[9];
        load_xc(A, s.cam, xc);
        f_times(s, xc, t0, t1);
        if constexpr (MODE == kBackSub) { t0 = s.b0 - t0; t1 = s.b1 - t1; }
        if (s.valid) { part[0] = s.e[0] * t0 + s.e[3] * t1; part[1] = s.e[1] * t0 + s.e[4] * t1; part[2] = s.e[2] * t0 + s.e[5] * t1; }
      } else if constexpr (MODE == kJtJx || MODE == kJtb) {
        double z0 = s.b0, z1 = s.b1;
        if constexpr (MODE == kJtJx) {
          xp[0] = A.x_e[po]; xp[1] = A.x_e[po + 1]; xp[2] = A.x_e[po + 2];
          double xc[9];
          load_xc(A, s.cam, xc);
          f_times(s, xc, z0, z1);
          z0 += s.e[0] * xp[0] + s.e[1] * xp[1] + s.e[2] * xp[2];
          z1 += s.e[3] * xp[0] + s.e[4] * xp[1] + s.e[5] * xp[2];
        }
        scatter_ft<LDS>(A, s, acc, z0, z1);
        if (s.valid) { part[0] = s.e[0] * z0 + s.e[3] * z1; part[1] = s.e[1] * z0 + s.e[4] * z1; part[2] = s.e[2] * z0 + s.e[5] * z1; }
      } else {
        static_assert(MODE == kInit || MODE == kEte || MODE == kCgnrInit, "set-up modes");
        if constexpr (MODE == kCgnrInit) scatter_ft<LDS>(A, s, acc, s.b0, s.b1);
        if (s.valid) {
          double a[6];
          ete_of(s, a);
#pragma unroll
          for (int i = 0; i < 6; ++i) part[i] = a[i];
          part[6] = s.e[0] * s.b0 + s.e[3] * s.b1; part[7] = s.e[1] * s.b0 + s.e[4] * s.b1; part[8] = s.e[2] * s.b0 + s.e[5] * s.b1;
        }
      }
      wave_allreduce<NX>(part);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[sub][wave][i] = part[i];
      }
    }
    __syncthreads();
    double tot[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) tot[i] = 0.0;
    if ((active || flag) && !apply_round) {   // (an idle wave of a sum round adds up too: it may hold a tile in the apply rounds)
      for (int k = 0; k < cnt; ++k) {
#pragma unroll
        for (int i = 0; i < NX; ++i) tot[i] += xr[sub][w0 + k][i];
      }
    }
    __syncthreads();
    if (flag & kRoundSum) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { carry[i] += tot[i]; tot[i] = carry[i]; }
      if (!(flag & kRoundLast)) continue;
      if constexpr (!kTwoPhase) {
#pragma unroll
        for (int i = 0; i < NX; ++i) carry[i] = 0.0;
      }
    } else if (apply_round) {
#pragma unroll
      for (int i = 0; i < NX; ++i) tot[i] = carry[i];
      if (flag & kRoundLast) {
#pragma unroll
        for (int i = 0; i < NX; ++i) carry[i] = 0.0;
      }
      if constexpr (!kTwoPhase) continue;
    }
    if (!active) continue;
    // ---- the point's sums are in tot.  `finish`: what is stored once per point (first wave of the point; a point of more than
    // kRoundWaves tiles: in its last sum round).  `apply`: what every tile does with the sums (such a point: in its apply rounds).
    const bool finish = wave == w0 && lane == 0 && !apply_round;
    const bool apply = flag == 0 || apply_round;
    if constexpr (MODE == kColNorm || MODE == kJtb) {
      if (finish) { A.y_e[po] = tot[0]; A.y_e[po + 1] = tot[1]; A.y_e[po + 2] = tot[2]; }
    } else if constexpr (MODE == kJtJx) {
      if (finish) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          double d = 0;
          if (A.D_e) { d = A.D_e[po + j]; d = d * d * xp[j]; }
          A.y_e[po + j] = tot[j] + d;
          lane_acc += xp[j] * (tot[j] + d);
        }
      }
    } else if constexpr (MODE == kSx || MODE == kSpseZ || MODE == kBackSub) {
      double ei[6], v[3];
      load_ete_inverse(A, pt, ei);
      const double u[3] = {tot[0], tot[1], tot[2]};
      sym3_mul(ei, u, v);
      if constexpr (MODE == kBackSub) {
        if (finish) {
          const double sg = A.negate_out ? -1.0 : 1.0;
          A.y_e[po] = sg * v[0]; A.y_e[po + 1] = sg * v[1]; A.y_e[po + 2] = sg * v[2];
          if (A.negate_out && !isfinite(v[0] + v[1] + v[2])) atomicAdd(A.nonfinite, 1);
        }
        if (!A.scalar_out || !apply) continue;
      } else if (!apply) continue;
      if (apply_round) {   // the sums came from earlier rounds: this tile's F x again
        double xc[9];
        load_xc(A, s.cam, xc);
        f_times(s, xc, t0, t1);
        if constexpr (MODE == kBackSub) { t0 = s.b0 - t0; t1 = s.b1 - t1; }
      }
      const double ev0 = s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
      const double ev1 = s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
      if constexpr (MODE == kBackSub) {   // fused model cost change: J x = F z + E x_e with F z = b - t
        const double m0 = (s.b0 - t0) + ev0, m1 = (s.b1 - t1) + ev1;
        if (s.valid) lane_acc += m0 * (s.b0 - 0.5 * m0) + m1 * (s.b1 - 0.5 * m1);
      } else {
        scatter_ft<LDS>(A, s, acc, MODE == kSx ? t0 - ev0 : ev0, MODE == kSx ? t1 - ev1 : ev1);
      }
    } else {
      double a[6] = {tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]}, ei[6];
      add_e_diagonal(A, pt, po, a, finish);
      invert_spd3(a, ei);
      if constexpr (MODE == kCgnrInit) {
        if (finish) {
          A.y_e[po] = tot[6]; A.y_e[po + 1] = tot[7]; A.y_e[po + 2] = tot[8];
          if (A.point_blocks) store_ete_inverse(A, pt, ei);
        }
      } else {
        if (finish) store_ete_inverse(A, pt, ei);
      }
      if constexpr (MODE == kInit) {
        if (apply) {
          const double g[3] = {tot[6], tot[7], tot[8]};
          init_apply<LDS>(A, s, tile * kTile + lane, s.b0, s.b1, ei, g, acc);
        }
      }
    }
  }
}

template <int MODE, bool LDS, int BLOCK, bool F32>
__global__ __launch_bounds__(BLOCK) void bal_fused_kernel(BalArgs A) {
  extern __shared__ double lds_acc[];
  if (A.status && *A.status != 0) return;  // CG already terminated: nothing to do
  if (A.run_after_cg && !CgStatusAllowsSolution(*A.run_after_cg)) return;  // speculative tail: CG has not ended (or failed)
  constexpr bool kScatters = (MODE == kSx || MODE == kJtJx || MODE == kJtb || MODE == kInit || MODE == kCgnrInit || MODE == kColNorm || MODE == kSpseZ);
  double* acc = nullptr;
  if constexpr (kScatters) {
    // every camera's accumulator row (LDS), or the hybrid rows of this workgroup — hyb_rows of them, then one strip per wave (scatter_ft)
    acc = lds_acc;
    const int n_acc = LDS ? A.n_f9 : 9 * A.hyb_rows;
    for (int i = threadIdx.x; i < n_acc; i += BLOCK) acc[i] = 0.0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  double lane_acc = 0.0;
  if constexpr (MODE == kBackSub) {  // y_f = z (:238-242), a few KB: no separate copy launch
    if (A.copy_dst)
      for (int i = blockIdx.x * BLOCK + threadIdx.x; i < A.copy_n; i += gridDim.x * BLOCK) {
        const double z = A.copy_src[i];
        A.copy_dst[i] = A.negate_out ? -z : z;
        if (A.negate_out && !isfinite(z)) atomicAdd(A.nonfinite, 1);
      }
  }
  // grid-strided over [tile_begin, tile_end); hybrid accumulation: workgroup g owns the tiles of group g (plan.cc)
  const bool grouped = kScatters && !LDS && A.grp_tile_ptr != nullptr;
  const int64_t tile0 = grouped ? A.grp_tile_ptr[blockIdx.x] + (threadIdx.x >> 6) : A.tile_begin + logical_workgroup() * (BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = grouped ? BLOCK / 64 : int64_t(gridDim.x) * (BLOCK / 64);
  const int64_t tile_end = grouped ? A.grp_tile_ptr[blockIdx.x + 1] : (A.tile_end > 0 ? A.tile_end : A.n_tiles);
  for (int64_t tile = tile0; tile < tile_end; tile += nwaves) {
    const int kind = A.tile_kind[tile];
    const int aux = A.tile_aux[tile];
    if constexpr (MODE == kJx) {
      process_tile<MODE, LDS, F32>(A, tile, lane, 1, 0, acc, lane_acc);
    } else {
      if (kind == 2 || kind == 3) continue;   // 3: the head of a long point that is taken in rounds (below)
      if (kind == 0) process_tile<MODE, LDS, F32>(A, tile, lane, aux & 0xff, aux >> 8, acc, lane_acc);
      else process_long_point<MODE, LDS, F32>(A, tile, aux, lane, acc, lane_acc);
    }
  }
  if constexpr (MODE != kJx) {
    if (A.round_word) fused_long_rounds<MODE, LDS, BLOCK, F32>(A, lane, grouped, grouped ? int(blockIdx.x) : 0, acc, lane_acc);   // (workgroup-uniform)
  }
  double* const scalar_dst = MODE == kJtJx ? A.pq_out : A.scalar_out;
  if ((MODE == kJx || MODE == kBackSub || MODE == kJtJx) && scalar_dst) {  // one partial per workgroup, summed in fixed order by the caller
    __shared__ double red[16];
    double v = lane_acc;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
      for (int i = 0; i < BLOCK / 64; ++i) t += red[i];
      if (MODE == kJtJx && A.pq_accumulate) t += scalar_dst[blockIdx.x];
      scalar_dst[blockIdx.x] = t;
    }
  }
  if constexpr (kScatters && LDS) {
    __syncthreads();
    double* out = A.partials + int64_t(blockIdx.x) * A.n_f9;
    for (int i = threadIdx.x; i < A.n_f9; i += BLOCK) out[i] = acc[i];
  } else if constexpr (kScatters) {
    if (A.hyb_rows > 0) {  // this workgroup's accumulator rows join the spilled rows in the ring: the second pass sums both
      __syncthreads();
      double* out = A.zbuf + 9 * (A.z_flush_row0 + int64_t(blockIdx.x) * A.hyb_rows);
      for (int i = threadIdx.x; i < 9 * A.hyb_rows; i += BLOCK) out[i] = acc[i];
    }
  }
}

// Software-pipelined variant of the streaming modes (kSx, kSpseZ, kJtJx; packed fp64 tiles,
// points-then-cameras vector layout).  bal_fused_kernel runs load -> wait -> compute -> store per
// tile and leans on the other 3 waves of its SIMD to keep HBM busy; the SQ counters
// (profiles/r01d_pmc_sq_*.txt) show its waves parked in s_waitcnt 45-60 % of their cycles, and
// switching the 24 MB of point-space stores of JtJx off saved 12 % — vmcnt retires in order, so
// the write acknowledgement of tile N sat in front of the loads of tile N+1.  Here a wave always
// has the next tile in flight while it computes (8 waves per CU, 512 threads, <= 256 VGPRs):
//   stage N:  issue index words (N+2) | issue pairs (N+1) | issue aux (N+1), addressed through
//             index words (N+1) that were issued a stage ago | compute (N) | store (N)
// so compute(N) only waits for loads issued during stage N-1, and a store has a whole stage to
// retire before anything queued behind it is needed.
// s_waitcnt counts are static, so every path through a stage must issue the SAME loads in the
// same order or the compiler has to fall back to vmcnt(0): no load here sits under a
// wave-uniform condition (the host picks this kernel only when x_f is unpadded, the layout is
// contiguous and — kJtJx — D is given), a wave past its last tile re-issues that tile, and
// tiles of long points run through compute() with every lane invalid.
template <int MODE>
__device__ __forceinline__ void issue_aux(const BalArgs& A, const Slot& s, int lane, int npts, StreamAux& x) {
  const double* xf = A.x_f + 9 * int64_t(s.cam);
#pragma unroll
  for (int k = 0; k < 9; ++k) x.xc[k] = xf[k];
  x.xa = x.xb = x.da = x.db = 0.0;
  x.base = 0;
  x.coop = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) { x.xp[j] = 0.0; x.dd[j] = 0.0; }
  if constexpr (MODE == kSx || MODE == kSpseZ) {
    load_ete_inverse(A, s.pt, x.ei);
  } else {
    const int n3 = 3 * npts;
    x.base = 3 * int64_t(__builtin_amdgcn_readfirstlane(s.pt));
    // per-lane predicates only mask lanes, the four load instructions are always issued
    // (lanes past the range read scalar 0 of it; nothing downstream looks at those lanes, and
    // selecting zeros here would consume the loads — a wait — in the issue phase)
    const bool a = lane < n3, b = lane + 64 < n3;
    const int64_t ia = x.base + (a ? lane : 0), ib = x.base + (b ? 64 + lane : 0);
    x.xa = load_xe(A.x_e + ia); x.da = load_xe(A.D_e + ia); x.xb = load_xe(A.x_e + ib); x.db = load_xe(A.D_e + ib);
  }
}

// The rest of CG iteration `it` by ONE workgroup, for a camera space of at most kCgTailMax scalars (CgTail, device.h): q = S p from the
// workgroups' partial sums (+ D_f^2 p), then ConjugateGradientsSolver's step exactly as cg_update_kernel and
// cg_finalize_direction_kernel split it (I/conjugate_gradients_solver.h:190-290): p.q, alpha, x += alpha p, r -= alpha q, z = M^-1 r,
// Q1, |r|, the termination tests, beta, p = z + beta p — same scalars, same failure codes, same double-buffered rho / Q0.
// `lds`: the workgroup's accumulator array (n_f9 doubles, flushed already): scratch for q, then for the new r.
template <int BLOCK>
__device__ __forceinline__ void cg_iteration_tail(const BalArgs& A, double* lds) {
  constexpr int PER = (kCgTailMax + BLOCK - 1) / BLOCK;
  __shared__ double sh[3][BLOCK / 64];
  const CgTail& T = A.tail;
  CgScalars& S = *T.S;
  const int n = A.n_f9, it = T.it, nparts = int(gridDim.x);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto sum3 = [&](double& a, double& b, double& c) {   // over the workgroup, fixed order; every thread gets the sums
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); c += __shfl_xor(c, m, 64); }
    __syncthreads();
    if (lane == 0) { sh[0][wv] = a; sh[1][wv] = b; sh[2][wv] = c; }
    __syncthreads();
    a = b = c = 0.0;
    for (int w = 0; w < BLOCK / 64; ++w) { a += sh[0][w]; b += sh[1][w]; c += sh[2][w]; }
  };
  // everything this kernel reads of S, before anything of it is written
  const int staged = S.fail_dir;
  const double rho = S.rho_pp[it & 1], Q0 = S.Q0_pp[it & 1], q_tol = S.q_tol, tol_r = S.tol_r;
  const int min_it = S.min_it, max_it = S.max_it;
  // everything the step reads of the CG vectors and of M^-1 is requested first, in one go with the partial sums: the tail is a chain
  // of memory round trips otherwise (11 us, as long as the three launches it replaces)
  double pv[PER], dv[PER], xo[PER], ro[PER], bv[PER], mrow[PER][9];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * BLOCK;
    const bool in = i < n;
    const int ii = in ? i : 0;
    pv[k] = T.p[ii]; dv[k] = T.D_f ? T.D_f[ii] : 0.0; xo[k] = T.x[ii]; ro[k] = T.r[ii]; bv[k] = T.rhs[ii];
    const double* m = T.blocks + int64_t(9) * ii;   // row (i % 9) of camera i / 9: 81 c + 9 a = 9 i
#pragma unroll
    for (int j = 0; j < 9; ++j) mrow[k][j] = m[j];
    if (!in) pv[k] = 0.0;
  }
  // the partial sums, all threads at once and every load of a thread independent of the others: (element i, subset sp) adds up the
  // partials sp, sp + ns, ... (at most kCgTailLoads of them: run_cg's condition) — ONE memory round trip; a thread per element walking
  // the partials alone was 35 us, batches of four dependent on each other 6 us
  const int ns = max(1, BLOCK / n);
  double* part = lds + n;   // [ns][n], behind the accumulator array (launch_stream2 sizes the LDS for it)
  {
    const int idx = threadIdx.x;
    const bool mine = idx < ns * n;
    const int sp = mine ? idx / n : 0, i = mine ? idx - sp * n : 0;
    double v[kCgTailLoads];
#pragma unroll
    for (int j = 0; j < kCgTailLoads; ++j) {
      const int w = sp + j * ns;
      v[j] = (mine && w < nparts) ? A.partials[int64_t(w) * n + i] : 0.0;
    }
#pragma unroll
    for (int h = kCgTailLoads / 2; h >= 1; h >>= 1) {
#pragma unroll
      for (int j = 0; j < h; ++j) v[j] += v[j + h];
    }
    if (mine) part[idx] = v[0];
  }
  __syncthreads();
  double qv[PER];
  double pq = 0, unused0 = 0, unused1 = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * BLOCK;
    qv[k] = 0.0;
    if (i < n) {
      double q = 0;
      for (int sp = 0; sp < ns; ++sp) q += part[sp * n + i];
      q += dv[k] * dv[k] * pv[k];
      qv[k] = q;
      pq += pv[k] * q;
    }
  }
  sum3(pq, unused0, unused1);
  if (staged) pq = 1.0;
  int fail = staged;
  double alpha = 0;
  if (!fail) {
    if (pq <= 0 || isinf(pq)) fail = kCgIndefinite;
    else { alpha = rho / pq; if (isinf(alpha)) fail = kCgFailAlpha; }
  }
  if (threadIdx.x == 0) { S.pq = pq; S.alpha = alpha; S.rho_new = rho; S.fail_step = fail; }
  if (fail) {
    if (threadIdx.x == 0) S.status = fail;
    return;
  }
  double q1 = 0, rr = 0, rz = 0, rn[PER], zv[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * BLOCK;
    rn[k] = 0.0;
    if (i < n) {
      const double xv = xo[k] + alpha * pv[k];
      const double rv = ro[k] - alpha * qv[k];
      T.x[i] = xv; T.r[i] = rv;
      rn[k] = rv;
      lds[i] = rv;
      q1 -= xv * (bv[k] + rv);
      rr += rv * rv;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * BLOCK;
    zv[k] = 0.0;
    if (i < n) {
      const int c = i / 9;
      double t = 0;
#pragma unroll
      for (int j = 0; j < 9; ++j) t += mrow[k][j] * lds[9 * c + j];
      zv[k] = t;
      T.z[i] = t;
      rz += rn[k] * t;
    }
  }
  sum3(q1, rr, rz);
  const double Q1 = q1, rho_next = rz;
  const double norm_r = sqrt(rr);
  const double zeta = it * (Q1 - Q0) / Q1;
  int status = kCgRunning;
  if (zeta < q_tol && it >= min_it) status = kCgConvergedZeta;
  else if (norm_r <= tol_r && it >= min_it) status = kCgConvergedResidual;
  else if (it >= max_it) status = kCgMaxIterations;
  int fail_dir = 0;
  double beta = 0.0;
  if (status == kCgRunning) {
    auto zero_or_inf = [](double v) { return v == 0.0 || isinf(v); };
    if (zero_or_inf(rho_next)) fail_dir = kCgFailRho;
    else { beta = rho_next / rho; if (zero_or_inf(beta)) fail_dir = kCgFailBeta; }
    if (!fail_dir) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * BLOCK;
        if (i < n) T.p[i] = zv[k] + beta * pv[k];
      }
    }
  }
  if (threadIdx.x == 0) {
    S.Q1 = Q1; S.zeta = zeta; S.norm_r = norm_r;
    if (status == kCgConvergedZeta) { S.status = status; return; }   // Q0 keeps its value, as in the reference (:273-284)
    S.Q0 = Q1;
    S.Q0_pp[(it + 1) & 1] = Q1;
    if (status != kCgRunning) { S.status = status; return; }
    S.rho = rho; S.rho_new = rho_next; S.beta = beta;
    S.rho_pp[(it + 1) & 1] = rho_next;
    S.fail_dir = fail_dir;
    S.iter = it + 1;
  }
}

// One tile of a long point inside a ROUND (plan.cc): the kRoundWaves waves of the workgroup each hold one tile in registers, the waves
// [w0, w0 + cnt) of one point leave their tile sums in LDS (`red`, double-buffered by round parity: one barrier per round), every
// wave adds them up in the same order and finishes its own tile.  A point of up to 512 observations is read ONCE, a tile per wave in
// flight, where one wave walked the point's tiles twice with nothing in flight (process_long_point: 0.20 - 0.25 of the HBM peak on
// graphs of long tracks).  `word`: the wave's round word (kRoundIdle: no tile — every lane is invalid, the loads were issued anyway).
// A point of more than kRoundWaves tiles (flag != 0, the round is its own: waves [0, cnt0), cnt0 from the round's first word): in its
// kRoundSum rounds every wave adds the round's total to `carry`; its kRoundApply rounds finish the (L2-warm) tiles with that sum.
// JtJx has nothing to finish: its sum rounds do all the work, the last one stores y_e.
template <int MODE, bool LDS>
__device__ __forceinline__ void compute_long_round(const BalArgs& A, const Slot& s, int lane, const StreamAux& x, uint32_t word, uint32_t word0,
                                                   int flag, int wave, double (*red)[3], double (&carry)[3], double* acc, double& dot) {
  const bool active = word != kRoundIdle;
  const int w0 = flag ? 0 : (active ? int((word >> 26) & 7u) : wave);
  const int cnt = flag ? int((word0 >> 29) & 7u) + 1 : (active ? int((word >> 29) & 7u) + 1 : 1);
  if constexpr (MODE == kSx || MODE == kSpseZ) {
    double t0, t1;
    f_times(s, x.xc, t0, t1);
    double u[3] = {s.e[0] * t0 + s.e[3] * t1, s.e[1] * t0 + s.e[4] * t1, s.e[2] * t0 + s.e[5] * t1};
    if (!s.valid) { u[0] = u[1] = u[2] = 0; }
    if (!(flag & kRoundApply)) {
      wave_allreduce<3>(u);
      if (lane == 0) { red[wave][0] = u[0]; red[wave][1] = u[1]; red[wave][2] = u[2]; }
    }
    __syncthreads();
    if (flag & kRoundApply) {
      u[0] = carry[0]; u[1] = carry[1]; u[2] = carry[2];
      if (flag & kRoundLast) { carry[0] = carry[1] = carry[2] = 0; }
    } else {
      u[0] = u[1] = u[2] = 0;
      for (int k = 0; k < cnt; ++k) { u[0] += red[w0 + k][0]; u[1] += red[w0 + k][1]; u[2] += red[w0 + k][2]; }
      if (flag & kRoundSum) { carry[0] += u[0]; carry[1] += u[1]; carry[2] += u[2]; return; }
    }
    double v[3];
    sym3_mul(x.ei, u, v);
    const double ev0 = s.e[0] * v[0] + s.e[1] * v[1] + s.e[2] * v[2];
    const double ev1 = s.e[3] * v[0] + s.e[4] * v[1] + s.e[5] * v[2];
    scatter_ft<LDS>(A, s, acc, MODE == kSx ? t0 - ev0 : ev0, MODE == kSx ? t1 - ev1 : ev1);
  } else {
    static_assert(MODE == kJtJx, "streaming modes");
    if (flag & kRoundApply) { __syncthreads(); return; }
    // issue_aux with one point: lanes 0..2 hold the point's x_e and D_e
    const double xp[3] = {shfl_idx(x.xa, 0), shfl_idx(x.xa, 1), shfl_idx(x.xa, 2)};
    double z0, z1;
    f_times(s, x.xc, z0, z1);
    z0 += s.e[0] * xp[0] + s.e[1] * xp[1] + s.e[2] * xp[2];
    z1 += s.e[3] * xp[0] + s.e[4] * xp[1] + s.e[5] * xp[2];
    scatter_ft<LDS>(A, s, acc, z0, z1);
    double w[3] = {s.e[0] * z0 + s.e[3] * z1, s.e[1] * z0 + s.e[4] * z1, s.e[2] * z0 + s.e[5] * z1};
    if (!s.valid) { w[0] = w[1] = w[2] = 0; }
    wave_allreduce<3>(w);
    if (lane == 0) { red[wave][0] = w[0]; red[wave][1] = w[1]; red[wave][2] = w[2]; }
    __syncthreads();
    double t = 0;   // lane j < 3: component j of the point's sum
    if (lane < 3) {
      for (int k = 0; k < cnt; ++k) t += red[w0 + k][lane];
    }
    bool store = active && wave == w0 && lane < 3;   // the point's first wave stores y_e
    if (flag & kRoundSum) {
      const double c = (lane == 0 ? carry[0] : (lane == 1 ? carry[1] : carry[2])) + t;
      if (lane == 0) carry[0] = c; else if (lane == 1) carry[1] = c; else if (lane == 2) carry[2] = c;
      store = store && (flag & kRoundLast);
      t = c;
      if (flag & kRoundLast) { carry[0] = carry[1] = carry[2] = 0; }
    }
    if (store) {
      const double yv = t + x.da * x.da * x.xa;
      A.y_e[x.base + lane] = yv;
      dot += x.xa * yv;
    }
  }
}

template <int MODE, bool LDS, bool NT>
__global__ __launch_bounds__(512) void bal_stream_kernel(BalArgs A) {
  constexpr int BLOCK = 512;
  extern __shared__ double lds_acc[];
  if (A.status && *A.status != 0) return;
  double* acc = lds_acc;  // every camera's accumulator row (LDS), or the hybrid rows of this workgroup + one strip per wave (scatter_ft)
  {
    const int n_acc = LDS ? A.n_f9 : 9 * A.hyb_rows;
    for (int i = threadIdx.x; i < n_acc; i += BLOCK) acc[i] = 0.0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  double dot = 0.0;  // kJtJx: this lane's share of x_e . y_e
  // grid-strided over [tile_begin, tile_end); hybrid accumulation: workgroup g owns the tiles of group g (plan.cc)
  const bool grouped = !LDS && A.grp_tile_ptr != nullptr;
  const int64_t nwaves = grouped ? BLOCK / 64 : int64_t(gridDim.x) * (BLOCK / 64);
  // wave-uniform by construction; readfirstlane tells the compiler, so that the tile words are
  // scalar loads and the branches on them scalar branches
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int range = grouped ? int(blockIdx.x) : 0;   // the hybrid group, or everything (plan.cc)
  const int64_t range_end = grouped ? A.grp_tile_ptr[blockIdx.x + 1] : (A.tile_end > 0 ? A.tile_end : A.n_tiles);
  // long points sit behind the normal tiles of their range (unless the ring is chunked): the pipeline ends where they begin
  const int64_t long_begin = A.long_behind ? int64_t(A.long_ptr[range]) : range_end;
  const int64_t tile_end = min(range_end, long_begin);
  const int64_t wave_first = (grouped ? 0 : logical_workgroup() * (BLOCK / 64)) + wave;
  const int64_t wave0 = (grouped ? int64_t(A.grp_tile_ptr[blockIdx.x]) : A.tile_begin) + wave_first;
  // The pipelined loop's walk.  Default: grid-strided (all workgroups stream through one region of memory together).  BLOCKED
  // (A.flags & 2, experiment: CERES_HIP_TILE_WALK=blocked): workgroup b takes the tiles [b C, (b + 1) C), its waves in turn — the
  // point-space range a workgroup writes (JtJx's y_e) is then one contiguous run.
  const bool blocked = !grouped && (A.flags & 2) != 0;
  const int64_t chunk = blocked ? ((tile_end - A.tile_begin + gridDim.x - 1) / gridDim.x + 7) / 8 * 8 : 0;
  const int64_t pipe_end = blocked ? min(tile_end, A.tile_begin + (logical_workgroup() + 1) * chunk) : tile_end;
  const int64_t pipe0 = blocked ? A.tile_begin + logical_workgroup() * chunk + wave : wave0;
  const int64_t pstep = blocked ? BLOCK / 64 : nwaves;
  const int64_t last = pipe_end - 1;
  if (pipe0 < pipe_end) {
    // Two register sets in ping-pong: copying "next" into "current" would need the loaded
    // values to have arrived, which is exactly the wait this kernel exists to avoid.  (The
    // three index words are the exception: they are copied a stage after they were issued.)
    Slot sa, sb;
    StreamAux xa, xb;
    SlotIdx i1, i2;
    int64_t tile = pipe0;
    int kind_a = A.tile_kind[tile], aux_a = A.tile_aux[tile], kind_b = 2, aux_b = 0;
    // prologue in the steady-state issue order: index words (1), pairs (0), aux (0)
    issue_idx(A, tile, lane, i2);
    __builtin_amdgcn_sched_barrier(0);
    issue_idx(A, min(tile + pstep, last), lane, i1);
    __builtin_amdgcn_sched_barrier(0);
    issue_pairs<NT>(A, tile, lane, sa);
    __builtin_amdgcn_sched_barrier(0);
    sa.cam = i2.cam; sa.seg = i2.seg;
    finish_slot(sa, lane, A.tile_pt0[tile]);
    issue_aux<MODE>(A, sa, lane, kind_a == 0 ? aux_a >> 8 : 0, xa);
    __builtin_amdgcn_sched_barrier(0);
    bool more = true;
    // one pipeline stage: everything of the next tile is issued, then this one is computed from `c`
    auto stage = [&](Slot& c, StreamAux& cx, int ckind, int caux, Slot& n, StreamAux& nx, int& nkind, int& naux) {
      const int64_t next = min(tile + pstep, last);  // past the end: re-issue, the load count stays the same
      more = tile + pstep < pipe_end;
      nkind = A.tile_kind[next];
      naux = A.tile_aux[next];
      issue_idx(A, min(next + pstep, last), lane, i2);
      __builtin_amdgcn_sched_barrier(0);
      issue_pairs<NT>(A, next, lane, n);
      __builtin_amdgcn_sched_barrier(0);
      n.cam = i1.cam; n.seg = i1.seg;
      finish_slot(n, lane, A.tile_pt0[next]);
      issue_aux<MODE>(A, n, lane, nkind == 0 ? naux >> 8 : 0, nx);
      __builtin_amdgcn_sched_barrier(0);
      if (ckind != 0) c.valid = false;  // long points are handled below; their seg words carry no store bits
      compute_stream<MODE, LDS>(A, c, lane, caux & 0xff, ckind == 0 ? caux >> 8 : 0, cx, acc, dot);
      __builtin_amdgcn_sched_barrier(0);
      i1 = i2;
      tile = next;
    };
    while (true) {
      stage(sa, xa, kind_a, aux_a, sb, xb, kind_b, aux_b);
      if (!more) break;
      stage(sb, xb, kind_b, aux_b, sa, xa, kind_a, aux_a);
      if (!more) break;
    }
  }
  // Points with more than 64 observations own whole tiles (kind 3 / 1 = head, 2 = continuation).  Those of up to kRoundWaves
  // tiles (kind 3) are taken in ROUNDS, one tile per wave, with the same one-stage-ahead pipeline as above.
  __shared__ double round_red[2][kRoundWaves][3];
  if (A.round_word) {
    // this workgroup's SEQUENCES of rounds (plan.cc): one packed round, or all rounds of one point of more than kRoundWaves tiles
    const int64_t qe = grouped ? int64_t(A.round_ptr[range + 1]) : int64_t(A.n_seq);
    const int64_t qstride = grouped ? 1 : int64_t(gridDim.x);
    int64_t it_q = grouped ? int64_t(A.round_ptr[range]) : logical_workgroup();
    if (it_q < qe) {   // (workgroup-uniform: the rounds have barriers)
      // (a hybrid group's sequences are all its workgroup's: one run of rounds, no table look-up on the way)
      int64_t it_r = A.seq_ptr[it_q], it_end = A.seq_ptr[grouped ? qe : it_q + 1];
      if (grouped) it_q = qe - 1;
      bool it_real = true;
      // the iterator's round, then one step on; past the end it keeps returning the last round (same load count on every path)
      auto next_round = [&](bool* real) {
        const int64_t r = it_r;
        *real = it_real;
        if (it_real) {
          if (it_r + 1 < it_end) ++it_r;
          else if (it_q + qstride < qe) { it_q += qstride; it_r = A.seq_ptr[it_q]; it_end = A.seq_ptr[it_q + 1]; }
          else it_real = false;
        }
        return r;
      };
      Slot sa, sb;
      StreamAux xa, xb;
      SlotIdx i1, i2;
      double carry[3] = {0, 0, 0};
      // an idle wave issues the loads of the round's first tile, with every lane invalid
      auto word_at = [&](int64_t r) { return A.round_word[r * kRoundWaves + wave]; };
      auto tile_at = [&](int64_t r, uint32_t w) { return int64_t((w != kRoundIdle ? w : A.round_word[r * kRoundWaves]) & 0x3FFFFFFu); };
      auto finish = [&](Slot& c, const SlotIdx& i, int64_t tile, uint32_t w) {
        c.cam = i.cam; c.seg = i.seg;
        finish_slot(c, lane, A.tile_pt0[tile]);
        if (w == kRoundIdle) { c.valid = false; c.cam = 0; c.pt = 0; c.acc = kSlotSpill; }
      };
      bool real_c, real_n, real_nn;
      int64_t rc = next_round(&real_c), rn = next_round(&real_n);
      uint32_t wa = word_at(rc), wb = kRoundIdle;
      // the round's first word and flag travel with the pipeline (scalar loads a stage ahead of their use)
      uint32_t w0a = A.round_word[rc * kRoundWaves], w0b = kRoundIdle;
      int fa = A.round_flag[rc], fb = 0;
      int par = 0;
      {
        const int64_t tile = tile_at(rc, wa);
        issue_idx(A, tile, lane, i2);
        __builtin_amdgcn_sched_barrier(0);
        issue_idx(A, tile_at(rn, word_at(rn)), lane, i1);
        __builtin_amdgcn_sched_barrier(0);
        issue_pairs<NT>(A, tile, lane, sa);
        __builtin_amdgcn_sched_barrier(0);
        finish(sa, i2, tile, wa);
        issue_aux<MODE>(A, sa, lane, 1, xa);
        __builtin_amdgcn_sched_barrier(0);
      }
      auto stage = [&](Slot& c, StreamAux& cx, uint32_t cw, uint32_t cw0, int cf, Slot& n, StreamAux& nx, uint32_t& nw, uint32_t& nw0, int& nf) {
        const int64_t rnn = next_round(&real_nn);
        nw = word_at(rn);
        nw0 = A.round_word[rn * kRoundWaves];
        nf = A.round_flag[rn];
        const int64_t next = int64_t((nw != kRoundIdle ? nw : nw0) & 0x3FFFFFFu);
        issue_idx(A, tile_at(rnn, word_at(rnn)), lane, i2);
        __builtin_amdgcn_sched_barrier(0);
        issue_pairs<NT>(A, next, lane, n);
        __builtin_amdgcn_sched_barrier(0);
        finish(n, i1, next, nw);
        issue_aux<MODE>(A, n, lane, 1, nx);
        __builtin_amdgcn_sched_barrier(0);
        compute_long_round<MODE, LDS>(A, c, lane, cx, cw, cw0, cf, wave, round_red[par], carry, acc, dot);
        __builtin_amdgcn_sched_barrier(0);
        par ^= 1;
        i1 = i2;
        rc = rn; real_c = real_n;
        rn = rnn; real_n = real_nn;
      };
      while (true) {
        stage(sa, xa, wa, w0a, fa, sb, xb, wb, w0b, fb);
        if (!real_c) break;
        stage(sb, xb, wb, w0b, fb, sa, xa, wa, w0a, fa);
        if (!real_c) break;
      }
    }
  }
  // Long points outside the rounds (kind 1: the ring is chunked — they sit among the normal tiles then — or the plan has more
  // tiles than a round word holds): one wave per point, two sweeps (process_long_point).
  for (int64_t tile = (A.long_behind ? long_begin + wave_first : wave0); tile < range_end; tile += nwaves) {
    if (A.tile_kind[tile] == 1) process_long_point<MODE, LDS, false>(A, tile, A.tile_aux[tile], lane, acc, dot);
  }
  if (MODE == kJtJx && A.pq_out) {  // one partial of x_e . y_e per workgroup
    __shared__ double red[BLOCK / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (lane == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
      for (int i = 0; i < BLOCK / 64; ++i) t += red[i];
      if (A.pq_accumulate) t += A.pq_out[blockIdx.x];
      A.pq_out[blockIdx.x] = t;
    }
  }
  if constexpr (LDS) {
    __syncthreads();
    double* out = A.partials + int64_t(blockIdx.x) * A.n_f9;
    const bool tail = MODE == kSx && A.tail.enabled;   // (launch-uniform)
    if (!tail) {
      for (int i = threadIdx.x; i < A.n_f9; i += BLOCK) out[i] = acc[i];
    } else {
      // write-through stores (device scope): the partial sums must be readable from another XCD when the ticket is taken, and a
      // release fence instead — an L2 write-back per workgroup, queued behind those of the XCD's other workgroups — cost 12 us
      for (int i = threadIdx.x; i < A.n_f9; i += BLOCK) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(out + i), "v"(acc[i]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }
    if constexpr (MODE == kSx) {
      if (tail) {
        __shared__ int last;
        __syncthreads();      // the workgroup's partial sums are out ...
        if (threadIdx.x == 0) last = atomicAdd(A.tail.ticket, 1u) == gridDim.x - 1 ? 1 : 0;   // ... before its ticket is
        __syncthreads();
        if (last) {
          __threadfence();    // and every other workgroup's are in before they are read
          cg_iteration_tail<BLOCK>(A, acc);
          if (threadIdx.x == 0) *A.tail.ticket = 0u;
        }
      }
    }
  } else {
    if (A.hyb_rows > 0) {  // this workgroup's accumulator rows join the spilled rows in the ring: the second pass sums both
      __syncthreads();
      double* out = A.zbuf + 9 * (A.z_flush_row0 + int64_t(blockIdx.x) * A.hyb_rows);
      for (int i = threadIdx.x; i < 9 * A.hyb_rows; i += BLOCK) out[i] = acc[i];
    }
  }
}

// y_f[pos(i)] = sum over workgroup partials (+ D_f^2 x_f).  One thread per F scalar.
// Workgroup = 64 consecutive scalars x 8 waves, wave w sums partials w, w+8, w+16, ...; the eight
// wave sums are combined through LDS in a fixed order (deterministic).  A workgroup takes every
// gridDim.x-th group of 64 scalars.  pq_out: also the partial inner product x_f . y_f of the scalars this
// workgroup finished (CG's p.q, camera part: x_f and the finished y_f are both in registers here).
__global__ __launch_bounds__(512) void bal_reduce_partials_kernel(const double* __restrict__ partials, int nparts, int n_f9,
                                           const int32_t* __restrict__ cam_pos, const double* __restrict__ D_f,
                                           const double* __restrict__ x_f, double* __restrict__ y_f,
                                           const int* __restrict__ status, double* __restrict__ pq_out,
                                           const double* __restrict__ sum_in, int n_sum_in, double* __restrict__ sum_out) {
  __shared__ double sh[8][64];
  if (status && *status != 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // sharded CGNR: this rank's point-space share of p.q (the tile pass's per-workgroup partials) collapsed into ONE double that
  // travels with the camera vector through the same all-reduce (sum_out = the element after the vector); fixed order
  if (sum_out && blockIdx.x == gridDim.x - 1 && wv == 7) {
    double v = 0;
    for (int k = lane; k < n_sum_in; k += 64) v += sum_in[k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) *sum_out = v;
  }
  const int ngroups = (n_f9 + 63) / 64;
  double dot = 0;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int i = g * 64 + lane;
    double s0 = 0, s1 = 0;
    if (i < n_f9) {
      int w = wv;
      for (; w + 8 < nparts; w += 16) {
        s0 += partials[int64_t(w) * n_f9 + i];
        s1 += partials[int64_t(w + 8) * n_f9 + i];
      }
      if (w < nparts) s0 += partials[int64_t(w) * n_f9 + i];
    }
    sh[wv][lane] = s0 + s1;
    __syncthreads();
    if (wv == 0 && i < n_f9) {
      double s = ((sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane])) + ((sh[4][lane] + sh[5][lane]) + (sh[6][lane] + sh[7][lane]));
      const int o = cam_pos ? cam_pos[i / 9] + i % 9 : i;
      if (D_f) { const double d = D_f[o]; s += d * d * x_f[o]; }
      y_f[o] = s;
      if (pq_out) dot += x_f[o] * s;
    }
    __syncthreads();  // sh is reused by the next group
  }
  if (pq_out && wv == 0) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (lane == 0) pq_out[blockIdx.x] = dot;
  }
}

// HBM read-stream probe: the same tile walk and 16-byte-per-lane loads as the fused kernels,
// nothing else (one double per workgroup written).  Gives the ceiling the fused kernels are
// measured against next to the 8 TB/s datasheet figure.
__global__ __launch_bounds__(1024) void bal_stream_probe_kernel(const double2* __restrict__ J, int64_t n_tiles, double* __restrict__ out) {
  __shared__ double sh[16];
  const int lane = threadIdx.x & 63;
  const int64_t wave = int64_t(blockIdx.x) * 16 + (threadIdx.x >> 6), nwaves = int64_t(gridDim.x) * 16;
  double acc = 0;
  for (int64_t tile = wave; tile < n_tiles; tile += nwaves) {
    const double2* p = J + tile * kTilePitch + lane;
    double2 v[kPairsPerSlot];
#pragma unroll
    for (int j = 0; j < kPairsPerSlot; ++j) v[j] = p[j * kTile];
#pragma unroll
    for (int j = 0; j < kPairsPerSlot; ++j) acc += v[j].x + v[j].y;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < 16; ++i) s += sh[i]; out[blockIdx.x] = s; }
}


// y_f += D_f^2 x_f over the camera scalars (after an all-reduce of the raw sums); pq_out: partial x_f . y_f per workgroup.
__global__ __launch_bounds__(256) void bal_add_f_diagonal_kernel(int n_f9, const int32_t* __restrict__ cam_pos, const double* __restrict__ D_f,
                                          const double* __restrict__ x_f, double* __restrict__ y_f,
                                          const int* __restrict__ status, double* __restrict__ pq_out) {
  __shared__ double red[4];
  if (status && *status != 0) return;
  double dot = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_f9; i += gridDim.x * 256) {
    const int o = cam_pos ? cam_pos[i / 9] + i % 9 : i;
    double y = y_f[o];
    if (D_f) { const double d = D_f[o]; y += d * d * x_f[o]; y_f[o] = y; }
    if (pq_out) dot += x_f[o] * y;
  }
  if (pq_out) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) pq_out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Re-layout: caller's values (any cell.position) -> tiles.  One wavefront per tile.
template <bool F32>
__global__ __launch_bounds__(256) void bal_pack_kernel(const double* __restrict__ values, const double* __restrict__ b,
                                                       const int32_t* __restrict__ slot_epos,
                                                       const int32_t* __restrict__ slot_fpos,
                                                       const int32_t* __restrict__ slot_bpos, int64_t n_tiles,
                                                       double2* __restrict__ J, float4* __restrict__ Jf,
                                                       double2* __restrict__ bt) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int64_t sl = tile * kTile + lane;
  const int ep = slot_epos[sl], fp = slot_fpos[sl], bp = slot_bpos[sl];
  double v[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) v[i] = 0.0;
  double b0 = 0, b1 = 0;
  if (ep >= 0) {
    const double* e = values + ep;
    const double* f = values + fp;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = e[i];
#pragma unroll
    for (int i = 0; i < 18; ++i) v[6 + i] = f[i];
    if (b) { b0 = b[bp]; b1 = b[bp + 1]; }
  }
  if constexpr (F32) {
    float4* o = Jf + tile * (6 * kTile) + lane;
#pragma unroll
    for (int q = 0; q < 6; ++q) tile_store(o + q * kTile, make_float4(float(v[4 * q]), float(v[4 * q + 1]), float(v[4 * q + 2]), float(v[4 * q + 3])));
  } else {
    double2* o = J + tile * kTilePitch + lane;
#pragma unroll
    for (int j = 0; j < kPairsPerSlot; ++j) tile_store(o + j * kTile, make_double2(v[2 * j], v[2 * j + 1]));
  }
  if (b) bt[sl] = make_double2(b0, b1);
}

// ---- camera-major passes ---------------------------------------------------------------
// Unit of work = ITEM: at most kCamChunk consecutive observations of ONE camera in the
// camera-major list (plan.cc); camera degrees are heavily skewed, so heavy cameras are split.
// One WAVEFRONT per item (4 items per workgroup): lanes stride over the item's observations,
// gather F from the caller-layout values (144 contiguous bytes per observation) and reduce
// with __shfl_xor.  A camera covered by a single item is stored directly; split cameras are
// combined with global_atomic_add_f64 into a zeroed output (rounding-level order effects
// only there).

// Per-camera 9x9 blocks: sum of F^T M F, with M = I (JACOBI: block diagonal of F^T F) or M = I - E (E^T E)^-1 E^T read
// from [slot][4] (SCHUR_JACOBI: the diagonal blocks SchurEliminator::Eliminate writes into a block-diagonal lhs).
// Every ITEM leaves its 45 upper-triangle sums (and, SCHUR, the 9 column square sums of F) in parts[item][kCamPart]:
// no atomics, no zeroed output, and the per-camera combination (bal_camera_finish_kernel, or the load phase of
// bal_invert9_kernel) adds the items of a camera in a fixed order — bit-reproducible.  The first version combined split
// cameras with global fp64 atomics: 81 per item, which on a problem with few cameras (Dubrovnik: 16) or small items
// (Ladybug) cost several times the pass itself (55 / 105 us, profiles/r02b_kernel_stats_*).
// Loads: every lane reads "its" 144-byte cell (18 loads of 8 bytes, all issued before the first use).  A naive lane-per-cell
// gather LOOP with half a million lanes in flight (tools/probes/gather_probe.hip) fetches 3-5x its useful bytes by FETCH_SIZE,
// whatever the order of the cells (profiles/r02l_gather_probe_fetch_size_calibration.txt); this kernel's few, register-heavy
// waves stay near 2.2x (1.9 GB for 0.88 GB of F and M_o on the Venice shape).  Measured, no effect (r02af): nine 16-byte loads per
// cell instead of eighteen 8-byte ones.  Measured SLOWER (r02ae, 0.46 vs 0.33 ms): forming M_o here from the E cell and the point's
// packed inverse instead of reading the M_o record (it saves kInit 0.034 ms of writes and costs this pass 0.13 ms).
// Tried and measured SLOWER (r02m, 0.65 vs 0.37 ms on the Venice shape): cooperative loading, nine lanes x 16 bytes
// per cell into an LDS strip and each lane picking its observation up from there — one line request per line, but load ->
// LDS -> compute serialise inside an iteration and the wave count drops with the 36 KB of LDS per workgroup.
template <bool SCHUR>
__global__ __launch_bounds__(256) void bal_camera_items_kernel(const double* __restrict__ values, CamItems items,
                                                               const int32_t* __restrict__ cam_fpos,
                                                               const int32_t* __restrict__ cam_slot,
                                                               const double* __restrict__ Mo, double* __restrict__ parts) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items.count) return;
  const int beg = items.begin[item], end = items.end[item];
  double acc[45], sq[9];
#pragma unroll
  for (int i = 0; i < 45; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) sq[i] = 0.0;
  for (int q = beg + lane; q < end; q += 64) {
    const double* f = values + cam_fpos[q];
    double f0[9], f1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { f0[k] = f[k]; f1[k] = f[9 + k]; }
    double m00 = 1.0, m01 = 0.0, m11 = 1.0;
    if constexpr (SCHUR) {
      const double2* mo = reinterpret_cast<const double2*>(Mo + 4 * int64_t(cam_slot[q]));
      const double2 a = mo[0], b = mo[1];  // plain loads: non-temporal ones (no L1) made this pass 8 % slower (r02y)
      m00 = a.x; m01 = a.y; m11 = b.x;
#pragma unroll
      for (int k = 0; k < 9; ++k) sq[k] += f0[k] * f0[k] + f1[k] * f1[k];  // column norms of the camera columns (the blocks hold F^T M F, not F^T F)
    }
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double g0 = m00 * f0[a] + m01 * f1[a];  // row a of (F^T M): [g0 g1]
      const double g1 = m01 * f0[a] + m11 * f1[a];
#pragma unroll
      for (int bb = a; bb < 9; ++bb) acc[idx++] += g0 * f0[bb] + g1 * f1[bb];
    }
  }
  double* out = parts + int64_t(item) * kCamPart;
#pragma unroll
  for (int i = 0; i < 45; ++i) {
    double v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == i) out[i] = v;   // every lane holds every sum; lane i stores entry i: 360 contiguous bytes per item
  }
  if constexpr (SCHUR) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      double v = sq[k];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 45 + k) out[45 + k] = v;
    }
  }
}

// The same item sums on the matrix pipe: a camera's block IS a contraction, F^T (W F) with F the (2 n) x 9 stack of the item's
// observations and W = diag(M_o) (SCHUR_JACOBI) or I.  One v_mfma_f64_16x16x4_f64 takes four rows = two observations: lane l holds
// column l & 15 (zero beyond 9) of row l >> 4, which is both its A operand (F^T[col][row]) and, after the 2 x 2 weight mixes it
// with the observation's other row (lane l ^ 16), its B operand ((W F)[row][col]).  Against the lane-per-observation form above:
// a load instruction reads whole 72-byte rows of two records instead of one 8-byte word of 64 records (4 instead of 18 line
// look-ups per observation), and there is no 54-value wavefront reduction at the end — the sums ARE the accumulator, entry
// (a, b) in register a / 4 of lane 16 (a % 4) + b.  What short items (a camera of a video-like scene has tens of observations) and
// launch-bound shapes spent their time on.  LaunchBalCameraItems picks the kernel by the mean item length.
typedef double cam_v4f64 __attribute__((ext_vector_type(4)));
template <bool SCHUR>
__global__ __launch_bounds__(256) void bal_camera_items_mfma_kernel(const double* __restrict__ values, CamItems items,
                                                                    const int32_t* __restrict__ cam_fpos,
                                                                    const int32_t* __restrict__ cam_slot,
                                                                    const double* __restrict__ Mo, double* __restrict__ parts) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items.count) return;
  const int beg = items.begin[item], end = items.end[item];
  const int col = lane & 15, kk = lane >> 4, ob = kk >> 1, r = kk & 1;
  const bool cv = col < 9;
  constexpr int U = 8;  // matrix instructions (pairs of observations) per iteration: their loads are all in flight together
  cam_v4f64 acc = cam_v4f64{0.0, 0.0, 0.0, 0.0}, acc1 = cam_v4f64{0.0, 0.0, 0.0, 0.0};   // two chains: the pipe need not wait for its own result
  double sq = 0.0;
  int fp[U], sl[U];
  auto load_idx = [&](int q0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 2 * u + ob;
      const bool ok = q < end;
      fp[u] = ok ? cam_fpos[q] : -1;
      if constexpr (SCHUR) sl[u] = ok ? cam_slot[q] : 0;
    }
  };
  load_idx(beg);
  for (int q0 = beg; q0 < end; q0 += 2 * U) {
    double a[U], wd[U], wo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = (fp[u] >= 0 && cv) ? values[fp[u] + 9 * r + col] : 0.0;
      if constexpr (SCHUR) {
        // record [m00 m01 | m11 m01]: one 16-byte load gives row r its diagonal and off-diagonal weight
        const double2 m = *reinterpret_cast<const double2*>(Mo + 4 * int64_t(sl[u]) + 2 * r);
        wd[u] = m.x;
        wo[u] = m.y;
      }
    }
    load_idx(q0 + 2 * U);   // the next iteration's index words travel while this one computes
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double b = a[u];
      if constexpr (SCHUR) {
        const double other = __shfl_xor(a[u], 16, 64);   // the observation's other row, same column
        b = wd[u] * a[u] + wo[u] * other;
        sq += a[u] * a[u];
      }
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b, acc1, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b, acc, 0, 0, 0);
    }
  }
  acc += acc1;
  double* out = parts + int64_t(item) * kCamPart;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = kk + 4 * i;
    if (ra <= col && col < 9) out[ra * (19 - ra) / 2 + (col - ra)] = acc[i];
  }
  if constexpr (SCHUR) {
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (kk == 0 && cv) out[45 + col] = sq;
  }
}

// Row `i` of camera c's block from the per-item partial sums (packed upper triangle: entry (a, b), a <= b, at a (19 - a) / 2 + b - a),
// items of a camera in list order; sqsum = the camera's column-i square sum (SCHUR items only).
__device__ __forceinline__ void gather_camera_row(const double* __restrict__ parts, int item_lo, int item_hi, int i, double (&row)[9],
                                                  double& sqsum, bool want_sq) {
#pragma unroll
  for (int k = 0; k < 9; ++k) row[k] = 0.0;
  sqsum = 0.0;
  for (int it = item_lo; it < item_hi; ++it) {
    const double* p = parts + int64_t(it) * kCamPart;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int a = k < i ? k : i, b = k < i ? i : k;
      row[k] += p[a * (19 - a) / 2 + (b - a)];
    }
    if (want_sq) sqsum += p[45 + i];
  }
}

// blocks[c] = sum of the camera's items (full symmetric 9x9) + D_f^2 on the diagonal if D_f != nullptr; camsq likewise.
// Nine lanes per camera, seven cameras per wavefront.  Used when the raw sums are needed in memory (sharded all-reduce,
// not_inverted read-back); otherwise bal_invert9_kernel gathers the items itself.
__global__ __launch_bounds__(64) void bal_camera_finish_kernel(const double* __restrict__ parts, const int32_t* __restrict__ cam_item_ptr,
                                                               const double* __restrict__ D_f, const int32_t* __restrict__ cam_pos,
                                                               const int64_t* __restrict__ cam_diag_off, double* __restrict__ blocks,
                                                               double* __restrict__ camsq, int n_cameras, const double* __restrict__ extra) {
  const int lane = threadIdx.x;
  const int grp = lane / 9, i = lane - 9 * grp;
  const int c = blockIdx.x * 7 + grp;
  if (grp >= 7 || c >= n_cameras) return;
  double row[9], sqsum;
  gather_camera_row(parts, cam_item_ptr[c], cam_item_ptr[c + 1], i, row, sqsum, camsq != nullptr);
  if (extra) {  // rows outside the tiles (no point cell): their F^T F, whose diagonal is their share of the column square sums
    const double* x = extra + 81 * int64_t(c) + 9 * i;
#pragma unroll
    for (int k = 0; k < 9; ++k) { row[k] += x[k]; if (k == i) sqsum += x[k]; }
  }
  if (D_f) {
    const double d = D_f[(cam_pos ? cam_pos[c] : 9 * c) + i];
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == i) row[k] += d * d;
  }
  double* out = blocks + (cam_diag_off ? cam_diag_off[c] : int64_t(81) * c) + 9 * i;
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = row[k];
  if (camsq) camsq[9 * int64_t(c) + i] = sqsum;
}

// Camera-major pass of ONE CHUNK of tiles when the camera accumulators do not fit in LDS: the tile pass of the chunk left
// F_o^T z_o (9 doubles) per slot in a ring buffer that is small enough to still be in the Infinity Cache; a unit = up to kZUnit
// entries of one camera inside the chunk.  Nine lanes per unit (lane k sums component k: the 72 bytes of an entry are one
// contiguous access of the group), seven units per wavefront, four entries in flight per lane.  A camera covered by a single
// unit in this chunk is updated with a plain read-modify-write (chunks run one after the other on the stream); split cameras
// (more than kZUnit observations inside one chunk: the popular ones) combine with global_atomic_add_f64.
__global__ __launch_bounds__(256) void bal_camera_chunk_kernel(ZUnits U, const double* __restrict__ ring, double* __restrict__ acc,
                                                               const int* __restrict__ status) {
  if (status && *status != 0) return;
  const int lane = threadIdx.x & 63;
  const int g = lane / 9, k = lane - 9 * g;
  const int64_t u = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 7 + g;
  if (g >= 7 || u >= U.count) return;
  const int unit = U.first + int(u);
  const int c = U.cam[unit], beg = U.begin[unit], end = U.end[unit];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int e = beg;
  for (; e + 7 < end; e += 8) {  // eight index words, then eight entries in flight
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = U.slot[e + i];
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ring[9 * int64_t(a[i]) + k];
    s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
  }
  for (; e + 3 < end; e += 4) {
    const int a0 = U.slot[e], a1 = U.slot[e + 1], a2 = U.slot[e + 2], a3 = U.slot[e + 3];
    s0 += ring[9 * int64_t(a0) + k]; s1 += ring[9 * int64_t(a1) + k]; s2 += ring[9 * int64_t(a2) + k]; s3 += ring[9 * int64_t(a3) + k];
  }
  for (; e < end; ++e) s0 += ring[9 * int64_t(U.slot[e]) + k];
  const double v = (s0 + s1) + (s2 + s3);
  double* dst = acc + 9 * int64_t(c) + k;
  if (U.shared[unit]) unsafeAtomicAdd(dst, v);
  else *dst += v;
}

// In-place inverse of the 9x9 SPD camera blocks from their upper triangle (Cholesky +
// solves against I, like BlockRandomAccessDiagonalMatrix::Invert,
// I/block_random_access_diagonal_matrix.cc:90-100).  One thread per camera.
// Nine lanes per camera (seven cameras per wavefront): lane i of a group owns row i of the
// block.  The Cholesky factor is built column by column with the pivot row broadcast by
// shuffles; lane e then solves L L^T x = e_e for "its" column of the inverse.  Same operation
// order per entry as a scalar column-Cholesky (the reference: Eigen LLT on the upper triangle,
// I/block_random_access_diagonal_matrix.cc:106-127); one thread per camera, as this kernel first
// was, ran a ~1000-instruction dependent chain with 648-byte strided accesses: 31 us for 1778 cameras.
constexpr int kInvertGatherWaves = 4;
__global__ __launch_bounds__(64 * kInvertGatherWaves) void bal_invert9_kernel(double* __restrict__ blocks, const int64_t* __restrict__ cam_diag_off,
                                                         int n_cameras, int* fail_flag, LmFuse lm, CamGather gather) {
  __shared__ double csum[kInvertGatherWaves][kCamPart];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int grp = lane / 9, i = lane - 9 * grp;  // lane 63 idles
  // assembled blocks: seven cameras per wavefront (64-thread workgroups); per-item partial sums (gather.parts): ONE camera per
  // workgroup of kInvertGatherWaves wavefronts, whose 54 lanes each add up every kInvertGatherWaves-th item of the camera entry by
  // entry (coalesced 432-byte rows) — the kernel lasts as long as its most popular camera, whose item list one wavefront alone
  // walked for 46 us on the Venice shape —, then wave 0 combines the partial sums in a fixed order and lanes 0..8 take the rows
  // gather.few: cameras of a handful of items each (tens of thousands of cameras with tens of observations: video-like scenes) — seven
  // cameras per wavefront here too, every lane adding up its row over the camera's items (a workgroup per camera was 206 us for
  // 52 800 cameras of one item each, latency from end to end)
  const bool few = gather.parts != nullptr && gather.few != 0;
  const bool gathering = gather.parts != nullptr && !few;
  const int c = gathering ? int(blockIdx.x) : int(blockIdx.x) * 7 + grp;
  const bool active = (gathering ? grp == 0 : grp < 7) && c < n_cameras;
  const int cc = active ? c : 0;
  const int g0 = 9 * ((!gathering && grp < 7) ? grp : 0);        // first lane of the group
  double* a = blocks + (cam_diag_off ? cam_diag_off[cc] : int64_t(81) * cc);
  double row[9], sq_from_items = 0.0;
  if (few) {
    if (active) {
      gather_camera_row(gather.parts, gather.cam_item_ptr[c], gather.cam_item_ptr[c + 1], i, row, sq_from_items, gather.want_sq != 0);
      if (gather.extra) {
        const double* x = gather.extra + 81 * int64_t(c) + 9 * i;
#pragma unroll
        for (int k = 0; k < 9; ++k) { row[k] += x[k]; if (k == i) sq_from_items += x[k]; }
      }
      if (gather.D_f) {
        const double d = gather.D_f[(gather.cam_pos ? gather.cam_pos[c] : 9 * c) + i];
#pragma unroll
        for (int k = 0; k < 9; ++k) if (k == i) row[k] += d * d;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) row[k] = (k == i ? 1.0 : 0.0);
    }
  } else if (gathering) {  // the blocks are still per-item partial sums: combine them here (bal_camera_items_kernel)
    const int item_lo = gather.cam_item_ptr[blockIdx.x], item_hi = gather.cam_item_ptr[blockIdx.x + 1];
    if (lane < kCamPart) {
      double s0 = 0.0, s1 = 0.0;
      int it = item_lo + wv;
      for (; it + nwv < item_hi; it += 2 * nwv) {
        s0 += gather.parts[int64_t(it) * kCamPart + lane];
        s1 += gather.parts[int64_t(it + nwv) * kCamPart + lane];
      }
      if (it < item_hi) s0 += gather.parts[int64_t(it) * kCamPart + lane];
      csum[wv][lane] = s0 + s1;
    }
    __syncthreads();
    if (wv != 0) return;
    if (lane < kCamPart) {
      double t = csum[0][lane];
      for (int w = 1; w < nwv; ++w) t += csum[w][lane];
      csum[0][lane] = t;
    }
    __builtin_amdgcn_wave_barrier();
    if (active) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int ra = k < i ? k : i, rb = k < i ? i : k;
        row[k] = csum[0][ra * (19 - ra) / 2 + (rb - ra)];
      }
      sq_from_items = csum[0][45 + i];
      if (gather.extra) {  // rows outside the tiles (no point cell): their F^T F, whose diagonal is their share of the column square sums
        const double* x = gather.extra + 81 * int64_t(c) + 9 * i;
#pragma unroll
        for (int k = 0; k < 9; ++k) { row[k] += x[k]; if (k == i) sq_from_items += x[k]; }
      }
      if (gather.D_f) {
        const double d = gather.D_f[(gather.cam_pos ? gather.cam_pos[c] : 9 * c) + i];
#pragma unroll
        for (int k = 0; k < 9; ++k) if (k == i) row[k] += d * d;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) row[k] = (k == i ? 1.0 : 0.0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) row[k] = active ? a[i * 9 + k] : (k == i ? 1.0 : 0.0);
  }
  if (lm.radius > 0.0 && active) {  // fused LM diagonal of the camera columns: d = clamp(diag(F^T F)), D^2 = d / radius
    const int o = lm.cam_pos ? lm.cam_pos[c] : 9 * c;
    double dii = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == i) dii = row[k];
    const double v = lm.camsq ? lm.camsq[9 * int64_t(c) + i] : ((gather.parts && gather.want_sq) ? sq_from_items : dii);
    const double d = fmin(fmax(v, lm.min_d), lm.max_d), q = d / lm.radius;
    lm.diag_f[o + i] = d;
    lm.D_f[o + i] = sqrt(q);
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == i) row[k] += q;
  }
  // L row i in Lr[0..i]
  double Lr[9];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    // pivot row j (entries k < j are final) comes from lane g0 + j
    double pj[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pj[k] = (k < j) ? shfl_idx(Lr[k], g0 + j) : 0.0;
    // diagonal, computed by every lane the same way from row j's data: a_jj - sum L_jk^2
    double ajj = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == j) ajj = row[k];
    double d = shfl_idx(ajj, g0 + j);
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k < j) d -= pj[k] * pj[k];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    const double inv = 1.0 / d;
    // symmetric block: a_ji = a_ij is in this lane's row
    double sv = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == j) sv = row[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k < j) sv -= Lr[k] * pj[k];
    Lr[j] = (i == j) ? d : (i > j ? sv * inv : 0.0);
  }
  if (!ok && active && fail_flag) atomicExch(fail_flag, 1);
  // every lane needs all of L for its solve
  double L[45];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k <= r) L[r * (r + 1) / 2 + k] = shfl_idx(Lr[k], g0 + r);
  }
  double col[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {  // L y = e_i
    double sacc = (r == i) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k < r) sacc -= L[r * (r + 1) / 2 + k] * col[k];
    col[r] = sacc / L[r * (r + 1) / 2 + r];
  }
#pragma unroll
  for (int r = 8; r >= 0; --r) {  // L^T x = y
    double sacc = col[r];
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k > r) sacc -= L[k * (k + 1) / 2 + r] * col[k];
    col[r] = sacc / L[r * (r + 1) / 2 + r];
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < 9; ++r) a[r * 9 + i] = col[r];  // column i of the inverse; lanes of a group write 72 contiguous bytes
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
// The dynamic-LDS ceiling is a per-device attribute of a kernel: set it once per (kernel, device), whichever thread
// gets there first.  Keyed by the kernel's ADDRESS (all kernels here share one function type, so a static per
// template instantiation would be shared between them).
static hipError_t allow_max_lds(const void* kernel) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned long long> done;  // kernel -> mask of devices 0..63
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> lock(mu);
  unsigned long long& mask = done[kernel];
  if (mask & bit) return hipSuccess;
  // static + dynamic LDS must fit the CU's 160 KB: kernels with a few bytes of static __shared__ (cross-wave reductions) get that much less
  hipFuncAttributes fa;
  if (hipError_t e = hipFuncGetAttributes(&fa, kernel); e != hipSuccess) return e;
  const int dyn = int(kMaxLdsBytes) - int(fa.sharedSizeBytes);
  if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); e != hipSuccess) return e;
  mask |= bit;
  return hipSuccess;
}

template <int MODE, int BLOCK, bool F32>
static hipError_t launch_fused2(const BalArgs& A, bool lds, int grid, hipStream_t stream) {
  if (lds) {
    const size_t bytes = size_t(A.n_f9) * sizeof(double);
    auto k = bal_fused_kernel<MODE, true, BLOCK, F32>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(k)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), bytes, stream, A);
  } else {
    // scattering modes: the hybrid accumulator rows (none without a hybrid plan) + one 64 x 9 strip per wave for the spilled rows
    constexpr bool kScatters = (MODE == kSx || MODE == kJtJx || MODE == kJtb || MODE == kInit || MODE == kCgnrInit || MODE == kColNorm || MODE == kSpseZ);
    const size_t bytes = kScatters ? (size_t(A.hyb_rows) * 9 + size_t(BLOCK / 64) * kTile * 9) * sizeof(double) : 0;
    auto k = bal_fused_kernel<MODE, false, BLOCK, F32>;
    if (bytes > 0)
      if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(k)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), bytes, stream, A);
  }
  return hipGetLastError();
}
template <int MODE, int BLOCK>
static hipError_t launch_fused(const BalArgs& A, bool lds, int grid, hipStream_t stream) {
  return A.Jf ? launch_fused2<MODE, BLOCK, true>(A, lds, grid, stream) : launch_fused2<MODE, BLOCK, false>(A, lds, grid, stream);
}

// Threads per workgroup of the streaming kernels: 1024 (16 waves per CU) hides HBM latency
// best for the light modes; the set-up modes need more registers and run at 512.
int BalBlockFor(int mode) {
  static int forced = [] { const char* e = getenv("CERES_HIP_BAL_BLOCK"); return e ? atoi(e) : 0; }();
  if (mode == kInit || mode == kCgnrInit || mode == kColNorm) return 512;
  if (forced == 512 || forced == 1024) return forced;
  return 1024;
}

template <int MODE, bool NT>
static hipError_t launch_stream2(const BalArgs& A, bool lds, int grid, hipStream_t stream) {
  if (lds) {
    auto k = bal_stream_kernel<MODE, true, NT>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(k)); e != hipSuccess) return e;
    // (+ the scratch of the CG iteration tail: cg_iteration_tail)
    const size_t tail_bytes = (MODE == kSx && A.tail.enabled) ? size_t(512 + kCgTailMax) * sizeof(double) : 0;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), size_t(A.n_f9) * sizeof(double) + tail_bytes, stream, A);
  } else {
    auto k = bal_stream_kernel<MODE, false, NT>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(k)); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), (size_t(A.hyb_rows) * 9 + size_t(512 / 64) * kTile * 9) * sizeof(double), stream, A);
  }
  return hipGetLastError();
}
template <int MODE>
static hipError_t launch_stream(const BalArgs& A, bool lds, int grid, hipStream_t stream) {
  // tiles that fit the 256 MiB Infinity Cache are read with plain loads (see issue_pairs)
  const bool nt = A.n_tiles * int64_t(kTilePitch) * int64_t(sizeof(double2)) > (int64_t(200) << 20);
  return nt ? launch_stream2<MODE, true>(A, lds, grid, stream) : launch_stream2<MODE, false>(A, lds, grid, stream);
}

// CERES_HIP_PIPELINE=0 falls back to the unpipelined kernels (A/B measurements).
static bool UsePipeline() {
  static int v = [] { const char* e = getenv("CERES_HIP_PIPELINE"); return e ? atoi(e) : 1; }();
  return v != 0;
}

bool BalSxRunsPipelined(const BalArgs& A) { return UsePipeline() && !A.Jf && !A.src_values && !A.cam_pos && !(A.flags & 1); }

hipError_t LaunchBalFused(int mode, const BalArgs& A, bool lds, int grid, hipStream_t stream) {
  const bool big = BalBlockFor(mode) == 1024;
  const bool big_s = big && (lds || A.hyb_rows == 0);  // scattering modes with hybrid rows: the rows are sized for 8 waves' strips
  if (UsePipeline() && !A.Jf && !A.src_values && !A.cam_pos && !(A.flags & 1)) {
    // (S.x and the power-series operator index nothing of the caller's by point: pt_pos may be set — renumbered points, plan.cc)
    if (mode == kSx) return launch_stream<kSx>(A, lds, grid, stream);
    if (mode == kJtJx && A.D_e && !A.pt_pos) return launch_stream<kJtJx>(A, lds, grid, stream);
    if (mode == kSpseZ) return launch_stream<kSpseZ>(A, lds, grid, stream);
  }
  switch (mode) {
    case kSx: return big_s ? launch_fused<kSx, 1024>(A, lds, grid, stream) : launch_fused<kSx, 512>(A, lds, grid, stream);
    case kJtJx: return big_s ? launch_fused<kJtJx, 1024>(A, lds, grid, stream) : launch_fused<kJtJx, 512>(A, lds, grid, stream);
    case kJtb: return (big_s && !A.src_values) ? launch_fused<kJtb, 1024>(A, lds, grid, stream) : launch_fused<kJtb, 512>(A, lds, grid, stream);
    case kInit: return launch_fused<kInit, 512>(A, lds, grid, stream);
    case kEte: return big ? launch_fused<kEte, 1024>(A, false, grid, stream) : launch_fused<kEte, 512>(A, false, grid, stream);
    case kBackSub: return big ? launch_fused<kBackSub, 1024>(A, false, grid, stream) : launch_fused<kBackSub, 512>(A, false, grid, stream);
    case kCgnrInit: return launch_fused<kCgnrInit, 512>(A, lds, grid, stream);
    case kColNorm: return launch_fused<kColNorm, 512>(A, true, grid, stream);
    case kJx: return launch_fused<kJx, 1024>(A, false, grid, stream);
    case kSpseZ: return big_s ? launch_fused<kSpseZ, 1024>(A, lds, grid, stream) : launch_fused<kSpseZ, 512>(A, lds, grid, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t LaunchBalReducePartials(const double* partials, int nparts, int n_f9, const int32_t* cam_pos,
                                   const double* D_f, const double* x_f, double* y_f, const int* status,
                                   double* pq_out, int* n_pq, hipStream_t stream, const double* sum_in, int n_sum_in,
                                   double* sum_out) {
  const int grid = std::max(1, std::min((n_f9 + 63) / 64, kMaxVecGrid));
  if (!x_f) pq_out = nullptr;
  if (n_pq) *n_pq = pq_out ? grid : 0;
  hipLaunchKernelGGL(bal_reduce_partials_kernel, dim3(grid), dim3(512), 0, stream, partials, nparts,
                     n_f9, cam_pos, D_f, x_f, y_f, status, pq_out, sum_in, n_sum_in, sum_out);
  return hipGetLastError();
}

hipError_t LaunchBalStreamProbe(const double2* J, int64_t n_tiles, int grid, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(bal_stream_probe_kernel, dim3(grid), dim3(1024), 0, stream, J, n_tiles, out);
  return hipGetLastError();
}


hipError_t LaunchBalAddFDiagonal(int n_f9, const int32_t* cam_pos, const double* D_f, const double* x_f, double* y_f,
                                 const int* status, double* pq_out, int* n_pq, hipStream_t stream) {
  const int grid = std::max(1, std::min((n_f9 + 255) / 256, kMaxVecGrid));
  if (!x_f) pq_out = nullptr;
  if (n_pq) *n_pq = pq_out ? grid : 0;
  if (!D_f && !pq_out) return hipSuccess;
  hipLaunchKernelGGL(bal_add_f_diagonal_kernel, dim3(grid), dim3(256), 0, stream, n_f9, cam_pos, D_f,
                     x_f, y_f, status, pq_out);
  return hipGetLastError();
}

hipError_t LaunchBalPack(const double* values, const double* b, const int32_t* slot_epos, const int32_t* slot_fpos,
                         const int32_t* slot_bpos, int64_t n_tiles, double2* J, float4* Jf, double2* bt, hipStream_t stream) {
  if (n_tiles == 0) return hipSuccess;
  if (Jf)
    hipLaunchKernelGGL((bal_pack_kernel<true>), dim3(unsigned((n_tiles + 3) / 4)), dim3(256), 0, stream, values, b, slot_epos,
                       slot_fpos, slot_bpos, n_tiles, J, Jf, bt);
  else
    hipLaunchKernelGGL((bal_pack_kernel<false>), dim3(unsigned((n_tiles + 3) / 4)), dim3(256), 0, stream, values, b, slot_epos,
                       slot_fpos, slot_bpos, n_tiles, J, Jf, bt);
  return hipGetLastError();
}

hipError_t LaunchBalInvert9(double* blocks, const int64_t* cam_diag_off, int n_cameras, int* fail_flag, const LmFuse& lm,
                            const CamGather& gather, hipStream_t stream) {
  const bool wg_per_camera = gather.parts && !gather.few;
  if (n_cameras > 0) hipLaunchKernelGGL(bal_invert9_kernel, dim3(wg_per_camera ? n_cameras : (n_cameras + 6) / 7), dim3(wg_per_camera ? 64 * kInvertGatherWaves : 64), 0, stream, blocks, cam_diag_off, n_cameras, fail_flag, lm, gather);
  return hipGetLastError();
}

hipError_t LaunchBalCameraItems(bool schur, const double* values, const CamItems& items, const int32_t* cam_fpos,
                                const int32_t* cam_slot, const double* Mo, double* parts, hipStream_t stream) {
  if (items.count == 0) return hipSuccess;
  const dim3 grid((items.count + 3) / 4);
  // Items of a few hundred observations at most (small problems: plan.cc cuts them short so that they spread over the chip; cameras
  // of video-like scenes: tens of observations): the matrix-pipe kernel — Ladybug shape 0.079 -> 0.060 ms, Dubrovnik 0.034 -> 0.026,
  // 52 800 cameras of 38 observations 0.42 -> 0.19 (profiles/r03zj_*).  Full-length items (kCamChunk = 512 observations, the Venice
  // shape): the lane-per-observation kernel keeps 64 records in flight per wavefront and is 8 % ahead there.
  // CERES_HIP_CAM_ITEMS_MFMA=0 / 1 forces one of them (A/B).
  static const int forced = [] { const char* e = getenv("CERES_HIP_CAM_ITEMS_MFMA"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  const bool mfma = forced >= 0 ? forced == 1 : items.observations < int64_t(384) * items.count;
  if (mfma) {
    if (schur) hipLaunchKernelGGL((bal_camera_items_mfma_kernel<true>), grid, dim3(256), 0, stream, values, items, cam_fpos, cam_slot, Mo, parts);
    else hipLaunchKernelGGL((bal_camera_items_mfma_kernel<false>), grid, dim3(256), 0, stream, values, items, cam_fpos, cam_slot, Mo, parts);
  } else {
    if (schur) hipLaunchKernelGGL((bal_camera_items_kernel<true>), grid, dim3(256), 0, stream, values, items, cam_fpos, cam_slot, Mo, parts);
    else hipLaunchKernelGGL((bal_camera_items_kernel<false>), grid, dim3(256), 0, stream, values, items, cam_fpos, cam_slot, Mo, parts);
  }
  return hipGetLastError();
}

hipError_t LaunchBalCameraFinish(const double* parts, const int32_t* cam_item_ptr, const double* D_f, const int32_t* cam_pos,
                                 const int64_t* cam_diag_off, double* blocks, double* camsq, int n_cameras, hipStream_t stream, const double* extra) {
  if (n_cameras > 0) hipLaunchKernelGGL(bal_camera_finish_kernel, dim3((n_cameras + 6) / 7), dim3(64), 0, stream, parts, cam_item_ptr, D_f, cam_pos,
                                        cam_diag_off, blocks, camsq, n_cameras, extra);
  return hipGetLastError();
}

hipError_t LaunchBalCameraChunk(const ZUnits& units, const double* ring, double* acc, const int* status, hipStream_t stream) {
  if (units.count == 0) return hipSuccess;
  const int waves = (units.count + 6) / 7;
  hipLaunchKernelGGL(bal_camera_chunk_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, units, ring, acc, status);
  return hipGetLastError();
}

}  // namespace chip
