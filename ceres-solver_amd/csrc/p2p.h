// p2p.h — the device side of the one-shot peer-to-peer exchange, as a WAVE-level primitive that a producer kernel
// calls on the sums it has just finished: the camera-space reduction of a tile pass (bal_reduce_exchange_kernel), the camera-major
// pass's per-camera blocks (bal_invert_kernel), the step's final scalars (collect_scalars_kernel).  Round 6: with the collective a
// kernel of its own, one rank of eight spent more of a Venice-sized CG iteration in the all-reduce launch (28-46 us: a handful of
// workgroups each pushing a 2048-element chunk to eight peers) and the launches around it than in its tile pass (30 us;
// profiles/r06a_*).  Inside the producer the push of a chunk starts when that chunk is summed, chunk by chunk across the device.
//
// Buffers (solver.hip: ceres_hip_comm_p2p_prepare): every rank owns receive slots [parity][source rank][cap] doubles and arrival
// flags [parity][source rank][chunks_cap], mapped by all peers.  A chunk = up to 64 consecutive slot indices = one wavefront:
//   push   : lane l stores its value into slot index `idx` of [epoch & 1][my rank] at EVERY rank (its own included); one
//            system-scope fence by the wave; lane q < world then stores `epoch` into the chunk's flag at rank q (release);
//   wait   : lane q spins (system-scope acquire, s_sleep, wall-clock timeout) on the flag that source rank q sets in MY buffer;
//   reduce : every lane adds MY slots over source ranks in rank order: identical bits on every rank.
// Slots are double-buffered by epoch parity: a writer reaching epoch e + 2 has completed e + 1, which needed every peer's e + 1 flag,
// which a peer sets only after its epoch-e kernel (the reader of the slot) finished — so EVERY rank must run the exchange of EVERY
// epoch, whatever it thinks of the data (a kernel gated by the CG status word still shakes hands: callers pass zeros).
// Chunk indices of one epoch are the caller's own numbering (any numbering below chunks_cap, the same on every rank).
#ifndef CERES_HIP_P2P_H_
#define CERES_HIP_P2P_H_

#include <hip/hip_runtime.h>

#include "device.h"

namespace chip {

// Memory ordering without cache maintenance: every access to the slots and flags is itself a SYSTEM-scope relaxed atomic (sc0 sc1 on
// gfx950: written through / read past this device's caches), and program order between them is enforced by waiting for the wave's
// outstanding memory operations (s_waitcnt vmcnt(0): the pushes are acknowledged before the flags are stored; the flags are seen
// before the slots are loaded — a wave issues in order).  A system-scope release / acquire FENCE instead writes back and invalidates
// the whole L2 (buffer_wbl2 / buffer_inv sc0 sc1): with one exchange per wavefront, 1778 wavefronts doing that at once made one
// camera-block exchange 130 us (profiles/r06b_*).  CERES_HIP_P2P_FENCES=1 adds the fences back (P2pComm::fences; A/B and fall-back).
__device__ __forceinline__ void p2p_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One wavefront: `v` = this lane's contribution to slot index `idx` (active lanes hold distinct indices below cap; inactive lanes pass
// active = false), `chunk` = the flag index of this wave's chunk in this epoch.  Returns the sum over ranks (NaN after a timeout, which
// also raises the communicator's error flags).  All 64 lanes must call it together.
__device__ __forceinline__ double p2p_exchange_wave(const P2pComm& C, int chunk, long long idx, bool active, double v) {
  const int lane = threadIdx.x & 63;
  const int parity = int(C.epoch & 1ull);
  const long long mine_off = (static_cast<long long>(parity) * C.world + C.rank) * C.cap + idx;
  const bool broken = __hip_atomic_load(C.error_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (active && !broken) {
    for (int q = 0; q < C.world; ++q) __hip_atomic_store(C.peers.slots[q] + mine_off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (C.fences) __threadfence_system();
  p2p_drain();
  int timed_out = broken ? 1 : 0;
  if (lane < C.world && !broken) {
    const int q = lane;
    __hip_atomic_store(C.peers.flags[q] + (static_cast<long long>(parity) * C.world + C.rank) * C.chunks_cap + chunk, C.epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* f = C.peers.flags[C.rank] + (static_cast<long long>(parity) * C.world + q) * C.chunks_cap + chunk;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < C.epoch) {
      if (wall_clock64() - t0 > C.timeout_ticks) {   // a peer never arrived: do not hang the GPU
        __hip_atomic_store(C.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(C.error_seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  timed_out = __any(timed_out);
  p2p_drain();
  if (C.fences) __threadfence_system();
  double s = 0.0;
  if (active) {
    const double* my = C.peers.slots[C.rank] + static_cast<long long>(parity) * C.world * C.cap + idx;
    for (int q = 0; q < C.world; ++q) s += __hip_atomic_load(my + static_cast<long long>(q) * C.cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // an incomplete sum must not be consumed: NaN flows into the CG scalars, whose tests then end the solve, and the host reports
  // CERES_HIP_E_COMM at its next poll
  return timed_out ? __builtin_nan("") : s;
}

// K chunks by one wavefront in ONE round trip (the stand-alone all-reduce of a long vector; the camera-major pass of tens of thousands
// of cameras): chunk k < K has flag index chunk0 + k and holds the slot indices idx0 + stride k + l for lanes l < lanes; indices at or
// beyond n_end do not exist (the tail).  Same protocol per chunk as above.
template <int K>
__device__ __forceinline__ void p2p_exchange_wave_multi(const P2pComm& C, int chunk0, long long idx0, int stride, int lanes, long long n_end, double (&v)[K]) {
  const int lane = threadIdx.x & 63;
  const int parity = int(C.epoch & 1ull);
  const long long base = (static_cast<long long>(parity) * C.world + C.rank) * C.cap;
  const bool broken = __hip_atomic_load(C.error_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (!broken && lane < lanes) {
    for (int q = 0; q < C.world; ++q) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const long long i = idx0 + static_cast<long long>(stride) * k + lane;
        if (i < n_end) __hip_atomic_store(C.peers.slots[q] + base + i, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (C.fences) __threadfence_system();
  p2p_drain();
  int timed_out = broken ? 1 : 0;
  if (lane < C.world && !broken) {
    const int q = lane;
    unsigned long long* fo = C.peers.flags[q] + (static_cast<long long>(parity) * C.world + C.rank) * C.chunks_cap + chunk0;
    const unsigned long long* fi = C.peers.flags[C.rank] + (static_cast<long long>(parity) * C.world + q) * C.chunks_cap + chunk0;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (idx0 + static_cast<long long>(stride) * k < n_end) __hip_atomic_store(fo + k, C.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    for (int k = 0; k < K && !timed_out; ++k) {
      if (idx0 + static_cast<long long>(stride) * k >= n_end) break;
      while (__hip_atomic_load(fi + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < C.epoch) {
        if (wall_clock64() - t0 > C.timeout_ticks) {
          __hip_atomic_store(C.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(C.error_seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          timed_out = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  timed_out = __any(timed_out);
  p2p_drain();
  if (C.fences) __threadfence_system();
  const double* my = C.peers.slots[C.rank] + static_cast<long long>(parity) * C.world * C.cap;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const long long i = idx0 + static_cast<long long>(stride) * k + lane;
    double s = 0.0;
    if (i < n_end && lane < lanes)
      for (int q = 0; q < C.world; ++q) s += __hip_atomic_load(my + static_cast<long long>(q) * C.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v[k] = timed_out ? __builtin_nan("") : s;
  }
}

}  // namespace chip
#endif
