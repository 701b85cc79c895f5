// gather.h -- the device side of the replay sampler (rng.hip osrl_replay_gather / osrl_step_begin), shared with the
// one-launch regression step of mlp.hip: a row's index is a pure function of (seed, step, stream, row), so any kernel
// that knows the step draws the same minibatch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "philox.h"

#define OSRL_MAX_FIELDS 8

namespace osrl_gather {

struct GatherArgs {
  const float* src[OSRL_MAX_FIELDS];
  float* dst[OSRL_MAX_FIELDS];
  int32_t width[OSRL_MAX_FIELDS];
  float scale[OSRL_MAX_FIELDS];
  int32_t n_fields, batch;
  int64_t n_rows;
  int32_t* idx_out;
  uint32_t k0, k1, stream_id;
  const osrl_step_state_t* st;
};

// One row of every table, index `idx` -> batch row `b`, by L lanes (lane id l): ALL fields' loads are requested before
// the first store.  (As a loop "for each field: load, scale, store" the fields were one dependent round trip each --
// six for the CPQ tables -- and the step prologue, which nothing overlaps, took 10.8 us: profiles/r3_timeline.txt.)
// Columns beyond 2 L of a wide table take the plain loop.
template <int L, class AR>
__device__ __forceinline__ void gather_row(AR a, int64_t idx, int b, int l) {
  float v[OSRL_MAX_FIELDS][2];
  const int nf = a.n_fields;
#pragma unroll
  for (int f = 0; f < OSRL_MAX_FIELDS; ++f) {
    const bool on = f < nf;
    const int w = on ? a.width[f] : 0;
    const float* __restrict__ s = (on ? a.src[f] : a.src[0]) + (on ? (size_t)idx * w : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = l + j * L;
      const float x = s[c < w ? c : 0];  // (clamped address + select: no branch around the load)
      v[f][j] = c < w ? x : 0.f;
    }
  }
#pragma unroll
  for (int f = 0; f < OSRL_MAX_FIELDS; ++f) {
    if (f < nf) {
      const int w = a.width[f];
      const float sc = a.scale[f];
      float* __restrict__ d = a.dst[f] + (size_t)b * w;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = l + j * L;
        if (c < w) d[c] = v[f][j] * sc;
      }
      if (w > 2 * L) {
        const float* __restrict__ s = a.src[f] + (size_t)idx * w;
        for (int c = l + 2 * L; c < w; c += L) d[c] = s[c] * sc;
      }
    }
  }
}

// one wave per sampled row; lanes stride over the row's columns (coalesced both sides)
// AR: `const GatherArgs&` (kernel argument by value) or `const OSRL_CAS GatherArgs&` (device-resident block, argmem.h)
// NT: the workgroup size when it is a compile-time constant (0: read blockDim -- an s_load from the hidden kernarg block)
template <class AR, int NT = 0>
__device__ __forceinline__ void gather_body(AR a, uint32_t step, int block) {
  using namespace osrl_rng;
  const int lane = threadIdx.x & 63;
  const int b = block * (NT ? NT / 64 : (int)(blockDim.x >> 6)) + (threadIdx.x >> 6);
  if (b >= a.batch) return;
  const U4 r = philox4x32_10(U4{(uint32_t)b, 0x5eedu, step, a.stream_id}, a.k0, a.k1);
  // 64-bit multiply-shift maps a 64-bit uniform onto [0, n_rows) (bias < 2^-40 for n_rows < 2^24)
  const uint64_t u = ((uint64_t)r.x << 32) | r.y;
  const int64_t idx = (int64_t)__umul64hi(u, (uint64_t)a.n_rows);
  if (lane == 0 && a.idx_out) a.idx_out[b] = (int32_t)idx;
  gather_row<64, AR>(a, idx, b, lane);
}

// rows [16 tile, 16 tile + 16) by one 8-wave workgroup in ONE pass: half a wave per row (the same draws, the same rows
// as gather_body; two passes of a wave per row are two dependent index -> row round trips)
template <class AR>
__device__ __forceinline__ void gather_tile16(AR a, uint32_t step, int tile) {
  using namespace osrl_rng;
  const int lane = threadIdx.x & 63, l = lane & 31;
  const int b = tile * 16 + 2 * (int)(threadIdx.x >> 6) + (lane >> 5);
  if (b >= a.batch) return;
  const U4 r = philox4x32_10(U4{(uint32_t)b, 0x5eedu, step, a.stream_id}, a.k0, a.k1);
  const uint64_t u = ((uint64_t)r.x << 32) | r.y;
  const int64_t idx = (int64_t)__umul64hi(u, (uint64_t)a.n_rows);
  if (l == 0 && a.idx_out) a.idx_out[b] = (int32_t)idx;
  gather_row<32, AR>(a, idx, b, l);
}

// host: the descriptor of osrl_replay_gather's arguments (false: invalid)
inline bool fill(GatherArgs& a, int32_t n_fields, const float* const* src, float* const* dst, const int32_t* width,
                 const float* scale, int64_t n_rows, int32_t batch, uint64_t seed, uint32_t stream_id,
                 const osrl_step_state_t* st) {
  for (int f = 0; f < OSRL_MAX_FIELDS; ++f) {
    a.src[f] = f < n_fields ? src[f] : nullptr;
    a.dst[f] = f < n_fields ? dst[f] : nullptr;
    a.width[f] = f < n_fields ? width[f] : 0;
    a.scale[f] = (f < n_fields && scale) ? scale[f] : 1.0f;
    if (f < n_fields && (!a.src[f] || !a.dst[f] || a.width[f] < 1)) return false;
  }
  a.n_fields = n_fields;
  a.batch = n_fields > 0 ? batch : 0;
  a.n_rows = n_rows;
  a.idx_out = nullptr;
  a.k0 = (uint32_t)seed;
  a.k1 = (uint32_t)(seed >> 32);
  a.stream_id = stream_id;
  a.st = st;
  return true;
}

}  // namespace osrl_gather
