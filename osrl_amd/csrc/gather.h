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
  for (int f = 0; f < a.n_fields; ++f) {
    const int w = a.width[f];
    const float* __restrict__ s = a.src[f] + (size_t)idx * w;
    float* __restrict__ d = a.dst[f] + (size_t)b * w;
    const float sc = a.scale[f];
    for (int c = lane; c < w; c += 64) d[c] = s[c] * sc;
  }
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
  for (int f = 0; f < a.n_fields; ++f) {
    const int w = a.width[f];
    const float* __restrict__ s = a.src[f] + (size_t)idx * w;
    float* __restrict__ d = a.dst[f] + (size_t)b * w;
    const float sc = a.scale[f];
    for (int c = l; c < w; c += 32) d[c] = s[c] * sc;
  }
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
