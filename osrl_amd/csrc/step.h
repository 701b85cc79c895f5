// step.h -- the device step-state tick shared by osrl_step_tick (optim.hip) and osrl_step_begin (rng.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"

namespace osrl_step {

// commit the previous step's statistics into the ring (all threads of the calling workgroup; NT = its size when known
// at compile time: `blockDim` is a load from the hidden kernarg block)
template <int NT = 0>
__device__ __forceinline__ void commit_stats(int64_t t_old, const float* __restrict__ stats_cur,
                                             float* __restrict__ ring, int n_stats, int ring_len) {
  if (stats_cur && ring && t_old >= 1) {
    const int slot = (int)((t_old - 1) % ring_len);
    for (int i = threadIdx.x; i < n_stats; i += (NT ? NT : (int)blockDim.x)) ring[(size_t)slot * n_stats + i] = stats_cur[i];
  }
}

// everything derived from the step count t (bias corrections, warm-up scale)
struct Tick {
  float bc1, bc2_sqrt, lr_scale;
};
__device__ __forceinline__ Tick tick_values(int64_t t, float beta1, float beta2, int warmup) {
  Tick k;
  k.bc1 = (float)(1.0 - pow((double)beta1, (double)t));
  k.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)t));
  // LambdaLR(min((s+1)/warmup, 1)) with s = number of scheduler steps taken = t-1 (cdt.py:327-330,409)
  k.lr_scale = warmup > 0 ? (float)fmin((double)t / (double)warmup, 1.0) : 1.0f;
  return k;
}

// t = t_old + 1 and everything derived from it (ONE thread)
__device__ __forceinline__ void advance(osrl_step_state_t* st, int64_t t_old, float beta1, float beta2, int warmup) {
  const int64_t t = t_old + 1;
  const Tick k = tick_values(t, beta1, beta2, warmup);
  st->step = t;
  st->bc1 = k.bc1;
  st->bc2_sqrt = k.bc2_sqrt;
  st->lr_scale = k.lr_scale;
}

}  // namespace osrl_step
