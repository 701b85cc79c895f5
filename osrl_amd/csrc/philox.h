// philox.h -- Philox4x32-10 counter-based generator shared by rng.hip (Gaussian noise, replay indices) and
// cdt.hip (dropout masks).  Counter-based = a value is a pure function of (seed, step, stream, element), so a
// backward kernel regenerates exactly the mask its forward kernel used without storing it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osrl_rng {

struct U4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ inline float u01(uint32_t x) {  // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Dropout masks: element e of dropout site `site` at train step `step` is KEPT iff word (e & 3) of
// philox(counter = {e/4 lo, e/4 hi, step, kDropStream | site}, key = seed) >= thresh, thresh = p * 2^32.
constexpr uint32_t kDropStream = 0x40000000u;

__device__ inline U4 drop_words(uint64_t e4, uint32_t step, uint32_t site, uint32_t k0, uint32_t k1) {
  return philox4x32_10(U4{(uint32_t)e4, (uint32_t)(e4 >> 32), step, kDropStream | site}, k0, k1);
}
__host__ __device__ inline uint32_t drop_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

}  // namespace osrl_rng
