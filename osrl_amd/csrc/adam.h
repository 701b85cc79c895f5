// adam.h -- the per-element Adam update, shared by the streaming optimizer kernel (optim.hip) and the one-launch
// regression step (mlp.hip), so that both produce the same bits from the same gradient.
//     m  = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// (torch.optim.Adam.step: osrl/algorithms/bc.py:54-55, cpq.py:232-238, bcql.py:218-226)
#pragma once
#include <hip/hip_runtime.h>

namespace osrl_adam {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Coef {
  float b1, b2, eps;
  float step_size;  // lr_t / (1 - b1^t)
  float bc2s;       // sqrt(1 - b2^t)
};

// The roundings are spelled out (one fused multiply-add per moment, IEEE divide and square root, one fused
// multiply-add for the parameter): with plain operators the compiler contracts a*b + c*d one way for float4 operands
// (pk_mul + pk_fma) and the other way for scalars, and the two kernels would disagree in the last bit from the second
// step on (when m and v are no longer zero).  This is the float4 form's choice, i.e. what optim.hip has always computed.
__device__ __forceinline__ void update1(float& pv, float& mv, float& vv, const float g, const Coef c) {
  mv = __fmaf_rn(c.b1, mv, __fmul_rn(1.0f - c.b1, g));
  vv = __fmaf_rn(c.b2, vv, __fmul_rn(__fmul_rn(1.0f - c.b2, g), g));
  const float d = __fadd_rn(sqrtf(vv) / c.bc2s, c.eps);
  pv = __fmaf_rn(-c.step_size, mv / d, pv);
}

// Polyak target  t <- tau p + (1 - tau) t   (cpq.py:107-113 _soft_update), the rounding spelled out for the same reason:
// one product rounded, then one fused multiply-add -- in the streaming kernel and in the dW launch's optimizer epilogue
__device__ __forceinline__ float polyak1(const float tau, const float pv, const float tv) {
  return __fmaf_rn(tau, pv, __fmul_rn(1.0f - tau, tv));
}

__device__ __forceinline__ void update4(f32x4& pv, f32x4& mv, f32x4& vv, const f32x4 g, const Coef c) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float p = pv[k], m = mv[k], v = vv[k];
    update1(p, m, v, g[k], c);
    pv[k] = p;
    mv[k] = m;
    vv[k] = v;
  }
}

}  // namespace osrl_adam
