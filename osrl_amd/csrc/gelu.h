// GELU (exact, erf form: torch.nn.GELU() default, cdt.py's TransformerBlock mlp) and its derivative: one definition
// for every kernel that evaluates them.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_g(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
}
