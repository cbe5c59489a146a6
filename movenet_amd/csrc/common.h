// Shared host/device helpers for the movenet HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "movenet_hip.h"

namespace mvn {

constexpr int kWave = 64;
constexpr float kLeakySlope = 0.01f;  // F.leaky_relu default (movenet/modules.py:140-141)

// ---- host side ----------------------------------------------------------
void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);

inline int n_layers(const mvn_dims *d) { return d->layer_size * d->stack_size; }
inline int dilation_of(const mvn_dims *d, int l) { return 1 << (l % d->layer_size); }
inline long long dilation_sum(const mvn_dims *d) {
  return (long long)d->stack_size * ((1LL << d->layer_size) - 1);
}
int validate_dims(const mvn_dims *d);
// Raise a kernel's dynamic-LDS limit to the CU's 160 KiB on the CURRENT device, once per
// (kernel, device): hipFuncSetAttribute applies to the current device's copy of the code
// object, so a per-process flag would leave a second device at the 64 KiB default.
int ensure_max_dynamic_lds(const void *kernel, const char *what);

// ---- device side --------------------------------------------------------
__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : kLeakySlope * x; }

// tanh(f) * sigmoid(g), accurate libm forms (parity with the CPU path matters
// more than the ~100 cycles they cost once per layer).
__device__ __forceinline__ float gate(float f, float g) {
  return tanhf(f) * (1.0f / (1.0f + expf(-g)));
}
// tanh and sigmoid from v_exp_f32 / v_rcp_f32 (absolute error ~1e-7; ~8 VALU ops each vs
// ~40-100 for the libm forms): the full-sequence forward evaluates 2 x 64 of them per thread
__device__ __forceinline__ float tanh_fast(float f) {
  const float a = __expf(-2.0f * fabsf(f));  // in (0, 1]
  return copysignf((1.0f - a) * __builtin_amdgcn_rcpf(1.0f + a), f);
}
__device__ __forceinline__ float sigmoid_fast(float g) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-g));  // e^-g may overflow to +inf: 1/inf = 0 is right
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Philox4x32-10, counter = (c0,c1,c2,c3), key = 64-bit seed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1) {
  uint32_t c[4] = {c0, c1, 0x6d766e31u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

}  // namespace mvn
