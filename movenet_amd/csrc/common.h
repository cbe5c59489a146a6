// Shared host/device helpers for the movenet HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "movenet_hip.h"

// Timing / diagnostic builds only (python -m movenet_amd.csrc.build --stamps --exp=N; the fused layer backward's
// 71-74 and 76, fused_bwd_l.h): 0 = the product, which is all the tracked library ever is.
#ifndef MVN_EXP
#define MVN_EXP 0
#endif

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

// The A/B switches of the full-sequence path (cross-checks between kernel forms in the tests, same-box comparisons):
// parsed from the environment ONCE per process, at first use -- a forward call used to make ~20 getenv calls, several
// inside its per-layer loop.  mvn_reload_switches() parses them again (tests that change a switch inside one process;
// movenet_amd.ops calls it before each pass when MOVENET_DEBUG_GUARD is set, which the test suite does).
struct Switches {
  int forward_tile;            // MOVENET_HIP_FORWARD_TILE: 0 (strip kernel), 32 or 64 (the tile kernels)
  bool head_f32;               // MOVENET_HIP_HEAD_MFMA=f32: the head's fp32 kernels
  bool forward_f32;            // MOVENET_HIP_FORWARD_MFMA=f32: the fp32-MFMA strip forward
  bool wgrad_f32;              // MOVENET_HIP_WGRAD_MFMA=f32
  bool no_fused_forward;       // MOVENET_HIP_NO_FUSED_FORWARD=1
  bool no_persistent_forward;  // MOVENET_HIP_NO_PERSISTENT_FORWARD=1
  bool no_dense_strip;         // MOVENET_HIP_NO_DENSE_STRIP=1
  bool no_side_stream;         // MOVENET_HIP_NO_SIDE_STREAM=1
  bool no_fused_backward;      // MOVENET_HIP_NO_FUSED_BACKWARD=1: the generic two-kernel forms
  bool bwd_split;              // MOVENET_HIP_BWD_FORM=split: the two fused halves of r2 / r3
  bool embed_scalar;           // MOVENET_HIP_EMBED_GRAD=scalar
};
const Switches &switches();
void parse_switches();

// ---- device side --------------------------------------------------------
__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : kLeakySlope * x; }

// exp(x) for x <= 0 -- every softmax / cross-entropy kernel's exponential, the maximum already subtracted (r4c): 2^n times
// v_exp_f32 of the remainder, x log2(e) - n formed with the constant in two pieces, ~2 ulp.  Nine vector instructions where
// libm's expf compiles to ~25: the two loss kernels issued 3000 - 4100 of them per wave, most of it exponentials.  ONE
// form for all of them: the fused forward + loss node must return the bits of the plain forward's probabilities.
__device__ __forceinline__ float sm_exp(float x) {
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(x, 1.44269502162933349609375f, -n);   // log2(e), high part (24 bits)
  r = __builtin_fmaf(x, 1.925963033500011e-8f, r);               // ... its remainder
  const float p = __builtin_amdgcn_exp2f(r);
  return x < -103.0f ? 0.0f : __builtin_ldexpf(p, (int)n);       // (-inf and anything below the subnormals: exactly 0)
}

// Column access of the softmax / loss kernels (lane = column, a class row per register): the (Q, S) tensor of ONE sequence
// as a raw buffer, the lane's column as its per-lane byte offset, the class row as a SCALAR byte offset.  As 64-bit pointers
// the 64 loads and 64 stores of a thread cost ~380 vector instructions of address arithmetic (v_mad_u64_u32,
// v_lshl_add_u64) and ~130 registers of addresses (r4c: the loss kernels ran at two waves per SIMD).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t col_rsrc(const float *p) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7FFFFFFF, 0x00020000);  // raw buffer, 32-bit data format (gfx9)
}
__device__ __forceinline__ float col_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void col_st(float x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}

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
