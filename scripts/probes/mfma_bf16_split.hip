// Diagnostic probe (not part of the library): what would an fp32 product cost on the bf16 matrix cores?
// An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 mantissa bits); six of the nine partial
// products (hh, hm, mh, mm, hl, lh) carry everything above 2^-24 of the product.  Per 32 x 32 x 16 block:
// 6 v_mfma_f32_32x32x16_bf16 against 8 v_mfma_f32_32x32x2_f32.  Measured here, one workgroup on one CU:
//   (a) cycles per MFMA, back to back, 1 and 2 waves per SIMD, both instruction forms;
//   (b) a wave that also SPLITS its 8 fp32 B values per block (and/sub/perm: ~26 VALU instructions) between
//       the MFMAs -- same wave, and with a second wave on the SIMD;
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/b scripts/probes/mfma_bf16_split.hip && /tmp/b
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }
// three bf16 parts of 8 floats, packed as 3 x (4 registers of two bf16 each)
__device__ __forceinline__ void split8(const float *x, u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ah = hi16(a), bh = hi16(b);
    const float ar = a - __uint_as_float(ah), br = b - __uint_as_float(bh);
    const unsigned am = hi16(ar), bm = hi16(br);
    const float ar2 = ar - __uint_as_float(am), br2 = br - __uint_as_float(bm);
    h[i] = __builtin_amdgcn_perm(bh, ah, 0x07060302);  // {b.hi16, a.hi16}
    m[i] = __builtin_amdgcn_perm(bm, am, 0x07060302);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
  }
}

// MODE 0: fp32 MFMAs (8 per block, 4 row blocks); 1: bf16 MFMAs only (6 per block, 4 row blocks);
// 2: bf16 MFMAs + the split of the block's 8 B values in the same wave
template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, const float *src, long long *cycles, int iters, int waves) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float res = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wave < waves) {
    f32x16 acc[4];
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    if (MODE == 0) {
      float a = tid * 1e-3f, b = 1.0f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[blk], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
      }
    } else {
      u32x4 wh = {0x3f803f80u + tid, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, wm = wh, wl = wh;
      float x[8];
      for (int i = 0; i < 8; ++i) x[i] = src[tid * 8 + i];
      u32x4 h = wh, m = wh, l = wh;
      for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
          split8(x, h, m, l);
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), am = __builtin_bit_cast(bf16x8, wm), al = __builtin_bit_cast(bf16x8, wl);
          const bf16x8 bh = __builtin_bit_cast(bf16x8, h), bm = __builtin_bit_cast(bf16x8, m), bl = __builtin_bit_cast(bf16x8, l);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[blk], 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[blk], 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[blk], 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[blk], 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[blk], 0, 0, 0);
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[blk], 0, 0, 0);
        }
        asm volatile("" : "+v"(wh), "+v"(wm), "+v"(wl));
      }
    }
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 16; ++r) res += acc[b][r];
  }
  const long long t1 = __builtin_readcyclecounter();
  out[tid] = res;
  if (lane == 0) cycles[wave] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int mfmas) {
  float *out, *src;
  long long *cyc, h[8];
  hipMalloc(&out, 512 * 4);
  hipMalloc(&src, 512 * 8 * 4);
  hipMemset(src, 0x3f, 512 * 8 * 4);
  hipMalloc(&cyc, 64);
  const int iters = 2000;
  for (int waves : {4, 8}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, src, cyc, iters, waves);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, src, cyc, iters, waves);
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double c = (double)h[0] / iters;
    printf("%-58s %d wave(s) per SIMD: %8.1f cycles per 32x32x16 block x 4 row blocks (%d MFMAs: %.1f each); per SIMD and block %.1f\n",
           what, waves / 4, c, mfmas, c / mfmas, c / 4 * (waves / 4));
  }
  hipFree(out); hipFree(src); hipFree(cyc);
}

int main() {
  run<0>("8 x v_mfma_f32_32x32x2_f32 per block (today)", 32);
  run<1>("6 x v_mfma_f32_32x32x16_bf16 per block", 24);
  run<2>("6 x v_mfma_f32_32x32x16_bf16 per block + split of 8 B values", 24);
  return 0;
}
