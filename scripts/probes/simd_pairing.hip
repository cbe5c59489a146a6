// Probe: how do the two waves of a SIMD share it when both run [V vector instructions][M bf16 MFMAs] streams?
// One workgroup on one CU, 256 threads (one wave per SIMD) and 512 (two), EVERY wave timed (the older wave of a SIMD wins
// the arbitration: timing wave 0 alone flatters).  The stream of a wave is R x { V x v_sub_f32 on 8 independent registers ;
// M x v_mfma_f32_32x32x16_bf16 on ACCS accumulators in turn }, the groups kept in place by scheduling barriers; PRIO = 1
// raises the wave's priority for its vector block (s_setprio 1 ... 0), PRIO = 2 for its MFMA block; PRIO 10 .. 16 run the real
// stage of fused_bwd_l.h (split of eight values, then the six MFMAs it feeds): 10 / 11 one stage per loop iteration (mask as
// a literal / in an SGPR), 12 two, 13 four stages between branches, 14 - 16 four stages with a taken branch to the next
// instruction / s_sleep 0 / nops around such a branch between them, 17 / 18 four stages with s_setprio 1 around every MFMA block /
// every split.  Per "unit" = 44 vector instructions + 6 MFMAs: cycles
// of the SIMD.  Measured (MI355X, gpurun_out/simd_pairing.txt): vector-only 236 alone / 118 per unit with two waves; MFMA-only
// 204 / 197 (the pipe); the stage 400 alone and 222 per unit with two waves when every stage ends in the loop's branch, but 250 /
// 246 with two / four stages between branches -- then wave 0 runs at 400 - 408 and the younger wave gets what is left --
// and no yield point tried (14 - 16) brings the even sharing back; priorities change nothing (226 / 222 / 221) or hurt (17: 265,
// 18: 258).  ONE wave never overlaps its own vector instructions with its own MFMAs: [7 v_sub][1 MFMA] loops run at 72 cycles
// per iteration alone (= 7 x 5.4 + 32 + the branch), the stage at 360 - 400 (= split + 6 x 32): whatever hides under an MFMA is
// the OTHER wave's work, which is why two waves per SIMD (256 registers each) beat one wave of 512 for this arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the real split of fused_bwd_l.h (44 instructions per 8 values): mask as a 32-bit literal (8-byte v_and_b32) or in a register
template <bool MASK_REG>
__device__ __forceinline__ void split8(float *x, u32x4 &h, u32x4 &m, u32x4 &l, unsigned mask_r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned mk = MASK_REG ? mask_r : 0xffff0000u;
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ah = __float_as_uint(a) & mk, bh = __float_as_uint(b) & mk;
    const float ar = a - __uint_as_float(ah), br = b - __uint_as_float(bh);
    const unsigned am = __float_as_uint(ar) & mk, bm = __float_as_uint(br) & mk;
    const float ar2 = ar - __uint_as_float(am), br2 = br - __uint_as_float(bm);
    h[i] = __builtin_amdgcn_perm(bh, ah, 0x07060302);
    m[i] = __builtin_amdgcn_perm(bm, am, 0x07060302);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
  }
}

template <int V, int M, int ACCS, int PRIO>
__global__ __launch_bounds__(512) void probe(float *out, const float *src, long long *cycles, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  u32x4 pa = {0x3f803f80u + tid, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, pb = pa;
  float x[8], y = src[tid];
  for (int i = 0; i < 8; ++i) x[i] = src[tid * 8 + i];
  __syncthreads();
  unsigned mask_r = 0xffff0000u;
  asm volatile("" : "+s"(mask_r));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PRIO >= 10) {  // the real stage: split (PRIO 10: literal mask, 11: mask in an SGPR), then six MFMAs fed by it
      u32x4 h, m, l;
      if (PRIO == 18) __builtin_amdgcn_s_setprio(1);
      if (PRIO == 10) split8<false>(x, h, m, l, mask_r); else split8<true>(x, h, m, l, mask_r);
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO == 17) __builtin_amdgcn_s_setprio(1);
      if (PRIO == 18) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#define MFX(a_, b_) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc[0], 0, 0, 0)
      MFX(l, pa); MFX(h, pb); MFX(m, pb); MFX(m, pa); MFX(h, pb); MFX(h, pa);
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO == 17) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
      if (PRIO >= 12) {  // 12: a second stage per round on the other accumulator; 13: three more (four stages between branches)
#pragma unroll
        for (int rep = 0; rep < (PRIO == 12 ? 1 : 3); ++rep) {
          if (PRIO == 14) asm volatile("s_branch 1f\n1:");        // a taken branch to the next instruction: a yield point?
          if (PRIO == 15) __builtin_amdgcn_s_sleep(0);
          if (PRIO == 16) asm volatile("s_nop 0\n s_branch 1f\n s_nop 0\n1:");
          __builtin_amdgcn_sched_barrier(0);
          if (PRIO == 18) __builtin_amdgcn_s_setprio(1);
          split8<false>(x, h, m, l, mask_r);
          __builtin_amdgcn_sched_barrier(0);
          if (PRIO == 17) __builtin_amdgcn_s_setprio(1);
          if (PRIO == 18) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#define MFY(a_, b_) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc[1], 0, 0, 0)
          MFY(l, pa); MFY(h, pb); MFY(m, pb); MFY(m, pa); MFY(h, pb); MFY(h, pa);
          __builtin_amdgcn_sched_barrier(0);
          if (PRIO == 17) __builtin_amdgcn_s_setprio(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
        }
      }
      continue;
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(y));
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      acc[i % ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[i % ACCS], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = 0.f;
  for (int r = 0; r < 16; ++r) res += acc[0][r] + acc[1][r];
  for (int i = 0; i < 8; ++i) res += x[i];
  out[tid] = res;
  if (lane == 0) { cycles[2 * wave] = t0; cycles[2 * wave + 1] = t1; }
}

template <int V, int M, int ACCS, int PRIO>
static void run() {
  float *out, *src;
  long long *cyc, h[16];
  hipMalloc(&out, 512 * 4);
  hipMalloc(&src, 512 * 8 * 4);
  hipMemset(src, 0x3f, 512 * 8 * 4);
  hipMalloc(&cyc, sizeof h);
  const int iters = 4000 * 6 / (M ? M : 6);
  double res[2], w0[2];
  for (int k = 0; k < 2; ++k) {
    const int threads = 256 << k;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<V, M, ACCS, PRIO>), dim3(1), dim3(threads), 0, 0, out, src, cyc, iters);
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    long long b = h[0], e = h[1];
    for (int w = 1; w < threads / 64; ++w) { b = b < h[2 * w] ? b : h[2 * w]; e = e > h[2 * w + 1] ? e : h[2 * w + 1]; }
    const double units = (double)iters * (M ? M / 6.0 : V / 44.0) * (PRIO == 12 ? 2 : PRIO >= 13 ? 4 : 1);
    res[k] = (double)(e - b) / units / (k + 1);
    w0[k] = (double)(h[1] - h[0]) / units;
  }
  printf("V %3d  M %2d  accs %d  prio %d : one wave %6.1f | two waves: per unit and SIMD %6.1f (wave 0 alone %6.1f)\n", V, M, ACCS, PRIO, res[0],
         res[1], w0[1]);
  hipFree(out); hipFree(src); hipFree(cyc);
}

int main() {
  run<44, 6, 1, 10>();
  run<44, 6, 1, 11>();
  run<44, 6, 1, 12>();
  run<44, 6, 1, 13>();
  run<44, 6, 1, 14>();
  run<44, 6, 1, 15>();
  run<44, 6, 1, 16>();
  run<44, 6, 1, 17>();
  run<44, 6, 1, 18>();
  run<44, 0, 1, 0>();
  run<0, 6, 1, 0>();
  run<44, 6, 1, 0>();
  run<88, 12, 1, 0>();
  run<176, 24, 1, 0>();
  run<352, 48, 1, 0>();
  run<7, 1, 1, 0>();
  run<15, 2, 1, 0>();
  run<22, 3, 1, 0>();
  run<44, 6, 2, 0>();
  run<7, 1, 2, 0>();
  run<44, 6, 1, 1>();
  run<88, 12, 1, 1>();
  run<176, 24, 1, 1>();
  run<7, 1, 1, 1>();
  run<44, 6, 1, 2>();
  run<88, 12, 1, 2>();
  run<176, 24, 1, 2>();
  run<7, 1, 1, 2>();
  return 0;
}
