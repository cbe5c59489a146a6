// Diagnostic probe (not part of the library): what the fp32 matrix cores sustain when nothing else
// is in the way -- v_mfma_f32_32x32x2_f32 from registers, 4 independent accumulators per wave,
// every CU busy.  The roofline's 157.3 TFLOP/s is 256 CUs x 256 FLOP/clk x 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_peak.bin scripts/probes/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    asm volatile("" : "+v"(a), "+v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs, int threads, const char *what) {
  float *out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    printf("%-46s rep %d: %8.3f ms  %7.1f TFLOP/s\n", what, rep, ms, flop / ms * 1e-9);
  }
  hipFree(out);
}

int main() {
  run<4>(2048, 256, "4 waves/WG, 4 accumulators, 2048 WGs");
  run<1>(2048, 256, "4 waves/WG, 1 accumulator (dependent chain)");
  run<2>(2048, 256, "4 waves/WG, 2 accumulators");
  run<4>(256, 256, "one WG per CU (1 wave/SIMD), 4 accumulators");
  run<1>(256, 256, "one WG per CU (1 wave/SIMD), dependent chain");
  return 0;
}
