// Diagnostic probe (not part of the library): how many vector-ALU / LDS instructions of the SAME
// wave fit in the shadow of one v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles)?  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void probe(float *out, long long *cycles, int iters) {
  __shared__ v4f lds[512];
  const int tid = threadIdx.x;
  lds[tid] = v4f{1.f * tid, 2.f, 3.f, 4.f};
  lds[tid + 256] = v4f{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float a = tid * 1e-3f, b = 1.0f;
  v2f p[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  const v2f w = {1.0001f, 0.9999f}, x = {tid * 1e-3f, 1.f};
  v4f s = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) p[v & 3] = __builtin_elementwise_fma(w, x, p[v & 3]);
        else s += lds[(tid + 64 * v) & 511];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(a), "+v"(b));
  }
  const long long t1 = __builtin_readcyclecounter();
  float res = p[0].x + p[1].y + p[2].x + p[3].y + s.x + s.y + s.z + s.w;
  for (int r = 0; r < 16; ++r) res += acc0[r] + acc1[r];
  out[tid] = res;
  if ((tid & 63) == 0) cycles[tid >> 6] = t1 - t0;
}

template <int NV, int KIND>
static void run(const char *what) {
  float *out;
  long long *cyc, h[4];
  hipMalloc(&out, 256 * 4);
  hipMalloc(&cyc, 32);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("1 MFMA + %2d %-14s per step: %7.1f cycles per step\n", NV, what, (double)h[0] / iters / 8);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("v_pk_fma_f32");
  run<4, 0>("v_pk_fma_f32");
  run<8, 0>("v_pk_fma_f32");
  run<12, 0>("v_pk_fma_f32");
  run<16, 0>("v_pk_fma_f32");
  run<2, 1>("ds_read_b128");
  run<4, 1>("ds_read_b128");
  run<8, 1>("ds_read_b128");
  return 0;
}
