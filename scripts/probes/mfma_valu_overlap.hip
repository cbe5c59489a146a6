// Diagnostic probe (not part of the library): do the matrix core and the vector ALU of a SIMD run
// instructions of two different waves at the same time?  One workgroup of 8 waves on one CU
// (two per SIMD): waves 0-3 issue v_mfma_f32_32x32x2_f32 back to back, waves 4-7 one of
// {nothing, v_pk_fma_f32, v_exp_f32, ds_read_b128, global_load_dwordx4}.  Cycles per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/o scripts/probes/mfma_valu_overlap.hip && /tmp/o
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OTHER>
__global__ __launch_bounds__(512) void probe(float *out, const v4f *gsrc, long long *cycles, int iters, int mfma_on,
                                             int prio) {
  __shared__ v4f lds[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lds[tid] = v4f{1.f * tid, 2.f, 3.f, 4.f};
  lds[tid + 512] = v4f{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float res = 0.f;
  // prio 1: the non-MFMA waves raise their priority (s_setprio 3); prio 2: the MFMA waves also
  // yield after every MFMA pair (s_setprio 0 is the default)
  if (prio && wave >= 4) __builtin_amdgcn_s_setprio(3);
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (mfma_on) {
      f32x16 acc0, acc1;
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
      float a = tid * 1e-3f, b = 1.0f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
      }
      for (int r = 0; r < 16; ++r) res += acc0[r] + acc1[r];
    }
  } else {
    if (OTHER == 1) {  // packed FMAs, 4 chains: 64 per iteration
      v2f p0 = {0.f, 0.f}, p1 = p0, p2 = p0, p3 = p0;
      const v2f w = {1.0001f, 0.9999f}, x = {tid * 1e-3f, 1.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          p0 = __builtin_elementwise_fma(w, x, p0);
          p1 = __builtin_elementwise_fma(w, x, p1);
          p2 = __builtin_elementwise_fma(w, x, p2);
          p3 = __builtin_elementwise_fma(w, x, p3);
        }
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      }
      res = p0.x + p1.y + p2.x + p3.y;
    } else if (OTHER == 2) {  // transcendental: 32 v_exp_f32 per iteration, 4 chains
      float e0 = 0.1f, e1 = 0.2f, e2 = 0.3f, e3 = 0.4f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
          asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
          asm volatile("v_exp_f32 %0, %0" : "+v"(e2));
          asm volatile("v_exp_f32 %0, %0" : "+v"(e3));
        }
      }
      res = e0 + e1 + e2 + e3;
    } else if (OTHER == 3) {  // LDS: 16 ds_read_b128 per iteration
      v4f s = {0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const v4f v = lds[(lane + 64 * k) & 1023];
          s += v;
        }
        asm volatile("" : "+v"(s));
      }
      res = s.x + s.y + s.z + s.w;
    } else if (OTHER == 4) {  // global: 8 x 16-byte loads per iteration (L2-resident 1 MB)
      v4f s = {0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += gsrc[((it * 8 + k) * 256 + (tid - 256)) & 65535];
        asm volatile("" : "+v"(s));
      }
      res = s.x + s.y + s.z + s.w;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[tid] = res;
  if (lane == 0) cycles[wave] = t1 - t0;
}

template <int OTHER>
static void run(const char *what) {
  float *out;
  v4f *g;
  long long *cyc, h[8];
  hipMalloc(&out, 512 * 4);
  hipMalloc(&g, 65536 * 16);
  hipMemset(g, 0, 65536 * 16);
  hipMalloc(&cyc, 64);
  const int iters = 2000;
  for (int mode : {0, 1, 2}) {
    const int mfma_on = mode > 0, prio = mode == 2;
    hipLaunchKernelGGL(probe<OTHER>, dim3(1), dim3(512), 0, 0, out, g, cyc, iters, mfma_on, prio);
    hipLaunchKernelGGL(probe<OTHER>, dim3(1), dim3(512), 0, 0, out, g, cyc, iters, mfma_on, prio);
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-34s MFMA waves %s%s: mfma wave %8.1f cycles/iter (16 MFMAs = 1024), other wave %8.1f cycles/iter\n", what,
           mfma_on ? "ON " : "off", prio ? ", others s_setprio 3" : "                    ", (double)h[0] / iters,
           (double)h[4] / iters);
  }
  hipFree(out); hipFree(g); hipFree(cyc);
}

int main() {
  run<0>("nothing beside");
  run<1>("64 v_pk_fma_f32 beside");
  run<2>("32 v_exp_f32 beside");
  run<3>("16 ds_read_b128 beside");
  run<4>("8 global_load_dwordx4 beside");
  return 0;
}
