// Diagnostic probe (not part of the library): what bounds one mat-vec phase of the fp16 generator
// (256 rows x 128 inputs, one workgroup of 8 waves on one CU)?  Shader cycles per phase of
//   mode 0: 64 v_dot2c_f32_f16 per wave of waves 0-3 (the r2 form: one wave per SIMD)
//   mode 1: 32 v_dot2c_f32_f16 per wave of all 8 waves (the r3 all-waves form)
//   mode 2: 8 v_mfma_f32_16x16x32_f16 per wave of all 8 waves (two chains of four k-steps)
//   mode 3: mode 2 + the four ds_read_b128 of the operand vector
//   mode 4: mode 1 + its four ds_read_b128
//   mode 5: mode 3 / mode 4 half and half: four MFMAs + 16 v_dot2c per wave, all 8 waves (do the
//           fp16 matrix cores run BESIDE the vector ALU?)
//   mode 6: the same on waves 0-3 only (one wave per SIMD: eight MFMAs + 32 v_dot2c each)
//   mode 7: waves 0-3 only, 64 v_dot2c each + the four ds_read_b128 (the r2 form, measured properly)
//   mode 8: waves 0-3 eight MFMAs each (half of the rows), waves 4-7 32 v_dot2c each (the other half):
//           one MFMA wave and one VALU wave per SIMD -- do the two pipes run side by side ACROSS waves?
// each repeated `iters` times between two barriers (the barrier pair is timed too).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/h16_phase scripts/probes/h16_phase.hip && /tmp/h16_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float *out, long long *cycles, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 xv[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) xv[tid] = (_Float16)(0.001f * tid);
  __syncthreads();
  h8 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[i][e] = (_Float16)(0.01f * ((tid + i + e) & 7));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f4v d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    h8 x[4];
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = ((const h8 *)xv)[4 * (lane >> 4) + i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = w[(i + it) & 15];
    }
    if (MODE == 0) {
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[(i * 4 + e) & 7] = __builtin_amdgcn_fdot2(h2{w[i][2 * e], w[i][2 * e + 1]}, h2{x[i & 3][2 * e], x[i & 3][2 * e + 1]},
                                                         acc[(i * 4 + e) & 7], false);
      }
    } else if (MODE == 1 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[(i * 4 + e) & 7] = __builtin_amdgcn_fdot2(h2{w[i][2 * e], w[i][2 * e + 1]}, h2{x[i & 3][2 * e], x[i & 3][2 * e + 1]},
                                                       acc[(i * 4 + e) & 7], false);
    } else if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[k], x[k], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 + k], x[k], d1, 0, 0, 0);
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[k], x[k], d0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[e] = __builtin_amdgcn_fdot2(h2{w[4 + k][2 * e], w[4 + k][2 * e + 1]}, h2{x[k][2 * e], x[k][2 * e + 1]}, acc[e], false);
      }
    } else if (MODE == 6) {
      if (wave < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[k], x[k], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 + k], x[k], d1, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[e] = __builtin_amdgcn_fdot2(h2{w[8 + k][2 * e], w[8 + k][2 * e + 1]}, h2{x[k][2 * e], x[k][2 * e + 1]}, acc[e], false);
            acc[4 + e] = __builtin_amdgcn_fdot2(h2{w[12 + k][2 * e], w[12 + k][2 * e + 1]}, h2{x[k][2 * e], x[k][2 * e + 1]}, acc[4 + e], false);
          }
        }
      }
    } else if (MODE == 8) {
      if (wave < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[k], x[k], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4 + k], x[k], d1, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[(i * 4 + e) & 7] = __builtin_amdgcn_fdot2(h2{w[i][2 * e], w[i][2 * e + 1]}, h2{x[i & 3][2 * e], x[i & 3][2 * e + 1]},
                                                         acc[(i * 4 + e) & 7], false);
      }
    } else if (MODE == 7) {
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[(i * 4 + e) & 7] = __builtin_amdgcn_fdot2(h2{w[i][2 * e], w[i][2 * e + 1]}, h2{x[i & 3][2 * e], x[i & 3][2 * e + 1]},
                                                         acc[(i * 4 + e) & 7], false);
      }
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(d0), "+v"(d1));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[tid] = s + d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
  if (lane == 0) cycles[wave] = t1 - t0;
}

template <int MODE>
void run(const char *what, float *out, long long *cyc, int iters) {
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0;
  for (int i = 0; i < 8; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%-78s %8.1f cycles per phase (incl. barrier)\n", what, (double)mx / iters);
}

int main() {
  float *out;
  long long *cyc;
  hipMalloc(&out, 512 * sizeof(float));
  hipMalloc(&cyc, 8 * sizeof(long long));
  const int iters = 20000;
  run<0>("mode 0: 64 v_dot2c per wave, waves 0-3 only (r2 form)", out, cyc, iters);
  run<1>("mode 1: 32 v_dot2c per wave, all 8 waves (r3 all-waves form)", out, cyc, iters);
  run<2>("mode 2: 8 v_mfma_f32_16x16x32_f16 per wave, all 8 waves", out, cyc, iters);
  run<3>("mode 3: mode 2 + 4 ds_read_b128 of the vector", out, cyc, iters);
  run<4>("mode 4: mode 1 + 4 ds_read_b128 of the vector", out, cyc, iters);
  run<5>("mode 5: 4 MFMA + 16 v_dot2c per wave, all 8 waves, + 4 ds_read_b128", out, cyc, iters);
  run<6>("mode 6: 8 MFMA + 32 v_dot2c per wave, waves 0-3 only, + 4 ds_read_b128", out, cyc, iters);
  run<7>("mode 7: 64 v_dot2c per wave, waves 0-3 only, + 4 ds_read_b128 (r2 form)", out, cyc, iters);
  run<8>("mode 8: waves 0-3 8 MFMA, waves 4-7 32 v_dot2c (half the rows each), + reads", out, cyc, iters);
  return 0;
}
