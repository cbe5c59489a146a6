// Diagnostic probe (not part of the library): issue cost in shader cycles of the instruction
// forms the generator kernels choose between, one workgroup of 8 waves on one CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rates scripts/probes/issue_rates.hip && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float *out, long long *cycles, int iters, int active_waves) {
  __shared__ v4f lds[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lds[tid] = v4f{1.f * tid, 2.f, 3.f, 4.f};
  lds[tid + 512] = v4f{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float x = out[tid], w = 1.0001f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  v2f p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f};
  const v2f w2 = {w, w}, x2 = {x, x};
  long long t0 = 0, t1 = 0;
  if (wave < active_waves) {

    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {  // 16 packed FMAs, 4 chains
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          p0 = __builtin_elementwise_fma(w2, x2, p0);
          p1 = __builtin_elementwise_fma(w2, x2, p1);
          p2 = __builtin_elementwise_fma(w2, x2, p2);
          p3 = __builtin_elementwise_fma(w2, x2, p3);
        }
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      } else if (MODE == 10) {  // 16 packed FMAs, 8 chains
        v2f q0 = p0, q1 = p1, q2 = p2, q3 = p3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          p0 = __builtin_elementwise_fma(w2, x2, p0);
          p1 = __builtin_elementwise_fma(w2, x2, p1);
          p2 = __builtin_elementwise_fma(w2, x2, p2);
          p3 = __builtin_elementwise_fma(w2, x2, p3);
          q0 = __builtin_elementwise_fma(w2, x2, q0);
          q1 = __builtin_elementwise_fma(w2, x2, q1);
          q2 = __builtin_elementwise_fma(w2, x2, q2);
          q3 = __builtin_elementwise_fma(w2, x2, q3);
        }
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        p0 += q0; p1 += q1; p2 += q2; p3 += q3;
      } else if (MODE == 12) {  // 16 packed FMAs, 2 chains
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          p0 = __builtin_elementwise_fma(w2, x2, p0);
          p1 = __builtin_elementwise_fma(w2, x2, p1);
        }
        asm volatile("" : "+v"(p0), "+v"(p1));
      } else if (MODE == 13) {  // 16 packed FMAs, 1 chain
#pragma unroll
        for (int k = 0; k < 16; ++k) p0 = __builtin_elementwise_fma(w2, x2, p0);
        asm volatile("" : "+v"(p0));
      } else if (MODE == 14) {  // 16 v_fmac_f32, 1 chain
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(x), "v"(w));
      } else if (MODE == 15) {  // 16 v_fmac_f32, 16 chains
        float c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = a0;
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(c[k]) : "v"(x), "v"(w));
#pragma unroll
        for (int k = 0; k < 16; ++k) a1 += c[k];
      } else if (MODE == 16) {  // v_exp_f32 x4 dependent
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(a0));
      } else if (MODE == 17) {  // DPP quad add x2 dependent (the chain's lane sum)
        asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a0));
      } else if (MODE == 1) {  // 16 v_fmac_f32_dpp row_newbcast, 4 chains
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(x), "v"(w));
        }
      } else if (MODE == 2) {  // 16 v_fmac_f32_dpp quad_perm
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(x), "v"(w));
        }
      } else if (MODE == 3) {  // 16 plain v_fmac_f32
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(x), "v"(w));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(x), "v"(w));
        }
      } else if (MODE == 4) {  // 4 ds_read_b128, 4 distinct addresses per wave (the x-vector pattern)
        v4f r0, r1, r2, r3;
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) v4f *)lds + 64u * (lane & 3);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        a0 += r0.x + r1.y + r2.z + r3.w;
      } else if (MODE == 5) {  // 4 ds_read_b128, all lanes distinct addresses
        v4f r0, r1, r2, r3;
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) v4f *)lds + 16u * lane;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                     "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        a0 += r0.x + r1.y + r2.z + r3.w;
      } else if (MODE == 6) {  // 1 ds_read_b32
        float r0;
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) v4f *)lds + 4u * lane;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(addr) : "memory");
        a0 += r0;
      } else if (MODE == 7) {  // permlane16_swap + permlane32_swap pair
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
      } else if (MODE == 8) {  // 16 v_readlane + v_fmac with the SGPR
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "s"(sx), "v"(w));
        }
      } else if (MODE == 9) {  // 4 ds_read_b128, one address for the whole wave
        v4f r0, r1, r2, r3;
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) v4f *)lds + 64u * wave;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        a0 += r0.x + r1.y + r2.z + r3.w;
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  out[tid] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (lane == 0) cycles[wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int per_iter) {
  float *out;
  long long *cyc;
  hipMalloc(&out, 512 * 4);
  hipMalloc(&cyc, 8 * 8);
  hipMemset(out, 0, 512 * 4);
  const int iters = 2000;
  for (int waves : {1, 4, 8}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, iters, waves);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, iters, waves);
    long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    // __builtin_readcyclecounter = s_memtime: shader cycles
    long long mx = 0;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-52s waves %d: wave 0 %8.2f, slowest %8.2f cycles/iter (x%d ops)\n", name, waves, (double)h[0] / iters,
           (double)mx / iters, per_iter);
  }
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0>("16 v_pk_fma_f32 (4 chains)", 16);
  run<10>("16 v_pk_fma_f32 (8 chains)", 16);
  run<12>("16 v_pk_fma_f32 (2 chains)", 16);
  run<13>("16 v_pk_fma_f32 (1 chain)", 16);
  run<3>("16 v_fmac_f32 (4 chains)", 16);
  run<14>("16 v_fmac_f32 (1 chain)", 16);
  run<15>("16 v_fmac_f32 (16 chains) + 16 adds", 32);
  run<16>("4 v_exp_f32 dependent", 4);
  run<17>("2 dependent DPP quad adds", 2);
  run<1>("16 v_fmac_f32_dpp row_newbcast (4 chains)", 16);
  run<2>("16 v_fmac_f32_dpp quad_perm (4 chains)", 16);
  run<8>("16 x (v_readlane + v_fmac sgpr)", 16);
  run<7>("permlane16_swap + permlane32_swap", 2);
  run<4>("4 ds_read_b128, 4 addresses/wave + wait", 4);
  run<9>("4 ds_read_b128, 1 address/wave + wait", 4);
  run<5>("4 ds_read_b128, 64 addresses/wave + wait", 4);
  run<6>("1 ds_read_b32 + wait", 1);
  return 0;
}
