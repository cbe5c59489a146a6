// Probe: the residuals of the exact bf16 x 3 split (bf3.h) through v_dot2c_f32_bf16 instead of "widen the top half,
// subtract": with h = {top16(b), top16(a)} already PACKED (the plane register itself), a - float(top16(a)) is
// dot2(h, {-1, 0}) + a and b - float(top16(b)) is dot2(h, {0, -1}) + b: 7 vector instructions per pair of values
// instead of 11.  Checks that the planes are bit-identical to the reference split over random values of every
// magnitude (and counts the subnormal-residual cases the dot unit may flush), then times both forms, one and two
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 scripts/probes/dot2_split.hip -o dot2_split && ./dot2_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned top(float x) { return __float_as_uint(x) & 0xffff0000u; }
__device__ __forceinline__ void split2_ref(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned ah = top(a), bh = top(b);
  const float ar = a - __uint_as_float(ah), br = b - __uint_as_float(bh);
  const unsigned am = top(ar), bm = top(br);
  const float ar2 = ar - __uint_as_float(am), br2 = br - __uint_as_float(bm);
  h = __builtin_amdgcn_perm(bh, ah, 0x07060302);
  m = __builtin_amdgcn_perm(bm, am, 0x07060302);
  l = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
}
// (the selector of the LOW half lives in a register: written as a constant, 0x0000bf80 is folded into the inline
// operand "-1.0", which the instruction reads as the 32-bit 0xbf800000 -- the HIGH-half selector; first run of this probe)
__device__ __forceinline__ float dot_lo(unsigned p, float c) {  // c - float(low half of p)
  unsigned sel = 0x0000bf80u;
  asm volatile("" : "+v"(sel));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, sel), c, false);
}
__device__ __forceinline__ float dot_hi(unsigned p, float c) {  // c - float(high half of p)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0xbf800000u), c, false);
}
__device__ __forceinline__ void split2_dot(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302);
  const float ar = dot_lo(h, a), br = dot_hi(h, b);
  m = __builtin_amdgcn_perm(__float_as_uint(br), __float_as_uint(ar), 0x07060302);
  const float ar2 = dot_lo(m, ar), br2 = dot_hi(m, br);
  l = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
}

__global__ void check(const float *x, unsigned *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h, m, l, h2, m2, l2;
  split2_ref(x[2 * i], x[2 * i + 1], h, m, l);
  split2_dot(x[2 * i], x[2 * i + 1], h2, m2, l2);
  out[6 * i + 0] = h; out[6 * i + 1] = m; out[6 * i + 2] = l;
  out[6 * i + 3] = h2; out[6 * i + 4] = m2; out[6 * i + 5] = l2;
}

template <int FORM>
__global__ __launch_bounds__(1024) void timed(const float *x, unsigned *out, long long *cyc, int iters) {
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = x[threadIdx.x * 8 + j];
  unsigned acc = 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h, m, l;
      if (FORM == 0) split2_ref(v[2 * j], v[2 * j + 1], h, m, l);
      else split2_dot(v[2 * j], v[2 * j + 1], h, m, l);
      acc ^= h ^ m ^ l;
      v[2 * j] = __uint_as_float((__float_as_uint(v[2 * j]) ^ (l & 0x00010001u)));  // keeps the chain alive, two cheap ops
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  // every wave reports: the OLDEST wave of a SIMD has issue priority, timing it alone hides what the others wait
  if ((threadIdx.x & 63) == 0) {
    cyc[2 * (blockIdx.x * 16 + (threadIdx.x >> 6))] = t0;
    cyc[2 * (blockIdx.x * 16 + (threadIdx.x >> 6)) + 1] = t1;
  }
}

int main() {
  const int n = 1 << 22;
  float *hx = (float *)malloc(n * 4), *dx;
  unsigned *ho = (unsigned *)malloc((size_t)n * 3 * 4), *dout;
  srand(7);
  for (int i = 0; i < n; ++i) {
    unsigned bits;
    const int kind = i & 7;
    if (kind < 5) {  // any finite float
      do bits = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31); while (((bits >> 23) & 255) == 255);
    } else if (kind == 5) {  // activation-like
      float f = (float)rand() / RAND_MAX * 8.f - 4.f; memcpy(&bits, &f, 4);
    } else if (kind == 6) {  // tiny normals: residuals go subnormal
      bits = ((unsigned)(1 + rand() % 24) << 23) | ((unsigned)rand() & 0x7fffff) | ((unsigned)(rand() & 1) << 31);
    } else {  // subnormals and zeros
      bits = (rand() & 3) ? ((unsigned)rand() & 0x7fffff) | ((unsigned)(rand() & 1) << 31) : ((unsigned)(rand() & 1) << 31);
    }
    memcpy(&hx[i], &bits, 4);
  }
  hipMalloc(&dx, n * 4);
  hipMalloc(&dout, (size_t)n * 3 * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  check<<<n / 2 / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, (size_t)n * 3 * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0, sum_bad = 0;
  for (int i = 0; i < n / 2; ++i) {
    const unsigned *r = ho + 6 * i;
    if (r[0] != r[3] || r[1] != r[4] || r[2] != r[5]) {
      ++bad;
      // do the planes still sum to the value?  (the only thing the products need)
      for (int half = 0; half < 2; ++half) {
        auto f = [&](unsigned p) { unsigned b = half ? (p & 0xffff0000u) : (p << 16); float v; memcpy(&v, &b, 4); return v; };
        const float x = hx[2 * i + half];
        const float s = (f(r[5]) + f(r[4])) + f(r[3]);
        if (s != x) ++sum_bad;
        if (fabsf(x) > 1e-30f && s != x) {
          if (bad_normal < 8) printf("value %a: planes %08x %08x %08x vs %08x %08x %08x\n", x, r[0], r[1], r[2], r[3], r[4], r[5]);
          ++bad_normal;
        }
      }
    }
  }
  printf("pairs %d  planes differ %ld  plane sums != value %ld  of those |x| > 1e-30: %ld\n", n / 2, bad, sum_bad, bad_normal);

  static long long hc[512 * 16 * 2];  // (grid <= 512, <= 16 waves each: begin, end)
  long long *dc;
  hipMalloc(&dc, sizeof hc);
  const int iters = 4096;
  for (int form = 0; form < 2; ++form)
    for (int threads : {256, 512, 1024, 2048}) {  // 1 / 2 / 4 / 8 waves per SIMD (2048: two workgroups of 1024 per CU)
      const int tpb = threads > 1024 ? 1024 : threads, grid = 256 * (threads / tpb);
      for (int rep = 0; rep < 2; ++rep) {
        if (form == 0) timed<0><<<grid, tpb>>>(dx, dout, dc, iters);
        else timed<1><<<grid, tpb>>>(dx, dout, dc, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
      // per workgroup: first begin .. last end of its waves; with two workgroups per CU the pairing is unknown, so the
      // SIMD figure below assumes they overlap fully (threads / 256 waves per SIMD)
      double s = 0, s0 = 0;
      const int nw = tpb / 64;
      for (int i = 0; i < grid; ++i) {
        long long b = hc[2 * (i * 16)], e = hc[2 * (i * 16) + 1];
        s0 += (double)(e - b);
        for (int w = 1; w < nw; ++w) {
          b = b < hc[2 * (i * 16 + w)] ? b : hc[2 * (i * 16 + w)];
          e = e > hc[2 * (i * 16 + w) + 1] ? e : hc[2 * (i * 16 + w) + 1];
        }
        s += (double)(e - b);
      }
      const double per_wg = s / grid / iters, w0 = s0 / grid / iters;
      printf("form %s, %d waves per SIMD: wave 0 alone %.1f, all waves of a workgroup %.1f cycles per round (split of 8 values + 8 chain ops each) = %.1f cycles of the SIMD per wave and round\n",
             form ? "dot2" : "widen+sub", threads / 256, w0, per_wg, per_wg / (threads / 256));
    }
  return bad_normal != 0;
}
