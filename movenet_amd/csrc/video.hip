// Video encoder and learned upsampler of the local conditioning path
// (/root/reference/movenet/wavenet.py:94-118 construction, :149-156 upsample_video):
//
//   video (B, F, 64, 64, Cin) --permute--> Conv3d(Cin -> C, kernel (1,64,64)) -> (B, C, F)
//        --> 3 x ConvTranspose1d(C -> C, kernel 10, stride 10) --> (B, C, 1000 F)
//
// The Conv3d consumes a whole frame per output: per (b, f) a C x 4096*Cin mat-vec
// (video_conv_kernel: frames staged in LDS, weight rows streamed coalesced).  A
// transposed conv with stride == kernel has no overlap, so it IS a 1x1 product
// with 10*C output rows (row = (c_out, tap)) whose epilogue scatters row (co, j),
// column i to out[co][10 i + j]: the shared gemm_wx / wgrad MFMA families
// (gemm_family.h) do forward, data gradient and weight gradient.
#include <algorithm>

#include "common.h"
#include "gemm_family.h"

namespace mvn {

constexpr int kUp = 10;     // kernel == stride of every upsampling layer
constexpr int kPix = 4096;  // 64 x 64 pixels per frame

// ---- Conv3d(k = (1,64,64)) ----------------------------------------------------
// grid (ceil(F / FB), B), 256 threads; dynamic LDS = FB * Cin * 4096 floats
__global__ __launch_bounds__(256) void video_conv_kernel(const float *__restrict__ video,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ bias, Act enc,
                                                         int C, int Cin, int F, int FB) {
  extern __shared__ __attribute__((aligned(16))) float frames[];  // [FB][Cin][4096]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, f0 = blockIdx.x * FB;
  const int nf = min(FB, F - f0);
  for (int i = tid; i < nf * Cin * kPix; i += 256) {
    const int f = i / (Cin * kPix), r = i - f * Cin * kPix, cin = r / kPix, p = r - cin * kPix;
    frames[i] = video[(((size_t)b * F + f0 + f) * kPix + p) * Cin + cin];
  }
  __syncthreads();
  for (int c = wave; c < C; c += 4) {
    for (int f = 0; f < nf; ++f) {
      float acc = 0.f;
      for (int cin = 0; cin < Cin; ++cin) {
        const float *wr = w + ((size_t)c * Cin + cin) * kPix;
        const float *fr = frames + ((size_t)f * Cin + cin) * kPix;
#pragma unroll 8
        for (int p = lane; p < kPix; p += 64) acc = fmaf(wr[p], fr[p], acc);
      }
      acc = wave_sum(acc);
      if (lane == 0) *enc.at(b, c, f0 + f) = acc + bias[c];
    }
  }
}

// dW[c][cin][p] += sum_f denc[b][c][f] * video[b][f][p][cin]; grid (Cin*4096/256, ceil(C/8), B)
__global__ __launch_bounds__(256) void video_conv_wgrad_kernel(const float *__restrict__ video,
                                                               Act denc, float *__restrict__ dw,
                                                               int C, int Cin, int F) {
  const int k = blockIdx.x * 256 + threadIdx.x;  // cin * 4096 + p
  const int c0 = blockIdx.y * 8, b = blockIdx.z;
  if (k >= Cin * kPix) return;
  const int cin = k / kPix, p = k - cin * kPix;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < F; ++f) {
    const float v = video[(((size_t)b * F + f) * kPix + p) * Cin + cin];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) acc[j] = fmaf(*denc.at(b, c0 + j, f), v, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c0 + j < C) atomicAdd(dw + (size_t)(c0 + j) * Cin * kPix + k, acc[j]);
}

// db[c] += sum_{b,f} d[b][c][f]  (one wave per channel)
__global__ void rowsum_kernel(Act d, float *__restrict__ db, int batch, int n) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < batch; ++b)
    for (int i = lane; i < n; i += 64) s += *d.at(b, c, i);
  s = wave_sum(s);
  if (lane == 0) db[c] += s;
}

// ---- ConvTranspose1d(k = 10, stride = 10) as a 10C-row 1x1 product ---------------
// weight (C_in, C_out, 10): W(m = co*10 + j, k = ci) = w[(ci*C + co)*10 + j]
struct UpOp {
  int K, t_begin, t_end, C;  // t = input time i
  const float *wt, *bias;
  Act xin, out;
  __device__ __forceinline__ float w(int m, int k) const {
    return (m < kUp * C && k < C) ? wt[((size_t)k * C + m / kUp) * kUp + m % kUp] : 0.f;
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    return (k < C && t < t_end) ? *xin.at(b, k, t) : 0.f;
  }
  __device__ __forceinline__ void epilogue(int b, int mb, int t, int lane, const f32x16 &a0,
                                           const f32x16 &a1) const {
    // row m = U + 4 (lane >> 5) with U = 64 mb + 32 half + (r & 3) + 8 (r >> 2) uniform: the two
    // divisions by 10 per output run on the scalar unit, a lane only selects (as per-lane divisions
    // plus 64-bit addresses the 32 outputs of an epilogue spilled 66 registers)
    float *base = out.at(b, 0, 0);
    const bool hi = lane >= 32;
    const unsigned tcol = (unsigned)(kUp * t);
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int U = mb * 64 + 32 * half + (r & 3) + 8 * (r >> 2);
        const int co_a = U / kUp, j_a = U - co_a * kUp, co_b = (U + 4) / kUp, j_b = (U + 4) - co_b * kUp;
        const int co = hi ? co_b : co_a, j = hi ? j_b : j_a;
        if (U + (hi ? 4 : 0) < kUp * C)
          base[(unsigned)(co * out.ld + j) + tcol] = (half ? a1[r] : a0[r]) + bias[(unsigned)co];
      }
  }
};

// data gradient: dx[ci][i] = sum_{co,j} w[ci][co][j] * dout[co][10 i + j]
struct UpDxOp {
  int K, t_begin, t_end, C;
  const float *wt;
  Act dout, dx;
  __device__ __forceinline__ float w(int m, int k) const {
    return (m < C && k < kUp * C) ? wt[(size_t)m * C * kUp + k] : 0.f;  // k = co*10 + j
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= kUp * C || t >= t_end) return 0.f;
    const int co = k / kUp, j = k - co * kUp;
    return *dout.at(b, co, kUp * t + j);
  }
  __device__ __forceinline__ void epilogue(int b, int mb, int t, int lane, const f32x16 &a0,
                                           const f32x16 &a1) const {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m0 = mb * 64 + acc_row(r, lane), m1 = m0 + 32;
      if (m0 < C) *dx.at(b, m0, t) = a0[r];
      if (m1 < C) *dx.at(b, m1, t) = a1[r];
    }
  }
};

// weight gradient: dw[ci][co][j] += sum_{b,i} dout[co][10 i + j] * x[ci][i]; db[co] += sum dout
struct UpWgOp {
  int t_begin, t_end, C;
  Act dout, xin;
  float *dwt, *dbt;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    if (m >= kUp * C) return 0.f;
    const int co = m / kUp, j = m - co * kUp;
    return *dout.at(b, co, kUp * t + j);
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    return n < C ? *xin.at(b, n, t) : 0.f;
  }
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (m >= kUp * C || n >= C) return nullptr;
    const int co = m / kUp, j = m - co * kUp;
    return dwt + ((size_t)n * C + co) * kUp + j;
  }
  __device__ __forceinline__ float *db(int m) const { return m < kUp * C ? dbt + m / kUp : nullptr; }
};

// ----------------------------------------------------------------------------------------
// r4: the up-sampler's backward as ONE kernel per layer (C = 64).  The generic forms above read dout[co][10 i + j] as
// the operand "row (co, j), column i" -- a stride of ten floats between the lanes of every global load: the weight
// gradient ran at 3.5 % matrix-core utilisation and 330 GB/s, the data gradient at 6 % (profiles/r03_pmc_summary.json).
// Here a workgroup stages a tile of dout as it lies in memory -- 64 rows x 320 CONTIGUOUS columns (32 input steps) -- and
// x (64 x 32) in LDS, and the stride-ten access happens on LDS reads:
//   dW[(co, j)][ci] += sum_i dout[co][10 i + j] x[ci][i]   40 blocks of 32 x 32 over 8 waves (five each, 80 accumulators),
//   dx[ci][i]        = sum_(co, j) w[ci][(co, j)] dout[co][10 i + j]   K = 640 split over the waves (80 each, the wave's
//                      slice of w in 80 registers for the whole launch), the eight partial sums meet in LDS,
//   db[co]          += sum dout (row sums at staging).
// fp32 MFMAs (v_mfma_f32_32x32x2_f32): 160 per wave and tile.  The weight gradient leaves by atomic adds once per workgroup.
// ----------------------------------------------------------------------------------------
struct UpBwdArgs {
  int n;                 // input steps of the layer (its output has 10 n)
  const float *wt;       // (C_in, C_out, 10)
  Act dout, xin, dx;     // dout (B, C, >= 10 n), xin / dx (B, C, >= n)
  float *dwt, *dbt;
};
constexpr int UB_TT = 32, UB_DP = 10 * UB_TT + 4, UB_XP = UB_TT + 1;            // pitches: 324 (dout rows), 33 (x rows)
constexpr int UB_LDS_FLOATS = 64 * UB_DP + 64 * UB_XP + 8 * 2 * 16 * 64;         // tiles + the eight dx partial sums

__global__ __launch_bounds__(512, 1) void up_bwd64_kernel(UpBwdArgs a, int tiles_per_b, int tiles_per_wg,
                                                         float *__restrict__ part, float *__restrict__ bias_part) {
  constexpr int C = 64, TT = UB_TT, DP = UB_DP, XP = UB_XP;
  extern __shared__ __attribute__((aligned(16))) float ub_lds[];
  float *Dt = ub_lds;                    // [64 co][DP]: dout[co][10 t0 + u], u < 320
  float *Xt = Dt + 64 * DP;              // [64 ci][XP]
  float *Ps = Xt + 64 * XP;              // [8 waves][2 blocks][16][64 lanes]: partial dx
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kk = lane >> 5;
  // ---- this wave's slice of w as the dx product's A operand: rows ci = 32 blk + li, K = its 8 output channels x 10 taps in
  // the order step s -> (co = 8 wave + s / 5, j = 2 (s % 5) + kk): every LDS address of the product is then ONE per-lane
  // base plus a compile-time constant (as k = 80 wave + 2 s + kk the (co, j) pairs of a lane were 40 loop-invariant
  // registers, and 93 registers lived in scratch)
  float wreg[2][40];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int s = 0; s < 40; ++s)
      wreg[blk][s] = a.wt[(size_t)(32 * blk + li) * (kUp * C) + (8 * wave + s / 5) * kUp + 2 * (s % 5) + kk];
  const int dbase = 8 * wave * DP + kk + kUp * li;
  // ---- weight gradient: 40 blocks of 32 x 32 ((co, j) rows x ci columns), five per wave: m blocks wave and wave + 8 with
  // both ci blocks, and ci block wave & 1 of m block 16 + (wave >> 1)
  f32x16 accw[5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
  const int mb2 = 16 + (wave >> 1), nb2 = wave & 1;
  int aoff[3];  // A operands: row m = 32 mb + li -> (co, j): Dt[co * DP + j + 10 t]
  {
    const int m0 = 32 * wave + li, m1 = 32 * (wave + 8) + li, m2 = 32 * mb2 + li;
    aoff[0] = (m0 / kUp) * DP + (m0 % kUp) + kUp * kk;
    aoff[1] = (m1 / kUp) * DP + (m1 % kUp) + kUp * kk;
    aoff[2] = (m2 / kUp) * DP + (m2 % kUp) + kUp * kk;
  }
  const int xoff0 = li * XP + kk, xoff1 = (32 + li) * XP + kk, xoff2 = (32 * nb2 + li) * XP + kk;
  float bsum = 0.f;  // thread -> dout row tid >> 3 (eight threads per row)
  for (int it = 0; it < tiles_per_wg; ++it) {
    const int tile = blockIdx.x * tiles_per_wg + it;
    if (tile >= tiles_per_b) break;
    const int b = blockIdx.y, t0 = tile * TT;
    // ---- stage dout rows (contiguous: 320 floats = 80 float4 per row, 8 threads x 10) and x
    {
      const int row = tid >> 3, q8 = tid & 7;
      const float *src = a.dout.at(b, row, 0) + (size_t)kUp * t0;
      const int ulim = kUp * min(TT, a.n - t0);
#pragma unroll 5
      for (int i = 0; i < 10; ++i) {
        const int u = 4 * (q8 + 8 * i);
        f4 v = kZero4;
        if (u + 3 < ulim) v = *(const f4 *)(src + u);
        else {
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = u + k < ulim ? src[u + k] : 0.f;
          v = f4{e[0], e[1], e[2], e[3]};
        }
        *(f4 *)&Dt[row * DP + u] = v;
        bsum += (v.x + v.y) + (v.z + v.w);
      }
      for (int i = tid; i < 64 * TT; i += 512) {
        const int r = i >> 5, t = i & 31;
        Xt[r * XP + t] = t0 + t < a.n ? *a.xin.at(b, r, t0 + t) : 0.f;
      }
    }
    __syncthreads();
    // ---- weight gradient: K = the tile's 32 steps (step s: t = 2 s + kk)
#pragma unroll
    for (int s = 0; s < TT / 2; ++s) {
      const float x0 = Xt[xoff0 + 2 * s], x1 = Xt[xoff1 + 2 * s], x2 = Xt[xoff2 + 2 * s];
      const float d0 = Dt[aoff[0] + 2 * kUp * s], d1 = Dt[aoff[1] + 2 * kUp * s], d2 = Dt[aoff[2] + 2 * kUp * s];
      accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, x0, accw[0], 0, 0, 0);
      accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, x1, accw[1], 0, 0, 0);
      accw[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, x0, accw[2], 0, 0, 0);
      accw[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, x1, accw[3], 0, 0, 0);
      accw[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2, x2, accw[4], 0, 0, 0);
      if (s & 1) __builtin_amdgcn_sched_barrier(0);  // (two steps of operands in flight, not sixteen)
    }
    // ---- data gradient: this wave's 8 output channels x 10 taps; B[k = (co, j)][n = i] = Dt[co][10 i + j]
    f32x16 accd[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[blk][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 40; ++s) {
      const float d = Dt[dbase + (s / 5) * DP + 2 * (s % 5)];
      accd[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[0][s], d, accd[0], 0, 0, 0);
      accd[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[1][s], d, accd[1], 0, 0, 0);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ps[((wave * 2 + blk) * 16 + r) * 64 + lane] = accd[blk][r];
    __syncthreads();
    // ---- the eight partial sums meet: thread -> (block, register r, lane): 2048 outputs, four per thread
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = tid + 512 * q, ln = o & 63, r = (o >> 6) & 15, blk = o >> 10;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += Ps[((w * 2 + blk) * 16 + r) * 64 + ln];
      const int ci = 32 * blk + acc_row(r, ln), t = t0 + (ln & 31);
      if (t < a.n) *a.dx.at(b, ci, t) = v;
    }
    __syncthreads();
  }
  // ---- this workgroup's weight-gradient blocks and bias sums: a slab of its own (slab_reduce_kernel's format: 640 x 64,
  // summed in a fixed order) -- as atomic adds into the 41 k words of dw they took ~250 us per launch, five times the
  // rest of the kernel (timing build without them: 58 against 310 us) -- or, without scratch, the atomics
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mb = i < 2 ? wave : i < 4 ? wave + 8 : mb2, nb = i < 4 ? (i & 1) : nb2;
      const int m = 32 * mb + acc_row(r, lane), ci = 32 * nb + li;
      if (part) part[(wg * (kUp * C) + m) * C + ci] = accw[i][r];
      else atomicAdd(a.dwt + (size_t)ci * (kUp * C) + m, accw[i][r]);
    }
  {
    float v = bsum;  // eight threads share a row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if ((tid & 7) == 0) {
      if (bias_part) bias_part[wg * C + (tid >> 3)] = v;
      else atomicAdd(a.dbt + (tid >> 3), v);
    }
  }
}

// db[co] += the workgroups' row sums, in a fixed order
__global__ void up_bias_reduce_kernel(const float *__restrict__ bias_part, int nparts, float *__restrict__ dbt) {
  const int co = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int p = lane; p < nparts; p += 64) s += bias_part[(size_t)p * 64 + co];
  s = wave_sum(s);
  if (lane == 0) dbt[co] += s;
}

static void up_bwd64_geometry(int n, int batch, int *tiles_per_b, int *tiles_per_wg, int *wgs) {
  *tiles_per_b = (n + UB_TT - 1) / UB_TT;
  int cus = 256;
  {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0)
      cus = c;
  }
  // one round of workgroups: each keeps its slice of w in registers over its tiles and writes its gradient blocks once
  const int want = std::max(1, cus / std::max(batch, 1));
  *tiles_per_wg = std::max(1, (*tiles_per_b + want - 1) / want);
  *wgs = (*tiles_per_b + *tiles_per_wg - 1) / *tiles_per_wg;
}
static size_t up_bwd64_scratch_floats(int n, int batch) {
  int tpb, tpw, wgs;
  up_bwd64_geometry(n, batch, &tpb, &tpw, &wgs);
  return (size_t)wgs * batch * (kUp * 64 * 64 + 64);
}

static int launch_up_bwd64(const UpBwdArgs &a, const UpWgOp &op, int batch, float *scratch, size_t scratch_floats, hipStream_t s) {
  if (a.n <= 0 || batch <= 0) return MVN_OK;
  int tiles_per_b, tiles_per_wg, wgs;
  up_bwd64_geometry(a.n, batch, &tiles_per_b, &tiles_per_wg, &wgs);
  const int rc = ensure_max_dynamic_lds((const void *)up_bwd64_kernel, "hipFuncSetAttribute(up_bwd64)");
  if (rc) return rc;
  const size_t nwg = (size_t)wgs * batch, need = nwg * (kUp * 64 * 64 + 64);
  float *part = (scratch && need <= scratch_floats) ? scratch : nullptr;
  float *bias_part = part ? part + nwg * (kUp * 64 * 64) : nullptr;
  hipLaunchKernelGGL(up_bwd64_kernel, dim3(wgs, batch), dim3(512), UB_LDS_FLOATS * sizeof(float), s, a, tiles_per_b, tiles_per_wg,
                     part, bias_part);
  if (part) {
    hipLaunchKernelGGL(slab_reduce_kernel<UpWgOp>, dim3(kUp * 64 * 64 / 32), dim3(32 * RED_SEG), 0, s, op, part, (int)nwg, kUp * 64, 64);
    hipLaunchKernelGGL(up_bias_reduce_kernel, dim3(64), dim3(64), 0, s, bias_part, (int)nwg, a.dbt);
  }
  return MVN_OK;
}

static int check_video_args(const mvn_dims *dims, int batch, int frames, int cin) {
  int rc = validate_dims(dims);
  if (rc) return rc;
  if (batch < 0 || frames < 1 || cin < 1 || cin > 8) {
    set_error("video: batch %d / frames %d / channels %d out of range (channels <= 8)", batch, frames,
              cin);
    return MVN_ERR_BAD_ARG;
  }
  return MVN_OK;
}

}  // namespace mvn

using namespace mvn;

extern "C" {

int mvn_upsample_video(const mvn_dims *dims, const mvn_video_params *vp, const float *video, int batch,
                       int frames, int cin, float *enc, float *u1, float *u2, float *ctx,
                       int ctx_ld, void *stream_) {
  int rc = check_video_args(dims, batch, frames, cin);
  if (rc) return rc;
  if (!vp || !vp->conv_w || !vp->conv_b || !video || !enc || !u1 || !u2 || !ctx ||
      ctx_ld < 1000 * frames) {
    set_error("mvn_upsample_video: NULL buffer or ctx_ld < 1000*frames");
    return MVN_ERR_BAD_ARG;
  }
  for (int i = 0; i < 3; ++i)
    if (!vp->up_w[i] || !vp->up_b[i]) {
      set_error("mvn_upsample_video: NULL upsampler parameter");
      return MVN_ERR_BAD_ARG;
    }
  if (batch == 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = dims->residual_channels, F = frames;
  const int FB = cin == 1 ? 4 : 1;
  const size_t lds = sizeof(float) * (size_t)FB * cin * kPix;
  rc = ensure_max_dynamic_lds((const void *)video_conv_kernel, "hipFuncSetAttribute(video_conv)");
  if (rc) return rc;
  Act encv = act_view(enc, batch, C, mvn_padded_len(F));
  hipLaunchKernelGGL(video_conv_kernel, dim3((F + FB - 1) / FB, batch), dim3(256), lds, s, video,
                     vp->conv_w, vp->conv_b, encv, C, cin, F, FB);
  float *stage[4] = {enc, u1, u2, ctx};
  int len = F;
  for (int i = 0; i < 3; ++i) {
    UpOp u;
    u.K = C; u.t_begin = 0; u.t_end = len; u.C = C; u.wt = vp->up_w[i]; u.bias = vp->up_b[i];
    u.xin = act_view(stage[i], batch, C, mvn_padded_len(len));
    u.out = act_view(stage[i + 1], batch, C, i == 2 ? ctx_ld : mvn_padded_len(len * kUp));
    launch_gemm(u, kUp * C, batch, s);
    len *= kUp;
  }
  return check_hip(hipGetLastError(), "mvn_upsample_video");
}

size_t mvn_upsample_video_scratch_floats(const mvn_dims *dims, int batch, int frames) {
  if (!dims || dims->residual_channels != 64 || batch <= 0 || frames <= 0) return 0;
  return up_bwd64_scratch_floats(frames * kUp * kUp, batch);  // (the last layer's launch is the largest)
}

int mvn_upsample_video_backward(const mvn_dims *dims, const mvn_video_params *vp,
                                const mvn_video_grads *vg, const float *video, int batch, int frames,
                                int cin, const float *enc, const float *u1, const float *u2,
                                const float *dctx, int dctx_ld, float *d_u2, float *d_u1, float *d_enc,
                                float *scratch, size_t scratch_floats, void *stream_) {
  int rc = check_video_args(dims, batch, frames, cin);
  if (rc) return rc;
  if (!vp || !vg || !vg->conv_w || !vg->conv_b || !video || !enc || !u1 || !u2 || !dctx || !d_u2 ||
      !d_u1 || !d_enc || dctx_ld < 1000 * frames) {
    set_error("mvn_upsample_video_backward: NULL buffer");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = dims->residual_channels, F = frames;
  const float *acts[3] = {enc, u1, u2};
  float *dacts[3] = {d_enc, d_u1, d_u2};
  int lens[4] = {F, F * kUp, F * kUp * kUp, F * kUp * kUp * kUp};
  for (int i = 2; i >= 0; --i) {
    // layer i maps acts[i] (length lens[i]) to its output (length lens[i+1])
    Act dout = act_view(const_cast<float *>(i == 2 ? dctx : dacts[i + 1]), batch, C,
                        i == 2 ? dctx_ld : mvn_padded_len(lens[i + 1]));
    Act xin = act_view(const_cast<float *>(acts[i]), batch, C, mvn_padded_len(lens[i]));
    if (!vg->up_w[i] || !vg->up_b[i]) {
      set_error("mvn_upsample_video_backward: NULL gradient pointer");
      return MVN_ERR_BAD_ARG;
    }
    if (C == 64) {  // (r4: one kernel per layer, dout staged as it lies in memory)
      UpBwdArgs ua;
      ua.n = lens[i]; ua.wt = vp->up_w[i]; ua.dout = dout; ua.xin = xin;
      ua.dx = act_view(dacts[i], batch, C, mvn_padded_len(lens[i]));
      ua.dwt = vg->up_w[i]; ua.dbt = vg->up_b[i];
      UpWgOp wo;  // (where the slab words go: slab_reduce_kernel's view of dw (C_in, C_out, 10))
      wo.t_begin = 0; wo.t_end = lens[i]; wo.C = C; wo.dout = dout; wo.xin = xin; wo.dwt = vg->up_w[i]; wo.dbt = vg->up_b[i];
      rc = launch_up_bwd64(ua, wo, batch, scratch, scratch_floats, s);
      if (rc) return rc;
      continue;
    }
    UpWgOp wg;
    wg.t_begin = 0; wg.t_end = lens[i]; wg.C = C; wg.dout = dout; wg.xin = xin;
    wg.dwt = vg->up_w[i]; wg.dbt = vg->up_b[i];
    launch_wgrad(wg, kUp * C, C, batch, nullptr, s);  // bias: atomics (10 rows share a word)
    UpDxOp dx;
    dx.K = kUp * C; dx.t_begin = 0; dx.t_end = lens[i]; dx.C = C; dx.wt = vp->up_w[i];
    dx.dout = dout; dx.dx = act_view(dacts[i], batch, C, mvn_padded_len(lens[i]));
    launch_gemm(dx, C, batch, s);
  }
  Act denc = act_view(d_enc, batch, C, mvn_padded_len(F));
  hipLaunchKernelGGL(video_conv_wgrad_kernel, dim3((cin * kPix + 255) / 256, (C + 7) / 8, batch),
                     dim3(256), 0, s, video, denc, vg->conv_w, C, cin, F);
  hipLaunchKernelGGL(rowsum_kernel, dim3(C), dim3(64), 0, s, denc, vg->conv_b, batch, F);
  return check_hip(hipGetLastError(), "mvn_upsample_video_backward");
}

}  // extern "C"
