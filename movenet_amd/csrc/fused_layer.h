// One gated residual layer of the full-sequence forward as ONE kernel (C = K = 64, audio only):
// reference arithmetic movenet/modules.py:67-93 -- f,g = dilated k=2 convs; z = tanh(f) sigmoid(g);
// residual 1x1 + bias + input; skip 1x1 + bias (kept for t >= RF-1 and summed over layers).
//
// The two-kernel form (gemm_wx_staged_kernel<FgOpT> then <RsOp>) writes z to HBM and reads it
// back, reads the layer input three times (both row blocks of the f/g product, then the
// residual add) and runs at 40 % / 28 % matrix-core utilisation because a 64-row block has
// little arithmetic per loaded element (profiles/r02_pmc_summary.json).  Here a workgroup owns
// 128 time columns and ALL rows:
//   phase A  f|g (128 rows) = W_fg (128 x 128) . [x(t-d); x(t)]     8 k-chunks of 16 through LDS
//            gate in registers (f and g of a channel sit in one lane: rows r and r + 16 of a
//            32-row tile), tanh / sigmoid stored for the backward pass, z -> LDS (32 KB)
//   phase B  res|skip (128 rows) = W_rs (128 x 64) . z               z never leaves the CU
//            epilogue: x' = (acc + br) + x, skip (+)= acc + bs
// 384 v_mfma_f32_32x32x2_f32 per wave against 8 + 8 global dwords per thread and chunk: the
// kernel is bound by the matrix cores (floor 78 us per layer at config 2), two workgroups per
// CU cover each other's staging and epilogues.  Same k order as the two-kernel form: the same
// bits.
#pragma once
#include "common.h"
#include "gemm_family.h"

namespace mvn {

struct FusedLayerArgs {
  int t_begin, t_end, d, t_skip0, t_base, first_layer;
  const float *wpack;              // this layer's packed weights (fused_pack_kernel): Wfg_t[128 k][128 m] | Wrs_t[64 k][128 m]
  const float *br, *bs;            // (64)
  Act xin, xout, th, sg, skip;     // xout.p == NULL: last layer; th/sg.p == NULL: nothing saved
};

constexpr int FL_C = 64, FL_T = 128, FL_KC = 16;
constexpr int FL_PACK_F = 128 * 128 + 64 * 128;  // floats per layer

// Weights of one layer in the order the layer kernel stages them: k-major, 128 rows contiguous
// (a wave's 64 lanes read 256 contiguous bytes; straight from the (C,C,2) / (C,C,1) tensors a
// wave-instruction touched 64 cache lines, and the weight staging cost as much as the MFMAs).
//   Wfg_t[k][m]: k < 64 tap 0 (x(t-d)) | k >= 64 tap 1 (x(t)); rows m in 32-row groups of 16
//   filter + 16 gate rows of channels 16 (m >> 5) ..;  Wrs_t[k][m]: m < 64 residual | skip
__global__ void fused_pack_kernel(const float *wf, const float *wg, const float *wr, const float *ws,
                                  float *__restrict__ dst) {
  constexpr int C = FL_C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 128 * 128) {
    const int k = i >> 7, m = i & 127;
    const int q = m & 31, c = 16 * (m >> 5) + (q & 15);
    const int tap = k >= C, kc = k - tap * C;
    dst[i] = (q >= 16 ? wg : wf)[((size_t)c * C + kc) * 2 + tap];
  } else if (i < FL_PACK_F) {
    const int j = i - 128 * 128, k = j >> 7, m = j & 127;
    dst[i] = m < C ? wr[(size_t)m * C + k] : ws[(size_t)(m - C) * C + k];
  }
}

// 16-byte global access at dword alignment (t_begin and the dilation shift are not multiples of 4)
typedef float fl_v4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f4 fl_ld4(const float *p) {
  const fl_v4 v = *(const fl_v4 *)p;
  return f4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void fl_st4(float *p, const f4 &v) { *(fl_v4 *)p = fl_v4{v.x, v.y, v.z, v.w}; }
// four columns t .. t+3 of a row, clamped at t_end (n = number of live columns, 1..4)
__device__ __forceinline__ f4 fl_ld4_edge(const float *p, int n) {
  f4 v = {p[0], 0.f, 0.f, 0.f};
  if (n > 1) v.y = p[1];
  if (n > 2) v.z = p[2];
  return v;
}
__device__ __forceinline__ void fl_st4_edge(float *p, const f4 &v, int n) {
  p[0] = v.x;
  if (n > 1) p[1] = v.y;
  if (n > 2) p[2] = v.z;
}

__global__ __launch_bounds__(256, 2) void fused_layer64_kernel(FusedLayerArgs a) {
  // one array: the 32 KB of the W / X chunk buffers double as the staging tile of the epilogues
  __shared__ __attribute__((aligned(16))) float lds[16384];
  float (*Ws)[FL_KC][128] = (float (*)[FL_KC][128])lds;            // [2][16][128] W chunks, [k][row]
  float (*Xs)[FL_KC][FL_T] = (float (*)[FL_KC][FL_T])(lds + 4096);  // [2][16][128] x chunks, [k][t]
  float (*Zs)[FL_T] = (float (*)[FL_T])(lds + 8192);                // [64][128] gated activation
  float (*St)[FL_T] = (float (*)[FL_T])lds;                         // [64][128] epilogue staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int t0 = a.t_begin + blockIdx.x * FL_T;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int C = FL_C;

  // staging: 16 rows x 128 columns per chunk = 512 float4: thread -> float4 column (tid & 31),
  // rows (tid >> 5) and (tid >> 5) + 8.  Every address is (a pointer fixed per thread) + (a
  // wave-uniform term).
  const int c4 = tid & 31, srow = tid >> 5;
  const float *wfg_p = a.wpack + (size_t)srow * 128 + 4 * c4;               // Wfg_t[k = srow][m = 4 c4 ..]
  const float *wrs_p = a.wpack + 128 * 128 + (size_t)srow * 128 + 4 * c4;   // Wrs_t[k = srow][m = 4 c4 ..]
  const int tq = t0 + 4 * c4;             // first of this thread's four columns
  const int nlive = a.t_end - tq;         // <= 0: none, >= 4: all
  // (clamped: a dead thread reads the tile's first columns, its values are zeroed afterwards)
  const float *x_p = a.xin.p + (size_t)b * a.xin.sb + (size_t)srow * a.xin.ld + (nlive > 0 ? tq : t0);
  // Two chunks of global loads stay in flight (register sets 0 / 1, statically named through
  // full unrolling): with one, every 16-deep chunk waited out a whole memory latency behind its
  // 2048 cycles of MFMAs (SQ_WAIT_ANY 51 % of the wave cycles, 41 % matrix-core utilisation).
  f4 wq00, wq01, wq10, wq11, xq00, xq01, xq10, xq11;  // [set][row half]; named: arrays went to LDS
  auto ld_x = [&](const float *q) -> f4 {
    f4 v = nlive >= 4 ? fl_ld4(q) : fl_ld4_edge(q, max(nlive, 1));
    if (nlive < 4) v = f4{nlive < 1 ? 0.f : v.x, nlive < 2 ? 0.f : v.y, nlive < 3 ? 0.f : v.z, 0.f};
    return v;
  };
  // chunk cc of the 12: 0..3 tap 0 on x(t - d), 4..7 tap 1 on x(t), 8..11 the residual/skip matrix
#define FL_GLOAD(cc, S)                                                                     \
  do {                                                                                      \
    if ((cc) < 8) {                                                                         \
      const float *xr = x_p + (size_t)(((cc) & 3) * FL_KC) * a.xin.ld - (((cc) >> 2) ? 0 : a.d); \
      wq##S##0 = *(const f4 *)&wfg_p[(size_t)((cc) * FL_KC) * 128];                         \
      wq##S##1 = *(const f4 *)&wfg_p[(size_t)((cc) * FL_KC + 8) * 128];                     \
      xq##S##0 = ld_x(xr);                                                                  \
      xq##S##1 = ld_x(xr + (size_t)8 * a.xin.ld);                                           \
    } else if ((cc) < 12) {                                                                 \
      wq##S##0 = *(const f4 *)&wrs_p[(size_t)(((cc) - 8) * FL_KC) * 128];                   \
      wq##S##1 = *(const f4 *)&wrs_p[(size_t)(((cc) - 8) * FL_KC + 8) * 128];               \
    }                                                                                       \
  } while (0)
#define FL_LSTORE(cc, S, B)                                                                 \
  do {                                                                                      \
    *(f4 *)&Ws[B][srow][4 * c4] = wq##S##0;                                                 \
    *(f4 *)&Ws[B][srow + 8][4 * c4] = wq##S##1;                                             \
    if ((cc) < 8) {                                                                         \
      *(f4 *)&Xs[B][srow][4 * c4] = xq##S##0;                                               \
      *(f4 *)&Xs[B][srow + 8][4 * c4] = xq##S##1;                                           \
    }                                                                                       \
  } while (0)
  // the 8 x 4 MFMAs of one chunk: A operand rows from Ws[B], B operand from `bsrc` (rows kr)
#define FL_MFMA(B, bsrc)                                                                    \
  _Pragma("unroll") for (int kk = 0; kk < FL_KC / 2; ++kk) {                                \
    const int kr = 2 * kk + lh;                                                             \
    const float bv = (bsrc)[kr][tcol];                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[B][kr][32 * i + li], bv, acc[i], 0, 0, 0); \
  }

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int tcol = 32 * wave + li;  // this lane's column inside the tile
  FL_GLOAD(0, 0);
  FL_GLOAD(1, 1);
  FL_LSTORE(0, 0, 0);
  __syncthreads();
  // ---- phase A: f | g = W_fg . [x(t-d); x(t)].  Chunk cc sits in LDS buffer cc & 1; its
  // successor cc + 1 waits in register set (cc + 1) & 1 and cc + 2 is requested now.
  for (int cc = 0; cc < 8; cc += 2) {
    FL_MFMA(0, Xs[0]);
    FL_LSTORE(cc + 1, 1, 1);
    FL_GLOAD(cc + 2, 0);
    __syncthreads();
    FL_MFMA(1, Xs[1]);
    if (cc + 2 < 8) FL_LSTORE(cc + 2, 0, 0);  // (chunk 8 is stored after the gate)
    FL_GLOAD(cc + 3, 1);
    __syncthreads();
  }
  // ---- gate in registers; z -> LDS (operand of phase B); tanh and sigmoid leave through the
  // staging tile as whole-row float4 stores (a lane holds ONE column of 8 channels: stored
  // from the registers that is 64 dword stores per wave, and the kernel was bound by them)
  float sgv[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = 16 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float tv = tanh_fast(acc[i][r]), sv = sigmoid_fast(acc[i][r + 8]);
      Zs[ch][tcol] = tv * sv;  // masked columns: x = 0 -> f = g = 0 -> z = 0
      St[ch][tcol] = tv;
      sgv[i][r] = sv;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // row pass: thread -> float4 column c4, rows srow + 8 p
  auto store_rows = [&](const Act &dst) {
    if (nlive <= 0) return;
    float *base = dst.p + (size_t)b * dst.sb + tq;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = srow + 8 * p;
      const f4 v = *(const f4 *)&St[row][4 * c4];
      float *q = base + (size_t)row * dst.ld;
      if (nlive >= 4) fl_st4(q, v); else fl_st4_edge(q, v, nlive);
    }
  };
  __syncthreads();
  if (a.th.p) {
    store_rows(a.th);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 8; ++r) St[16 * i + (r & 3) + 8 * (r >> 2) + 4 * lh][tcol] = sgv[i][r];
    __syncthreads();
    store_rows(a.sg);
    __syncthreads();
  }
  // ---- phase B: res | skip = W_rs . z   (z stays in LDS; chunk 8 waits in register set 0,
  // chunk 9 in set 1)
  FL_LSTORE(8, 0, 0);
  FL_GLOAD(10, 0);
  __syncthreads();
  FL_MFMA(0, Zs);            // chunk 8
  FL_LSTORE(9, 1, 1);
  FL_GLOAD(11, 1);
  __syncthreads();
  FL_MFMA(1, Zs + FL_KC);     // chunk 9
  FL_LSTORE(10, 0, 0);
  __syncthreads();
  FL_MFMA(0, Zs + 2 * FL_KC); // chunk 10
  FL_LSTORE(11, 1, 1);
  __syncthreads();
  FL_MFMA(1, Zs + 3 * FL_KC); // chunk 11
  __syncthreads();
#undef FL_GLOAD
#undef FL_LSTORE
#undef FL_MFMA
  // ---- epilogue: residual rows (tiles 0, 1) through the staging tile: x' = (acc + br) + x
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) St[32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh][tcol] = acc[i][r];
  __syncthreads();
  if (a.xout.p && nlive > 0) {
    const float *xi = a.xin.p + (size_t)b * a.xin.sb + tq;
    float *xo = a.xout.p + (size_t)b * a.xout.sb + tq;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = srow + 8 * p;
      const f4 v = *(const f4 *)&St[row][4 * c4];
      const float bias = a.br[row];
      const float *qi = xi + (size_t)row * a.xin.ld;
      float *qo = xo + (size_t)row * a.xout.ld;
      const f4 x = nlive >= 4 ? fl_ld4(qi) : fl_ld4_edge(qi, nlive);
      const f4 o = f4{(v.x + bias) + x.x, (v.y + bias) + x.y, (v.z + bias) + x.z, (v.w + bias) + x.w};
      if (nlive >= 4) fl_st4(qo, o); else fl_st4_edge(qo, o, nlive);
    }
  }
  __syncthreads();
  // skip rows (tiles 2, 3): skip (+)= acc + bs for t >= t_skip0
#pragma unroll
  for (int i = 2; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) St[32 * (i - 2) + (r & 3) + 8 * (r >> 2) + 4 * lh][tcol] = acc[i][r];
  __syncthreads();
  {
    // live columns of this thread: [max(tq, t_skip0), min(tq + 4, t_end))
    const int lo = max(a.t_skip0 - tq, 0), hi = min(nlive, 4);
    if (hi > lo) {
      float *sk = a.skip.p + (size_t)b * a.skip.sb + (tq - a.t_base);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int row = srow + 8 * p;
        const f4 v = *(const f4 *)&St[row][4 * c4];
        const float bias = a.bs[row];
        float *q = sk + (size_t)row * a.skip.ld;
        if (lo == 0 && hi == 4) {
          f4 o = f4{v.x + bias, v.y + bias, v.z + bias, v.w + bias};
          if (!a.first_layer) {
            const f4 old = fl_ld4(q);
            o = f4{old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w};
          }
          fl_st4(q, o);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e >= lo && e < hi) {
              const float add = f4_get(v, e) + bias;
              q[e] = a.first_layer ? add : q[e] + add;
            }
        }
      }
    }
  }
}

static void launch_fused_layer64(const FusedLayerArgs &a, int batch, hipStream_t s) {
  const int nt = a.t_end - a.t_begin;
  if (nt <= 0 || batch <= 0) return;
  hipLaunchKernelGGL(fused_layer64_kernel, dim3((nt + FL_T - 1) / FL_T, batch), dim3(256), 0, s, a);
}

}  // namespace mvn
