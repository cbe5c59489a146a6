// One gated residual layer of the full-sequence forward, PERSISTENT form (C = K = 64, audio only, fp32):
// reference arithmetic movenet/modules.py:67-93, as fused_layer.h -- f,g = dilated k=2 convs;
// z = tanh(f) sigmoid(g); x' = x + Wr z + br; skip (+)= Ws z + bs for t >= RF - 1.
//
// fused_layer64_kernel (fused_layer.h) streams its 48 KB of weights from L2 through LDS for
// every 128-column tile and passes 21 barriers per tile: 43 % matrix-core utilisation, 174 us per
// layer at config 2.  This kernel is built like the backward halves (fused_bwd.h): a 512-thread
// workgroup per CU walks 64-column tiles of a 512-step chunk, and BOTH products run in the
// transposed form with their weights in REGISTERS for the whole launch:
//   F'[t][m] = sum_k X[k][t] W[k][m]   X = [x(t-d); x(t)] staged once per tile (128 x 64, pitch 68)
//     wave -> (32 t x 32 channels, tap): filter AND gate block of its channels over the 64 rows
//     of its tap (2 x 32 weights per lane); the two taps' partial sums meet through LDS, each
//     wave finishing half of the block: gate in registers, z / tanh / sigmoid to LDS tiles
//   Y'[t][m2] = sum_c Z[c][t] Wrs[c][m2]   wave -> one 32 x 32 block of [residual | skip] (32 weights)
//   epilogue through the staging tiles as whole-row float4 accesses: tanh, sigmoid, x' = (y + br) + x
//   (x from the staged tile), skip (+)= y + bs.
// The next tile's X waits in registers, then in the second X buffer; five barriers per tile.
// Summation order differs from fused_layer64_kernel (two 64-deep partial sums instead of one
// 128-deep chain): same values to fp32 rounding, not the same bits.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_family.h"
#include "fused_bwd.h"

namespace mvn {

struct FusedFwdPArgs {
  int t_begin, t_end, d, t_skip0, t_base, first_layer;
  const float *wf, *wg;          // (64 out, 64 in, 2 taps)
  const float *wr, *ws;          // (64 out, 64 in)
  const float *br, *bs;          // (64)
  Act xin, xout, th, sg, skip;   // xout.p == NULL: last layer; th/sg.p == NULL: nothing saved
  // conditioned layers (strip kernel only): f,g += Wc ctx(t) + bc   (modules.py:58-63, :75-77)
  const float *wcf = nullptr, *wcg = nullptr, *bcf = nullptr, *bcg = nullptr;  // (64, 64), (64)
  Act ctx = Act{nullptr, 0, 0};
  const float *wpack = nullptr;  // fused_fwd_bf3.h: the layer's LDS image, written once per forward call (or NULL)
};

// NTB = 32-step blocks per tile: 2 -> 64-column tiles, 512 threads, one workgroup per CU (122 KB of
// LDS); 1 -> 32-column tiles, 256 threads, TWO workgroups per CU (65 KB each) whose phases run
// under each other's MFMAs -- the same registers per wave either way (a wave's blocks are 32 wide).
template <int NTB>
__global__ __launch_bounds__(256 * NTB, NTB == 1 ? 2 : 1) void fused_layer64p_kernel(FusedFwdPArgs a, int chunks_per_b,
                                                                               int chunk_t) {
  constexpr int C = 64, TT = 32 * NTB, LD = TT + 4, NTH = 256 * NTB;
  extern __shared__ __attribute__((aligned(16))) float fp_lds[];
  float (*X)[128][LD] = (float (*)[128][LD])fp_lds;              // [2]: x(t - d) rows | x(t) rows
  float (*Z)[LD] = (float (*)[LD])(fp_lds + 2 * 128 * LD);       // gated activation
  float (*S1)[LD] = (float (*)[LD])(fp_lds + (2 * 128 + 64) * LD);   // f partials -> tanh -> x'
  float (*S2)[LD] = (float (*)[LD])(fp_lds + (2 * 128 + 128) * LD);  // g partials -> sigmoid -> skip
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5, h4 = 4 * lh;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_begin, a.t_skip0);

  // ---- the weights reach their registers through LDS: a lane's values lie 512 (256) bytes apart
  // in the (out, in, tap) / (out, in) tensors -- fetched straight from global memory every wave
  // instruction touched 64 cache lines, 49 000 line requests per workgroup: longer than its tiles
  // (three passes of 32 KB through the tile buffers: filter, gate, residual | skip)
  const int tt = (wave >> 1) % NTB, cc = wave & 1, kh = wave / (2 * NTB);   // first product: block (tt, cc), tap kh
  const int t2 = wave >> 2, mt = wave & 3;                                  // second product: block (t2, mt)
  float wfr[32], wgr[32], wrs[32];
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const float *src = pass == 0 ? a.wf : pass == 1 ? a.wg : a.wr;
    for (int i = tid; i < (pass < 2 ? 2048 : 1024); i += NTH) *(f4 *)&fp_lds[4 * i] = *(const f4 *)&src[4 * i];
    if (pass == 2)
      for (int i = tid; i < 1024; i += NTH) *(f4 *)&fp_lds[4096 + 4 * i] = *(const f4 *)&a.ws[4 * i];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      if (pass < 2) {
        const float v = fp_lds[((32 * cc + li) * C + 2 * kk + lh) * 2 + kh];
        if (pass == 0) wfr[kk] = v; else wgr[kk] = v;
      } else {
        wrs[kk] = fp_lds[(mt < 2 ? 0 : 4096) + (32 * (mt & 1) + li) * C + 2 * kk + lh];
      }
    }
    __syncthreads();
  }
  const float bias2 = (mt < 2 ? a.br : a.bs)[32 * (mt & 1) + li];

  // ---- staging: thread -> rows (tid / TPR) + 32 p, columns 4 (tid % TPR) .. +3
  constexpr int TPR = TT / 4;  // threads per row
  const int srow = tid / TPR, st = 4 * (tid % TPR);
  f4 xreg[4], kreg[2];
  auto interior = [&](int t0) { return t0 >= a.t_begin && t0 + TT <= te; };
  auto gload_x = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
    if (interior(t0)) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        xreg[p] = ldg4(a.xin.at(b, (32 * p + srow_q) & (C - 1), 0) + t - (p < 2 ? a.d : 0));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        xreg[p] = ld4_edge(a.xin.at(b, (32 * p + srow_q) & (C - 1), 0) - (p < 2 ? a.d : 0), t, a.t_begin, te);
    }
  };
  // the skip accumulator's old values for this tile (columns t - t_base, live from skip_lo)
  auto gload_skip = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
    if (a.first_layer || t0 + TT <= skip_lo) {
      kreg[0] = kreg[1] = kZero4;
    } else if (t0 >= skip_lo && t0 + TT <= te) {
#pragma unroll
      for (int p = 0; p < 2; ++p) kreg[p] = ldg4(a.skip.at(b, 32 * p + srow_q, 0) + (t - a.t_base));
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) kreg[p] = ld4_edge(a.skip.at(b, 32 * p + srow_q, 0) - a.t_base, t, skip_lo, te);
    }
  };
  auto lstore_x = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *(f4 *)&X[buf][32 * p + srow][st] = xreg[p];
  };
  // rows srow + 32 p (p < 2) of a 64-row staging tile -> dst, columns inside [lo, te)
  auto store_rows = [&](const Act &dst, float (*S)[LD], int t0, int lo, int col_shift) {
    const int t = t0 + st;
    float *base = dst.p + (size_t)b * dst.sb + (t - col_shift);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = 32 * p + srow;
      const f4 v = *(const f4 *)&S[row][st];
      float *q = base + (size_t)row * dst.ld;
      if (t >= lo && t + 3 < te) {
        *(f4 *)q = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e >= lo && t + e < te) q[e] = f4_get(v, e);
      }
    }
  };

  gload_x(tb);
  lstore_x(0);
  __syncthreads();
  int cur = 0;
  for (int t0 = tb; t0 < te; t0 += TT, cur ^= 1) {
    const bool more = t0 + TT < te;
    gload_skip(t0);
    if (more) gload_x(t0 + TT);
    __builtin_amdgcn_sched_barrier(0);
    // ---- f | g partial sums of this wave's tap: 2 x 32 MFMAs, LDS operands one step ahead
    f32x16 accf, accg;
#pragma unroll
    for (int r = 0; r < 16; ++r) accf[r] = accg[r] = 0.f;
    {
      float av[2][8];
      auto fetch = [&](int g, int S) {
#pragma unroll
        for (int i = 0; i < 8; ++i) av[S][i] = X[cur][64 * kh + 2 * (8 * g + i) + lh][32 * tt + li];
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int S = g & 1;
        if (g + 1 < 4) fetch(g + 1, S ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          accf = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][i], wfr[8 * g + i], accf, 0, 0, 0);
          accg = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][i], wgr[8 * g + i], accg, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- the taps meet: this wave finishes t-groups q = 2 kh, 2 kh + 1 of its block and hands the
    // other two to its partner (the same block, other tap) through the staging tiles
    // (the t-group index is a compile-time constant on either side of the branch: indexed at run
    // time the accumulator registers were moved through s_set_gpr_idx, 24 switches per tile)
    auto hand_over = [&](auto QB) {
      constexpr int q0 = decltype(QB)::value;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq, tc = 32 * tt + 8 * q + h4;
        *(f4 *)&S1[32 * cc + li][tc] = f4{accf[4 * q], accf[4 * q + 1], accf[4 * q + 2], accf[4 * q + 3]};
        *(f4 *)&S2[32 * cc + li][tc] = f4{accg[4 * q], accg[4 * q + 1], accg[4 * q + 2], accg[4 * q + 3]};
      }
    };
    if (kh == 0) hand_over(std::integral_constant<int, 2>()); else hand_over(std::integral_constant<int, 0>());
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads above have had the MFMAs to land (cf. fused_bwd.h)
    __syncthreads();
    auto finish = [&](auto QB) {
      constexpr int q0 = decltype(QB)::value;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq, tc = 32 * tt + 8 * q + h4;
        const f4 pf = *(const f4 *)&S1[32 * cc + li][tc], pg = *(const f4 *)&S2[32 * cc + li][tc];
        f4 tv, sv;
        tv.x = tanh_fast(accf[4 * q] + pf.x);     sv.x = sigmoid_fast(accg[4 * q] + pg.x);
        tv.y = tanh_fast(accf[4 * q + 1] + pf.y); sv.y = sigmoid_fast(accg[4 * q + 1] + pg.y);
        tv.z = tanh_fast(accf[4 * q + 2] + pf.z); sv.z = sigmoid_fast(accg[4 * q + 2] + pg.z);
        tv.w = tanh_fast(accf[4 * q + 3] + pf.w); sv.w = sigmoid_fast(accg[4 * q + 3] + pg.w);
        *(f4 *)&Z[32 * cc + li][tc] = f4{tv.x * sv.x, tv.y * sv.y, tv.z * sv.z, tv.w * sv.w};
        *(f4 *)&S1[32 * cc + li][tc] = tv;  // (the slots this lane has just read)
        *(f4 *)&S2[32 * cc + li][tc] = sv;
      }
    };
    if (kh == 0) finish(std::integral_constant<int, 0>()); else finish(std::integral_constant<int, 2>());
    if (more) lstore_x(cur ^ 1);
    __syncthreads();
    // ---- tanh / sigmoid leave for the backward pass; residual | skip block of this wave
    // (their four row stores per thread are issued between the MFMA steps below)
    auto store_row = [&](const Act &dst, float (*S)[LD], int p) {
      const int t = t0 + st, row = 32 * p + srow;
      const f4 v = *(const f4 *)&S[row][st];
      float *q = dst.p + (size_t)b * dst.sb + (size_t)row * dst.ld + t;
      if (t >= a.t_begin && t + 3 < te) {
        *(f4 *)q = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e >= a.t_begin && t + e < te) q[e] = f4_get(v, e);
      }
    };
    f32x16 accy;
#pragma unroll
    for (int r = 0; r < 16; ++r) accy[r] = 0.f;
    {
      float zv[2][8];
      auto fetch = [&](int g, int S) {
#pragma unroll
        for (int i = 0; i < 8; ++i) zv[S][i] = Z[2 * (8 * g + i) + lh][32 * t2 + li];
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int S = g & 1;
        if (g + 1 < 4) fetch(g + 1, S ^ 1);
        if (a.th.p) store_row(g < 2 ? a.th : a.sg, g < 2 ? S1 : S2, g & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) accy = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[S][i], wrs[8 * g + i], accy, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();  // tanh / sigmoid have been read: the staging tiles take x' and the skip term
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tc = 32 * t2 + 8 * q + h4, row = 32 * (mt & 1) + li;
      f4 v = f4{accy[4 * q] + bias2, accy[4 * q + 1] + bias2, accy[4 * q + 2] + bias2, accy[4 * q + 3] + bias2};
      if (mt < 2) {
        const f4 x = *(const f4 *)&X[cur][C + row][tc];
        *(f4 *)&S1[row][tc] = f4{v.x + x.x, v.y + x.y, v.z + x.z, v.w + x.w};
      } else {
        *(f4 *)&S2[row][tc] = v;
      }
    }
    __syncthreads();
    if (a.xout.p) store_rows(a.xout, S1, t0, a.t_begin, 0);
    {
      // skip (+)= term, columns t - t_base, live from skip_lo
      const int t = t0 + st;
      float *base = a.skip.p + (size_t)b * a.skip.sb + (t - a.t_base);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = 32 * p + srow;
        const f4 v = *(const f4 *)&S2[row][st], o = kreg[p];
        const f4 r = a.first_layer ? v : f4{o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w};
        float *q = base + (size_t)row * a.skip.ld;
        if (t >= skip_lo && t + 3 < te) {
          *(f4 *)q = r;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= skip_lo && t + e < te) q[e] = f4_get(r, e);
        }
      }
    }
    __syncthreads();  // the staging tiles are free for the next tile's partial sums
  }
}

// ----------------------------------------------------------------------------------------
// STRIP form: one WAVE owns a strip of 32 time columns and runs the whole layer on it without a
// single barrier (r2b timing builds: the tile kernels above spend 45 % of their time with waves
// waiting at barriers for a sibling whose SIMD is held by another wave's MFMA burst).
//   * both weight matrices sit in LDS for the whole launch (96 KB, read-only), laid out per MFMA
//     block and lane so that one ds_read_b128 feeds four MFMAs (A operand: lane -> output row);
//   * the strip's input, [x(t-d); x(t)] for 32 columns, goes from global memory straight into 64
//     registers per lane as the B operand (lane -> column); the k order pairs channel c (lane
//     half 0) with c + 4 (lane half 1), c running over the channels with bit 2 clear -- which
//     are exactly the channels a lane of that half holds in the ACCUMULATOR layout (rows
//     (r & 3) + 8 (r >> 2) + 4 lh), so that
//       - f and g of a channel meet in one lane: gate in registers, no exchange;
//       - z, computed in accumulator order, IS the B operand of the second product;
//       - the residual add finds x(t) of its channel in the lane's own input registers;
//   * tanh, sigmoid, x' and the skip term leave as dword stores, 128 contiguous bytes per channel
//     and lane half (a whole cache line); the skip accumulator's old values are fetched under
//     the second product into the dead x(t-d) registers.
// Per strip and wave: 384 MFMAs, 96 ds_read_b128, ~290 vector-memory instructions.  The two
// waves of a SIMD are independent: one's loads and stores run under the other's MFMAs.
// ----------------------------------------------------------------------------------------
// HAS_CTX: the conditioned layer -- the context is a third K block of the first product (96 KB of
// f|g weights), its 32 registers are the ones the unconditioned kernel uses to fetch x(t) a strip ahead.
// Strips start at multiples of 32 columns of the absolute time axis (rows are 256-byte aligned: every
// 128-byte row segment a store instruction writes is ONE cache line, not two halves: 148.6 -> 142.4 us
// per layer), and tanh / sigmoid -- written once, read by the backward pass much later -- leave with
// the non-temporal hint (-> 135.8 us; on x' and the skip sums, which the next layer reads, it costs 5 us).
constexpr int FS_AUX_SAVE = 2;  // nt
template <bool HAS_CTX>
__global__ __launch_bounds__(512, 1) void fused_layer64s_kernel(FusedFwdPArgs a, int chunks_per_b, int chunk_t) {
  constexpr int C = 64, NK1 = HAS_CTX ? 24 : 16, W1_F = 4 * NK1 * 256;
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  float *W1 = fs_lds, *W2 = fs_lds + W1_F, *BI = fs_lds + W1_F + 8192;  // BI: br | bs | bcf | bcg
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_begin, a.t_skip0);
  // ---- weights into LDS: [block][k-step / 4][lane][k-step % 4].  The loop runs over the SOURCE
  // elements (coalesced reads of the (out, in, tap) / (out, in) tensors) and scatters into LDS.
  for (int sI = tid; sI < 2 * 8192; sI += 512) {
    const int g = sI >> 13, r = sI & 8191;        // g: 0 filter, 1 gate
    const int tap = r & 1, kc = (r >> 1) & 63, cm = r >> 7;
    const int lhs = (kc >> 2) & 1, j = (kc & 3) + 4 * (kc >> 3), kk = j + 32 * tap;
    const int blk = 2 * g + (cm >> 5), ln = (cm & 31) + 32 * lhs;
    W1[((blk * NK1 + (kk >> 2)) * 64 + ln) * 4 + (kk & 3)] = (g ? a.wg : a.wf)[r];
  }
  if (HAS_CTX) {
    for (int sI = tid; sI < 2 * 4096; sI += 512) {
      const int g = sI >> 12, r = sI & 4095;      // g: 0 filter, 1 gate context conv
      const int kc = r & 63, cm = r >> 6;
      const int lhs = (kc >> 2) & 1, kk = 64 + (kc & 3) + 4 * (kc >> 3);
      const int blk = 2 * g + (cm >> 5), ln = (cm & 31) + 32 * lhs;
      W1[((blk * NK1 + (kk >> 2)) * 64 + ln) * 4 + (kk & 3)] = (g ? a.wcg : a.wcf)[r];
    }
    if (tid < 128) BI[128 + tid] = tid < 64 ? a.bcf[tid] : a.bcg[tid - 64];
  }
  for (int sI = tid; sI < 2 * 4096; sI += 512) {
    const int g = sI >> 12, r = sI & 4095;        // g: 0 residual, 1 skip
    const int kc = r & 63, m2 = r >> 6;
    const int lhs = (kc >> 2) & 1, kk = (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (m2 >> 5), ln = (m2 & 31) + 32 * lhs;
    W2[((blk * 8 + (kk >> 2)) * 64 + ln) * 4 + (kk & 3)] = (g ? a.ws : a.wr)[r];
  }
  if (tid < 128) BI[tid] = tid < 64 ? a.br[tid] : a.bs[tid - 64];
  __syncthreads();
  const unsigned w1a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)W1 + 16u * lane;
  const unsigned w2a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)W2 + 16u * lane;
  typedef float fsv4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) fsv4 lds_v4;
  // channel of accumulator register r of block h in this lane: 32 h + (r & 3) + 8 (r >> 2) + 4 lh
  const int cbase = 4 * lh;

  // Every global access below is a raw BUFFER access: (a resource per tensor and sequence:
  // four scalar registers) + (row * ld: one scalar offset) + (one of five per-lane byte offsets).
  // Formed as 64-bit vector addresses the ~290 accesses of a strip spilled 443 registers; as
  // scalar row pointers + vector offsets they still cost a 64-bit vector add each and spilled the
  // scalar file (130 v_writelane / v_readlane per strip).
  constexpr int RSRC = 0x00020000;  // raw buffer, 32-bit data format (gfx9)
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xin.p + (size_t)b * a.xin.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t thb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.th.p + (size_t)b * a.th.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t sgb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.sg.p + (size_t)b * a.sg.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t xob = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xout.p + (size_t)b * a.xout.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t skb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.skip.p + (size_t)b * a.skip.sb - a.t_base), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t cb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.ctx.p + (size_t)b * a.ctx.sb), 0, 0x7FFFFFFF, RSRC);
  constexpr bool st_ok = true;
  const bool save = st_ok && a.th.p != nullptr, has_out = st_ok && a.xout.p != nullptr;
  int xld4 = 4 * a.xin.ld, thld4 = 4 * a.th.ld, xold4 = 4 * a.xout.ld, skld4 = 4 * a.skip.ld, cld4 = 4 * a.ctx.ld;
  // (row offsets = row * ld are re-formed where they are used, behind a fence on ld: hoisted out of
  // the strip loop the ~120 products filled the scalar file and were spilled to vector lanes)
#define FS_FENCE(x) asm volatile("" : "+s"(x))

  // x(t) of a strip (the tap-1 half of the input, 32 registers) is fetched one strip AHEAD, under
  // the second product of the strip before; the first product consumes it first, which gives the
  // x(t - d) half, requested at the top of the strip, 128 MFMAs to arrive.  (Without it the wave
  // counters showed 51 % of the wave cycles waiting for memory: a wave that issues MFMAs back to
  // back starves the other wave of its SIMD, so the two fall into step and wait together.)
  auto column = [&](int t0_, bool &live_, int &tc_) {
    const int t_ = t0_ + li;
    live_ = t_ >= a.t_begin && t_ < te;
    tc_ = live_ ? t_ : a.t_begin;  // (clamped: dead lanes read a valid column, zeroed afterwards)
  };
  float xb1[32];  // x(t) of the current strip: B operand of k-steps 32..63, residual input
  if (!HAS_CTX) {
    bool lv;
    int tcc;
    column(tb + 32 * wave, lv, tcc);
    const int o1 = 4 * (cbase * a.xin.ld + tcc);
    FS_FENCE(xld4);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      xb1[j] = lv ? v : 0.f;
    }
  }

  // Lanes that must not store (columns outside [t_begin, te), no output tensor) carry an offset
  // beyond the buffer's num_records: the hardware drops the access -- no exec-mask branch per store.
  // (Tried: holding a strip's x' and skip sums in registers until after the next strip's first
  // product, so that no wait on a prefetched load sits right behind 64 fresh stores -- on gfx9 loads
  // and stores retire through one in-order counter and the compiler's wait at the loop edge was
  // vmcnt(0).  Measured 131.4 against 131.9 us per layer: the second wave of the SIMD covers it.)
  constexpr int FS_OOB = (int)0x80000000;
  for (int t0 = tb + 32 * wave; t0 < te; t0 += 32 * 8) {
    const int t = t0 + li;
    bool live;
    int tc;
    column(t0, live, tc);
    const bool skip_live = st_ok && t >= skip_lo && t < te;
    // per-lane byte offsets (channel part 4 lh of the row + the column)
    const int ox0 = 4 * (cbase * a.xin.ld + tc - a.d);
    const int oth = (save && live) ? 4 * (cbase * a.th.ld + tc) : FS_OOB, oxo = 4 * (cbase * a.xout.ld + tc);
    const int osk = 4 * (cbase * a.skip.ld + (skip_live ? t : skip_lo));
    float xn1[32];  // without context: x(t) of the NEXT strip; with: ctx(t) of this one (k-steps 64..95)
    if (HAS_CTX) {
      const int o1 = 4 * (cbase * a.xin.ld + tc), oc = 4 * (cbase * a.ctx.ld + tc);
      FS_FENCE(xld4);
      FS_FENCE(cld4);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
        xb1[j] = live ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cb, oc, ((j & 3) + 8 * (j >> 2)) * cld4, 0));
        xn1[j] = live ? v : 0.f;
      }
    }
    // ---- x(t - d) of channel kc(j) + 4 lh: B operand of k-steps 0..31; later the skip accumulator's old values
    float xa0[32];
    FS_FENCE(xld4);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, ox0, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      xa0[j] = v;
    }
    // (interior strips -- wave-uniform test -- need no masking: 32 selects less per strip)
    if (!(t0 >= a.t_begin && t0 + 32 <= te)) {
#pragma unroll
      for (int j = 0; j < 32; ++j) xa0[j] = live ? xa0[j] : 0.f;
    }
    // ---- f | g: four 32 x 32 blocks (f c<32, f c>=32, g c<32, g c>=32), K = 128, the x(t) half first
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int kq = 0; kq < NK1; ++kq) {
      const int k4 = kq < 8 ? 8 + kq : kq < 16 ? kq - 8 : kq;  // x(t), x(t - d), context
      fsv4 aw[4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) aw[blk] = *(const lds_v4 *)(uintptr_t)(w1a + 4u * (unsigned)((blk * NK1 + k4) * 256));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              aw[blk][e], k4 >= 16 ? xn1[4 * (k4 - 16) + e] : k4 >= 8 ? xb1[4 * (k4 - 8) + e] : xa0[4 * k4 + e], acc[blk], 0, 0, 0);
    }
    // ---- gate in registers; tanh / sigmoid leave; z in accumulator order = the next B operand
    float z[32];
    FS_FENCE(thld4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cg_ = 32 * h + (r & 3) + 8 * (r >> 2) + cbase;
        const float tv = tanh_fast(HAS_CTX ? acc[h][r] + BI[128 + cg_] : acc[h][r]);
        const float sv = sigmoid_fast(HAS_CTX ? acc[2 + h][r] + BI[192 + cg_] : acc[2 + h][r]);
        z[16 * h + r] = tv * sv;
        const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tv), thb, oth, c0 * thld4, FS_AUX_SAVE);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv), sgb, oth, c0 * thld4, FS_AUX_SAVE);
      }
    // the skip accumulator's old values, into the x(t - d) registers (dead now), and the NEXT strip's
    // x(t), both under the MFMAs below
    FS_FENCE(skld4);
    if (!a.first_layer) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xa0[16 * h + r] = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(skb, osk, (32 * h + (r & 3) + 8 * (r >> 2)) * skld4, 0));
    }
    const bool more = !HAS_CTX && t0 + 32 * 8 < te;
    if (HAS_CTX) {
    } else if (more) {
      bool lv;
      int tcc;
      column(t0 + 32 * 8, lv, tcc);
      const int o1 = 4 * (cbase * a.xin.ld + tcc);
      FS_FENCE(xld4);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
        xn1[j] = v;
      }
      if (!(t0 + 32 * 8 >= a.t_begin && t0 + 32 * 8 + 32 <= te)) {
#pragma unroll
        for (int j = 0; j < 32; ++j) xn1[j] = lv ? xn1[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) xn1[j] = 0.f;
    }
    // ---- residual | skip: four blocks (res c<32, res c>=32, skip k<32, skip k>=32), K = 64
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      fsv4 aw[4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) aw[blk] = *(const lds_v4 *)(uintptr_t)(w2a + 4u * (unsigned)((blk * 8 + k4) * 256));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[blk][e], z[4 * k4 + e], acc[blk], 0, 0, 0);
    }
    // ---- x' = (y + br) + x(t): x(t) of this lane's channel is input register 32 + 16 h + r;
    // skip (+)= y + bs, columns t - t_base, live from skip_lo
    FS_FENCE(xold4);
    FS_FENCE(skld4);
    {
      const int oxo_m = (has_out && live) ? oxo : FS_OOB, osk_m = skip_live ? osk : FS_OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((acc[h][r] + BI[c0 + cbase]) + xb1[16 * h + r]), xob, oxo_m,
                                                c0 * xold4, 0);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = 32 * h + (r & 3) + 8 * (r >> 2);
          const float v = acc[2 + h][r] + BI[64 + k0 + cbase];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a.first_layer ? v : xa0[16 * h + r] + v), skb, osk_m, k0 * skld4, 0);
        }
    }
    if (!HAS_CTX) {
#pragma unroll
      for (int j = 0; j < 32; ++j) xb1[j] = xn1[j];
    }
  }
}

static int launch_fused_layer64s(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch, 1, &chunks, &chunk_t, 256);  // a chunk: whole rounds of the 8 waves' strips
  const bool has_ctx = a.ctx.p != nullptr;
  const void *fn = has_ctx ? (const void *)fused_layer64s_kernel<true> : (const void *)fused_layer64s_kernel<false>;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(fused_layer64s)");
  if (rc) return rc;
  const size_t lds = ((has_ctx ? 24576 : 16384) + 8192 + 256) * sizeof(float);
  if (has_ctx)
    hipLaunchKernelGGL(fused_layer64s_kernel<true>, dim3(chunks * batch), dim3(512), lds, s, a, chunks, chunk_t);
  else
    hipLaunchKernelGGL(fused_layer64s_kernel<false>, dim3(chunks * batch), dim3(512), lds, s, a, chunks, chunk_t);
  return MVN_OK;
}

// ----------------------------------------------------------------------------------------
// The strip form for the head's 1x1 convs with 64 x 256 weights (conv1 forward: 64 -> 256 rows,
// leaky-ReLU in and out; its data gradient: 256 -> 64 rows through W^T, times the leaky-ReLU
// derivative): the generic kernel reads a 64-row input once per 64-row OUTPUT block (conv1: four
// times) and ran at 1.5 TB/s.  Same recipe as fused_layer64s_kernel: weights in LDS (64 KB), the
// strip's input in registers as the B operand, raw buffer accesses, no barrier.
//   IN: 0 identity, 1 leaky-ReLU on load;  OUT: 1 leaky(y + bias), 2 y * leaky'(ref)
// ----------------------------------------------------------------------------------------
struct DenseStripArgs {
  int t_begin, t_end, t_out_end;
  const float *wmat;  // W[m][k] = TRANSPOSED ? wmat[k * ldw + m] : wmat[m * ldw + k]
  int ldw;
  const float *bias;
  Act xin, yout, ref;
};

// M = output rows per workgroup (blockIdx.y selects the row block): 64 x 256 weights for conv1
// (one block of 256 rows), 256 x 128 for conv2 and its data gradient (two blocks: the input is
// read twice instead of four times, 128 of its 256 rows' weights = 128 KB in LDS).
template <int K, int M, int IN, int OUT, bool TRANSPOSED>
__global__ __launch_bounds__(512, 1) void dense_strip_kernel(DenseStripArgs a, int chunks_per_b, int chunk_t) {
  static_assert(K * M <= 32768 && K % 8 == 0 && M % 32 == 0, "at most 128 KB of weights");
  constexpr int NB = M / 32, NK4 = K / 8;
  extern __shared__ __attribute__((aligned(16))) float ds_lds[];
  float *W = ds_lds, *BI = ds_lds + K * M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int m_base = blockIdx.y * M;
  const int li = lane & 31, lh = lane >> 5;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  for (int r = tid; r < K * M; r += 512) {  // source order: coalesced
    const int m = TRANSPOSED ? r % M : r / K, k = TRANSPOSED ? r / M : r % K;
    const int kk = k >> 1;
    const float v = TRANSPOSED ? a.wmat[(size_t)k * a.ldw + m_base + m] : a.wmat[(size_t)(m_base + m) * a.ldw + k];
    W[(((m >> 5) * NK4 + (kk >> 2)) * 64 + (m & 31) + 32 * (k & 1)) * 4 + (kk & 3)] = v;
  }
  if (OUT != 2 && tid < M) BI[tid] = a.bias[m_base + tid];
  __syncthreads();
  const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)W + 16u * lane;
  typedef float dsv4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) dsv4 lds_v4;
  constexpr int RSRC = 0x00020000;
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xin.p + (size_t)b * a.xin.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.yout.p + (size_t)b * a.yout.sb + (size_t)m_base * a.yout.ld), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.ref.p + (size_t)b * a.ref.sb + (size_t)m_base * a.ref.ld), 0, 0x7FFFFFFF, RSRC);
  int xld4 = 4 * a.xin.ld, yld4 = 4 * a.yout.ld, rld4 = 4 * a.ref.ld;
  for (int t0 = tb + 32 * wave; t0 < te; t0 += 32 * 8) {
    const int t = t0 + li;
    const bool live = t >= a.t_begin && t < te, out_live = live && t < a.t_out_end;
    const int tc = live ? t : a.t_begin;
    const int ox = 4 * (lh * a.xin.ld + tc), oy = 4 * (4 * lh * a.yout.ld + tc), orf = 4 * (4 * lh * a.ref.ld + tc);
    float xr[K / 2];
    FS_FENCE(xld4);
#pragma unroll
    for (int kk = 0; kk < K / 2; ++kk) {
      float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, ox, 2 * kk * xld4, 0));
      if (IN == 1) v = leaky(v);
      xr[kk] = live ? v : 0.f;
    }
    // (all of the strip's loads are issued before its first MFMA: the scheduler otherwise requests
    // a K = 256 input just in time, 65 registers and a memory round trip every few MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < NK4; ++k4) {
      dsv4 aw[NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) aw[blk] = *(const lds_v4 *)(uintptr_t)(wa + 4u * (unsigned)((blk * NK4 + k4) * 256));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[blk][e], xr[4 * k4 + e], acc[blk], 0, 0, 0);
    }
    FS_FENCE(yld4);
    FS_FENCE(rld4);
    if (out_live) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m0 = 32 * blk + (r & 3) + 8 * (r >> 2);
          float y = acc[blk][r];
          if (OUT == 0) {
            y = y + BI[m0 + 4 * lh];
          } else if (OUT == 1) {
            y = leaky(y + BI[m0 + 4 * lh]);
          } else {
            const float rv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, orf, m0 * rld4, 0));
            y = y * (rv > 0.f ? 1.0f : kLeakySlope);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), yb, oy, m0 * yld4, 0);
        }
    }
  }
}

// `m_total` output rows = m_total / M row blocks
template <int K, int M, int IN, int OUT, bool TRANSPOSED>
static int launch_dense_strip(const DenseStripArgs &a, int m_total, int batch, hipStream_t s) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch * (m_total / M), 1, &chunks, &chunk_t, 256);
  const void *fn = (const void *)dense_strip_kernel<K, M, IN, OUT, TRANSPOSED>;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(dense_strip)");
  if (rc) return rc;
  hipLaunchKernelGGL((dense_strip_kernel<K, M, IN, OUT, TRANSPOSED>), dim3(chunks * batch, m_total / M), dim3(512),
                     (K * M + M) * sizeof(float), s, a, chunks, chunk_t);
  return MVN_OK;
}

template <int NTB>
static int launch_fused_layer64p_t(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  constexpr int TT = 32 * NTB, LDS_BYTES = (2 * 128 + 3 * 64) * (TT + 4) * (int)sizeof(float);
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch, NTB == 1 ? 2 : 1, &chunks, &chunk_t, TT);  // one round of workgroups (fused_bwd.h)
  const void *fn = (const void *)fused_layer64p_kernel<NTB>;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(fused_layer64p)");
  if (rc) return rc;
  hipLaunchKernelGGL(fused_layer64p_kernel<NTB>, dim3(chunks * batch), dim3(256 * NTB), LDS_BYTES, s, a, chunks, chunk_t);
  return MVN_OK;
}
// The strip kernel is the default; MOVENET_HIP_FORWARD_TILE=32 / 64 select the tile kernels (A/B, tests)
static int launch_fused_layer64p(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  if (a.ctx.p) return launch_fused_layer64s(a, batch, s);  // conditioned: the strip kernel only (the caller checks the row length)
  const int tile = switches().forward_tile;
  if (tile == 64) return launch_fused_layer64p_t<2>(a, batch, s);
  if (tile == 32) return launch_fused_layer64p_t<1>(a, batch, s);
  // (the strip kernel's buffer resources span 2 GB from a sequence's base: rows of more than 4 M
  // columns -- 25 x the reference's MAX_AUDIO_FRAMES -- go to the tile kernel)
  if (a.xin.ld > (1 << 22) || a.skip.ld > (1 << 22)) return launch_fused_layer64p_t<1>(a, batch, s);
  return launch_fused_layer64s(a, batch, s);
}

}  // namespace mvn
