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
  const int tb = (a.t_begin & ~3) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
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

template <int NTB>
static int launch_fused_layer64p_t(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  constexpr int TT = 32 * NTB, LDS_BYTES = (2 * 128 + 3 * 64) * (TT + 4) * (int)sizeof(float);
  const int nt = a.t_end - (a.t_begin & ~3);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch, NTB == 1 ? 2 : 1, &chunks, &chunk_t, TT);  // one round of workgroups (fused_bwd.h)
  const void *fn = (const void *)fused_layer64p_kernel<NTB>;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(fused_layer64p)");
  if (rc) return rc;
  hipLaunchKernelGGL(fused_layer64p_kernel<NTB>, dim3(chunks * batch), dim3(256 * NTB), LDS_BYTES, s, a, chunks, chunk_t);
  return MVN_OK;
}
// MOVENET_HIP_FORWARD_TILE=64: the one-workgroup-per-CU form (A/B)
static int launch_fused_layer64p(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  const char *e = getenv("MOVENET_HIP_FORWARD_TILE");
  if (e && e[0] == '6') return launch_fused_layer64p_t<2>(a, batch, s);
  return launch_fused_layer64p_t<1>(a, batch, s);
}

}  // namespace mvn
