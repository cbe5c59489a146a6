// Backward of one gated residual layer as ONE kernel (C = K = 64), r4: the two halves of fused_bwd.h
// joined so that df | dg never crosses HBM (it was written once and read twice: 384 of the 896 floats a
// layer's backward moved per position).  The reference arithmetic is movenet/modules.py:67-93 differentiated:
//   dz  = Wr^T dxo + Ws^T dskip,   df = dz sg (1 - th^2),   dg = dz th sg (1 - sg)
//   dWr += dxo z^T, dWs += dskip z^T, dbr += sum_t dxo, dbs += sum_t dskip           (z = th sg)
//   dWf|dWg[o][c][tap 1] += dfg[o][t] x[c][t],   [tap 0] += dfg[o][t] x[c][t - d]
//   dx[u] = [u >= t_lo] (dxo[u] + W1^T dfg[u]) + [u + d < T] W0^T dfg[u + d]
// The last line is what kept the halves apart: dx[u] needs dfg at u AND at u + d, i.e. another tile.  Here the
// gradient w.r.t. a layer's input leaves in SCATTER form, as two tensors
//   A'[u] = dxo[u] + W1^T dfg[u]   (u in [t_lo, T)),     P0[t] = W0^T dfg[t]   (t in [t_lo, T)),
// both products of the SAME dfg tile, and the layer below stages its dxo as A'[t] (t >= up_lo) + P0[t + up_d]
// (t + up_d < T): the dilation shift moves from the producer's operand (two tiles of dfg) to the consumer's load
// (a second 64-row tensor).  Per position the backward of a layer reads A', P0, dskip, tanh, sigmoid, x(t), x(t - d)
// (448 floats) and writes A', P0 (128): 576 floats against 896.
//
// A 512-thread workgroup (one per CU) owns 64-column tiles; LDS: U (128 x 64: [dxo; dskip], later [x(t - d); x(t)]),
// G (tanh | sigmoid, overwritten in place by df | dg), O ([dxo -> A'; P0]: the output staging tile), all fp32 with
// pitch 68, and [Wr | Ws]^T as three bf16 planes (48 KB, the dz product's A operand): 150 KB.  Every product runs on
// the bf16 matrix cores in the bf16 x 3 form (bf3.h), each phase as an OPERAND PIPELINE -- the LDS reads of stage s + 1
// are issued ahead of the split and the six MFMAs of stage s:
//   phase 1 (12 slots per wave): wave (K half, t block, c block) forms its half of a 32 c x 32 t block of dz (K = 64
//            of the tile's 128 rows, read across rows; weights from LDS) and ONE block of the residual / skip weight
//            gradient (K = time); the tile's x rows are in flight meanwhile and land in U behind the barrier;
//   gate:    the two K halves of a block meet through the idle P0 staging rows, each wave handing over the half of its
//            partial sums the other one finishes: all eight waves turn tanh | sigmoid into df | dg in place;
//   phase 2 (20 slots): wave (tap, K half, channel block) forms both 32-step blocks of its tap's product with its 48 plane
//            registers of W_tap, every wave owns two blocks of the filter / gate weight gradient; the next tile's loads
//            are issued in parts between the slots; the K halves meet in O.
// Both phases are software-pipelined INSIDE the wave (r4c, bf3_slot in bf3.h): a slot issues the LDS reads of the next
// operand, then the six MFMAs of this one with, behind them, this operand's m and l planes and the next operand's h plane.
// Five barriers per tile (the two halves had three each).  Slabs and bias partial sums leave in wgrad2's format;
// reduce_layer64_kernel adds them up.
//
// Where its time goes (timing builds 71-74 and the stamps of build 76, config 2, same box, us per layer): 198-209 as built,
// 190 without any global access, 130 without MFMAs -- an ON-CHIP time.  By the stamps a tile costs 23.6 k cycles, 16.8 k of
// them in the two phases, where every stage is a split of eight values (44 vector instructions) feeding six MFMAs (54.9 M
// vector instructions per launch = 1900 per wave and tile, matrix pipe busy 36 %).  The two waves of a SIMD do NOT share it
// evenly: waves 0-3 (the older ones) finish phase 2 in 8.1 k cycles and wait 3.5 k at the barrier, waves 4-7 take 10.9 k
// (phase 1: 4.8 k / 6.1 k) -- the SIMD issues from the older wave whenever it can, and a straight-line stream of MFMAs and
// vector instructions always can.  Tried on that (r4c; scripts/probes/simd_pairing.hip, dot2_split.hip;
// same-box timing builds, none kept): waves 4-7 entering a phase 128 .. 320 cycles late (no change, not even the sleep's
// cost: the partner fills in); static or per-stage alternating s_setprio (+1 .. +3 %); idle cycles in front of every MFMA so
// that the partner's MFMAs fit in between (+7 .. +40 %: they do not); the residuals through v_dot2c_f32_bf16 (28
// instructions per split instead of 44, but the instruction runs at a quarter of the rate).  Kept: the split of operand s + 1
// and the low planes of operand s BEHIND the MFMAs of operand s (bf3_slot): waves 0-3 phase 2 8.1 k -> 6.6 k cycles, waves
// 4-7 10.9 k -> 10.5 k, the kernel 1 - 2 % (203.9 -> 202.1 us on one box, 213.9 -> 209.2 on another).  In the probe two
// waves running [44 vector instructions][6 MFMAs] loops share a SIMD at 222 cycles per stage (pipe bound 192) when every
// stage ends in a branch, 246 - 250 with four stages between branches, and this kernel's unrolled phases run at 262 - 272.
// Measured earlier, none kept: the split's residuals as packed
// subtractions (36 instructions instead of 44: 212.6 against 205.5 us -- a v_pk_add_f32 costs more than the two adds it
// replaces); two wave roles (waves 0-3 the data path over full K with 96 plane registers of tap weights and no K halves
// to meet, waves 4-7 both weight gradients as 2 x 2 blocks: 52 splits per SIMD and tile instead of 64, four barriers
// instead of five) -- 209 against 198 us: the prefetched tile no longer fits the register file and waits in scratch.
// Operands split ONCE into planes in LDS was built too (r4b, a whole second kernel, parity green on its first run): tiles as
// [row][64 t] bf16 x three planes written by the staging threads and by the gate, row operands fetched with one ds_read_b128
// per plane, operands read across the tile's rows with two ds_read_b64_tr_b16 per plane (the CDNA4 transposed read;
// scripts/probes/tr_read.hip pins its lane map: element j of lane (li, lh) = tile[r0 + 8 lh + j][c0 + li]), 700 vector
// instructions per wave and tile instead of 1900.  The planes are 1.5 x the bytes, so the dz weights leave LDS for
// registers, and 96 plane registers of resident weights do not fit beside the accumulators: the compiler keeps part of them
// in scratch and reloads them in front of the MFMAs that need them (225 us per layer); streamed from packed images per tile
// instead (dz weights: 204 us; tap weights too: 234 us -- the tile's own loads queue behind 144 KB of weight loads per tile
// and CU).  Same box, the form above: 187-195 us.  With 256 registers per wave at two waves per SIMD and 160 KB of LDS the
// two forms meet at the same place from opposite sides.  (Four waves of 512 registers per CU would NOT be the next step: a
// wave never overlaps its own vector instructions with its own MFMAs -- scripts/probes/simd_pairing.hip -- only the SIMD's
// other wave's work hides under an MFMA.)
#pragma once
#include "fused_bwd.h"
#include "fused_fwd.h"
#include "fused_fwd_bf3.h"

namespace mvn {

struct FusedBwdLArgs {
  int t_lo, t_end, d;      // outputs and dfg cover [t_lo = A_{l+1}, t_end = T); d = this layer's dilation
  int t_skip0, t_base;
  int up_lo, up_d;         // the layer above: its A' holds values from up_lo = t_lo + up_d, its P0 is read at t + up_d
  const float *wr, *ws;    // (64 out, 64 in) each
  const float *wf, *wg;    // (64 out, 64 in, 2 taps) each
  Act ga, gp;              // A', P0 of the layer above; ga.p == NULL: last layer (its residual output is unused)
  Act dskip, th, sg, xin;
  Act oa, op;              // A'[t], P0[t] of this layer, t in [t_lo, t_end)
  Act dfg;                 // conditioned layers: df | dg written for the context pass (bwd_dctx_wgctx64_kernel)
};

constexpr int FBL_TILE_F = 128 * W2_LD;                 // floats per LDS tile
constexpr int FBL_WIMG_BYTES = 2 * 8 * 3 * 1024;        // [c block][k-step][plane][lane][8 bf16]
constexpr int FBL_LDS_BYTES = 3 * FBL_TILE_F * 4 + FBL_WIMG_BYTES;  // 150 KB

__device__ __forceinline__ f4 f4_add(const f4 &x, const f4 &y) { return f4{x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w}; }

template <bool WRITE_DFG>
__global__ __launch_bounds__(512, 1) void bwd_layer64_kernel(FusedBwdLArgs a, int chunks_per_b, int chunk_t,
                                                            float *__restrict__ rs_bias_part, float *__restrict__ rs_part,
                                                            float *__restrict__ fg_part) {
  constexpr int C = FB_C, LD = W2_LD, TT = W2_T;
  extern __shared__ __attribute__((aligned(16))) float fbl_lds[];
  float (*U)[LD] = (float (*)[LD])fbl_lds;                      // [dxo; dskip], then [x(t - d); x(t)]
  float (*Gt)[LD] = (float (*)[LD])(fbl_lds + FBL_TILE_F);      // tanh | sigmoid, then df | dg
  float (*O)[LD] = (float (*)[LD])(fbl_lds + 2 * FBL_TILE_F);   // dxo -> A' | P0
  unsigned short *wimg = (unsigned short *)(fbl_lds + 3 * FBL_TILE_F);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5, h4 = 4 * lh;
  const int tb = (a.t_lo & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_lo, a.t_skip0);
  const bool has_dxo = a.ga.p != nullptr;
  // timing builds (wrong results; python -m movenet_amd.csrc.build --stamps --exp=N): 71 no global loads after a
  // workgroup's first tile, 72 no global stores, 73 both, 74 no MFMAs
#if MVN_EXP == 71 || MVN_EXP == 73
  const bool ld_ok = a.t_end < 0;  // (never: the loads stay in the code, none is executed)
#else
  constexpr bool ld_ok = true;
#endif
#if MVN_EXP == 72 || MVN_EXP == 73
  const bool st_ok = a.t_end < 0;
#else
  constexpr bool st_ok = true;
#endif

  // ---- [Wr | Ws]^T as the dz product's A operand: lane -> row c = 32 cb + (lane & 31), element j of k-step ks ->
  // o = 16 ks + 8 (lane >> 5) + j (o < 64: residual rows, else skip rows), three planes 1 KB apart
  for (int i = tid; i < 2 * C * C; i += 512) {
    const int o = i >> 6, c = i & 63;
    const float w = o < C ? a.wr[(size_t)o * C + c] : a.ws[(size_t)(o - C) * C + c];
    unsigned short h, m, l;
    bf3_split1(w, h, m, l);
    const int at = ((((c >> 5) * 8 + (o >> 4)) * 3) * 64 + (c & 31) + 32 * ((o >> 3) & 1)) * 8 + (o & 7);
    wimg[at] = h;
    wimg[at + 512] = m;
    wimg[at + 1024] = l;
  }
  const unsigned wa0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short *)wimg + 16u * lane;

  // ---- phase-2 roles (bwd_dx_wgfg64_kernel's): tap half, K half, channel block; B operand W_tap[o][32 wc + li]
  // for o = 64 kh + 16 j + 8 lh + e as planes
  const int half = wave >> 2, kh = (wave >> 1) & 1, wc = wave & 1;
  u32x4 wp[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = 64 * kh + 16 * j + 8 * lh + e;
      const float *src = o < C ? a.wf : a.wg;
      wv[e] = src[((size_t)(o & (C - 1)) * C + 32 * wc + li) * 2 + (half ? 0 : 1)];
    }
    bf3_split8(wv, wp[j][0], wp[j][1], wp[j][2]);
  }
  const int wm = wave >> 1, wn = wave & 1;  // filter / gate weight gradient: rows [32 wm, +32), columns [64 wn, +64)
  f32x16 accw[2], accr;                     // accr: residual / skip weight gradient, block (rows [32 wm, +32), channels [32 wn, +32))
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[0][r] = accw[1][r] = accr[r] = 0.f;
  // phase-1 roles: dz block (channels [32 cb1, +32), steps [32 tq1, +32)) over the K half kh1 of the tile's 128 rows
  const int kh1 = wave >> 2, tq1 = (wave >> 1) & 1, cb1 = wave & 1;
  float *Zx = &O[C][0];  // the first K halves' partial dz (4 blocks x 16 x 64 floats) in the P0 staging rows, idle in phase 1

  // ---- staging: thread -> rows (tid >> 4) + 32 p, columns 4 (tid & 15) .. +3
  const int srow = tid >> 4, st = 4 * (tid & 15);
  f4 r_ga[2], r_gp[2], r_ds[2], r_th[2], r_sg[2], r_x[4];
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  bool z_ga = false, z_ds = false;  // the staged tile's dxo / dskip rows are absent (zeroed at the LDS store)
  const __amdgpu_buffer_rsrc_t gab = fb_rsrc(a.ga.p + (size_t)b * a.ga.sb);
  const __amdgpu_buffer_rsrc_t gpb = fb_rsrc(a.gp.p + (size_t)b * a.gp.sb);
  const __amdgpu_buffer_rsrc_t dskb = fb_rsrc(a.dskip.p + (size_t)b * a.dskip.sb);
  const __amdgpu_buffer_rsrc_t thb = fb_rsrc(a.th.p + (size_t)b * a.th.sb);
  const __amdgpu_buffer_rsrc_t sgb = fb_rsrc(a.sg.p + (size_t)b * a.sg.sb);
  const __amdgpu_buffer_rsrc_t xinb = fb_rsrc(a.xin.p + (size_t)b * a.xin.sb);
  const __amdgpu_buffer_rsrc_t oab = fb_rsrc(a.oa.p + (size_t)b * a.oa.sb);
  const __amdgpu_buffer_rsrc_t opb = fb_rsrc(a.op.p + (size_t)b * a.op.sb);
  const __amdgpu_buffer_rsrc_t dfgb = fb_rsrc(a.dfg.p + (size_t)b * a.dfg.sb);
  // (every tensor but dskip is a (B, ch, Tp) view with the same row pitch -- checked by the launcher -- so one
  // per-lane offset and one scalar row pitch serve them all)
  const int vo_t = 4 * (srow * a.th.ld + st), vo_ds = 4 * (srow * a.dskip.ld + st);
  // a tile is INTERIOR when every operand row either covers it or misses it altogether (wave-uniform): raw
  // 16-byte buffer loads then; edge tiles take the masked form
  auto interior = [&](int t0) {
    const bool x_full = t0 >= a.t_lo && t0 + TT <= te;
    const bool ga_ok = !has_dxo || t0 >= a.up_lo || t0 + TT <= a.up_lo;
    const bool gp_ok = !has_dxo || t0 + TT + a.up_d <= a.t_end || t0 + a.up_d >= a.t_end;
    const bool s_ok = t0 >= skip_lo || t0 + TT <= skip_lo;
    return x_full && ga_ok && gp_ok && s_ok;
  };
  auto gload_r_part = [&](int t0, int part) {  // part 0..4: A', P0 (shifted), dskip, tanh, sigmoid of tile t0
    int ld_t = 4 * a.th.ld, ld_ds = 4 * a.dskip.ld;
    asm volatile("" : "+s"(ld_t), "+s"(ld_ds));  // (formed here, not hoisted)
    const int c4 = 4 * t0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (part == 0) {
        // (absent rows load a row that IS valid and are zeroed at the LDS store)
        r_ga[p] = (has_dxo && t0 >= a.up_lo) ? fb_load16(gab, vo_t, 32 * p * ld_t + c4) : fb_load16(thb, vo_t, c4);
      } else if (part == 1) {
        r_gp[p] = (has_dxo && t0 + a.up_d < a.t_end) ? fb_load16(gpb, vo_t, 32 * p * ld_t + c4 + 4 * a.up_d) : kZero4;
      } else if (part == 2) {
        r_ds[p] = t0 >= skip_lo ? fb_load16(dskb, vo_ds, 32 * p * ld_ds + c4 - 4 * a.t_base) : fb_load16(sgb, vo_t, c4);
      } else if (part == 3) {
        r_th[p] = fb_load16(thb, vo_t, 32 * p * ld_t + c4);
      } else {
        r_sg[p] = fb_load16(sgb, vo_t, 32 * p * ld_t + c4);
      }
    }
  };
  auto gload_r_edge = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = 32 * p + srow_q;
      r_ga[p] = has_dxo ? ld4_edge(a.ga.at(b, row, 0), t, a.up_lo, te) : kZero4;
      r_gp[p] = has_dxo ? ld4_edge(a.gp.at(b, row, 0) + a.up_d, t, a.t_lo, min(a.t_end - a.up_d, te)) : kZero4;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = 32 * p + srow_q;
      r_ds[p] = ld4_edge(a.dskip.at(b, row, 0) - a.t_base, t, skip_lo, te);
      r_th[p] = ld4_edge(a.th.at(b, row, 0), t, a.t_lo, te);
      r_sg[p] = ld4_edge(a.sg.at(b, row, 0), t, a.t_lo, te);
    }
  };
  auto set_zero_flags = [&](int t0, bool inter) {  // which row groups of tile t0 are absent (interior tiles only)
    z_ga = inter && (!has_dxo || t0 < a.up_lo);
    z_ds = inter && t0 < skip_lo;
  };
  auto gload_x = [&](int t0) {  // rows [0, 64): x(t - d); [64, 128): x(t)
    if (t0 >= a.t_lo && t0 + TT <= te) {
      int ld_x = 4 * a.th.ld;
      asm volatile("" : "+s"(ld_x));
#pragma unroll
      for (int p = 0; p < 4; ++p) r_x[p] = fb_load16(xinb, vo_t, 32 * (p & 1) * ld_x + 4 * t0 - (p < 2 ? 4 * a.d : 0));
    } else {
      int srow_q = srow;
      asm volatile("" : "+v"(srow_q));
      const int t = t0 + st;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        r_x[p] = ld4_edge(a.xin.at(b, 32 * (p & 1) + srow_q, 0) - (p < 2 ? a.d : 0), t, a.t_lo, te);
    }
  };
  auto lstore_r = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      // (absent A': P0 alone; P0 past the end of the sequence or no dxo at all: r_gp is zero)
      const f4 dxo = z_ga ? r_gp[p] : f4_add(r_ga[p], r_gp[p]);
      const f4 dsk = z_ds ? kZero4 : r_ds[p];
      *(f4 *)&U[32 * p + srow][st] = dxo;
      *(f4 *)&O[32 * p + srow][st] = dxo;
      *(f4 *)&U[C + 32 * p + srow][st] = dsk;
      bsum[p] += (dxo.x + dxo.y) + (dxo.z + dxo.w);  // bias gradients = row sums
      bsum[2 + p] += (dsk.x + dsk.y) + (dsk.z + dsk.w);
      *(f4 *)&Gt[32 * p + srow][st] = r_th[p];
      *(f4 *)&Gt[C + 32 * p + srow][st] = r_sg[p];
    }
  };

  {
    const bool inter = interior(tb);
    if (inter) {
#pragma unroll
      for (int part = 0; part < 5; ++part) gload_r_part(tb, part);
    } else {
      gload_r_edge(tb);
    }
    set_zero_flags(tb, inter);
  }
  lstore_r();
  __syncthreads();  // (covers the weight image too)
#if MVN_EXP == 76  // (diagnostic build: cycles per section of a tile, summed over the workgroup's tiles, printed by two waves)
  unsigned tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (unsigned)__builtin_readcyclecounter(), ntile = 0;
#define FBL_STAMP(i_) { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); tsum[i_] += now_ - tlast; tlast = now_; }
#else
#define FBL_STAMP(i_)
#endif
  for (int t0 = tb; t0 < te; t0 += TT) {
    const bool more = t0 + TT < te;
    FBL_STAMP(7)
    if (ld_ok || t0 == tb) gload_x(t0);  // this tile's x rows fly under phase 1
    __builtin_amdgcn_sched_barrier(0);
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    {
      // ---- phase 1: 12 slots
      //   stages 0-3:  dz (32 c x 32 t), this wave's K half = [Wr | Ws]^T (A: planes in LDS) x [dxo; dskip] (B: read
      //                across the tile's rows), one k-step each;
      //   stages 4-11: residual / skip weight gradient, rows [32 wm, +32) of [dxo; dskip] x z^T (channels [32 wn, +32)),
      //                K = time: per 16 steps the row operand (even stage), then z = tanh x sigmoid and the product
      float ob[2][8], sgv[8];
      auto fetch1 = [&](int sI, float (&o)[8]) {
        if (sI < 4) {
          const int ks = 4 * kh1 + sI;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = U[16 * ks + 8 * lh + e][32 * tq1 + li];
        } else {
          const int G = (sI - 4) >> 1;
          if (((sI - 4) & 1) == 0) {
            const f4 a0 = *(const f4 *)&U[32 * wm + li][16 * G + 2 * h4], a1 = *(const f4 *)&U[32 * wm + li][16 * G + 2 * h4 + 4];
            o[0] = a0.x; o[1] = a0.y; o[2] = a0.z; o[3] = a0.w; o[4] = a1.x; o[5] = a1.y; o[6] = a1.z; o[7] = a1.w;
          } else {
            const f4 t0v = *(const f4 *)&Gt[32 * wn + li][16 * G + 2 * h4], t1v = *(const f4 *)&Gt[32 * wn + li][16 * G + 2 * h4 + 4];
            const f4 s0v = *(const f4 *)&Gt[C + 32 * wn + li][16 * G + 2 * h4], s1v = *(const f4 *)&Gt[C + 32 * wn + li][16 * G + 2 * h4 + 4];
            o[0] = t0v.x; o[1] = t0v.y; o[2] = t0v.z; o[3] = t0v.w; o[4] = t1v.x; o[5] = t1v.y; o[6] = t1v.z; o[7] = t1v.w;
            sgv[0] = s0v.x; sgv[1] = s0v.y; sgv[2] = s0v.z; sgv[3] = s0v.w; sgv[4] = s1v.x; sgv[5] = s1v.y; sgv[6] = s1v.z; sgv[7] = s1v.w;
          }
        }
      };
      // software-pipelined like phase 2 (bf3_slot): slot s reads operand s + 1 (and the next k-step's weight planes), then
      // issues the MFMAs of operand s with its m and l planes and the h plane of operand s + 1 formed behind them
      typedef __attribute__((address_space(3))) u32x4 lds_u4;
      u32x4 wA[2][3], hc, hn, mc, lc, ah, am, al;
      auto fetchA = [&](int sI, u32x4 (&w)[3]) {
        const unsigned wa = wa0 + 3072u * (unsigned)(cb1 * 8 + 4 * kh1 + sI);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w[pl] = *(const lds_u4 *)(uintptr_t)(wa + 1024u * pl);
      };
      fetch1(0, ob[0]);
      fetchA(0, wA[0]);
      bf3_peel(ob[0], hc);
      __builtin_amdgcn_sched_barrier(0);
#if MVN_EXP == 74
#define FBL_MF(acc_, a_, b_) acc_[0] += __uint_as_float(a_[0] ^ b_[0])
#else
#define FBL_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
#endif
#pragma clang loop unroll(full)
      for (int sI = 0; sI < 12; ++sI) {
        float (&xc)[8] = ob[sI & 1];
        float (&xn)[8] = ob[(sI + 1) & 1];
        if (sI + 1 < 12) fetch1(sI + 1, xn);
        if (sI + 1 < 4) fetchA(sI + 1, wA[(sI + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (sI < 4) {
          const u32x4 &wh = wA[sI & 1][0], &wmid = wA[sI & 1][1], &wl = wA[sI & 1][2];
          bf3_slot(true, xc, xn, mc, lc, hn,
                   [&]() { FBL_MF(accd, wl, hc); }, [&]() { FBL_MF(accd, wmid, hc); }, [&]() { FBL_MF(accd, wh, hc); },
                   [&]() { FBL_MF(accd, wmid, mc); }, [&]() { FBL_MF(accd, wh, mc); }, [&]() { FBL_MF(accd, wh, lc); });
        } else if (((sI - 4) & 1) == 0) {
          // the row operand's planes; then z = tanh x sigmoid of the next slot and its h plane
          ah = hc;
          bf3_peel(xc, am);
          bf3_pack_l(xc, al);
#pragma unroll
          for (int e = 0; e < 8; ++e) xn[e] *= sgv[e];
          bf3_peel(xn, hn);
        } else {
          bf3_slot(sI + 1 < 12, xc, xn, mc, lc, hn,
                   [&]() { FBL_MF(accr, al, hc); }, [&]() { FBL_MF(accr, am, hc); }, [&]() { FBL_MF(accr, ah, hc); },
                   [&]() { FBL_MF(accr, am, mc); }, [&]() { FBL_MF(accr, ah, mc); }, [&]() { FBL_MF(accr, ah, lc); });
        }
        hc = hn;
        __builtin_amdgcn_sched_barrier(0);
      }
#undef FBL_MF
      // the two K halves of a block meet through the P0 staging rows, each wave handing over the HALF of its partial
      // sums the other one finishes (registers 8 (1 - kh1) .. +8), so that all eight waves share the gate derivative
#pragma unroll
      for (int r = 0; r < 8; ++r) Zx[(((2 * tq1 + cb1) * 2 + kh1) * 8 + r) * 64 + lane] = kh1 ? accd[r] : accd[8 + r];
    }
    __builtin_amdgcn_sched_barrier(0);
    FBL_STAMP(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the x rows (and the previous tile's stores)
    __syncthreads();  // B1: U and tanh | sigmoid have been read; the partial sums of dz are staged
    FBL_STAMP(1)
    {
      // ---- gate derivative in place: this lane holds dz of channels 32 cb1 + acc_row(r) at t = 32 tq1 + li, r in [8 kh1, +8)
      float tv[8], sv[8], dzv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int rr = 8 * kh1 + r, c = 32 * cb1 + (rr & 3) + 8 * (rr >> 2) + h4, tc = 32 * tq1 + li;
        tv[r] = Gt[c][tc];
        sv[r] = Gt[C + c][tc];
        dzv[r] = Zx[(((2 * tq1 + cb1) * 2 + (1 - kh1)) * 8 + r) * 64 + lane];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int rr = 8 * kh1 + r, c = 32 * cb1 + (rr & 3) + 8 * (rr >> 2) + h4, tc = 32 * tq1 + li;
        const float dz = dzv[r] + (kh1 ? accd[8 + r] : accd[r]);
        Gt[c][tc] = dz * sv[r] * (1.0f - tv[r] * tv[r]);
        Gt[C + c][tc] = dz * tv[r] * sv[r] * (1.0f - sv[r]);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) *(f4 *)&U[32 * p + srow][st] = r_x[p];
    __syncthreads();  // B2: df | dg and the x rows are staged
    FBL_STAMP(2)
    if (WRITE_DFG && st_ok) {
      const int t = t0 + st;
      if (t >= a.t_lo && t + 3 < te) {
#pragma unroll
        for (int p = 0; p < 4; ++p) fb_store16(*(const f4 *)&Gt[32 * p + srow][st], dfgb, vo_t, 4 * (32 * p * a.th.ld + t0));
      } else {
        float *base = a.dfg.p + (size_t)b * a.dfg.sb + t;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= a.t_lo && t + e < te) base[(size_t)(32 * p + srow) * a.dfg.ld + e] = Gt[32 * p + srow][st + e];
      }
    }
    const bool spread = more && ld_ok && interior(t0 + TT);
    if (more && ld_ok && !spread) gload_r_edge(t0 + TT);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: this wave's K half of its tap's product for both 32-step blocks, and its two blocks of the
    // filter / gate weight gradient; the next tile's loads in parts between the steps
    f32x16 accd2[2];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd2[ub][r] = 0.f;
    {
      // 20 slots, five per 16 time steps -- the weight gradient's row operand of dfg (planes only), its two x operands, then
      // this wave's k-step of the tap product for both 32-step blocks (operand read ACROSS the tile's rows) -- software-
      // pipelined INSIDE the wave (r4c): slot s issues the LDS reads of operand s + 1, then the six MFMAs of operand s with,
      // behind them, the m and l planes of operand s and the h plane of operand s + 1 (bf3_slot): no vector instruction
      // stands in front of an MFMA it does not feed.
      float xb[2][8];
      auto fetch2 = [&](int sI, float (&o)[8]) {
        const int G = sI / 5, k = sI - 5 * G;
        if (k < 3) {
          const float *row = k == 0 ? &Gt[32 * wm + li][0] : &U[64 * wn + 32 * (k - 1) + li][0];
          const f4 a0 = *(const f4 *)(row + 16 * G + 2 * h4), a1 = *(const f4 *)(row + 16 * G + 2 * h4 + 4);
          o[0] = a0.x; o[1] = a0.y; o[2] = a0.z; o[3] = a0.w; o[4] = a1.x; o[5] = a1.y; o[6] = a1.z; o[7] = a1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = Gt[64 * kh + 16 * G + 8 * lh + e][32 * (k - 3) + li];
        }
      };
      u32x4 hc, hn, mc, lc, ah, am, al;
      fetch2(0, xb[0]);
      bf3_peel(xb[0], hc);
      __builtin_amdgcn_sched_barrier(0);
#if MVN_EXP == 74
#define FBL_MF(acc_, a_, b_) acc_[0] += __uint_as_float(a_[0] ^ b_[0])
#else
#define FBL_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
#endif
#pragma clang loop unroll(full)
      for (int sI = 0; sI < 20; ++sI) {
        const int G = sI / 5, k = sI - 5 * G;
        float (&xc)[8] = xb[sI & 1];
        if (sI + 1 < 20) fetch2(sI + 1, xb[(sI + 1) & 1]);
        if (spread && k == 0) {
          gload_r_part(t0 + TT, G);
          if (G == 3) gload_r_part(t0 + TT, 4);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the reads stay in front; the planes formed from them wait where they are used)
        if (k == 0) {
          ah = hc;
          bf3_peel(xc, am);
          bf3_pack_l(xc, al);
          bf3_peel(xb[(sI + 1) & 1], hn);
        } else if (k < 3) {
          f32x16 &acc = accw[k - 1];
          bf3_slot(sI + 1 < 20, xc, xb[(sI + 1) & 1], mc, lc, hn,
                   [&]() { FBL_MF(acc, al, hc); }, [&]() { FBL_MF(acc, am, hc); }, [&]() { FBL_MF(acc, ah, hc); },
                   [&]() { FBL_MF(acc, am, mc); }, [&]() { FBL_MF(acc, ah, mc); }, [&]() { FBL_MF(acc, ah, lc); });
        } else {
          f32x16 &acc = accd2[k - 3];
          const u32x4 &wh = wp[G][0], &wmid = wp[G][1], &wl = wp[G][2];
          bf3_slot(sI + 1 < 20, xc, xb[(sI + 1) & 1], mc, lc, hn,
                   [&]() { FBL_MF(acc, hc, wl); }, [&]() { FBL_MF(acc, hc, wmid); }, [&]() { FBL_MF(acc, hc, wh); },
                   [&]() { FBL_MF(acc, mc, wmid); }, [&]() { FBL_MF(acc, mc, wh); }, [&]() { FBL_MF(acc, lc, wh); });
        }
        hc = hn;
        __builtin_amdgcn_sched_barrier(0);
      }
#undef FBL_MF
    }
    FBL_STAMP(3)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next tile's loads, ahead of this tile's stores
    // ---- the K halves meet in O: rows [0, 64) hold dxo (tap 1: A' = dxo + W1^T dfg), rows [64, 128) take P0
    if (kh == 0) {
#pragma unroll
      for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 *p4 = (f4 *)&O[64 * half + 32 * wc + li][32 * ub + 8 * q + h4];
          const f4 v = f4{accd2[ub][4 * q], accd2[ub][4 * q + 1], accd2[ub][4 * q + 2], accd2[ub][4 * q + 3]};
          *p4 = half ? v : f4_add(*p4, v);
        }
    }
    __syncthreads();  // B3: U and G have been read; the first K halves are in O
    FBL_STAMP(4)
    if (kh == 1) {
#pragma unroll
      for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 *p4 = (f4 *)&O[64 * half + 32 * wc + li][32 * ub + 8 * q + h4];
          *p4 = f4_add(*p4, f4{accd2[ub][4 * q], accd2[ub][4 * q + 1], accd2[ub][4 * q + 2], accd2[ub][4 * q + 3]});
        }
    }
    __syncthreads();  // B4
    FBL_STAMP(5)
    if (st_ok) {
      // whole-row float4 stores: rows srow + 32 p (A' rows, then P0 rows), columns t0 + st .. +3 inside [t_lo, te)
      const int t = t0 + st;
      if (t >= a.t_lo && t + 3 < te) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          fb_store16(*(const f4 *)&O[32 * p + srow][st], oab, vo_t, 4 * (32 * p * a.th.ld + t0));
          fb_store16(*(const f4 *)&O[C + 32 * p + srow][st], opb, vo_t, 4 * (32 * p * a.th.ld + t0));
        }
      } else {
        float *ba = a.oa.p + (size_t)b * a.oa.sb + t, *bp = a.op.p + (size_t)b * a.op.sb + t;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= a.t_lo && t + e < te) {
              ba[(size_t)(32 * p + srow) * a.oa.ld + e] = O[32 * p + srow][st + e];
              bp[(size_t)(32 * p + srow) * a.op.ld + e] = O[C + 32 * p + srow][st + e];
            }
      }
    }
    // (a thread's lstore_r() overwrites exactly the O elements the same thread has just read; U and G were
    // last read before B3)
    if (more) {
      set_zero_flags(t0 + TT, spread || !ld_ok);
      lstore_r();
    }
    __syncthreads();  // B5
    FBL_STAMP(6)
#if MVN_EXP == 76
    ++ntile;
#endif
  }
#if MVN_EXP == 76
  if (blockIdx.x == 3 && (tid == 0 || tid == 320))
    printf("FBL wave %d tiles %u: phase1 %u | wait+B1 %u | gate+X %u | phase2 %u | wait+kh0+B3 %u | kh1+B4 %u | out+lstore+B5 %u | top %u\n",
           tid >> 6, ntile, tsum[0] / ntile, tsum[1] / ntile, tsum[2] / ntile, tsum[3] / ntile, tsum[4] / ntile, tsum[5] / ntile,
           tsum[6] / ntile, tsum[7] / ntile);
#endif
  // ---- this workgroup's slabs and bias partial sums (wgrad2_kernel's formats: 128 x 128, 128 x 64, 128)
  {
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, wm_e = wave_e >> 1, wn_e = wave_e & 1;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * wm_e + acc_row(r, lane_e), n = 64 * wn_e + 32 * ni + (lane_e & 31);
        fg_part[((size_t)blockIdx.x * 128 + m) * 128 + n] = accw[ni][r];
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * wm_e + acc_row(r, lane_e), n = 32 * wn_e + (lane_e & 31);
      rs_part[((size_t)blockIdx.x * 128 + m) * 64 + n] = accr[r];
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float v = bsum[p];  // 16 lanes share a row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    // bsum[p]: dxo rows 32 p + srow (p < 2), dskip rows 32 (p - 2) + srow
    if ((tid & 15) == 0) rs_bias_part[(size_t)blockIdx.x * 128 + (p < 2 ? 32 * p : C + 32 * (p - 2)) + srow] = v;
  }
}

// The gradient w.r.t. the FIRST layer's input in dense form, for the embedding / causal-conv gradient:
// dx0[t] = [t >= t_lo] A'[t] + [t + d < T] P0[t + d],  t in [0, T).
__global__ __launch_bounds__(256) void bwd_scatter_combine_kernel(Act oa, Act op, Act dx, int C, int t_lo, int d, int T) {
  const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T || c >= C) return;
  const float va = t >= t_lo ? *oa.at(b, c, t) : 0.f;
  const float vp = t + d < T ? *op.at(b, c, t + d) : 0.f;
  *dx.at(b, c, t) = va + vp;
}

// Launch geometry of the fused layer backward (one round of one workgroup per CU): false when the scratch cannot
// hold its slabs (the caller then keeps the two-half form for the whole pass).
struct FusedBwdLPlan {
  int chunks = 0, chunk_t = 0;
  float *bias = nullptr, *rs = nullptr, *fg = nullptr;
  size_t used = 0;  // floats of `slab` taken
};
static bool bwd_layer64_plan(int t_lo, int t_end, int batch, float *bias_scratch, size_t bias_floats, float *slab,
                             size_t slab_floats, FusedBwdLPlan *pl) {
  const int nt = t_end - (t_lo & ~TILE_ALIGN);
  if (t_end <= t_lo || batch <= 0) return true;
  fb_chunks(nt, batch, 1, &pl->chunks, &pl->chunk_t);
  if (!bias_scratch || !slab) return false;
  // short sequences (the parity tests' sizes): fewer, longer chunks so that the slabs fit the scratch
  const size_t per_wg = 128 * 64 + 128 * 128;
  const size_t n_max = std::min(slab_floats / per_wg, bias_floats / 128) / (size_t)batch;
  if (n_max < 1) return false;
  if ((size_t)pl->chunks > n_max) {
    const int tiles = (nt + W2_T - 1) / W2_T;
    const int chunk_tiles = (tiles + (int)n_max - 1) / (int)n_max;
    pl->chunks = (tiles + chunk_tiles - 1) / chunk_tiles;
    pl->chunk_t = chunk_tiles * W2_T;
  }
  const size_t n = (size_t)pl->chunks * batch;
  pl->used = n * per_wg;
  if (pl->used > slab_floats || n * 128 > bias_floats) return false;
  pl->bias = bias_scratch;
  pl->rs = slab;
  pl->fg = slab + n * 128 * 64;
  return true;
}
template <class RsOp, class FgOp>
static int launch_bwd_layer64(const FusedBwdLArgs &a, const RsOp &rs, const FgOp &fg, int batch, const FusedBwdLPlan &pl,
                              hipStream_t s) {
  if (pl.chunks <= 0) return MVN_OK;
  const bool wd = a.dfg.p != nullptr;
  const int ldt = a.th.ld;
  if (a.sg.ld != ldt || a.xin.ld != ldt || a.oa.ld != ldt || a.op.ld != ldt || a.gp.ld != ldt || (a.ga.p && a.ga.ld != ldt) ||
      (wd && a.dfg.ld != ldt)) {
    set_error("bwd_layer64: the (B, ch, Tp) views must share one row pitch");
    return MVN_ERR_BAD_ARG;
  }
  const int n = pl.chunks * batch;
  const void *fn = wd ? (const void *)bwd_layer64_kernel<true> : (const void *)bwd_layer64_kernel<false>;
  constexpr size_t lds_bytes = FBL_LDS_BYTES;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(bwd_layer64)");
  if (rc) return rc;
  void *args[] = {(void *)&a, (void *)&pl.chunks, (void *)&pl.chunk_t, (void *)&pl.bias, (void *)&pl.rs, (void *)&pl.fg};
  if (check_hip(hipLaunchKernel(fn, dim3(n), dim3(512), args, lds_bytes, s), "bwd_layer64")) return MVN_ERR_LAUNCH;
  hipLaunchKernelGGL((reduce_layer64_kernel<RsOp, FgOp>), dim3(260 + 128 * 128 / 32), dim3(32 * RED_SEG), 0, s, rs, pl.rs,
                     pl.bias, n, fg, pl.fg, n);
  return MVN_OK;
}

}  // namespace mvn
