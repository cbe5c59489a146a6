// PIPE generator with FP16 OPERANDS and FP32 ACCUMULATION, C = K = 128, Q = 256: the form
// BASELINE configs[4] names ("60-layer WaveNet, residual_ch=128, fp16 ... 1x1 convs,
// autoregressive generate of 1 s audio, LDS ring-buffer stress").  Reference precedent for
// reduced precision: torch.autocast in movenet/trainer.py:124; arithmetic restated:
// movenet/wavenet.py:217-237, movenet/modules.py:19-30, :67-93, :139-142 (see generate.hip).
//
// Same structure as gen_pipe_kernel<128> (generate_pipe.hip): a private pipeline of
// workgroups per sequence, weights resident, activations handed on as {value, epoch}
// granules.  What changes with 16-bit operands:
//   * every weight matrix is stored as halves (rounded to nearest even once, at pack time):
//     a thread's two rows x 64 inputs are 64 VGPRs instead of 128, a layer's past-tap matrix
//     64 KB of LDS instead of 128 -- so a stage holds TWO layers and 60 layers are 30 + 1
//     stages, which fit ONE XCD (fp32: 61 stages over two XCDs);
//   * the vector operand of every product (residual stream, popped queue entry, context
//     column, gated activation, head activations) is rounded to fp16 when it is written to
//     LDS, and a lane reads its 64 inputs with 8 ds_read_b128 instead of 16;
//   * products and sums run in fp16 x fp16 -> fp32: v_mfma_f32_16x16x32_f16 in the layer stages (r3,
//     the default: "fp16 MFMA 1x1 convs" as BASELINE configs[4] is written), v_dot2c_f32_f16 in the
//     head and in the layer stages' dot-product form (MOVENET_H16_FORM=dot2).
// Everything else stays fp32: the residual stream itself, the skip sum, the past-tap
// partial sums, biases, gating, the embedding rows (a gather, no product), the dilation
// queues in HBM/L2, logits and the double softmax.
// Tolerance against the fp32 path: DESIGN.md section 2 / tests/test_fp16_gpu.py.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gen_common.h"
#include "pipe_common.h"

namespace mvn {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));  // 16 bytes

namespace h16 {
constexpr int C = 128, Q = 256, NT = 512;
constexpr int LPS = 2;              // layers per stage, dot-product form
// layers per stage, matrix-core form: THREE when a pipeline serves one sequence (21 stages instead of
// 31: ten hops less on the chain; the third layer's past taps and the skip tiles stream from L2
// behind the hand-on), two when it serves several in turn (there the stages' service time per turn
// is what bounds the step, and L2 streaming per turn costs more than ten hops save)
template <bool MULTI>
struct LpsM {
  static constexpr int value = MULTI ? 2 : 3;
};
constexpr int KQ = 2;               // lanes sharing one channel's two rows
constexpr int KPER = C / KQ;        // 64 inputs per lane
constexpr int NV = KPER / 8;        // 8 h8 vectors per row per lane
constexpr int MAT_H = 2 * C * C;    // halves per 2C x C matrix
constexpr int MAT_F = MAT_H / 2;    // ... in float units of the packed blob
constexpr int LAYER_F = 3 * MAT_F + 2 * C;  // WC | WP | WR (halves) | residual, skip biases (fp32)
constexpr int CTX_LAYER_F = MAT_F + 2 * C;
constexpr int EMB_F = 2 * Q * C;    // fp32 tables
constexpr int W1_F = Q * C / 2, W2_F = Q * Q / 2;
constexpr int HEAD_F = W1_F + Q + W2_F + Q;
constexpr int GRAN = 2 * C;         // granules per inbox: C residual stream | C running skip sum
constexpr int GL = C / 64;          // the chain polls the residual granules only: one 16-byte load per lane
constexpr int W1NV = (C / 2) / 8;   // conv1: 2 threads per row, 64 inputs each = 8 vectors
// LDS bytes: layer stage LPS * 64 KB of past-tap weights + vectors; head stage 64 KB of conv1
constexpr int LDS_BYTES = LPS * MAT_H * 2 + 8192;
// r3, several sequences per pipeline (as gen_fold_kernel<true>): a pipeline serves up to GMAX
// sequences in turn; what survives from a sequence's turn to its next is the past-tap part of its
// f/g sums (pf, pg of the LPS layers: four floats per channel), kept in LDS behind the vectors
constexpr int GMAX = 8;
constexpr int PFS_F = 2 * LPS;      // floats per channel and sequence
constexpr int PFS_FM = PFS_F;       // ... matrix-core form (two layers per stage when MULTI)
constexpr int LDS_BYTES_MULTI = LDS_BYTES + GMAX * C * PFS_F * 4;
// matrix-core form: two layers' past taps in LDS (a third streams from L2) + 4 KB of vectors
constexpr int LDS_BYTES_M = 2 * MAT_H * 2 + 4096;
constexpr int LDS_BYTES_M_MULTI = LDS_BYTES_M + GMAX * C * PFS_FM * 4;
static_assert(LDS_BYTES_M_MULTI <= 160 * 1024 && LDS_BYTES_MULTI <= 160 * 1024, "one CU's LDS");
}  // namespace h16

__device__ __forceinline__ float dot8(const h8 w, const h8 x, float acc) {
  acc = __builtin_amdgcn_fdot2(h2{w[0], w[1]}, h2{x[0], x[1]}, acc, false);
  acc = __builtin_amdgcn_fdot2(h2{w[2], w[3]}, h2{x[2], x[3]}, acc, false);
  acc = __builtin_amdgcn_fdot2(h2{w[4], w[5]}, h2{x[4], x[5]}, acc, false);
  return __builtin_amdgcn_fdot2(h2{w[6], w[7]}, h2{x[6], x[7]}, acc, false);
}
// two rows (registers) against the same LDS vector of 8*N halves: four independent chains
// per row, combined in a fixed order
template <int N>
__device__ __forceinline__ void dot2_rows(const h8 (&wa)[N], const h8 (&wb)[N], const _Float16 *xp, float &ra,
                                          float &rb) {
  // the vector is fetched four h8 at a time (16 registers live, not 32: next to 128 weight
  // registers the whole vector spills), the second half requested before the first is used
  constexpr int CH = 4;
  static_assert(N % CH == 0, "whole chunks");
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  h8 x[2][CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[0][i] = ((const h8 *)xp)[i];
#pragma unroll
  for (int ch = 0; ch < N / CH; ++ch) {
    if (ch + 1 < N / CH) {
#pragma unroll
      for (int i = 0; i < CH; ++i) x[(ch + 1) & 1][i] = ((const h8 *)xp)[CH * (ch + 1) + i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      a[i & 3] = dot8(wa[CH * ch + i], x[ch & 1][i], a[i & 3]);
      b[i & 3] = dot8(wb[CH * ch + i], x[ch & 1][i], b[i & 3]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  ra = (a[0] + a[2]) + (a[1] + a[3]);
  rb = (b[0] + b[2]) + (b[1] + b[3]);
}
// one row whose weights are fetched on the fly (LDS or L2): [N][stride] h8 at column idx.
// Four vectors at a time behind scheduling fences: hoisted together, the 2 x N vectors of a
// call (and of the next call) would push resident weight registers to scratch.
template <int N>
__device__ __forceinline__ float dot_stream_h(const h8 *wsrc, int stride, int idx, const _Float16 *xp) {
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i0 = 0; i0 < N; i0 += 4) {
    h8 w[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = wsrc[(i0 + i) * stride + idx];
      x[i] = ((const h8 *)xp)[i0 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = dot8(w[i], x[i], a[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  return (a[0] + a[2]) + (a[1] + a[3]);
}
__device__ __forceinline__ float pair_sum(float v) { return v + dpp_mov<DPP_XOR1>(v); }

// MULTI = false: one sequence per pipeline (nseq == nb).  MULTI = true: pipeline b serves sequences
// b, b + nb, b + 2 nb, ... < nseq in turn.
// MFMA = true (r3): every product of a layer stage on the matrix cores (see the layer stage below).
template <bool MULTI, bool MFMA>
__global__ __launch_bounds__(512, 2) void gen_pipe_h16_kernel(GenArgs a, u64 *hand, unsigned *err, int NS,
                                                             int nb, int nseq) {
  using namespace h16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (xcd, slot) -> (sequence, stage): same placement as gen_pipe_kernel (speed only; every edge
  // verifies its own placement below)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int b, s;
  if (NS <= PIPE_XCD_CUS) {
    b = xcd + 8 * (slot / NS);
    s = slot % NS;
  } else {
    const int XS = (NS + PIPE_XCD_CUS - 1) / PIPE_XCD_CUS, SPX = (NS + XS - 1) / XS;
    b = xcd / XS;
    s = (xcd % XS) * SPX + slot;
    if (slot >= SPX || s >= NS) return;
  }
  if (b >= nb) return;
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;  // sticky status
  const int L = a.L;
  const int s_next = s + 1 == NS ? 0 : s + 1;
  const int G = MULTI ? (nseq - b + nb - 1) / nb : 1;  // sequences of this pipeline
  int bq = b;                                           // the sequence whose turn it is
  u64 *inbox = hand + ((size_t)b * NS + s) * GRAN;
  u64 *outbox = hand + ((size_t)b * NS + s_next) * GRAN;
  constexpr int LDSB = MFMA ? LDS_BYTES_M : LDS_BYTES;
  int *iflag = (int *)(smem_b + LDSB - 64);  // [0] ok flag, [3] fast-edge flag
  float *pfs = (float *)(smem_b + LDSB);     // MULTI: [GMAX][C][PFS_F(M)] (layer stages), head: indices
  bool fast_edge = false;
  {
    unsigned *xcc = err + 16;
    const unsigned mine = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1;  // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
      __hip_atomic_store(xcc + b * NS + s, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned other = 0;
      for (unsigned spins = 0; spins < (1u << 20) && other == 0; ++spins) {
        other = __hip_atomic_load(xcc + b * NS + s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other == 0) __builtin_amdgcn_s_sleep(8);
      }
      iflag[3] = (other == mine) ? 1 : 0;
    }
    __syncthreads();
    fast_edge = iflag[3] != 0;
    __syncthreads();
  }

  if (MFMA && s < NS - 1) {
    // ================= layer stage, MATRIX-CORE form: layers l0 .. l0+nl-1 =================
    // A mat-vec wastes 15 of an MFMA's 16 columns, and still: scripts/probes/h16_phase.hip times one
    // 256 x 128 phase (vector reads, products, barrier) at 412 cycles with eight
    // v_mfma_f32_16x16x32_f16 per wave against 464 (all waves) / 520 (one wave per SIMD, the form
    // below) with v_dot2c -- and the matrix cores return every row's COMPLETE sum to a lane, so the
    // DPP lane sums go as well.  (Half of the rows each way is slower than either: 545.)
    // Wave w owns channels [16 w, 16 w + 16): four 16-row tiles per layer -- filter, gate, residual,
    // skip -- x four k-steps of 32 inputs, the A operands in registers (lane l: row l % 16, inputs
    // 32 kk + 8 (l / 16) .. + 8: 64 registers per layer, as many as the dot-product form); the B
    // operand is the stage's vector in every column (lane l reads inputs 32 kk + 8 (l / 16) .. + 8:
    // four ds_read_b128, the same for the 16 lanes of a row group); the accumulator of lane l holds
    // rows 4 (l / 16) + r, r < 4, in every column: lane 16 q + r (r < 4) post-processes channel
    // 16 w + 4 q + r -- gate, residual add, queue traffic, skip lane -- and is that channel's lead.
    constexpr int LPSM = LpsM<MULTI>::value;
    constexpr bool WS_REG = LPSM == 2;  // the skip tiles in registers (two layers) or streamed from L2 (three)
    const int l0 = s * LPSM, nl = min(LPSM, L - l0);
    const int q = lane >> 4, r4 = lane & 3;
    const bool lead = (lane & 15) < 4;
    const int c = 16 * wave + 4 * q + r4;                    // (every lane: the channel its row group's lane r4 leads)
    // past-tap f|g weights: the first LPW layers' in LDS ([LPW][tile 2][kk 4][512] h8), the third
    // layer's streamed from L2 (off the chain: 64 KB per step of a stage that idles 35 us of it)
    constexpr int LPW = 2;
    h8 *wp = (h8 *)smem_b;
    float *cur = (float *)(smem_b + LPW * MAT_H * 2);        // [C] residual stream (fp32)
    _Float16 *curh = (_Float16 *)(cur + 2 * C);              // [C] the stream as the products' operand
    const u64 *skbox = inbox + C + c;                        // this channel's granule of the skip lane
    _Float16 *zbh = curh + C;                                // [LPSM][C] gated activations (kept for the skip tiles)
    _Float16 *pasth = zbh + LPSM * C;                         // [LPSM][C] popped queue entries
    _Float16 *ctxh = pasth + LPSM * C;                        // [C] context column
    float *ring = a.state + (size_t)b * a.state_per_seq;
    auto bind = [&](int g) {  // MULTI: the pointers of sequence b + g nb
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      skbox = inbox + C + c;
      ring = a.state + (size_t)bq * a.state_per_seq;
    };
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto sel = [&](const f4v &v) { return r4 == 0 ? v[0] : r4 == 1 ? v[1] : r4 == 2 ? v[2] : v[3]; };
    // a 16-row tile x 128 inputs against the vector at `xp` (halves in LDS): four MFMAs
    // (one accumulator over the four k-steps; two accumulators of two steps each measured no faster:
    // the phase is not waiting for MFMA passes)
    auto tile = [&](const h8 (&w)[4], const h8 (&x)[4]) {
      f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kk], x[kk], acc, 0, 0, 0);
      return acc;
    };
    auto vec = [&](h8 (&x)[4], const _Float16 *xp) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) x[kk] = ((const h8 *)xp)[4 * kk + q];
    };

    // (the skip tiles are not on the chain: with three layers per stage their 48 registers are what
    // does not fit -- they then stream from L2 behind the hand-on, 96 KB per step of a stage)
    h8 wF[LPSM][4], wG[LPSM][4], wR[LPSM][4], wS[WS_REG ? LPSM : 1][4];
    float bias_r[LPSM], bias_s[LPSM], pf[LPSM], pg[LPSM], xs[LPSM];
    int doff[LPSM], dmask[LPSM];
#pragma unroll
    for (int j = 0; j < LPSM; ++j) {
      bias_r[j] = 0.f; bias_s[j] = 0.f; pf[j] = 0.f; pg[j] = 0.f; xs[j] = 0.f;
      doff[j] = 0; dmask[j] = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        wF[j][kk] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        wG[j][kk] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        wR[j][kk] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        if (WS_REG) wS[WS_REG ? j : 0][kk] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      if (j < nl) {
        const float *lw = a.w + EMB_F + (size_t)(l0 + j) * LAYER_F;
        const h8 *wc8 = (const h8 *)lw, *wp8 = (const h8 *)(lw + MAT_F), *wr8 = (const h8 *)(lw + 2 * MAT_F);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {  // packed as [tile 2][kk 4][wave 8][lane 64] = [tile][kk][tid]
          wF[j][kk] = wc8[kk * NT + tid];
          wG[j][kk] = wc8[(4 + kk) * NT + tid];
          wR[j][kk] = wr8[kk * NT + tid];
          if (WS_REG) wS[WS_REG ? j : 0][kk] = wr8[(4 + kk) * NT + tid];
        }
        if (j < LPW) {
#pragma unroll
          for (int i = 0; i < 8; ++i) wp[(j * 8 + i) * NT + tid] = wp8[i * NT + tid];
        }
        bias_r[j] = lw[3 * MAT_F + c];
        bias_s[j] = lw[3 * MAT_F + C + c];
        const int l = l0 + j;
        dmask[j] = (1 << (l % a.layer_size)) - 1;
        doff[j] = ring_offset(l, a.layer_size, C);
      }
    }

    // Off the critical path: queue push / pop (lead lanes), then the past-tap half of step tn's f/g
    // sums: the same tiles, A operands streamed from LDS (context convs: from L2)
    auto precompute = [&](int tn, bool push) {
      int tq = tid;
      asm volatile("" : "+v"(tq));  // addresses rebuilt per step: the chain needs the registers
      const int lq = (tq >> 4) & 3, cq = 16 * (tq >> 6) + 4 * lq + (tq & 3);
      if ((tq & 15) < 4) {
#pragma unroll
        for (int j = 0; j < LPSM; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + cq;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j] : ring_load(base + (tn & dmask[j]) * C);
            pasth[j * C + cq] = (_Float16)pv;
          }
      }
      if (a.ctx_tm && tq < C) ctxh[tq] = (_Float16)a.ctx_tm[(size_t)bq * a.ctx_stride_b + (size_t)tn * C + tq];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < LPSM; ++j)
        if (j < nl) {
          h8 x[4], w[4];
          vec(x, pasth + j * C);
          // (layers beyond the LDS-resident ones: their past-tap weights come from the packed blob)
          const h8 *wsrc = j < LPW ? (const h8 *)(wp + j * 8 * NT)
                                   : (const h8 *)(a.w + EMB_F + (size_t)(l0 + j) * LAYER_F + MAT_F);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) w[kk] = wsrc[kk * NT + tq];
          pf[j] = sel(tile(w, x));
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) w[kk] = wsrc[(4 + kk) * NT + tq];
          pg[j] = sel(tile(w, x));
          __builtin_amdgcn_sched_barrier(0);
          if (a.ctx_tm) {
            // 1x1 context convs (modules.py:58-63, :75-77), weights streamed from L2
            const float *wc = a.wctx + (size_t)(l0 + j) * CTX_LAYER_F;
            vec(x, ctxh);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = ((const h8 *)wc)[kk * NT + tq];
            pf[j] += sel(tile(w, x)) + wc[MAT_F + cq];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = ((const h8 *)wc)[(4 + kk) * NT + tq];
            pg[j] += sel(tile(w, x)) + wc[MAT_F + C + cq];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    };
    auto save_pf = [&](int g) {
      if (MULTI && lead) {
        v2f *q2 = (v2f *)(pfs + ((size_t)g * C + c) * PFS_FM);
#pragma unroll
        for (int j = 0; j < LPSM; ++j) q2[j] = v2f{pf[j], pg[j]};
      }
    };
    auto load_pf = [&](int g) {
      if (MULTI) {
        const v2f *q2 = (const v2f *)(pfs + ((size_t)g * C + c) * PFS_FM);
#pragma unroll
        for (int j = 0; j < LPSM; ++j) {
          const v2f v = q2[j];
          pf[j] = v.x;
          pg[j] = v.y;
        }
      }
    };
    __syncthreads();
    if (MULTI) {
      for (int g = 0; g < G; ++g) {
        bind(g);
        precompute(a.t_begin, false);
        save_pf(g);
        __syncthreads();
      }
    } else {
      precompute(a.t_begin, false);
    }

    bool alive = true;
    for (int ts = a.t_begin; ts < a.t_end && alive; ++ts)
    for (int g = 0; g < G; ++g) {
      if (MULTI) bind(g);
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      if (wave == 0) {
        float v[GL];
        const bool ok = wait_inbox<GL>(inbox, epoch, err, v);  // the C residual granules
        if (ok) {
          cur[2 * lane] = v[0];
          cur[2 * lane + 1] = v[1];
          *(h2 *)(curh + 2 * lane) = h2{(_Float16)v[0], (_Float16)v[1]};
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      load_pf(g);
      const u64 sk_peek = lead ? peek_granule(skbox) : 0;  // skip lane: requested here, used at the hand-on
      float skipacc = 0.f;
#pragma unroll
      for (int j = 0; j < LPSM; ++j)
        if (j < nl) {
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 0, 0);
          h8 x[4];
          vec(x, curh);
          const f4v aF = tile(wF[j], x), aG = tile(wG[j], x);
          const float z = gate_fast(sel(aF) + pf[j], sel(aG) + pg[j]);
          if (lead) zbh[j * C + c] = (_Float16)z;
          const float old = cur[c];  // this layer's input: residual add below, queue push later
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 1, 0);
          lds_barrier();
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 2, 0);
          vec(x, zbh + j * C);
          const f4v aR = tile(wR[j], x);  // (the skip tile is not on the chain: after the hand-on, below)
          if (lead) {
            xs[j] = old;
            const float outv = (sel(aR) + bias_r[j]) + old;
            cur[c] = outv;
            curh[c] = (_Float16)outv;
            // the stage's last layer: hand the activation on before anything else
            if (j == nl - 1) put_granule(outbox + c, epoch, outv, fast_edge);
          }
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 3, 0);
          if (j == nl - 1) MVN_FINE(b, s, ts - a.t_begin, 5, 0);
          lds_barrier();
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 4, 0);
        }
      // ---- the skip lane, off the chain: sk' = sk + sum_j (Ws_j z_j + bs_j)
#pragma unroll
      for (int j = 0; j < LPSM; ++j)
        if (j < nl) {
          h8 x[4];
          vec(x, zbh + j * C);
          if (WS_REG) {
            skipacc += sel(tile(wS[WS_REG ? j : 0], x)) + bias_s[j];
          } else {
            h8 w[4];
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const h8 *ws8 = (const h8 *)(a.w + EMB_F + (size_t)(l0 + j) * LAYER_F + 2 * MAT_F);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = ws8[(4 + kk) * NT + tq];
            skipacc += sel(tile(w, x)) + bias_s[j];
          }
        }
      if (lead) {
        const float skin = (unsigned)(sk_peek >> 32) == epoch ? __uint_as_float((unsigned)sk_peek)
                                                              : wait_granule(skbox, epoch, err);
        put_granule(outbox + C + c, epoch, skin + skipacc, fast_edge);
      }
      MVN_STAMP(b, s, ts - a.t_begin, 1);
      if (iflag[0] == 0) {  // hand-off timed out (checked after the step: off the chain)
        alive = false;
        break;
      }
      if (ts + 1 < a.t_end) {
        precompute(ts + 1, true);
        save_pf(g);
      } else if (lead) {
        // last step of the launch: push only (the next launch pops in its prologue)
#pragma unroll
        for (int j = 0; j < LPSM; ++j)
          if (j < nl) ring[doff[j] + c + (ts & dmask[j]) * C] = xs[j];
      }
    }
    return;
  }

  if (!MFMA && s < NS - 1) {
    // ================= layer stage, dot-product form (r2 + the r3 skip lane): layers l0 .. l0+nl-1 =================
    const int l0 = s * LPS, nl = min(LPS, L - l0);
    const bool fg_group = tid < 256;
    const int t = tid & 255, c = t / KQ, kq = t % KQ;
    const bool lead = kq == 0;
    h8 *wp = (h8 *)smem_b;                                   // [LPS][2*NV][256] h8: past-tap f|g weights
    float *cur = (float *)(smem_b + LPS * MAT_H * 2);        // [C] residual stream (fp32)
    _Float16 *curh = (_Float16 *)(cur + 2 * C);              // [C] the stream as the products' operand
    // r3: the running skip sum travels on a lane of its own (as in gen_fold_kernel): the chain waits
    // for the C residual granules only -- ONE 16-byte poll per lane through poll16 -- and the lane
    // that owns a channel fetches its skip granule off the chain and adds it when the stage hands on
    const u64 *skbox = inbox + C + c;  // (rebound per turn when MULTI)
    _Float16 *zbh = curh + C;                                // [C] gated activation
    _Float16 *pasth = zbh + C;                               // [LPS][C] popped queue entries
    _Float16 *ctxh = pasth + LPS * C;                        // [C] context column
    float *ring = a.state + (size_t)b * a.state_per_seq;
    auto bind = [&](int g) {  // MULTI: the pointers of sequence b + g nb
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      skbox = inbox + C + c;
      ring = a.state + (size_t)bq * a.state_per_seq;
    };

    h8 wa[LPS][NV], wb[LPS][NV];  // FG: f_c | g_c current-tap rows; RS: res_c | skip_c
    float bias_r[LPS], bias_s[LPS], pf[LPS], pg[LPS], xs[LPS];
    int doff[LPS], dmask[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
      bias_r[j] = 0.f; bias_s[j] = 0.f; pf[j] = 0.f; pg[j] = 0.f; xs[j] = 0.f;
      doff[j] = 0; dmask[j] = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        wa[j][i] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        wb[j][i] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      if (j < nl) {
        const float *lw = a.w + EMB_F + (size_t)(l0 + j) * LAYER_F;
        const h8 *wc8 = (const h8 *)lw, *wp8 = (const h8 *)(lw + MAT_F), *wr8 = (const h8 *)(lw + 2 * MAT_F);
        if (fg_group) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            wa[j][i] = wc8[i * 256 + t];
            wb[j][i] = wc8[(NV + i) * 256 + t];
          }
#pragma unroll
          for (int i = 0; i < 2 * NV; ++i) wp[(j * 2 * NV + i) * 256 + t] = wp8[i * 256 + t];
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            wa[j][i] = wr8[i * 256 + t];
            wb[j][i] = wr8[(NV + i) * 256 + t];
          }
          bias_r[j] = lw[3 * MAT_F + c];
          bias_s[j] = lw[3 * MAT_F + C + c];
        }
        const int l = l0 + j;
        dmask[j] = (1 << (l % a.layer_size)) - 1;
        doff[j] = ring_offset(l, a.layer_size, C);
      }
    }

    // Off the critical path: push this step's layer inputs into the dilation queues, pop the
    // entries step tn needs (RS lead lanes), then the past-tap half of step tn's f/g sums
    auto precompute = [&](int tn, bool push) {
      int tq = t;
      asm volatile("" : "+v"(tq));  // addresses rebuilt per step: the chain needs the registers
      const int cq = tq / KQ, kk = tq % KQ;
      if (!fg_group && lead) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + cq;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j] : ring_load(base + (tn & dmask[j]) * C);
            pasth[j * C + cq] = (_Float16)pv;
          }
      }
      if (a.ctx_tm && fg_group && tq < C)
        ctxh[tq] = (_Float16)a.ctx_tm[(size_t)bq * a.ctx_stride_b + (size_t)tn * C + tq];
      __syncthreads();
      if (fg_group) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            const h8 *wpj = wp + j * 2 * NV * 256;
            pf[j] = pair_sum(dot_stream_h<NV>(wpj, 256, tq, pasth + j * C + KPER * kk));
            pg[j] = pair_sum(dot_stream_h<NV>(wpj + NV * 256, 256, tq, pasth + j * C + KPER * kk));
            if (a.ctx_tm) {
              // 1x1 context convs (modules.py:58-63, :75-77), weights streamed from L2
              const float *wc = a.wctx + (size_t)(l0 + j) * CTX_LAYER_F;
              pf[j] += pair_sum(dot_stream_h<NV>((const h8 *)wc, 256, tq, ctxh + KPER * kk)) + wc[MAT_F + cq];
              pg[j] += pair_sum(dot_stream_h<NV>((const h8 *)wc + NV * 256, 256, tq, ctxh + KPER * kk)) +
                       wc[MAT_F + C + cq];
            }
          }
      }
    };
    // MULTI: a sequence's pf / pg between its turns (FG lead lanes write, both lanes of a channel read)
    auto save_pf = [&](int g) {
      if (MULTI && fg_group && lead) *(f4 *)(pfs + ((size_t)g * C + c) * PFS_F) = f4{pf[0], pg[0], pf[1], pg[1]};
    };
    auto load_pf = [&](int g) {
      if (MULTI && fg_group) {
        const f4 v = *(const f4 *)(pfs + ((size_t)g * C + c) * PFS_F);
        pf[0] = v.x; pg[0] = v.y; pf[1] = v.z; pg[1] = v.w;
      }
    };
    static_assert(LPS == 2, "save_pf / load_pf move two layers' sums as one float4");
    __syncthreads();
    if (MULTI) {
      for (int g = 0; g < G; ++g) {
        bind(g);
        precompute(a.t_begin, false);
        save_pf(g);
        __syncthreads();  // the FG waves have read this sequence's popped entries: the next one's may land
      }
    } else {
      precompute(a.t_begin, false);
    }

    bool alive = true;
    for (int ts = a.t_begin; ts < a.t_end && alive; ++ts)
    for (int g = 0; g < G; ++g) {
      if (MULTI) bind(g);
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      if (wave == 0) {
        float v[GL];
        const bool ok = wait_inbox<GL>(inbox, epoch, err, v);  // the C residual granules
        if (ok) {
          cur[2 * lane] = v[0];
          cur[2 * lane + 1] = v[1];
          *(h2 *)(curh + 2 * lane) = h2{(_Float16)v[0], (_Float16)v[1]};
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      load_pf(g);
      // skip lane: the granule was sent with the residual ones; its load is issued here and used
      // when the stage hands on (the spin loop is only the fallback)
      const u64 sk_peek = (!fg_group && lead) ? peek_granule(skbox) : 0;
      float skipacc = 0.f;
#pragma unroll
      for (int j = 0; j < LPS; ++j)
        if (j < nl) {
          float old = 0.f;
          if (fg_group) {
            float f, g;
            dot2_rows<NV>(wa[j], wb[j], curh + KPER * kq, f, g);
            f = pair_sum(f) + pf[j];
            g = pair_sum(g) + pg[j];
            const float z = gate_fast(f, g);
            if (lead) zbh[c] = (_Float16)z;
          } else if (lead) {
            old = cur[c];  // this layer's input: residual add below, queue push later
          }
          lds_barrier();
          if (!fg_group) {
            float r, k;
            dot2_rows<NV>(wa[j], wb[j], zbh + KPER * kq, r, k);
            r = pair_sum(r);
            k = pair_sum(k);
            if (lead) {
              xs[j] = old;
              const float outv = (r + bias_r[j]) + old;
              cur[c] = outv;
              curh[c] = (_Float16)outv;
              skipacc += k + bias_s[j];
              if (j == nl - 1) {
                // the stage's last layer: hand the activation on before anything else
                put_granule(outbox + c, epoch, outv, fast_edge);
                const float skin = (unsigned)(sk_peek >> 32) == epoch ? __uint_as_float((unsigned)sk_peek)
                                                                      : wait_granule(skbox, epoch, err);
                put_granule(outbox + C + c, epoch, skin + skipacc, fast_edge);
              }
            }
          }
          lds_barrier();
        }
      MVN_STAMP(b, s, ts - a.t_begin, 1);
      if (iflag[0] == 0) {  // hand-off timed out (checked after the step: off the chain)
        alive = false;
        break;
      }
      if (ts + 1 < a.t_end) {
        precompute(ts + 1, true);
        save_pf(g);
      } else if (!fg_group && lead) {
        // last step of the launch: push only (the next launch pops in its prologue)
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) ring[doff[j] + c + (ts & dmask[j]) * C] = xs[j];
      }
    }
    return;
  }

  // ============================ head stage ============================
  {
    h8 *big = (h8 *)smem_b;                                   // conv1 weights [W1NV][512] h8 (64 KB)
    _Float16 *a0h = (_Float16 *)(smem_b + W1NV * NT * 16);    // [C]   lrelu(skip)
    _Float16 *a1h = a0h + C;                                  // [Q]   lrelu(conv1)
    float *lgb = (float *)(a1h + Q);                          // [Q]   logits
    const float *E0 = a.w, *E1 = E0 + Q * C;
    const float *hw = a.w + EMB_F + (size_t)L * LAYER_F;
    const h8 *W1p = (const h8 *)hw, *W2p = (const h8 *)(hw + W1_F + Q);
    const float *b1 = hw + W1_F, *b2 = hw + W1_F + Q + W2_F;
    int32_t *samples = a.samples + (size_t)b * a.stride;
    int *hidx = (int *)pfs;  // MULTI: [GMAX][2] = {idx_cur, idx_prev} of each sequence between its turns
    auto bind = [&](int g) {
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      samples = a.samples + (size_t)bq * a.stride;
    };

    // conv1: thread (o1 = tid>>1, q1 = tid&1), 64 inputs; conv2: thread (og = tid>>3, q2 = tid&7),
    // 4 outputs x 32 inputs
    const int o1 = tid >> 1, q1 = tid & 1, og = tid >> 3, q2 = tid & 7;
    for (int i = tid; i < W1NV * NT; i += NT) big[i] = W1p[i];
    h8 w2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) w2[r][i] = W2p[(r * 4 + i) * NT + tid];
    const float b1r = b1[o1];
    const float b2r = b2[4 * og + (q2 & 3)];
    __syncthreads();

    int idx_cur = 0, idx_prev = -1;
    auto send_h0 = [&](unsigned ep) {  // wave 0: granules c = residual, C + c = skip sum 0
      const int ic = min(max(idx_cur, 0), Q - 1), ip = min(idx_prev, Q - 1);
#pragma unroll
      for (int j = 0; j < C / 64; ++j) {
        const int ch = lane + 64 * j;
        float v = E1[ic * C + ch];
        if (ip >= 0) v += E0[ip * C + ch];
        put_granule(outbox + ch, ep, v, fast_edge);
        put_granule(outbox + C + ch, ep, 0.f, fast_edge);
      }
    };
    for (int g = 0; g < G; ++g) {
      if (MULTI) bind(g);
      if (wave == 0) {
        idx_cur = samples[a.t_begin];
        idx_prev = a.t_begin > 0 ? samples[a.t_begin - 1] : -1;
        if (a.t_begin < a.t_end) send_h0(1u);
        MVN_STAMP(b, s, 0, 1);
        if (MULTI && lane == 0) {
          hidx[2 * g] = idx_cur;
          hidx[2 * g + 1] = idx_prev;
        }
      }
    }
    if (MULTI) __syncthreads();

    bool alive = true;
    for (int ts = a.t_begin; ts < a.t_end && alive; ++ts)
    for (int g = 0; g < G; ++g) {
      if (MULTI) {
        bind(g);
        if (wave == 0) {  // (written by this wave's lane 0 a whole round ago)
          idx_cur = hidx[2 * g];
          idx_prev = hidx[2 * g + 1];
        }
      }
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      const int u = ts + 1;
      const bool want_out = (a.logits_out || a.choices_out) && u >= a.logits_t0;
      const bool do_head = u < a.n_total && (u >= a.n_given || want_out);  // block-uniform
      int next_idx = 0;
      // this step's Philox uniform, formed while the step's input is still on its way (the fence
      // keeps it from being sunk to its use behind the head's barriers)
      float uni = 0.f;
      if (wave == 0 && a.temperature > 0.f) {
        uni = philox_uniform(a.seed, (uint32_t)u, (uint32_t)bq);
        asm volatile("" : "+v"(uni));
      }
      if (wave == 0) {
        if (u < a.n_given) next_idx = samples[u];  // prompt / teacher forcing
        float v[GL];
        const bool ok = wait_inbox<GL>(inbox + C, epoch, err, v);  // only the skip sum feeds the head
        if (ok) *(h2 *)(a0h + 2 * lane) = h2{(_Float16)leaky(v[0]), (_Float16)leaky(v[1])};
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      if (do_head) {
        {
          float hsum = dot_stream_h<W1NV>(big, NT, tid, a0h + (C / 2) * q1);
          hsum = pair_sum(hsum);
          if (q1 == 0) a1h[o1] = (_Float16)leaky(hsum + b1r);
        }
        lds_barrier();
        {
          h8 x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = ((const h8 *)(a1h + 32 * q2))[i];
          float sv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = dot8(w2[r][0], x[0], 0.f), e1 = dot8(w2[r][1], x[1], 0.f);
            const float e2 = dot8(w2[r][2], x[2], 0.f), e3 = dot8(w2[r][3], x[3], 0.f);
            float t4 = (e0 + e2) + (e1 + e3);
            t4 = quad_sum(t4);
            sv[r] = t4 + other_quad(t4);
          }
          const int sel = q2 & 3;
          if (q2 < 4) lgb[4 * og + sel] = (sel == 0 ? sv[0] : sel == 1 ? sv[1] : sel == 2 ? sv[2] : sv[3]) + b2r;
        }
        lds_barrier();
      }
      if (wave == 0) {
        if (do_head) {
          const f4 lv = ((const f4 *)lgb)[lane];
          const float lg[4] = {lv.x, lv.y, lv.z, lv.w};
          if (a.logits_out && u >= a.logits_t0)
            ((f4 *)(a.logits_out + ((size_t)bq * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * Q))[lane] = lv;
          const int pick = choose_class(lg, a.temperature, uni, lane, Q);
          if (u >= a.n_given) next_idx = pick;
          idx_prev = idx_cur;
          idx_cur = next_idx;
          if (ts + 1 < a.t_end) send_h0(epoch + 1);
          MVN_STAMP(b, s, ts + 1 - a.t_begin, 1);
          if (lane == 0) {
            if (a.choices_out && u >= a.logits_t0) a.choices_out[(size_t)bq * a.n_total + u] = pick;
            if (u >= a.n_given) samples[u] = pick;
          }
        } else {
          idx_prev = idx_cur;
          idx_cur = next_idx;
          if (ts + 1 < a.t_end) send_h0(epoch + 1);
          MVN_STAMP(b, s, ts + 1 - a.t_begin, 1);
        }
        if (MULTI && lane == 0) {
          hidx[2 * g] = idx_cur;
          hidx[2 * g + 1] = idx_prev;
        }
      }
      if (iflag[0] == 0) {  // hand-off timed out
        alive = false;
        break;
      }
    }
  }
}

// ---- packing: state_dict layouts (fp32) -> per-thread register order, halves ------------
// each 2C x C matrix: [2*NV][t (256)] vectors of 8 halves; thread t = KQ*c + kq owns rows
// (c, C+c) x inputs k = KPER*kq + 8*(iv % NV) + e: vectors 0..NV-1 row c, NV..2NV-1 row C+c
__device__ __forceinline__ void h16_matrix_index(int h, int &row, int &k, bool mfma) {
  using namespace h16;
  const int e = h & 7, v = h >> 3;
  if (mfma) {
    // MFMA A-operand order: [tile 2][kk 4][wave 8][lane 64] vectors; lane l of wave w holds row
    // 16 w + l % 16 of the tile's half (filter | gate, residual | skip), inputs 32 kk + 8 (l / 16) + e
    const int l = v & 63, w = (v >> 6) & 7, kk = (v >> 9) & 3, t = v >> 11;
    row = t * C + 16 * w + (l & 15);
    k = 32 * kk + 8 * (l >> 4) + e;
    return;
  }
  const int t = v & 255, iv = v >> 8;
  row = (iv / NV) * C + t / KQ;
  k = KPER * (t % KQ) + 8 * (iv % NV) + e;
}
__global__ void pack_layer_h16_kernel(const float *fw, const float *gw, const float *rw, const float *rb,
                                      const float *sw, const float *sb, float *__restrict__ dst, bool mfma) {
  using namespace h16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // half index for the matrices
  _Float16 *dh = (_Float16 *)dst;
  if (i < 3 * MAT_H) {
    const int region = i / MAT_H;
    int row, k;
    h16_matrix_index(i - region * MAT_H, row, k, mfma);
    const float v = region < 2 ? fg_elem(fw, gw, C, row, region == 0 ? C + k : k)  // WC current, WP past tap
                               : rs_elem(rw, sw, C, row, k);
    dh[i] = (_Float16)v;
  } else if (i < 3 * MAT_H + 2 * C) {
    const int o = i - 3 * MAT_H;
    dst[3 * MAT_F + o] = o < C ? rb[o] : sb[o - C];
  }
}
__global__ void pack_ctx_h16_kernel(const float *wcf, const float *bcf, const float *wcg, const float *bcg,
                                    float *__restrict__ dst, bool mfma) {
  using namespace h16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  _Float16 *dh = (_Float16 *)dst;
  if (i < MAT_H) {
    int row, k;
    h16_matrix_index(i, row, k, mfma);
    dh[i] = (_Float16)(row < C ? wcf[(size_t)row * C + k] : wcg[(size_t)(row - C) * C + k]);
  } else if (i < MAT_H + 2 * C) {
    const int o = i - MAT_H;
    dst[MAT_F + o] = o < C ? bcf[o] : bcg[o - C];
  }
}
__global__ void pack_head_h16_kernel(const float *w1, const float *b1, const float *w2, const float *b2,
                                     float *__restrict__ dst) {
  using namespace h16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  _Float16 *d1 = (_Float16 *)dst, *d2 = (_Float16 *)(dst + W1_F + Q);
  if (i < Q * C) {
    // conv1: [W1NV][tid (512)] vectors; thread (o1 = tid>>1, q1 = tid&1) owns inputs 64 q1 + 8 iv + e
    const int e = i & 7, v = i >> 3, tid = v & (NT - 1), iv = v >> 9;
    d1[i] = (_Float16)w1[(size_t)(tid >> 1) * C + (C / 2) * (tid & 1) + 8 * iv + e];
  } else if (i < Q * C + Q * Q) {
    // conv2: [4 r][4 iv][tid (512)] vectors; thread (og = tid>>3, q2 = tid&7): output 4 og + r,
    // inputs 32 q2 + 8 iv + e
    const int ii = i - Q * C;
    const int e = ii & 7, v = ii >> 3, tid = v & (NT - 1), rest = v >> 9, r = rest >> 2, iv = rest & 3;
    d2[ii] = (_Float16)w2[(size_t)(4 * (tid >> 3) + r) * Q + 32 * (tid & 7) + 8 * iv + e];
  } else if (i < Q * C + Q * Q + Q) {
    const int o = i - Q * C - Q * Q;
    dst[W1_F + o] = b1[o];
    dst[W1_F + Q + W2_F + o] = b2[o];
  }
}
__global__ void pack_embed_h16_kernel(const float *__restrict__ causal_w, float *__restrict__ dst) {
  using namespace h16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= EMB_F) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, qq = r / C, c = r - qq * C;
  dst[i] = causal_w[((size_t)c * Q + qq) * 2 + tap];
}

// Which layer-stage form packs and runs: the matrix-core form unless MOVENET_H16_FORM=dot2 (the
// packed weight order differs, so pack and launch read the same switch; read once per process).
static bool h16_mfma_form() {
  static const bool on = [] {
    const char *e = getenv("MOVENET_H16_FORM");
    return !(e && e[0] == 'd');
  }();
  return on;
}

bool pipe_h16_ok(const mvn_dims *d) {
  return d->residual_channels == 128 && d->skip_channels == 128 && d->input_channels == 256 &&
         n_layers(d) >= 1;
}
// stages per pipeline for SIZING (hand-off area, co-residency): the two-layer forms' count, the larger one
int pipe_h16_stages(const mvn_dims *d) { return (n_layers(d) + h16::LPS - 1) / h16::LPS + 1; }
int pipe_h16_pipelines(const mvn_dims *d) {  // co-resident pipelines
  const int NS = pipe_h16_stages(d);
  return NS <= PIPE_XCD_CUS ? 8 * (PIPE_XCD_CUS / NS) : 8 / ((NS + PIPE_XCD_CUS - 1) / PIPE_XCD_CUS);
}
int pipe_h16_max_batch(const mvn_dims *d) { return h16::GMAX * pipe_h16_pipelines(d); }  // GMAX sequences each
size_t pipe_h16_weights_floats(const mvn_dims *d) {
  return (size_t)h16::EMB_F + (size_t)n_layers(d) * h16::LAYER_F + h16::HEAD_F;
}
size_t pipe_h16_ctx_layer_floats() { return h16::CTX_LAYER_F; }

int pipe_h16_pack(const mvn_dims *d, const mvn_params *p, float *packed, bool has_ctx, hipStream_t s) {
  using namespace h16;
  const int L = n_layers(d);
  hipLaunchKernelGGL(pack_embed_h16_kernel, dim3((EMB_F + 255) / 256), dim3(256), 0, s, p->causal_w, packed);
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL(pack_layer_h16_kernel, dim3((3 * MAT_H + 2 * C + 255) / 256), dim3(256), 0, s,
                       p->filter_w[l], p->gate_w[l], p->residual_w[l], p->residual_b[l], p->skip_w[l],
                       p->skip_b[l], packed + EMB_F + (size_t)l * LAYER_F, h16_mfma_form());
  float *head = packed + EMB_F + (size_t)L * LAYER_F;
  hipLaunchKernelGGL(pack_head_h16_kernel, dim3((Q * C + Q * Q + Q + 255) / 256), dim3(256), 0, s, p->head1_w,
                     p->head1_b, p->head2_w, p->head2_b, head);
  if (has_ctx) {
    float *ctx = packed + pipe_h16_weights_floats(d);
    for (int l = 0; l < L; ++l)
      hipLaunchKernelGGL(pack_ctx_h16_kernel, dim3((MAT_H + 2 * C + 255) / 256), dim3(256), 0, s,
                         p->ctx_filter_w[l], p->ctx_filter_b[l], p->ctx_gate_w[l], p->ctx_gate_b[l],
                         ctx + (size_t)l * CTX_LAYER_F, h16_mfma_form());
  }
  return check_hip(hipGetLastError(), "pipe_h16_pack");
}

// `hand`: the hand-off area of the generator state (gen_common.h: hand_status_offset): this
// variant uses the first batch * NS inboxes and placement words of it.
int pipe_h16_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_total,
                    size_t status_off, hipStream_t s) {
  using namespace h16;
  const int pipes = std::min(batch, pipe_h16_pipelines(d));
  const bool multi = batch > pipes;
  int dev = 0, cus = 0, per_cu = 0, coop = 0;
  const bool mm = h16_mfma_form();
  // (a pipeline that serves one sequence runs three layers per stage in the matrix-core form)
  const int lps = mm ? (multi ? LpsM<true>::value : LpsM<false>::value) : LPS;
  int NS = (n_layers(d) + lps - 1) / lps + 1;
  const void *fn = multi ? (mm ? (const void *)gen_pipe_h16_kernel<true, true> : (const void *)gen_pipe_h16_kernel<true, false>)
                         : (mm ? (const void *)gen_pipe_h16_kernel<false, true> : (const void *)gen_pipe_h16_kernel<false, false>);
  const int lds_bytes = mm ? (multi ? LDS_BYTES_M_MULTI : LDS_BYTES_M) : (multi ? LDS_BYTES_MULTI : LDS_BYTES);
  int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(gen_pipe_h16)");
  if (rc) return rc;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice") ||
      check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev),
                "hipDeviceGetAttribute(CUs)") ||
      check_hip(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev),
                "hipDeviceGetAttribute(cooperative)") ||
      check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NT, lds_bytes),
                "hipOccupancyMaxActiveBlocksPerMultiprocessor(gen_pipe_h16)"))
    return MVN_ERR_LAUNCH;
  const int XS = (NS + PIPE_XCD_CUS - 1) / PIPE_XCD_CUS;
  const int slots = NS <= PIPE_XCD_CUS ? (pipes + 7) / 8 * NS : (NS + XS - 1) / XS;
  if (cus < 8 * PIPE_XCD_CUS || batch > pipe_h16_max_batch(d) || per_cu < 1 || slots * 8 > per_cu * cus) {
    set_error("PIPE_F16 variant: %d stages per pipeline, %d pipelines of at most %d sequences each on %d CUs "
              "(batch %d asked for)", NS, cus < 8 * PIPE_XCD_CUS ? 0 : pipe_h16_pipelines(d), GMAX, cus, batch);
    return MVN_ERR_UNSUPPORTED;
  }
  if ((size_t)batch * NS * GRAN * 2 > status_off || status_off + 16 + (size_t)batch * NS > hand_total) {
    set_error("PIPE_F16 variant: hand-off area too small");
    return MVN_ERR_BAD_ARG;
  }
  unsigned *err = (unsigned *)(hand + status_off);
  const size_t tail_floats = hand_total - status_off - 16;
  rc = check_hip(hipMemsetAsync(hand, 0, (size_t)batch * NS * GRAN * 2 * sizeof(float), s),
                 "hipMemsetAsync(granules)");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(err + 16, 0, tail_floats * sizeof(float), s), "hipMemsetAsync(placement words)");
  if (rc) return rc;
  u64 *gran = (u64 *)hand;
  GenArgs args = a;
  int nb = pipes, nseq = batch;
  if (coop && pipe_cooperative_launch()) {
    void *kargs[] = {(void *)&args, (void *)&gran, (void *)&err, (void *)&NS, (void *)&nb, (void *)&nseq};
    return check_hip(hipLaunchCooperativeKernel(fn, dim3(slots * 8), dim3(NT), kargs, (unsigned)lds_bytes, s),
                     "mvn_generate(pipe_f16, cooperative launch)");
  }
  void *kargs[] = {(void *)&args, (void *)&gran, (void *)&err, (void *)&NS, (void *)&nb, (void *)&nseq};
  return check_hip(hipLaunchKernel(fn, dim3(slots * 8), dim3(NT), kargs, (size_t)lds_bytes, s), "mvn_generate(pipe_f16)");
}

}  // namespace mvn

#ifdef MVN_PIPE_STAMPS
extern "C" int mvn_debug_read_stamps_h16(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_stamps) / 8) n = sizeof(mvn::g_stamps) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_stamps), n * 8), "read stamps");
}
extern "C" int mvn_debug_read_fine_h16(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_fine) / 8) n = sizeof(mvn::g_fine) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_fine), n * 8), "read fine");
}
#endif
