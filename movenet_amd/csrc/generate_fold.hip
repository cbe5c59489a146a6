// FOLD variant of the pipelined generator (C = K = 64, Q = 256): the dependent chain of a
// layer is ONE mat-vec + gate + LDS exchange instead of two.
//
// In the PIPE kernel (generate_pipe.hip) a layer costs two dependent phases: z = gate(Wc x + ..)
// and then x' = x + Wr z + br, each with its own LDS exchange and barrier (1244 cycles per layer,
// DESIGN.md 4.1).  Here the residual 1x1 of layer j is folded into the current-tap filter/gate
// matrix of layer j+1 (reference arithmetic: movenet/modules.py:67-93 -- f,g = conv(x), x' = x +
// conv_residual(z) -- restated for the product of the two linear maps):
//
//     fg_{j+1} = Wc_{j+1} x_{j+1} + past-tap part
//              = (Wc_{j+1} Wr_j) z_j + [ Wc_{j+1} x_j + Wc_{j+1} br_j + past-tap part ]
//                `-- critical: one mat-vec on z_j --'  `-- known one phase earlier: helper waves --'
//
// A stage holds THREE layers (10 layer stages + the head = 11 workgroups per sequence).  Its
// inbox carries 3C granules {xp, zl, sk}: xp = the stream WITHOUT the previous stage's last
// residual term, zl = that stage's last gated activation -- the two the chain waits for -- and
// sk = the running skip sum, which travels on a lane of its own (sent and awaited off the chain,
// after the stage's third phase).  Waves 0-3 (the chain) compute
//     z_0 = gate(Wc_0 xp + (Wc_0 Wr_prev) zl + pf_0)
//     z_1 = gate((Wc_1 Wr_0) z_0 + base_1 + pf_1)          and (Wc_2 Wr_0) z_0 for the helpers' base_2
//     z_2 = gate((Wc_2 Wr_1) z_1 + base_2 + pf_2)          -> sent on as the next stage's zl
// with their five matrices in registers; waves 4-7 (the helpers) compute one phase ahead
//     base_1 = Wc_1 xp + (Wc_1 Wr_prev) zl,    x_0 = xp + Wr_prev zl
//     base_2 = Wc_2 x_0  [+ the chain's term],  x_1 = x_0 + Wr_0 z_0 + br_0
//                                               x_2 = x_1 + Wr_1 z_1 + br_1   -> xp' = x_2 + br_2 sent early
// and, after the third phase, sk' = sk + Ws_prev zl + Ws_0 z_0 + Ws_1 z_1 + biases (the skip 1x1
// of a stage's LAST layer is added by the next stage, the head for the last one).  The constant
// vectors Wc_1 br_0 and Wc_2 (br_0 + br_1) are packed once and ride in pf_1 / pf_2.  The
// explicit x_0, x_1, x_2 feed the dilation queues off the critical path, exactly as in PIPE.
//
// The products Wc Wr are formed once at pack time (fp64 sums, rounded to fp32).  The folded
// form is the same real-number function as the reference's; its fp32 rounding differs from
// the layer-by-layer order at the 1e-7 level (logits within 2e-5 of the range, greedy
// fixtures bit-exact: tests/test_generate_gpu.py).
#include <cstdlib>

#include "common.h"
#include "gen_common.h"
#include "pipe_common.h"

namespace mvn {

namespace fold {
constexpr int C = 64, Q = 256, NT = 512, LPS = 3;
constexpr int KQ = 4, KPER = 16, NF4 = 4;  // thread 4c + kq owns rows (c, C + c) x 16 inputs
constexpr int MAT_F = 2 * C * C;           // one 2C x C matrix (or two C x C ones) in the per-thread order
// a stage section: 14 matrices, then the vectors
enum { M_A0 = 0, M_A0P, M_A1, M_A2, M_B2P,       // chain waves: registers
       M_B1, M_B1P, M_B2, M_RR, M_RS,            // helper waves: registers; RR = {Wr_prev; Wr_0}, RS = {Wr_1; Ws_1}
       M_WP0, M_WP1, M_WP2, M_SS,                // LDS: past taps of the three layers, SS = {Ws_prev; Ws_0}
       N_MAT };
constexpr int N_LDS_MAT = 4;
// vectors: cb1[2C] = Wc_1 br_0, cb2[2C] = Wc_2 (br_0 + br_1), then bs_prev, br_0, bs_0, br_1, bs_1, br_2 [C each]
constexpr int V_CB1 = 0, V_CB2 = 2 * C, V_BSP = 4 * C, V_BR0 = 5 * C, V_BS0 = 6 * C, V_BR1 = 7 * C,
              V_BS1 = 8 * C, V_BR2 = 9 * C, VEC_F = 10 * C;
constexpr int STAGE_F = N_MAT * MAT_F + VEC_F;
constexpr int CTX_LAYER_F = MAT_F + 2 * C;  // as in the PIPE variant
constexpr int EMB_F = 2 * Q * C;
constexpr int W1_F = Q * C, W2_F = Q * Q, WSL_F = C * C;
constexpr int HEAD_F = W1_F + Q + W2_F + Q + WSL_F + C;
constexpr int GRAN = 3 * C;                 // granules per inbox: xp | zl | sk
// LDS floats: layer stage matrices + vectors; head stage the embedding tables (32768) + vectors
constexpr int LDS_FLOATS = N_LDS_MAT * MAT_F + 1536 + 16;
// r3, several sequences per pipeline ("rounds"): a stage is busy ~1.1 us of a 15 us step, so a
// pipeline serves up to GMAX sequences in turn -- weights shared, one inbox per sequence and
// stage, the LDS step vectors shared (they live within one turn).  What must survive from a
// sequence's turn to its next is the past-tap part of its f/g sums (pf, pg: six floats per
// channel), kept in LDS behind the vectors; the head keeps two class indices per sequence.
constexpr int GMAX = 8;
constexpr int PFS_F = 8;                    // floats per channel and sequence: pf0 pg0 pf1 pg1 | pf2 pg2 - -
constexpr int LDS_FLOATS_MULTI = LDS_FLOATS + GMAX * C * PFS_F;
}  // namespace fold

// A thread's share of a matrix: 32 floats = 8 float4 of the packed order.  Two formats:
//  * PAIR (the 2C x C matrices): w[k] = {row c, row C + c} at input k of the thread's 16: one
//    packed FMA against {x_k, x_k} (an op_sel broadcast) advances BOTH rows, the four
//    accumulators end as {f, g} -- 16 FMAs + 3 packed adds for the two rows, where separate
//    rows cost 16 + 6 + 2 (the kernel is bound by VALU issue: scripts/probes/issue_rates.hip)
//  * SPLIT (two unrelated C x C matrices): a() = w[0..8) is row c of the first against
//    inputs {2i, 2i + 1}, b() = w[8..16) the same row of the second
struct FoldMat {
  v2f w[16];
};
__device__ __forceinline__ void fold_load(FoldMat &m, const float *sec, int t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f4 v = ((const f4 *)sec)[i * 256 + t];
    m.w[2 * i] = v2f{v.x, v.y};
    m.w[2 * i + 1] = v2f{v.z, v.w};
  }
}
// acc (+)= w * {x.y, x.y}: the compiler finds the op_sel form for the high element of a
// register pair only every other time (it copies .w to a fresh register first), so it is spelled out
__device__ __forceinline__ v2f pk_fma_hi(v2f w, v2f xpair, v2f acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w), "v"(xpair));
  return acc;
}
__device__ __forceinline__ v2f pk_mul_hi(v2f w, v2f xpair) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(w), "v"(xpair));
  return r;
}
// PAIR format: acc[j] (+)= w[4i + j] * {x[i][j], x[i][j]}
template <bool INIT>
__device__ __forceinline__ void pair_acc(v2f (&acc)[4], const v2f (&w)[16], const f4 (&x)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (INIT && i == 0) {
      acc[0] = w[0] * v2f{x[0].x, x[0].x};
      acc[1] = w[1] * v2f{x[0].y, x[0].y};
      acc[2] = w[2] * v2f{x[0].z, x[0].z};
      acc[3] = pk_mul_hi(w[3], v2f{x[0].z, x[0].w});
    } else {
      acc[0] = __builtin_elementwise_fma(w[4 * i], v2f{x[i].x, x[i].x}, acc[0]);
      acc[1] = __builtin_elementwise_fma(w[4 * i + 1], v2f{x[i].y, x[i].y}, acc[1]);
      acc[2] = __builtin_elementwise_fma(w[4 * i + 2], v2f{x[i].z, x[i].z}, acc[2]);
      acc[3] = pk_fma_hi(w[4 * i + 3], v2f{x[i].z, x[i].w}, acc[3]);
    }
  }
}
__device__ __forceinline__ v2f pair_sum(const v2f (&acc)[4]) { return (acc[0] + acc[2]) + (acc[1] + acc[3]); }
// SPLIT format, half H: one row against the 16 inputs
template <int H>
__device__ __forceinline__ float split_dot(const v2f (&w)[16], const f4 (&x)[4]) {
  v2f a0 = w[8 * H] * v2f{x[0].x, x[0].y}, a1 = w[8 * H + 1] * v2f{x[0].z, x[0].w};
  v2f a2 = w[8 * H + 2] * v2f{x[1].x, x[1].y}, a3 = w[8 * H + 3] * v2f{x[1].z, x[1].w};
  a0 = __builtin_elementwise_fma(w[8 * H + 4], v2f{x[2].x, x[2].y}, a0);
  a1 = __builtin_elementwise_fma(w[8 * H + 5], v2f{x[2].z, x[2].w}, a1);
  a2 = __builtin_elementwise_fma(w[8 * H + 6], v2f{x[3].x, x[3].y}, a2);
  a3 = __builtin_elementwise_fma(w[8 * H + 7], v2f{x[3].z, x[3].w}, a3);
  const v2f t = (a0 + a2) + (a1 + a3);
  return t.x + t.y;
}

// LDS accesses by byte address: base register + immediate (an address formed from the dynamic
// LDS symbol is a register per address; 20 of them spilled the chain's weights)
typedef __attribute__((address_space(3))) float lds_f;
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) vf4 lds_f4;
__device__ __forceinline__ f4 lds_load4(unsigned addr) {
  const vf4 v = *(const lds_f4 *)(uintptr_t)addr;
  f4 r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const float *p) { return (unsigned)(uintptr_t)(lds_f *)p; }
#define LDSF(addr, off) (*(lds_f *)(uintptr_t)((addr) + 4u * (unsigned)(off)))
template <int N4>
__device__ __forceinline__ void ldsv(f4 (&x)[N4], unsigned addr, int off) {
#pragma unroll
  for (int i = 0; i < N4; ++i) x[i] = lds_load4(addr + 4u * (unsigned)off + 16u * i);
}

// PAIR-format matrix streamed from LDS (eight float4 at waddr + 4096 i) against 16 inputs at xaddr
__device__ __forceinline__ v2f pair_dot_lds(unsigned waddr, unsigned xaddr) {
  v2f acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f4 wa = lds_load4(waddr + 4096u * (2 * i)), wb = lds_load4(waddr + 4096u * (2 * i + 1));
    const f4 x = lds_load4(xaddr + 16u * i);
    if (i == 0) {
      acc[0] = v2f{wa.x, wa.y} * v2f{x.x, x.x};
      acc[1] = v2f{wa.z, wa.w} * v2f{x.y, x.y};
      acc[2] = v2f{wb.x, wb.y} * v2f{x.z, x.z};
      acc[3] = pk_mul_hi(v2f{wb.z, wb.w}, v2f{x.z, x.w});
    } else {
      acc[0] = __builtin_elementwise_fma(v2f{wa.x, wa.y}, v2f{x.x, x.x}, acc[0]);
      acc[1] = __builtin_elementwise_fma(v2f{wa.z, wa.w}, v2f{x.y, x.y}, acc[1]);
      acc[2] = __builtin_elementwise_fma(v2f{wb.x, wb.y}, v2f{x.z, x.z}, acc[2]);
      acc[3] = pk_fma_hi(v2f{wb.z, wb.w}, v2f{x.z, x.w}, acc[3]);
    }
  }
  return pair_sum(acc);
}
// SPLIT-format half H of a matrix streamed from LDS
template <int H>
__device__ __forceinline__ float split_dot_lds(unsigned waddr, unsigned xaddr) {
  f4 w[4], x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[i] = lds_load4(waddr + 4096u * (4 * H + i));
    x[i] = lds_load4(xaddr + 16u * i);
  }
  v2f a0 = v2f{w[0].x, w[0].y} * v2f{x[0].x, x[0].y}, a1 = v2f{w[0].z, w[0].w} * v2f{x[0].z, x[0].w};
  v2f a2 = v2f{w[1].x, w[1].y} * v2f{x[1].x, x[1].y}, a3 = v2f{w[1].z, w[1].w} * v2f{x[1].z, x[1].w};
  a0 = __builtin_elementwise_fma(v2f{w[2].x, w[2].y}, v2f{x[2].x, x[2].y}, a0);
  a1 = __builtin_elementwise_fma(v2f{w[2].z, w[2].w}, v2f{x[2].z, x[2].w}, a1);
  a2 = __builtin_elementwise_fma(v2f{w[3].x, w[3].y}, v2f{x[3].x, x[3].y}, a2);
  a3 = __builtin_elementwise_fma(v2f{w[3].z, w[3].w}, v2f{x[3].z, x[3].w}, a3);
  const v2f t = (a0 + a2) + (a1 + a3);
  return t.x + t.y;
}

// MULTI = false: one sequence per pipeline (nseq == nb), the form every config-2 figure up to
// batch 16 is measured on.  MULTI = true: pipeline b serves sequences b, b + nb, b + 2 nb, ... < nseq.
template <bool MULTI>
__global__ __launch_bounds__(512, 2) void gen_fold_kernel(GenArgs a, u64 *hand, unsigned *err, int NS, int nb, int nseq) {
  using namespace fold;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (xcd, slot) -> (pipeline, stage): a pipeline of NS <= 32 stages sits in one XCD (speed only;
  // every edge verifies its own placement below).  The CUs an XCD has left over behind its whole pipelines
  // (config 2: 32 - 2 x 11 = 10) form further pipelines ACROSS XCDs (80 CUs: seven more) -- slower hops, which
  // does not matter where they are used: launches whose step is bound by the stages' service time per turn.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int in_xcd = (PIPE_XCD_CUS / NS) * NS;  // slots of an XCD's whole pipelines
  int b, s;
  if (slot < in_xcd) {
    b = xcd + 8 * (slot / NS);
    s = slot % NS;
  } else {
    const int idx = xcd * (PIPE_XCD_CUS - in_xcd) + (slot - in_xcd);
    b = 8 * (PIPE_XCD_CUS / NS) + idx / NS;
    s = idx % NS;
  }
  if (b >= nb) return;
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;  // sticky status
  const int L = a.L;
  const int s_next = s + 1 == NS ? 0 : s + 1;
  const int G = MULTI ? (nseq - b + nb - 1) / nb : 1;  // sequences of this pipeline
  int bq = b;                                           // the sequence whose turn it is
  u64 *inbox = hand + ((size_t)b * NS + s) * GRAN;
  u64 *outbox = hand + ((size_t)b * NS + s_next) * GRAN;
  int *iflag = (int *)(smem + LDS_FLOATS - 16);  // [0] ok flag, [3] fast-edge flag
  float *pfs = smem + LDS_FLOATS;                // MULTI: [GMAX][C][PFS_F] (layer stages), head: indices
  bool fast_edge = false;
  {
    unsigned *xcc = err + 16;  // [nb * NS] words, zeroed by the launch's memset
    const unsigned mine = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1;  // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
      __hip_atomic_store(xcc + b * NS + s, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned other = 0;
      for (unsigned spins = 0; spins < (1u << 20) && other == 0; ++spins) {
        other = __hip_atomic_load(xcc + b * NS + s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other == 0) __builtin_amdgcn_s_sleep(8);
      }
      iflag[3] = (other == mine) ? 1 : 0;  // unknown (time-out) => the safe form
    }
    __syncthreads();
    fast_edge = iflag[3] != 0;
    __syncthreads();
  }

  if (s < NS - 1) {
    // ================= layer stage: layers l0 .. l0 + nl - 1 =================
    const int l0 = s * LPS, nl = min(LPS, L - l0);
    const bool chain = tid < 256;
    const int t = tid & 255, c = t >> 2, kq = t & 3;
    const bool lead = kq == 0;
    float *lmat = smem;                  // WP0 | WP1 | WP2 | SS
    // LDS vectors, addressed as constant offsets from two per-lane bases (kept opaque so that
    // every access is base register + immediate instead of a register per address)
    constexpr int O_XP = 0;              // [C] inbox: stream without the last residual term
    constexpr int O_ZL = C;              // [C] inbox: previous stage's last gated activation
    constexpr int O_Z0 = 2 * C, O_Z1 = 3 * C;
    constexpr int O_X0 = 4 * C;          // [C] layer 0's input, explicit
    constexpr int O_B1 = 5 * C, O_B2 = 7 * C;  // [2C] each, helper -> chain
    constexpr int O_PAST = 9 * C;        // [LPS][C] popped queue entries
    constexpr int O_CTX = 12 * C;        // [C] context vector of the step being prepared
    constexpr int O_VEC = 13 * C;        // [VEC_F] the stage's constant vectors
    float *vec = smem + N_LDS_MAT * MAT_F;
    unsigned vc = lds_addr(vec) + 4u * c;          // this lane's channel
    unsigned vq = lds_addr(vec) + 4u * KPER * kq;  // this lane's 16 inputs
    asm volatile("" : "+v"(vc), "+v"(vq));
    u64 *ob = outbox + c;                // granules of this lane's channel: xp' | zl' (+ C) | sk' (+ 2C)
    const u64 *ibs = inbox + 2 * C + c;
    float *ring = a.state + (size_t)b * a.state_per_seq;
    auto bind = [&](int g) {  // MULTI: the pointers of sequence b + g nb
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      ob = outbox + c;
      ibs = inbox + 2 * C + c;
      ring = a.state + (size_t)bq * a.state_per_seq;
    };
    const float *sec = a.w + EMB_F + (size_t)s * STAGE_F;
    const float *vecs = sec + N_MAT * MAT_F;

    // chain waves: A0, A0', A1, A2, B2'; helper waves: B1, B1', B2, RR, RS
    FoldMat m0, m1, m2, m3, m4;
    {
      const int first = chain ? M_A0 : M_B1;
      fold_load(m0, sec + (size_t)(first + 0) * MAT_F, t);
      fold_load(m1, sec + (size_t)(first + 1) * MAT_F, t);
      fold_load(m2, sec + (size_t)(first + 2) * MAT_F, t);
      fold_load(m3, sec + (size_t)(first + 3) * MAT_F, t);
      fold_load(m4, sec + (size_t)(first + 4) * MAT_F, t);
      const f4 *src = (const f4 *)(sec + (size_t)M_WP0 * MAT_F);
      for (int i = tid; i < N_LDS_MAT * MAT_F / 4; i += NT) ((f4 *)lmat)[i] = src[i];
    }
    // (the two groups never need each other's per-lane values: one set of registers serves
    // both -- chain: pf/pg of the three layers and the four folded-bias constants; helpers:
    // the explicit inputs xs of the three layers (queue pushes) and the six biases)
    float pf[LPS], pg[LPS];
    float (&xs)[LPS] = pf;
    int doff[LPS], dmask[LPS];
    // the stage's bias vectors live in LDS behind the step vectors (read where needed: the six
    // per-lane copies cost the chain its registers)
    for (int i = tid; i < VEC_F; i += NT) vec[O_VEC + i] = vecs[i];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
      pf[j] = 0.f; pg[j] = 0.f;
      const int l = l0 + j;
      dmask[j] = j < nl ? (1 << (l % a.layer_size)) - 1 : 0;
      doff[j] = j < nl ? ring_offset(l, a.layer_size, C) : 0;
    }

    // Off the critical path: push this step's layer inputs into the dilation queues, pop the
    // entries step tn needs (helper lead lanes), then the past-tap half of step tn's f/g sums
    // (chain waves, matrices streamed from LDS).  Addresses are rebuilt here behind an
    // optimisation fence (the chain needs the registers, this code has slack).
    // MULTI: a turn's service time is what bounds the step beyond ~5 sequences per pipeline, and
    // most of it was the round trip of these pops (128 sequences' queues are 100 MB: MALL / HBM,
    // not L2).  The entries step tn needs were pushed at least two steps ago (dilation 1 takes
    // this step's value instead), so they are requested at the TOP of the turn and arrive under
    // the wait for the inbox and the chain.
    float popv[LPS] = {0.f, 0.f, 0.f};
    auto prefetch_pops = [&](int tn) {
      if (MULTI && !chain && lead) {
        int cq = c;
        asm volatile("" : "+v"(cq));
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) popv[j] = ring_load(ring + doff[j] + cq + (tn & dmask[j]) * C);
      }
    };
    auto precompute = [&](int tn, bool push, bool popped) {
      int tq = t;
      asm volatile("" : "+v"(tq));
      const int cq = tq >> 2, kk = tq & 3;
      if (!chain && lead) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + cq;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j]
                             : (MULTI && popped) ? popv[j] : ring_load(base + (tn & dmask[j]) * C);
            vec[O_PAST + j * C + cq] = pv;
          }
      }
      if (a.ctx_tm && chain && tq < C) vec[O_CTX + tq] = a.ctx_tm[(size_t)bq * a.ctx_stride_b + (size_t)tn * C + tq];
      __syncthreads();
      if (chain) {
        const unsigned wm = lds_addr(lmat) + 16u * tq, xq = lds_addr(vec) + 4u * (O_PAST + KPER * kk);
        const v2f p0 = pair_dot_lds(wm, xq);
        const v2f p1 = pair_dot_lds(wm + 4u * MAT_F, xq + 4u * C);
        const v2f p2 = pair_dot_lds(wm + 8u * MAT_F, xq + 8u * C);
        pf[0] = chan_sum<KQ>(p0.x);
        pg[0] = chan_sum<KQ>(p0.y);
        pf[1] = chan_sum<KQ>(p1.x) + vec[O_VEC + V_CB1 + cq];
        pg[1] = chan_sum<KQ>(p1.y) + vec[O_VEC + V_CB1 + C + cq];
        pf[2] = chan_sum<KQ>(p2.x) + vec[O_VEC + V_CB2 + cq];
        pg[2] = chan_sum<KQ>(p2.y) + vec[O_VEC + V_CB2 + C + cq];
        if (a.ctx_tm) {
          // 1x1 context convs (modules.py:58-63, :75-77), weights streamed from L2
#pragma unroll
          for (int j = 0; j < LPS; ++j)
            if (j < nl) {
              const float *wc = a.wctx + (size_t)(l0 + j) * CTX_LAYER_F;
              pf[j] += chan_sum<KQ>(dot_stream<NF4>((const f4 *)wc, 256, tq, vec + O_CTX + KPER * kk)) + wc[MAT_F + cq];
              pg[j] += chan_sum<KQ>(dot_stream<NF4>((const f4 *)wc + NF4 * 256, 256, tq, vec + O_CTX + KPER * kk)) +
                       wc[MAT_F + C + cq];
            }
        }
      }
    };
    // MULTI: a sequence's pf / pg between its turns (lead lanes of the chain waves write, the four
    // lanes of a channel read the same 32 bytes)
    auto save_pf = [&](int g) {
      if (MULTI && chain && lead) {
        float *q = pfs + ((size_t)g * C + c) * PFS_F;
        *(f4 *)q = f4{pf[0], pg[0], pf[1], pg[1]};
        *(v2f *)(q + 4) = v2f{pf[2], pg[2]};
      }
    };
    auto load_pf = [&](int g) {
      if (MULTI && chain) {
        const float *q = pfs + ((size_t)g * C + c) * PFS_F;
        const f4 v = *(const f4 *)q;
        const v2f w = *(const v2f *)(q + 4);
        pf[0] = v.x; pg[0] = v.y; pf[1] = v.z; pg[1] = v.w; pf[2] = w.x; pg[2] = w.y;
      }
    };
    __syncthreads();
    if (MULTI) {
      for (int g = 0; g < G; ++g) {
        bind(g);
        precompute(a.t_begin, false, false);
        save_pf(g);
        __syncthreads();  // the chain waves have read this sequence's popped entries: the next one's may land
      }
    } else {
      precompute(a.t_begin, false, false);
    }

    bool alive = true;
    for (int ts = a.t_begin; ts < a.t_end && alive; ++ts)
    for (int g = 0; g < G; ++g) {
      if (MULTI) {
        bind(g);
        if (ts + 1 < a.t_end) prefetch_pops(ts + 1);
      }
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      if (wave == 0) {
        // 2C granules: lane i takes 2i, 2i + 1 (xp for i < 32, zl above) with one 16-byte load
        float v[2];
        const bool ok = wait_inbox<2>(inbox, epoch, err, v);
        if (ok) {
          float *dst = vec + 2 * lane;  // O_XP and O_ZL are adjacent
          dst[0] = v[0];
          dst[1] = v[1];
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      load_pf(g);
      float xv = 0.f;  // helper lead lanes: the explicit stream
      MVN_FINE(b, s, ts - a.t_begin, 0, 0);
      // ---- phase 0
      if (chain) {
        f4 xa[NF4], xb[NF4];
        ldsv<NF4>(xa, vq, O_XP);
        ldsv<NF4>(xb, vq, O_ZL);
        v2f acc[4];
        pair_acc<true>(acc, m0.w, xa);
        pair_acc<false>(acc, m1.w, xb);
        const v2f fg = pair_sum(acc);
        const float f = chan_sum<KQ>(fg.x) + pf[0];
        const float g = chan_sum<KQ>(fg.y) + pg[0];
        const float z = gate_fast(f, g);
        if (lead) LDSF(vc, O_Z0) = z;
        MVN_FINE(b, s, ts - a.t_begin, 1, 0);
      } else {
        f4 xa[NF4], xb[NF4];
        ldsv<NF4>(xa, vq, O_XP);
        ldsv<NF4>(xb, vq, O_ZL);
        v2f acc[4];
        pair_acc<true>(acc, m0.w, xa);
        pair_acc<false>(acc, m1.w, xb);
        const v2f fg = pair_sum(acc);
        float r = split_dot<0>(m3.w, xb);                          // Wr_prev zl
        const float f = chan_sum<KQ>(fg.x);
        const float g = chan_sum<KQ>(fg.y);
        r = chan_sum<KQ>(r);
        if (lead) {
          xv = LDSF(vc, O_XP) + r;               // x_0 (br_prev already inside xp)
          LDSF(vc, O_X0) = xv;
          xs[0] = xv;
          LDSF(vc, O_B1) = f;
          LDSF(vc, O_B1 + C) = g;
        }
        MVN_FINE(b, s, ts - a.t_begin, 6, 256);
      }
      lds_barrier();
      MVN_FINE(b, s, ts - a.t_begin, 2, 0);
      // ---- phase 1
      v2f b2 = {0.f, 0.f};  // chain: (Wc_2 Wr_0) z_0, this thread's 16 inputs
      if (chain) {
        f4 xz[NF4];
        ldsv<NF4>(xz, vq, O_Z0);
        v2f acc[4], acc2[4];
        pair_acc<true>(acc, m2.w, xz);
        pair_acc<true>(acc2, m4.w, xz);  // (same block as the first: the broadcasts fold into op_sel)
        const v2f fg = pair_sum(acc);
        b2 = pair_sum(acc2);
        asm volatile("" : "+v"(b2));  // not to be sunk past the gate (into a block where the broadcasts become moves)
        const float f = chan_sum<KQ>(fg.x) + (LDSF(vc, O_B1) + pf[1]);
        const float g = chan_sum<KQ>(fg.y) + (LDSF(vc, O_B1 + C) + pg[1]);
        const float z = gate_fast(f, g);
        if (lead) LDSF(vc, O_Z1) = z;
        MVN_FINE(b, s, ts - a.t_begin, 3, 0);
      } else {
        f4 xa[NF4], xz[NF4];
        ldsv<NF4>(xa, vq, O_X0);
        ldsv<NF4>(xz, vq, O_Z0);
        const float br0 = LDSF(vc, O_VEC + V_BR0);
        v2f acc[4];
        pair_acc<true>(acc, m2.w, xa);                             // Wc_2 x_0
        const v2f fg = pair_sum(acc);
        float r = split_dot<1>(m3.w, xz);                          // Wr_0 z_0
        const float f = chan_sum<KQ>(fg.x);
        const float g = chan_sum<KQ>(fg.y);
        r = chan_sum<KQ>(r);
        if (lead) {
          LDSF(vc, O_B2) = f;
          LDSF(vc, O_B2 + C) = g;
          xv = (xv + br0) + r;           // x_1
          xs[1] = xv;
        }
        MVN_FINE(b, s, ts - a.t_begin, 7, 256);
      }
      lds_barrier();
      MVN_FINE(b, s, ts - a.t_begin, 4, 0);
      // ---- phase 2: the chain's z_2 and the helpers' xp' leave as granules; the skip sum follows
      if (chain) {
        f4 xz[NF4];
        ldsv<NF4>(xz, vq, O_Z1);
        v2f acc[4];
        pair_acc<true>(acc, m3.w, xz);
        const v2f fg = pair_sum(acc) + b2;
        const float f = chan_sum<KQ>(fg.x) + (LDSF(vc, O_B2) + pf[2]);
        const float g = chan_sum<KQ>(fg.y) + (LDSF(vc, O_B2 + C) + pg[2]);
        const float z = gate_fast(f, g);
        if (lead) put_granule(ob + C, epoch, z, fast_edge);
        MVN_FINE(b, s, ts - a.t_begin, 5, 0);
      } else {
        const u64 sk_peek = peek_granule(ibs);
        f4 xz[NF4];
        ldsv<NF4>(xz, vq, O_Z1);
        const float br1 = LDSF(vc, O_VEC + V_BR1), br2 = LDSF(vc, O_VEC + V_BR2);
        float r = split_dot<0>(m4.w, xz);                          // Wr_1 z_1
        r = chan_sum<KQ>(r);
        if (lead) {
          xv = (xv + br1) + r;          // x_2
          xs[2] = xv;
          put_granule(ob, epoch, xv + br2, fast_edge);
        }
        // skip lane: sk' = sk + (Ws_prev zl + bs_prev) + (Ws_0 z_0 + bs_0) + (Ws_1 z_1 + bs_1)
        float k1 = split_dot<1>(m4.w, xz);
        unsigned wss = lds_addr(lmat + 3 * MAT_F) + 16u * t;  // SS = {Ws_prev; Ws_0}
        asm volatile("" : "+v"(wss));
        float kp = split_dot_lds<0>(wss, vq + 4u * O_ZL);
        float k0 = split_dot_lds<1>(wss, vq + 4u * O_Z0);
        kp = chan_sum<KQ>(kp);
        k0 = chan_sum<KQ>(k0);
        k1 = chan_sum<KQ>(k1);
        if (lead) {
          const float bsp = LDSF(vc, O_VEC + V_BSP), bs0 = LDSF(vc, O_VEC + V_BS0), bs1 = LDSF(vc, O_VEC + V_BS1);
          const float skin = (unsigned)(sk_peek >> 32) == epoch ? __uint_as_float((unsigned)sk_peek)
                                                                : wait_granule(ibs, epoch, err);
          put_granule(ob + 2 * C, epoch, ((skin + (kp + bsp)) + (k0 + bs0)) + (k1 + bs1), fast_edge);
        }
      }
      MVN_STAMP(b, s, ts - a.t_begin, 1);
      lds_barrier();  // iflag and the step's LDS vectors are settled for everyone
      if (iflag[0] == 0) {  // hand-off timed out (checked after the step: off the chain)
        alive = false;
        break;
      }
      if (ts + 1 < a.t_end) {
        precompute(ts + 1, true, true);
        save_pf(g);
      } else if (!chain && lead) {
        // last step of the launch: push only (the next launch pops in its prologue)
        int cq = c;
        asm volatile("" : "+v"(cq));
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) ring[doff[j] + cq + (ts & dmask[j]) * C] = xs[j];
      }
    }
    return;
  }

  // ============================ head stage ============================
  {
    float *tab = smem;                    // embedding tables [2][Q][C] (128 KB)
    float *vec = smem + N_LDS_MAT * MAT_F;
    float *zlb = vec;                     // [C] last layer's gated activation
    float *a0 = zlb + C;                  // [C] lrelu(skip)
    float *a1 = a0 + C;                   // [Q]
    float *lgb = a1 + Q;                  // [Q] logits
    const float *E0 = tab, *E1 = tab + Q * C;
    const float *hw = a.w + EMB_F + (size_t)(NS - 1) * STAGE_F;
    const f4 *W1p = (const f4 *)hw, *W2p = (const f4 *)(hw + W1_F + Q);
    const float *b1 = hw + W1_F, *b2 = hw + W1_F + Q + W2_F;
    const f4 *WSp = (const f4 *)(hw + W1_F + Q + W2_F + Q);
    const float *bsl = hw + W1_F + Q + W2_F + Q + WSL_F;
    int32_t *samples = a.samples + (size_t)b * a.stride;
    int *hidx = (int *)pfs;  // MULTI: [GMAX][2] = {idx_cur, idx_prev} of each sequence between its turns
    auto bind = [&](int g) {
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      samples = a.samples + (size_t)bq * a.stride;
    };

    // last layer's skip 1x1: thread (cs = tid >> 3, q8 = tid & 7), 8 inputs;
    // conv1: thread (o1 = tid >> 1, q1 = tid & 1), 32 inputs; conv2: thread (og = tid >> 3,
    // q2 = tid & 7), 4 outputs x 32 inputs
    const int o1 = tid >> 1, q1 = tid & 1, og = tid >> 3, q2 = tid & 7;
    v2f wsl[4], w1[16], w2[4][16];
    {
      const f4 *src = (const f4 *)a.w;
      f4 *dst = (f4 *)tab;
      for (int i = tid; i < EMB_F / 4; i += NT) dst[i] = src[i];
      loadn<2>(wsl, WSp, NT, tid);
      loadn<8>(w1, W1p, NT, tid);
#pragma unroll
      for (int r = 0; r < 4; ++r) loadn<8>(w2[r], W2p + r * 8 * NT, NT, tid);
    }
    const float b1r = b1[o1];
    const float b2r = b2[4 * og + (q2 & 3)];
    const float bslr = bsl[og];
    __syncthreads();

    int idx_cur = 0, idx_prev = -1;
    auto send_h0 = [&](unsigned ep) {  // wave 0: xp = the causal conv's two embedding rows, zl = sk = 0
      const int ic = min(max(idx_cur, 0), a.Q - 1), ip = min(idx_prev, a.Q - 1);
      float v = E1[ic * C + lane];
      if (ip >= 0) v += E0[ip * C + lane];
      put_granule(outbox + lane, ep, v, fast_edge);
      put_granule(outbox + C + lane, ep, 0.f, fast_edge);
      put_granule(outbox + 2 * C + lane, ep, 0.f, fast_edge);
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
        float v[2];
        const bool ok = wait_inbox64(inbox + C, epoch, err, v);  // zl (the last stage's xp' is not used)
        if (ok && lane < 32) {
          zlb[2 * lane] = v[0];
          zlb[2 * lane + 1] = v[1];
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      const u64 sk_peek = peek_granule(inbox + 2 * C + og);
      float sv = 0.f;
      if (do_head) {
        // the last layer's skip 1x1 (modules.py:90-91)
        f4 x[2];
        ldsn<2>(x, zlb + 8 * q2);
        sv = dotn<2>(wsl, x);
        sv = quad_sum(sv);
        sv += other_quad(sv);
      }
      // the skip lane ends here, every step (the step's skip sums have then been consumed all
      // along the pipeline before the next sample enters it)
      float skin = 0.f;
      if (q2 == 0)
        skin = (unsigned)(sk_peek >> 32) == epoch ? __uint_as_float((unsigned)sk_peek)
                                                  : wait_granule(inbox + 2 * C + og, epoch, err);
      if (do_head) {
        // skip sum, then the head's first leaky-ReLU (modules.py:140)
        if (q2 == 0) a0[og] = leaky(skin + (sv + bslr));
        lds_barrier();
        {
          f4 x[8];
          ldsn<8>(x, a0 + 32 * q1);
          float hsum = dotn<8>(w1, x);
          hsum += dpp_mov<DPP_XOR1>(hsum);
          if (q1 == 0) a1[o1] = leaky(hsum + b1r);
        }
        lds_barrier();
        {
          f4 x[8];
          ldsn<8>(x, a1 + 32 * q2);
          float s0 = dotn<8>(w2[0], x), s1 = dotn<8>(w2[1], x);
          float s2 = dotn<8>(w2[2], x), s3 = dotn<8>(w2[3], x);
          s0 = quad_sum(s0); s0 += other_quad(s0);
          s1 = quad_sum(s1); s1 += other_quad(s1);
          s2 = quad_sum(s2); s2 += other_quad(s2);
          s3 = quad_sum(s3); s3 += other_quad(s3);
          const int sel = q2 & 3;
          if (q2 < 4) lgb[4 * og + sel] = (sel == 0 ? s0 : sel == 1 ? s1 : sel == 2 ? s2 : s3) + b2r;
        }
        lds_barrier();
      }
      if (wave == 0) {
        if (do_head) {
          const f4 lv = ((const f4 *)lgb)[lane];
          const float lg[4] = {lv.x, lv.y, lv.z, lv.w};
          if (a.logits_out && u >= a.logits_t0 && 4 * lane < a.Q)  // (rows of a.Q logits: the padding is not written)
            ((f4 *)(a.logits_out + ((size_t)bq * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * a.Q))[lane] = lv;
          const int pick = choose_class(lg, a.temperature, uni, lane, a.Q);
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

// ---- packing ---------------------------------------------------------------------------
// Per-layer parameter pointers of the (up to) four layers a stage section is built from:
// index 0 = the previous stage's last layer (NULL for stage 0), 1..3 = the stage's own layers
// (NULL where the model has fewer layers: such a layer is packed as zeros, which makes it the
// identity -- z = gate(0, 0) = 0, no residual, no skip contribution).
struct FoldLayers {
  const float *fw[4], *gw[4], *rw[4], *rb[4], *sw[4], *sb[4];
};
__device__ __forceinline__ double fold_wc(const FoldLayers &p, int j, int row, int k) {  // current tap
  return p.fw[j] ? (double)fg_elem(p.fw[j], p.gw[j], fold::C, row, fold::C + k) : 0.0;
}
__device__ __forceinline__ double fold_wr(const FoldLayers &p, int j, int o, int k) {
  return p.rw[j] ? (double)p.rw[j][(size_t)o * fold::C + k] : 0.0;
}
__device__ __forceinline__ double fold_br(const FoldLayers &p, int j, int o) { return p.rb[j] ? (double)p.rb[j][o] : 0.0; }

__global__ void pack_fold_stage_kernel(FoldLayers p, float *__restrict__ dst) {
  using namespace fold;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= STAGE_F) return;
  if (i >= N_MAT * MAT_F) {
    const int o = i - N_MAT * MAT_F;
    double v = 0.0;
    if (o < V_CB2) {            // cb1 = Wc_1 br_0
      for (int m = 0; m < C; ++m) v += fold_wc(p, 2, o - V_CB1, m) * fold_br(p, 1, m);
    } else if (o < V_BSP) {     // cb2 = Wc_2 (br_0 + br_1)
      for (int m = 0; m < C; ++m) v += fold_wc(p, 3, o - V_CB2, m) * (fold_br(p, 1, m) + fold_br(p, 2, m));
    } else if (o < V_BR0) v = p.sb[0] ? p.sb[0][o - V_BSP] : 0.0;
    else if (o < V_BS0) v = fold_br(p, 1, o - V_BR0);
    else if (o < V_BR1) v = p.sb[1] ? p.sb[1][o - V_BS0] : 0.0;
    else if (o < V_BS1) v = fold_br(p, 2, o - V_BR1);
    else if (o < V_BR2) v = p.sb[2] ? p.sb[2][o - V_BS1] : 0.0;
    else v = fold_br(p, 3, o - V_BR2);
    dst[i] = (float)v;
    return;
  }
  // matrix element in the per-thread order: [2 NF4][t (256)] float4; thread t = 4c + kq owns rows
  // (c, C + c) x inputs 16 kq .. 16 kq + 15 (FoldMat)
  const int mat = i / MAT_F, r = i - mat * MAT_F;
  const int e = r & 3, v4 = r >> 2, t = v4 & 255, i4 = v4 >> 8;
  const bool split = mat == M_RR || mat == M_RS || mat == M_SS;
  // SPLIT: float4 i4 of half i4 / 4 holds inputs 4 (i4 % 4) + e of row t / KQ;
  // PAIR: float4 i4 holds {row c, row C + c} at inputs 2 i4 and 2 i4 + 1
  const int row = split ? (i4 / NF4) * C + t / KQ : (e & 1) * C + t / KQ;
  const int k = KPER * (t % KQ) + (split ? 4 * (i4 % NF4) + e : 2 * i4 + (e >> 1));
  double v = 0.0;
  auto prod = [&](int jc, int jr) {  // (Wc_jc Wr_jr)[row][k]
    double acc = 0.0;
    for (int m = 0; m < C; ++m) acc += fold_wc(p, jc, row, m) * fold_wr(p, jr, m, k);
    return acc;
  };
  auto ws = [&](int j, int o) { return p.sw[j] ? (double)p.sw[j][(size_t)o * C + k] : 0.0; };
  auto wp = [&](int j) { return p.fw[j] ? (double)fg_elem(p.fw[j], p.gw[j], C, row, k) : 0.0; };  // past tap
  switch (mat) {
    case M_A0: v = fold_wc(p, 1, row, k); break;
    case M_A0P: v = prod(1, 0); break;
    case M_A1: v = prod(2, 1); break;
    case M_A2: v = prod(3, 2); break;
    case M_B2P: v = prod(3, 1); break;
    case M_B1: v = fold_wc(p, 2, row, k); break;
    case M_B1P: v = prod(2, 0); break;
    case M_B2: v = fold_wc(p, 3, row, k); break;
    case M_RR: v = row < C ? fold_wr(p, 0, row, k) : fold_wr(p, 1, row - C, k); break;
    case M_RS: v = row < C ? fold_wr(p, 2, row, k) : ws(2, row - C); break;
    case M_WP0: v = wp(1); break;
    case M_WP1: v = wp(2); break;
    case M_WP2: v = wp(3); break;
    case M_SS: v = row < C ? ws(0, row) : ws(1, row - C); break;
  }
  dst[i] = (float)v;
}

// `qm`: the MODEL's class count (64, 128 or 256).  The head always runs 256 classes wide: classes >= qm are padding --
// zero rows and columns, conv2 bias -inf, so that their logits are -inf (probability 0 in the first softmax;
// choose_class masks them in the second) and they are never picked.
__global__ void pack_fold_head_kernel(const float *w1, const float *b1, const float *w2, const float *b2,
                                      const float *sw_last, const float *sb_last, float *__restrict__ dst, int qm) {
  using namespace fold;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W1_F) {
    // conv1: [8][tid (512)] float4, thread (o1 = tid >> 1, q1 = tid & 1) owns 32 inputs
    const int e = i & 3, v = i >> 2, tid = v & (NT - 1), i4 = v >> 9;
    const int o = tid >> 1;
    dst[i] = o < qm ? w1[(size_t)o * C + 32 * (tid & 1) + 4 * i4 + e] : 0.f;
  } else if (i < W1_F + Q) {
    dst[i] = i - W1_F < qm ? b1[i - W1_F] : 0.f;
  } else if (i < W1_F + Q + W2_F) {
    const int ii = i - W1_F - Q;
    const int e = ii & 3, v = ii >> 2, tid = v & (NT - 1), rest = v >> 9, r = rest >> 3, i8 = rest & 7;
    const int o = 4 * (tid >> 3) + r, k = 32 * (tid & 7) + 4 * i8 + e;
    dst[i] = (o < qm && k < qm) ? w2[(size_t)o * qm + k] : 0.f;
  } else if (i < W1_F + Q + W2_F + Q) {
    const int o = i - W1_F - Q - W2_F;
    dst[i] = o < qm ? b2[o] : -INFINITY;
  } else if (i < W1_F + Q + W2_F + Q + WSL_F) {
    // last layer's skip 1x1: [2][tid (512)] float4, thread (cs = tid >> 3, q8 = tid & 7) owns 8 inputs
    const int ii = i - (W1_F + Q + W2_F + Q);
    const int e = ii & 3, v = ii >> 2, tid = v & (NT - 1), i2 = v >> 9;
    dst[i] = sw_last ? sw_last[(size_t)(tid >> 3) * C + 8 * (tid & 7) + 4 * i2 + e] : 0.f;
  } else if (i < HEAD_F) {
    dst[i] = sb_last ? sb_last[i - (W1_F + Q + W2_F + Q + WSL_F)] : 0.f;
  }
}

__global__ void pack_fold_embed_kernel(const float *__restrict__ causal_w, float *__restrict__ dst, int qm) {
  using namespace fold;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= EMB_F) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, qq = r / C, c = r - qq * C;
  dst[i] = qq < qm ? causal_w[((size_t)c * qm + qq) * 2 + tap] : 0.f;
}

bool fold_ok(const mvn_dims *d) {
  // (r4: Q in {64, 128, 256} -- the head runs 256 classes wide, a smaller model's are padded at pack time)
  return d->residual_channels == 64 && d->skip_channels == 64 && head_q_ok(d->input_channels) &&
         n_layers(d) >= 1 && (n_layers(d) + fold::LPS - 1) / fold::LPS + 1 <= PIPE_XCD_CUS;
}
int fold_stages(const mvn_dims *d) { return (n_layers(d) + fold::LPS - 1) / fold::LPS + 1; }
int fold_pipelines(const mvn_dims *d) { return 8 * (PIPE_XCD_CUS / fold_stages(d)); }  // co-resident pipelines
int fold_pipelines_max(const mvn_dims *d) {  // ... plus the pipelines the XCDs' left-over CUs form across XCDs
  const int NS = fold_stages(d);
  return fold_pipelines(d) + 8 * (PIPE_XCD_CUS - (PIPE_XCD_CUS / NS) * NS) / NS;
}
int fold_max_batch(const mvn_dims *d) { return fold::GMAX * fold_pipelines_max(d); }     // GMAX sequences each
// Pipelines a launch of `batch` sequences runs on: one sequence each up to fold_pipelines(d) (the fastest step);
// rounds on those while the pipeline's latency bounds the step (up to FOLD_LATENCY_ROUNDS each); every pipeline
// the chip holds beyond that, where the step is rounds x the stages' service time per turn
constexpr int FOLD_LATENCY_ROUNDS = 5;
int fold_launch_pipelines(const mvn_dims *d, int batch) {
  const int p = fold_pipelines(d);
  if (batch <= p) return batch;
  return batch <= FOLD_LATENCY_ROUNDS * p ? p : fold_pipelines_max(d);
}
size_t fold_weights_floats(const mvn_dims *d) {
  return (size_t)fold::EMB_F + (size_t)(fold_stages(d) - 1) * fold::STAGE_F + fold::HEAD_F;
}
size_t fold_hand_floats(const mvn_dims *d, int batch) {
  const size_t n = (size_t)batch * fold_stages(d);
  return n * fold::GRAN * 2 + (16 + n + 63) / 64 * 64;
}

int fold_pack(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s) {
  using namespace fold;
  const int L = n_layers(d), NSL = fold_stages(d) - 1;
  hipLaunchKernelGGL(pack_fold_embed_kernel, dim3((EMB_F + 255) / 256), dim3(256), 0, s, p->causal_w, packed, d->input_channels);
  for (int st = 0; st < NSL; ++st) {
    FoldLayers fl;
    for (int j = 0; j < 4; ++j) {
      const int l = st * LPS - 1 + j;
      const bool ok = l >= 0 && l < L;
      fl.fw[j] = ok ? p->filter_w[l] : nullptr;
      fl.gw[j] = ok ? p->gate_w[l] : nullptr;
      fl.rw[j] = ok ? p->residual_w[l] : nullptr;
      fl.rb[j] = ok ? p->residual_b[l] : nullptr;
      fl.sw[j] = ok ? p->skip_w[l] : nullptr;
      fl.sb[j] = ok ? p->skip_b[l] : nullptr;
    }
    hipLaunchKernelGGL(pack_fold_stage_kernel, dim3((STAGE_F + 255) / 256), dim3(256), 0, s, fl,
                       packed + EMB_F + (size_t)st * STAGE_F);
  }
  // the head adds the skip 1x1 of the last stage's THIRD layer (index NSL * LPS - 1); when the
  // model ends earlier that slot is a zero layer (packed as zeros here) and the real last
  // layer's skip term was already added inside its stage
  const int l_tail = NSL * LPS - 1;
  const bool tail_real = l_tail < L;
  hipLaunchKernelGGL(pack_fold_head_kernel, dim3((HEAD_F + 255) / 256), dim3(256), 0, s, p->head1_w, p->head1_b,
                     p->head2_w, p->head2_b, tail_real ? p->skip_w[l_tail] : nullptr,
                     tail_real ? p->skip_b[l_tail] : nullptr, packed + EMB_F + (size_t)NSL * STAGE_F, d->input_channels);
  return check_hip(hipGetLastError(), "fold_pack");
}

// context section: the PIPE variant's per-layer layout (same thread mapping), packed by it
//
// Up to fold_pipelines(d) sequences: one sequence per pipeline (gen_fold_kernel<false>).  More:
// the pipelines serve ceil(batch / pipelines) sequences each in turn (gen_fold_kernel<true>).
int fold_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_floats_total,
                size_t status_offset_floats, hipStream_t s) {
  using namespace fold;
  int NS = fold_stages(d);
  const int pipes = fold_launch_pipelines(d, batch);
  const bool multi = batch > pipes;
  const void *fn = multi ? (const void *)gen_fold_kernel<true> : (const void *)gen_fold_kernel<false>;
  int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(gen_fold)");
  if (rc) return rc;
  const size_t lds_bytes = (multi ? LDS_FLOATS_MULTI : LDS_FLOATS) * sizeof(float);
  int dev = 0, cus = 0, per_cu = 0, coop = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice") ||
      check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute(CUs)") ||
      check_hip(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev),
                "hipDeviceGetAttribute(cooperative)") ||
      check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NT, lds_bytes),
                "hipOccupancyMaxActiveBlocksPerMultiprocessor(gen_fold)"))
    return MVN_ERR_LAUNCH;
  const int slots = pipes <= fold_pipelines(d) ? (pipes + 7) / 8 * NS : PIPE_XCD_CUS;
  if (cus < 8 * PIPE_XCD_CUS || batch > fold_max_batch(d) || per_cu < 1 || slots * 8 > per_cu * cus) {
    set_error("FOLD variant: %d stages per pipeline, %d pipelines of at most %d sequences each on %d CUs (batch %d "
              "asked for)", NS, cus < 8 * PIPE_XCD_CUS ? 0 : fold_pipelines_max(d), GMAX, cus, batch);
    return MVN_ERR_UNSUPPORTED;
  }
  // hand-off area layout of the generator state: [granules ...][16 flag words at
  // status_offset][placement words]; this variant's granules and placement words must fit
  const size_t gran_floats = (size_t)batch * NS * GRAN * 2;
  if (gran_floats > status_offset_floats || status_offset_floats + 16 + (size_t)batch * NS > hand_floats_total) {
    set_error("FOLD variant: hand-off area too small (%zu granule floats, status word at %zu of %zu)",
              gran_floats, status_offset_floats, hand_floats_total);
    return MVN_ERR_BAD_ARG;
  }
  unsigned *err = (unsigned *)(hand + status_offset_floats);
  rc = check_hip(hipMemsetAsync(hand, 0, gran_floats * sizeof(float), s), "hipMemsetAsync(granules)");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(err + 16, 0, (hand_floats_total - status_offset_floats - 16) * sizeof(float), s),
                 "hipMemsetAsync(placement words)");
  if (rc) return rc;
  u64 *gran = (u64 *)hand;
  GenArgs args = a;
  int nb = pipes, nseq = batch;
  if (coop && pipe_cooperative_launch()) {
    void *kargs[] = {(void *)&args, (void *)&gran, (void *)&err, (void *)&NS, (void *)&nb, (void *)&nseq};
    return check_hip(hipLaunchCooperativeKernel(fn, dim3(slots * 8), dim3(NT), kargs, (unsigned)lds_bytes, s),
                     "mvn_generate(fold, cooperative launch)");
  }
  if (multi)
    hipLaunchKernelGGL(gen_fold_kernel<true>, dim3(slots * 8), dim3(NT), lds_bytes, s, args, gran, err, NS, nb, nseq);
  else
    hipLaunchKernelGGL(gen_fold_kernel<false>, dim3(slots * 8), dim3(NT), lds_bytes, s, args, gran, err, NS, nb, nseq);
  return check_hip(hipGetLastError(), "mvn_generate(fold)");
}

}  // namespace mvn

#ifdef MVN_PIPE_STAMPS
extern "C" int mvn_debug_read_stamps_fold(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_stamps) / 8) n = sizeof(mvn::g_stamps) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_stamps), n * 8), "read stamps");
}
extern "C" int mvn_debug_read_fine_fold(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_fine) / 8) n = sizeof(mvn::g_fine) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_fine), n * 8), "read fine");
}
#endif
