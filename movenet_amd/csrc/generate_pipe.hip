// PIPE variant of the generator: a layer pipeline across CUs with RESIDENT weights.
//
// The STREAM variant re-reads 3.3 MB of weights from L2 for every generated
// sample (one CU per sequence, ~42 GB/s => 78 us per step).  Here each sequence
// gets a private pipeline of NS = ceil(L/4)+1 workgroups (512 threads, one per
// CU): stage s keeps the weights of layers 4s..4s+3 in registers (waves 0-3 the
// current-tap filter|gate matrices, waves 4-7 the residual|skip matrices, 128
// VGPRs per thread) and LDS (past-tap matrix),
// the last stage keeps the dense head in registers and the embedding tables in
// LDS.  Nothing is re-read per step; what moves is the activation: 64 residual
// + 64 skip-sum floats travel from stage to stage as 128 eight-byte
// {value, epoch} granules written with write-through (sc1) stores and polled
// by one wave of the consumer (the data is the flag: no fence, no separate
// flag).  Off the critical path each stage pushes/pops its dilation queues
// and pre-computes the past-tap half of next step's f/g sums.
//
// Per step per sequence: ~30 x 0.3 us of dependent layer arithmetic + NS hops of
// ~0.9 us.  All spins are bounded; a time-out raises the error word in the
// hand-off area and every stage drains.
//
// Reference arithmetic: see generate.hip (movenet/wavenet.py:217-237,
// movenet/modules.py:19-30, :67-93, :139-142).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gen_common.h"
#include "pipe_common.h"

namespace mvn {

// Shape traits.  C = K = 64: 4 layers per stage, a thread owns 2 rows x 16 inputs (4 lanes
// per channel).  C = K = 128 (BASELINE config 5): the 3 x 128 KB of a single layer already
// fill a CU, so 1 layer per stage, 2 rows x 64 inputs per thread (2 lanes per channel).
template <int CC>
struct PipeCfg {
  static constexpr int C = CC, Q = 256, NT = 512;
  static constexpr int LPS = CC == 64 ? 4 : 1;    // layers per stage
  static constexpr int KQ = 256 / CC;             // lanes sharing one channel's two rows
  static constexpr int KPER = CC / KQ;            // inputs per lane
  static constexpr int NF4 = KPER / 4;            // float4 per row per lane
  static constexpr int MAT_F = 2 * CC * CC;       // one 2C x C matrix
  static constexpr int LAYER_F = 3 * MAT_F + 2 * CC;   // WC | WP | WR | biases
  static constexpr int CTX_LAYER_F = MAT_F + 2 * CC;   // context section per layer
  static constexpr int EMB_F = 2 * Q * CC;
  static constexpr int W1_F = Q * CC, W2_F = Q * Q;
  static constexpr int HEAD_F = W1_F + Q + W2_F + Q;
  static constexpr int GRAN = 2 * CC;             // granules per inbox: residual | skip sum
  static constexpr int GL = GRAN / 64;            // granules per polling lane
  static constexpr int W1N = CC / 2;              // head conv1 inputs per thread (2 threads per row)
  // LDS floats: layer stage LPS*MAT_F + (4 + LPS)*C; head stage: tables or conv1 weights
  // (32768 either way) + a0[C] + a1[Q] + logits[Q]; + 16 flag words
  static constexpr int LDS_FLOATS = 32768 + 8 * CC + 2 * Q + 64 + 16;
  // Rounds (r3, as in generate_fold.hip): a pipeline serves up to GMAX sequences in turn -- weights
  // shared, one inbox per sequence and stage; between its turns a sequence's past-tap sums wait in
  // LDS behind the step vectors, PFS_F floats per channel (16 KB either way)
  static constexpr int GMAX = CC == 64 ? 8 : 16;
  static constexpr int PFS_F = 2 * LPS;
  static constexpr int LDS_FLOATS_MULTI = LDS_FLOATS + GMAX * CC * PFS_F;
};
// Workgroup = 8 waves.  Waves 0-3 ("FG group") own the filter/gate matrices, waves
// 4-7 ("RS group") the residual/skip matrices: at any moment ONE wave per SIMD is
// issuing, so the dependent chain is not slowed by a co-resident wave replaying the
// same bookkeeping instructions, and each thread keeps LPS x 2 x KPER weights = 128
// VGPRs resident.
//
// MULTI = false: one sequence per pipeline (nseq == nb).  MULTI = true: pipeline b serves sequences
// b, b + nb, b + 2 nb, ... < nseq in turn, one step of each per round.
template <int CC, bool MULTI>
__global__ __launch_bounds__(512, 2) void gen_pipe_kernel(GenArgs a, u64 *hand, unsigned *err, int NS,
                                                         int nb, int nseq) {
  using P = PipeCfg<CC>;
  constexpr int C = P::C, Q = P::Q, NT = P::NT, LPS = P::LPS, KQ = P::KQ, KPER = P::KPER;
  constexpr int NF4 = P::NF4, MAT_F = P::MAT_F, GRAN = P::GRAN, GL = P::GL, PFS_F = P::PFS_F;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup i is dispatched to XCD i % 8 (observed; every edge verifies its placement
  // below, so this is speed only -- but co-residency needs <= 32 workgroups per XCD, which
  // the host checks with the same arithmetic).  A pipeline of NS <= 32 stages sits in
  // one XCD so that its hops stay inside one L2; a longer one spans XS adjacent XCDs.
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
  // The status word is STICKY: raised by a timed-out hand-off, cleared only when the host
  // zeroes the generator state (RingGenerator.reset).  A launch that finds it raised does
  // nothing, so a failure in one advance() chunk cannot be papered over by the next one.
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const int L = a.L;
  const int s_next = s + 1 == NS ? 0 : s + 1;
  const int G = MULTI ? (nseq - b + nb - 1) / nb : 1;  // sequences of this pipeline
  int bq = b;                                           // the sequence whose turn it is
  u64 *inbox = hand + ((size_t)b * NS + s) * GRAN;
  u64 *outbox = hand + ((size_t)b * NS + s_next) * GRAN;
  int *iflag = (int *)(smem + P::LDS_FLOATS - 16);  // [0] ok flag, [3] fast-edge flag
  float *pfs = smem + P::LDS_FLOATS;                // MULTI: [GMAX][C][PFS_F] (layer stages), head: indices
  // placement handshake: publish my XCC id (+1), read my consumer's
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
    // ================= layer stage: layers l0 .. l0+nl-1 =================
    // FG group thread t: channel c = t / KQ, rows f_c and g_c, inputs k in [KPER*kq, +KPER);
    // RS group thread t: channel c, rows res_c and skip_c, same k split.  A row is finished
    // by a KQ-lane DPP sum; each lane fetches KPER inputs (NF4 x ds_read_b128).
    const int l0 = s * LPS, nl = min(LPS, L - l0);
    const bool fg_group = tid < 256;
    const int t = tid & 255, c = t / KQ, kq = t % KQ;
    const bool lead = kq == 0;
    float *wp = smem;                  // [LPS][2*NF4][256] float4: past-tap f|g weights (FG group)
    float *cur = smem + LPS * MAT_F;   // [C] residual stream
    float *zb = cur + C;               // [C] gated activation
    float *pastb = zb + C;             // [LPS][C] popped queue entries
    float *skin = pastb + LPS * C;     // [C] running skip sum as received
    float *ctxb = skin + C;            // [C] context vector of the step being prepared
    float *ring = a.state + (size_t)b * a.state_per_seq;
    auto bind = [&](int g) {  // MULTI: the pointers of sequence b + g nb
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      ring = a.state + (size_t)bq * a.state_per_seq;
    };

    v2f wa[LPS][2 * NF4], wb[LPS][2 * NF4];  // FG: f_c | g_c current-tap rows; RS: res_c | skip_c
    float bias_r[LPS], bias_s[LPS], pf[LPS], pg[LPS], xs[LPS];
    int doff[LPS], dmask[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
      bias_r[j] = 0.f; bias_s[j] = 0.f; pf[j] = 0.f; pg[j] = 0.f; xs[j] = 0.f;
      doff[j] = 0; dmask[j] = 0;
#pragma unroll
      for (int i = 0; i < 2 * NF4; ++i) { wa[j][i] = v2f{0.f, 0.f}; wb[j][i] = v2f{0.f, 0.f}; }
      if (j < nl) {
        const float *lw = a.w + P::EMB_F + (size_t)(l0 + j) * P::LAYER_F;
        if (fg_group) {
          loadn<NF4>(wa[j], (const f4 *)lw, 256, t);
          loadn<NF4>(wb[j], (const f4 *)lw + NF4 * 256, 256, t);
          const f4 *wp4 = (const f4 *)(lw + MAT_F);
#pragma unroll
          for (int i = 0; i < 2 * NF4; ++i)
            ((f4 *)wp)[(j * 2 * NF4 + i) * 256 + t] = wp4[i * 256 + t];
        } else {
          loadn<NF4>(wa[j], (const f4 *)(lw + 2 * MAT_F), 256, t);
          loadn<NF4>(wb[j], (const f4 *)(lw + 2 * MAT_F) + NF4 * 256, 256, t);
          bias_r[j] = lw[3 * MAT_F + c];
          bias_s[j] = lw[3 * MAT_F + C + c];
        }
        const int l = l0 + j;
        dmask[j] = (1 << (l % a.layer_size)) - 1;
        doff[j] = ring_offset(l, a.layer_size, C);
      }
    }

    // Off the critical path: push this step's layer inputs into the dilation queues,
    // pop the entries step tn needs (RS lead lanes), then the past-tap half of step
    // tn's f/g pre-activations (FG group).
    auto precompute = [&](int tn, bool push) {
      // `tq` is the thread index behind an optimisation fence: every address below is
      // recomputed here instead of being kept in registers across the step loop (the
      // critical path needs those registers; this code has slack)
      int tq = t;
      asm volatile("" : "+v"(tq));
      const int cq = tq / KQ, kk = tq % KQ;
      if (!fg_group && lead) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + cq;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j] : ring_load(base + (tn & dmask[j]) * C);
            pastb[j * C + cq] = pv;
          }
      }
      if (a.ctx_tm && fg_group && tq < C)
        ctxb[tq] = a.ctx_tm[(size_t)bq * a.ctx_stride_b + (size_t)tn * C + tq];
      __syncthreads();
      if (fg_group) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            const f4 *wpj = (const f4 *)wp + j * 2 * NF4 * 256;
            pf[j] = chan_sum<KQ>(dot_stream<NF4>(wpj, 256, tq, pastb + j * C + KPER * kk));
            pg[j] = chan_sum<KQ>(dot_stream<NF4>(wpj + NF4 * 256, 256, tq, pastb + j * C + KPER * kk));
            if (a.ctx_tm) {
              // 1x1 context convs (modules.py:58-63, :75-77): their weights are streamed
              // from L2 here, off the critical path
              const float *wc = a.wctx + (size_t)(l0 + j) * P::CTX_LAYER_F;
              pf[j] += chan_sum<KQ>(dot_stream<NF4>((const f4 *)wc, 256, tq, ctxb + KPER * kk)) +
                       wc[MAT_F + cq];
              pg[j] += chan_sum<KQ>(dot_stream<NF4>((const f4 *)wc + NF4 * 256, 256, tq, ctxb + KPER * kk)) +
                       wc[MAT_F + C + cq];
            }
          }
      }
    };
    // MULTI: a sequence's pf / pg between its turns (only the lead lane of a channel uses them)
    auto save_pf = [&](int g) {
      if (MULTI && fg_group && lead) {
        float *q = pfs + ((size_t)g * C + c) * PFS_F;
#pragma unroll
        for (int j = 0; j < LPS; ++j) *(v2f *)(q + 2 * j) = v2f{pf[j], pg[j]};
      }
    };
    auto load_pf = [&](int g) {
      if (MULTI && fg_group && lead) {
        const float *q = pfs + ((size_t)g * C + c) * PFS_F;
#pragma unroll
        for (int j = 0; j < LPS; ++j) {
          const v2f v = *(const v2f *)(q + 2 * j);
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
        __syncthreads();  // the filter/gate waves have read this sequence's popped entries
      }
    } else {
      precompute(a.t_begin, false);
    }

    bool alive = true;
    for (int ts = a.t_begin; ts < a.t_end && alive; ++ts)
    for (int g = 0; g < G; ++g) {
      if (MULTI) bind(g);
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      // The step starts as soon as the C residual-stream granules are in; the running skip sum
      // (sent a little later by the producer, see below) is awaited by wave 4 while the filter/gate
      // waves already work on the first layer.  (r3: C = 128 too -- one 16-byte poll per lane through
      // poll16 instead of the two-load spin loop over 2C granules, and only the residual row of the
      // residual|skip product on the chain.)
      constexpr bool SPLIT = true;
      if (wave == 0) {
        bool ok;
        if constexpr (SPLIT && CC == 128) {
          float v[2];
          ok = wait_inbox<2>(inbox, epoch, err, v);
          if (ok) {
            cur[2 * lane] = v[0];
            cur[2 * lane + 1] = v[1];
          }
        } else if constexpr (SPLIT) {
          float v[2];
          ok = wait_inbox64(inbox, epoch, err, v);
          if (ok && lane < 32) {
            cur[2 * lane] = v[0];
            cur[2 * lane + 1] = v[1];
          }
        } else {
          float v[GL];
          ok = wait_inbox<GL>(inbox, epoch, err, v);
          if (ok) {
            // granules 0..C-1 residual stream, C..2C-1 running skip sum
            cur[2 * lane] = v[0];
            cur[2 * lane + 1] = v[1];
            skin[2 * lane] = v[GL - 2];
            skin[2 * lane + 1] = v[GL - 1];
          }
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      load_pf(g);
      if constexpr (SPLIT) {
        if (wave == 4) {  // off the chain: the residual/skip waves idle during the first f/g phase
          float v[2];
          bool ok;
          if constexpr (CC == 128) {
            ok = wait_inbox<2>(inbox + C, epoch, err, v);
            if (ok) {
              skin[2 * lane] = v[0];
              skin[2 * lane + 1] = v[1];
            }
          } else {
            ok = wait_inbox64(inbox + C, epoch, err, v);
            if (ok && lane < 32) {
              skin[2 * lane] = v[0];
              skin[2 * lane + 1] = v[1];
            }
          }
          if (lane == 0) iflag[1] = ok ? 1 : 0;
        }
      }
      float skipacc = 0.f;
      if constexpr (!SPLIT) skipacc = (!fg_group && lead) ? skin[c] : 0.f;
#pragma unroll
      for (int j = 0; j < LPS; ++j)
        if (j < nl) {
          float old = 0.f;
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 0, 0);
          if (fg_group) {
            float f, g;
            dot2_lds<NF4>(wa[j], wb[j], cur + KPER * kq, f, g);
            f = chan_sum<KQ>(f) + pf[j];
            g = chan_sum<KQ>(g) + pg[j];
            const float z = gate_fast(f, g);
            if (lead) zb[c] = z;
            if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 1, 0);
          } else if (lead) {
            old = cur[c];  // this layer's input: residual add below, queue push later
          }
          lds_barrier();
          if constexpr (SPLIT) {
            // Only the residual row is on the chain: it is finished and published first; the
            // skip row (same z, kept in registers) follows behind the barrier, while the
            // filter/gate waves are already on the next layer.
            f4 xz[CC == 64 ? NF4 : 1];  // C = 64: z stays in registers for the skip row; C = 128: re-read
            if (!fg_group) {
              if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 2, 256);
              if (j == 0 && lead) skipacc = skin[c];  // wave 4 stored it before this barrier
              float r;
              if constexpr (CC == 64) {
                ldsn<NF4>(xz, zb + KPER * kq);
                r = chan_sum<KQ>(dotn<NF4>(wa[j], xz));
              } else {
                r = chan_sum<KQ>(dot1_lds<NF4>(wa[j], zb + KPER * kq));
              }
              if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 3, 256);
              if (lead) {
                xs[j] = old;
                const float outv = (r + bias_r[j]) + old;
                cur[c] = outv;
                if (j == nl - 1) put_granule(outbox + c, epoch, outv, fast_edge);  // hand on at once
              }
              if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 4, 256);
            }
            lds_barrier();
            if (!fg_group) {
              // (C = 128: one layer per stage, so zb is untouched until the next step's first phase)
              float k;
              if constexpr (CC == 64) k = chan_sum<KQ>(dotn<NF4>(wb[j], xz));
              else k = chan_sum<KQ>(dot1_lds<NF4>(wb[j], zb + KPER * kq));
              if (lead) {
                skipacc += k + bias_s[j];
                if (j == nl - 1) put_granule(outbox + C + c, epoch, skipacc, fast_edge);
              }
            }
          } else {
            if (!fg_group) {
              float r, k;
              dot2_lds<NF4>(wa[j], wb[j], zb + KPER * kq, r, k);
              r = chan_sum<KQ>(r);
              k = chan_sum<KQ>(k);
              if (lead) {
                xs[j] = old;
                const float outv = (r + bias_r[j]) + old;
                cur[c] = outv;
                skipacc += k + bias_s[j];
                if (j == nl - 1) {
                  // the stage's last layer: hand the activation on before anything else
                  put_granule(outbox + c, epoch, outv, fast_edge);
                  put_granule(outbox + C + c, epoch, skipacc, fast_edge);
                }
              }
            }
            lds_barrier();
          }
          if (j == 0) MVN_FINE(b, s, ts - a.t_begin, 5, 0);
        }
      MVN_STAMP(b, s, ts - a.t_begin, 1);
      // hand-off timed out (checked after the step: off the chain; iflag[1] was written by wave 4
      // before the first f/g -> r/s barrier of this step)
      if (iflag[0] == 0 || (SPLIT && iflag[1] == 0)) {
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
    // C = 64: the embedding tables (128 KB) live in LDS and conv1's weights in registers.
    // C = 128: the tables are 256 KB, so they stay in L2 (two 512-B rows per step) and LDS
    // holds conv1's weights instead (its 64 inputs per thread would not fit the registers
    // next to conv2's 128).
    constexpr bool TABLES_IN_LDS = CC == 64;
    constexpr int W1N4 = P::W1N / 4;         // float4 of conv1 weights per thread
    float *big = smem;                        // tables [2][Q][C] or conv1 weights [W1N4][512] f4
    float *a0 = smem + 32768;                 // [C]
    float *a1 = a0 + C;                       // [256]
    float *lgb = a1 + Q;                      // [256] logits
    const float *E0 = TABLES_IN_LDS ? big : a.w, *E1 = E0 + Q * C;
    const float *hw = a.w + P::EMB_F + (size_t)L * P::LAYER_F;
    const f4 *W1p = (const f4 *)hw, *W2p = (const f4 *)(hw + P::W1_F + Q);
    const float *b1 = hw + P::W1_F, *b2 = hw + P::W1_F + Q + P::W2_F;
    int32_t *samples = a.samples + (size_t)b * a.stride;
    int *hidx = (int *)pfs;  // MULTI: [GMAX][2] = {idx_cur, idx_prev} of each sequence between its turns
    auto bind = [&](int g) {
      bq = b + g * nb;
      inbox = hand + ((size_t)bq * NS + s) * GRAN;
      outbox = hand + ((size_t)bq * NS + s_next) * GRAN;
      samples = a.samples + (size_t)bq * a.stride;
    };

    // conv1: thread (o1 = tid>>1, q1 = tid&1), C/2 inputs; conv2: thread (og = tid>>3,
    // q2 = tid&7), 4 outputs x 32 inputs
    const int o1 = tid >> 1, q1 = tid & 1, og = tid >> 3, q2 = tid & 7;
    v2f w1[TABLES_IN_LDS ? 2 * W1N4 : 2], w2[4][16];
    if constexpr (TABLES_IN_LDS) {
      const f4 *src = (const f4 *)a.w;
      f4 *dst = (f4 *)big;
      for (int i = tid; i < P::EMB_F / 4; i += NT) dst[i] = src[i];
      loadn<W1N4>(w1, W1p, NT, tid);
    } else {
      for (int i = tid; i < W1N4 * NT; i += NT) ((f4 *)big)[i] = W1p[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) loadn<8>(w2[r], W2p + r * 8 * NT, NT, tid);
    const float b1r = b1[o1];
    const float b2r = b2[4 * og + (q2 & 3)];
    __syncthreads();

    // Wave 0 closes every step alone (softmax, choice) and immediately opens the next
    // one: it gathers the two embedding rows of the causal conv (modules.py:28-30 on a
    // one-hot input) and hands them to stage 0, so no barrier sits between the choice
    // and the next step's first hop.
    int idx_cur = 0, idx_prev = -1;
    auto send_h0 = [&](unsigned ep) {  // wave 0: granules c = residual, C + c = skip sum 0
      const int ic = min(max(idx_cur, 0), a.Q - 1), ip = min(idx_prev, a.Q - 1);
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
        const bool ok = wait_inbox<GL>(inbox, epoch, err, v);
        if (ok) {
          // only the skip sum (granules C..2C-1) feeds the head
          if (GL == 2) {
            if (lane >= 32) {
              a0[2 * (lane - 32)] = leaky(v[0]);
              a0[2 * (lane - 32) + 1] = leaky(v[1]);
            }
          } else {
            a0[2 * lane] = leaky(v[GL - 2]);
            a0[2 * lane + 1] = leaky(v[GL - 1]);
          }
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      if (do_head) {
        {
          float hsum;
          if constexpr (TABLES_IN_LDS) {
            f4 x[W1N4];
            ldsn<W1N4>(x, a0 + P::W1N * q1);
            hsum = dotn<W1N4>(w1, x);
          } else {
            hsum = dot_stream<W1N4>((const f4 *)big, NT, tid, a0 + P::W1N * q1);
          }
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
          // lane i owns classes 4i..4i+3; every reduction is intra-wave (DPP + readlane)
          const f4 lv = ((const f4 *)lgb)[lane];
          float lg[4] = {lv.x, lv.y, lv.z, lv.w};
          if (a.logits_out && u >= a.logits_t0 && 4 * lane < a.Q)  // (rows of a.Q logits: the padding is not written)
            ((f4 *)(a.logits_out +
                    ((size_t)bq * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * a.Q))[lane] = lv;
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

// ---- packing: state_dict layouts -> per-thread register order -----------------
template <int CC>
__global__ void pack_layer_pipe_kernel(const float *fw, const float *gw, const float *rw,
                                       const float *rb, const float *sw, const float *sb,
                                       float *__restrict__ dst) {
  using P = PipeCfg<CC>;
  constexpr int C = P::C, MAT_F = P::MAT_F, NF4 = P::NF4, KQ = P::KQ, KPER = P::KPER;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P::LAYER_F) return;
  if (i >= 3 * MAT_F) {
    const int o = i - 3 * MAT_F;
    dst[i] = o < C ? rb[o] : sb[o - C];
    return;
  }
  // each matrix: [2*NF4][t (256)] float4; thread t = KQ*c + kq owns rows (c, C+c) x inputs
  // k = KPER*kq + 4*(i4 % NF4) + e: float4 0..NF4-1 belong to row c, NF4..2NF4-1 to row C+c
  const int region = i / MAT_F, r = i - region * MAT_F;
  const int e = r & 3, v = r >> 2, t = v & 255, i4 = v >> 8;
  const int row = (i4 / NF4) * C + t / KQ, k = KPER * (t % KQ) + 4 * (i4 % NF4) + e;
  if (region < 2)
    dst[i] = fg_elem(fw, gw, C, row, region == 0 ? C + k : k);  // WC: current tap, WP: past tap
  else
    dst[i] = rs_elem(rw, sw, C, row, k);
}

// `qm`: the model's class count (64, 128 or 256); the head runs 256 wide, classes >= qm are padding (zero rows and
// columns, conv2 bias -inf: see pack_fold_head_kernel)
template <int CC>
__global__ void pack_head_pipe_kernel(const float *w1, const float *b1, const float *w2,
                                      const float *b2, float *__restrict__ dst, int qm) {
  using P = PipeCfg<CC>;
  constexpr int C = P::C, Q = P::Q, NT = P::NT, W1N = P::W1N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P::W1_F) {
    // conv1: [W1N/4][tid (512)] float4, thread (o1 = tid>>1, q1 = tid&1) owns W1N inputs
    const int e = i & 3, v = i >> 2, tid = v & (NT - 1), i4 = v >> 9;
    const int o = tid >> 1;
    dst[i] = o < qm ? w1[(size_t)o * C + W1N * (tid & 1) + 4 * i4 + e] : 0.f;
  } else if (i < P::W1_F + Q) {
    dst[i] = i - P::W1_F < qm ? b1[i - P::W1_F] : 0.f;
  } else if (i < P::W1_F + Q + P::W2_F) {
    const int ii = i - P::W1_F - Q;
    const int e = ii & 3, v = ii >> 2, tid = v & (NT - 1), rest = v >> 9, r = rest >> 3, i8 = rest & 7;
    const int o = 4 * (tid >> 3) + r, k = 32 * (tid & 7) + 4 * i8 + e;
    dst[i] = (o < qm && k < qm) ? w2[(size_t)o * qm + k] : 0.f;
  } else if (i < P::HEAD_F) {
    const int o = i - P::W1_F - Q - P::W2_F;
    dst[i] = o < qm ? b2[o] : -INFINITY;
  }
}

template <int CC>
__global__ void pack_embed_pipe_kernel(const float *__restrict__ causal_w, float *__restrict__ dst, int qm) {
  using P = PipeCfg<CC>;
  constexpr int C = P::C, Q = P::Q;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P::EMB_F) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, qq = r / C, c = r - qq * C;
  dst[i] = qq < qm ? causal_w[((size_t)c * qm + qq) * 2 + tap] : 0.f;
}

template <int CC>
__global__ void pack_ctx_pipe_kernel(const float *wcf, const float *bcf, const float *wcg,
                                     const float *bcg, float *__restrict__ dst) {
  using P = PipeCfg<CC>;
  constexpr int C = P::C, MAT_F = P::MAT_F, NF4 = P::NF4, KQ = P::KQ, KPER = P::KPER;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P::CTX_LAYER_F) return;
  if (i >= MAT_F) {
    const int o = i - MAT_F;
    dst[i] = o < C ? bcf[o] : bcg[o - C];
    return;
  }
  // same thread mapping as the current-tap matrix: thread t = KQ*c + kq owns rows (c, C+c)
  const int e = i & 3, v = i >> 2, t = v & 255, i4 = v >> 8;
  const int c = t / KQ, k = KPER * (t % KQ) + 4 * (i4 % NF4) + e;
  dst[i] = (i4 / NF4) ? wcg[(size_t)c * C + k] : wcf[(size_t)c * C + k];
}

bool pipe_ok(const mvn_dims *d) {
  const int c = d->residual_channels;
  return (c == 64 || c == 128) && d->skip_channels == c && head_q_ok(d->input_channels) &&
         n_layers(d) >= 1;
}
static int pipe_lps(const mvn_dims *d) { return d->residual_channels == 64 ? 4 : 1; }
int pipe_stages(const mvn_dims *d) { return (n_layers(d) + pipe_lps(d) - 1) / pipe_lps(d) + 1; }
int pipe_pipelines(const mvn_dims *d) {
  // the kernel's placement: NS <= 32 -> floor(32/NS) pipelines in each of the 8 XCDs,
  // otherwise one pipeline per group of ceil(NS/32) XCDs
  const int NS = pipe_stages(d);
  return NS <= PIPE_XCD_CUS ? 8 * (PIPE_XCD_CUS / NS) : 8 / ((NS + PIPE_XCD_CUS - 1) / PIPE_XCD_CUS);
}
int pipe_max_batch(const mvn_dims *d) {  // each pipeline serves up to GMAX sequences in turn
  return (d->residual_channels == 64 ? PipeCfg<64>::GMAX : PipeCfg<128>::GMAX) * pipe_pipelines(d);
}
size_t pipe_hand_floats(const mvn_dims *d, int batch) {
  // batch * NS inboxes of 2C granules (2 floats each), then 16 flag words (error word
  // first) and batch * NS placement words, padded to 64 floats
  const size_t n = (size_t)batch * pipe_stages(d);
  return n * 2 * d->residual_channels * 2 + (16 + n + 63) / 64 * 64;
}
size_t pipe_weights_floats(const mvn_dims *d) {
  const size_t L = n_layers(d);
  return d->residual_channels == 64 ? PipeCfg<64>::EMB_F + L * PipeCfg<64>::LAYER_F + PipeCfg<64>::HEAD_F
                                    : PipeCfg<128>::EMB_F + L * PipeCfg<128>::LAYER_F + PipeCfg<128>::HEAD_F;
}

template <int CC>
static int pipe_pack_t(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s) {
  using P = PipeCfg<CC>;
  const int L = n_layers(d);
  hipLaunchKernelGGL(pack_embed_pipe_kernel<CC>, dim3((P::EMB_F + 255) / 256), dim3(256), 0, s,
                     p->causal_w, packed, d->input_channels);
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL(pack_layer_pipe_kernel<CC>, dim3((P::LAYER_F + 255) / 256), dim3(256), 0, s,
                       p->filter_w[l], p->gate_w[l], p->residual_w[l], p->residual_b[l], p->skip_w[l],
                       p->skip_b[l], packed + P::EMB_F + (size_t)l * P::LAYER_F);
  hipLaunchKernelGGL(pack_head_pipe_kernel<CC>, dim3((P::HEAD_F + 255) / 256), dim3(256), 0, s,
                     p->head1_w, p->head1_b, p->head2_w, p->head2_b,
                     packed + P::EMB_F + (size_t)L * P::LAYER_F, d->input_channels);
  return check_hip(hipGetLastError(), "pipe_pack");
}
int pipe_pack(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s) {
  return d->residual_channels == 64 ? pipe_pack_t<64>(d, p, packed, s) : pipe_pack_t<128>(d, p, packed, s);
}

template <int CC>
static int pipe_pack_ctx_t(const mvn_dims *d, const mvn_params *p, float *ctx_section, hipStream_t s) {
  using P = PipeCfg<CC>;
  for (int l = 0; l < n_layers(d); ++l)
    hipLaunchKernelGGL(pack_ctx_pipe_kernel<CC>, dim3((P::CTX_LAYER_F + 255) / 256), dim3(256), 0, s,
                       p->ctx_filter_w[l], p->ctx_filter_b[l], p->ctx_gate_w[l], p->ctx_gate_b[l],
                       ctx_section + (size_t)l * P::CTX_LAYER_F);
  return check_hip(hipGetLastError(), "pipe_pack_ctx");
}
int pipe_pack_ctx(const mvn_dims *d, const mvn_params *p, float *ctx_section, hipStream_t s) {
  return d->residual_channels == 64 ? pipe_pack_ctx_t<64>(d, p, ctx_section, s)
                                    : pipe_pack_ctx_t<128>(d, p, ctx_section, s);
}

// Co-residency: every stage of every pipeline polls its predecessor, so ALL workgroups of the
// grid must be resident at once.  The launch is cooperative -- the runtime checks the grid
// against the kernel's occupancy (hipErrorCooperativeLaunchTooLarge instead of a silent
// dead wait) and does not run it next to another kernel of this process -- and the host
// repeats the same check with the occupancy query.  Another PROCESS holding CUs can still
// starve a stage: then the bounded spins raise the sticky status word and the caller
// (WaveNet.generate) reruns the call on a kernel without hand-offs.
template <int CC>
static int pipe_launch_t(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_total,
                         size_t status_off, hipStream_t s) {
  using P = PipeCfg<CC>;
  int NS = pipe_stages(d);
  // up to pipe_pipelines(d) sequences: one per pipeline; more: ceil(batch / pipelines) each, in turn
  const int pipes = std::min(batch, pipe_pipelines(d));
  const bool multi = batch > pipes;
  const void *fn = multi ? (const void *)gen_pipe_kernel<CC, true> : (const void *)gen_pipe_kernel<CC, false>;
  int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(gen_pipe)");
  if (rc) return rc;
  const size_t lds_bytes = (multi ? P::LDS_FLOATS_MULTI : P::LDS_FLOATS) * sizeof(float);
  // grid: 8 workgroups (one per XCD) per slot; see the kernel's (xcd, slot) -> (b, s) map
  const int XS = (NS + PIPE_XCD_CUS - 1) / PIPE_XCD_CUS;
  const int slots = NS <= PIPE_XCD_CUS ? (pipes + 7) / 8 * NS : (NS + XS - 1) / XS;
  int dev = 0, cus = 0, per_cu = 0, coop = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice") ||
      check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev),
                "hipDeviceGetAttribute(CUs)") ||
      check_hip(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev),
                "hipDeviceGetAttribute(cooperative)") ||
      check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, P::NT, lds_bytes),
                "hipOccupancyMaxActiveBlocksPerMultiprocessor(gen_pipe)"))
    return MVN_ERR_LAUNCH;
  if (per_cu < 1 || slots * 8 > per_cu * cus) {
    set_error("PIPE variant: grid of %d workgroups is not co-resident (%d per CU x %d CUs)",
              slots * 8, per_cu, cus);
    return MVN_ERR_UNSUPPORTED;
  }
  // Every polled word is re-initialised by memset nodes ahead of each launch: the granules
  // and the placement words -- NOT the 16 flag words between them (the sticky status word).
  const size_t gran_floats = (size_t)batch * NS * P::GRAN * 2;
  if (gran_floats > status_off || status_off + 16 + (size_t)batch * NS > hand_total) {
    set_error("PIPE variant: hand-off area too small");
    return MVN_ERR_BAD_ARG;
  }
  unsigned *err = (unsigned *)(hand + status_off);
  const size_t tail_floats = hand_total - status_off - 16;
  rc = check_hip(hipMemsetAsync(hand, 0, gran_floats * sizeof(float), s), "hipMemsetAsync(granules)");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(err + 16, 0, tail_floats * sizeof(float), s),
                 "hipMemsetAsync(placement words)");
  if (rc) return rc;
  u64 *gran = (u64 *)hand;
  GenArgs args = a;
  int nb = pipes, nseq = batch;
  void *kargs[] = {(void *)&args, (void *)&gran, (void *)&err, (void *)&NS, (void *)&nb, (void *)&nseq};
  if (coop && pipe_cooperative_launch())
    return check_hip(hipLaunchCooperativeKernel(fn, dim3(slots * 8), dim3(P::NT), kargs,
                                                (unsigned)lds_bytes, s),
                     "mvn_generate(pipe, cooperative launch)");
  return check_hip(hipLaunchKernel(fn, dim3(slots * 8), dim3(P::NT), kargs, lds_bytes, s), "mvn_generate(pipe)");
}

int pipe_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_total, size_t status_off,
                hipStream_t s) {
  const int NS = pipe_stages(d);
  int dev = 0, cus = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return MVN_ERR_LAUNCH;
  if (check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev),
                "hipDeviceGetAttribute"))
    return MVN_ERR_LAUNCH;
  if (cus < 8 * PIPE_XCD_CUS || batch > pipe_max_batch(d)) {
    set_error("PIPE variant: %d stages per pipeline, %d pipelines of at most %d sequences each on %d CUs "
              "(batch %d asked for)", NS, cus < 8 * PIPE_XCD_CUS ? 0 : pipe_pipelines(d),
              pipe_max_batch(d) / std::max(1, pipe_pipelines(d)), cus, batch);
    return MVN_ERR_UNSUPPORTED;
  }
  return d->residual_channels == 64 ? pipe_launch_t<64>(a, d, batch, hand, hand_total, status_off, s)
                                    : pipe_launch_t<128>(a, d, batch, hand, hand_total, status_off, s);
}

}  // namespace mvn

extern "C" int mvn_gen_launch_is_cooperative(void) { return mvn::pipe_cooperative_launch() ? 1 : 0; }


#ifdef MVN_PIPE_STAMPS
extern "C" int mvn_debug_read_stamps(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_stamps) / 8) n = sizeof(mvn::g_stamps) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_stamps), n * 8), "read stamps");
}
extern "C" int mvn_debug_read_fine(unsigned long long *out, size_t n) {
  if (n > sizeof(mvn::g_fine) / 8) n = sizeof(mvn::g_fine) / 8;
  return mvn::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(mvn::g_fine), n * 8), "read fine");
}
#endif
