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
#include "common.h"
#include "gen_common.h"

namespace mvn {

typedef unsigned long long u64;
typedef float4 f4;

namespace p64 {
constexpr int C = 64, Q = 256, NT = 512, LPS = 4;
constexpr int MAT_F = 8192;                  // one 128x64 matrix
constexpr int LAYER_F = 3 * MAT_F + 128;     // WC | WP | WR | brs
constexpr int EMB_F = 2 * Q * C;
constexpr int W1_F = Q * C, W2_F = Q * Q;
constexpr int HEAD_F = W1_F + Q + W2_F + Q;
constexpr int CTX_LAYER_F = MAT_F + 128;     // context section per layer: Wcf|Wcg rows + biases
constexpr int GRAN = 128;                    // granules per inbox
constexpr int LDS_FLOATS = 33408;            // max(layer stage, head stage) + 16 flag words
constexpr unsigned SPIN_LIMIT = 1u << 23;
}  // namespace p64

#ifdef MVN_PIPE_STAMPS
// Diagnostic build only (python -m movenet_amd.csrc.build --stamps): wall-clock
// (s_memrealtime, 100 MHz) stamps of "inbox complete" and "outbox sent" per stage
// for the first 64 steps of a launch; read back with mvn_debug_read_stamps().
__device__ unsigned long long g_stamps[16][16][64][4];
#define MVN_STAMP(bb, ss, step, which)                                              \
  do {                                                                              \
    if ((bb) < 16 && (ss) < 16 && (step) < 64 && threadIdx.x == 0) {                \
      g_stamps[bb][ss][step][which] = __builtin_amdgcn_s_memrealtime();             \
      g_stamps[bb][ss][step][2 + (which)] = __builtin_amdgcn_s_memtime();           \
    }                                                                               \
  } while (0)
// finer shader-clock stamps inside the first layer of a stage (thread `who`)
__device__ unsigned long long g_fine[16][16][64][8];
#define MVN_FINE(bb, ss, step, slot, who)                                           \
  do {                                                                              \
    if ((bb) < 16 && (ss) < 16 && (step) < 64 && threadIdx.x == (who))              \
      g_fine[bb][ss][step][slot] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
#else
#define MVN_STAMP(bb, ss, step, which) do {} while (0)
#define MVN_FINE(bb, ss, step, slot, who) do {} while (0)
#endif

// `same_xcd`: producer and consumer were FOUND (from HW_REG_XCC_ID, exchanged at kernel
// start) to sit on one XCD.  They then share one L2, so a plain store (write-through L1,
// line kept in that L2) is seen by the consumer's L1-bypassing polls: ~0.33 us per hop
// instead of ~0.6.  Otherwise the granule is stored sc1 (write-through to memory), the
// placement-independent form.  Placement only ever selects between two correct forms.
__device__ __forceinline__ void put_granule(u64 *g, unsigned epoch, float v, bool same_xcd) {
  const u64 x = ((u64)epoch << 32) | (u64)__float_as_uint(v);
  if (same_xcd)
    *g = x;
  else
    __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0 only.  Lane i owns granules 2i and 2i+1 of the inbox.  Returns false on
// time-out / raised error word (wave-uniform).
__device__ __forceinline__ bool wait_inbox(const u64 *in, unsigned epoch, unsigned *err, float &v0,
                                           float &v1) {
  const int lane = threadIdx.x & 63;
  for (unsigned spins = 1;; ++spins) {
    const u64 g0 = __hip_atomic_load(in + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 g1 = __hip_atomic_load(in + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = (unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch;
    if (__all(ok)) {
      v0 = __uint_as_float((unsigned)g0);
      v1 = __uint_as_float((unsigned)g1);
      return true;
    }
    if ((spins & 255u) == 0) {
      const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || spins > p64::SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---- cross-lane moves as DPP (one VALU op) instead of ds_bpermute (an LDS round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each group of 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each row of 16

// sum over each aligned group of 4 lanes, result in all 4
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<DPP_XOR1>(v);
  v += dpp_mov<DPP_XOR2>(v);
  return v;
}
// after quad_sum: the value held by the OTHER quad of the same group of 8
__device__ __forceinline__ float other_quad(float v) { return dpp_mov<DPP_HALF_MIRROR>(v); }

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = quad_sum(v);
  v += dpp_mov<DPP_HALF_MIRROR>(v);
  v += dpp_mov<DPP_MIRROR>(v);  // every lane of a row of 16 now holds the row sum
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// tanh(f) * sigmoid(g) from two v_exp_f32 and one reciprocal-based division:
//   tanh(f) = (1 - e^-2|f|) / (1 + e^-2|f|) * sign(f),  sigmoid(g) = 1 / (1 + e^-g)
// Absolute error ~1e-7 (fp32 rounding of an O(1) value); ~12 VALU ops vs ~100 for
// the libm forms, and this sits on the per-layer critical path.
__device__ __forceinline__ float gate_fast(float f, float g) {
  const float a = __expf(-2.0f * fabsf(f));       // in (0, 1]
  const float e = __expf(-g);                      // may overflow to +inf: 1/inf = 0 is right
  const float num = copysignf(1.0f - a, f);
  const float den = (1.0f + a) * (1.0f + e);
  return num * __builtin_amdgcn_rcpf(den);  // v_rcp_f32: 1 ulp
}

typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef MVN_EXP
#define MVN_EXP 0   // timing experiments of scripts/pipe_stamps.py; 0 = the product
#endif

// 16-term dot product as 8 packed FMAs (v_pk_fma_f32) in two independent chains,
// combined in a fixed order.  w: 8 float2, x: 16 consecutive floats in LDS (already
// fetched as four float4 by the caller).
__device__ __forceinline__ float dot16(const v2f (&w)[8], const f4 (&x)[4]) {
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a0 = __builtin_elementwise_fma(w[2 * i], v2f{x[i].x, x[i].y}, a0);
    a1 = __builtin_elementwise_fma(w[2 * i + 1], v2f{x[i].z, x[i].w}, a1);
  }
  const v2f t = a0 + a1;
  return t.x + t.y;
}
// 32-term dot product as 16 packed FMAs in four chains (head)
__device__ __forceinline__ float dot32(const v2f (&w)[16], const float *x) {
  const f4 *x4 = (const f4 *)x;
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const f4 xv = x4[i], yv = x4[i + 1];
    a0 = __builtin_elementwise_fma(w[2 * i], v2f{xv.x, xv.y}, a0);
    a1 = __builtin_elementwise_fma(w[2 * i + 1], v2f{xv.z, xv.w}, a1);
    a2 = __builtin_elementwise_fma(w[2 * i + 2], v2f{yv.x, yv.y}, a2);
    a3 = __builtin_elementwise_fma(w[2 * i + 3], v2f{yv.z, yv.w}, a3);
  }
  const v2f t = (a0 + a1) + (a2 + a3);
  return t.x + t.y;
}
__device__ __forceinline__ void load32(v2f (&w)[16], const f4 *src, int stride, int idx) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f4 v = src[i * stride + idx];
    w[2 * i] = v2f{v.x, v.y};
    w[2 * i + 1] = v2f{v.z, v.w};
  }
}
// two rows x 16 inputs: wa <- float4 0..3, wb <- float4 4..7 of a [8][stride] block
__device__ __forceinline__ void load2x16(v2f (&wa)[8], v2f (&wb)[8], const f4 *src, int stride,
                                         int idx) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f4 u = src[i * stride + idx], v = src[(4 + i) * stride + idx];
    wa[2 * i] = v2f{u.x, u.y};
    wa[2 * i + 1] = v2f{u.z, u.w};
    wb[2 * i] = v2f{v.x, v.y};
    wb[2 * i + 1] = v2f{v.z, v.w};
  }
}

// (value, index) arg-max combine: larger value wins, smaller index on ties
__device__ __forceinline__ void argmax_take(float &bv, int &bi, float ov, int oi) {
  if (ov > bv || (ov == bv && oi < bi)) {
    bv = ov;
    bi = oi;
  }
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Workgroup = 8 waves.  Waves 0-3 ("FG group") own the filter/gate matrices, waves
// 4-7 ("RS group") the residual/skip matrices: at any moment ONE wave per SIMD is
// issuing, so the dependent chain is not slowed by a co-resident wave replaying the
// same bookkeeping instructions, and each thread keeps 4 layers x 32 weights = 128
// VGPRs resident.
__global__ __launch_bounds__(512, 2) void gen_pipe64_kernel(GenArgs a, u64 *hand, unsigned *err,
                                                           int NS, int nb) {
  using namespace p64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups i and i+8 are observed to land on the same XCD: lay the NS stages of a
  // pipeline out with stride 8 so that its hops stay inside one L2 (speed only; every
  // edge verifies its placement below).
  const int slot = blockIdx.x >> 3;
  const int b = (blockIdx.x & 7) + 8 * (slot / NS), s = slot - (slot / NS) * NS;
  if (b >= nb) return;
  const int L = a.L;
  const int s_next = s + 1 == NS ? 0 : s + 1;
  u64 *inbox = hand + ((size_t)b * NS + s) * GRAN;
  u64 *outbox = hand + ((size_t)b * NS + s_next) * GRAN;
  int *iflag = (int *)(smem + LDS_FLOATS - 16);  // [0] ok flag, [1] idx_cur, [2] idx_prev
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
    // FG group thread t: channel c = t>>2, rows f_c and g_c, inputs k in [16kq, 16kq+16);
    // RS group thread t: channel c = t>>2, rows res_c and skip_c, same k split.  A row is
    // finished by a 4-lane DPP sum; each lane fetches 16 inputs (4 x ds_read_b128).
    const int l0 = s * LPS, nl = min(LPS, L - l0);
    const bool fg_group = tid < 256;
    const int t = tid & 255, c = t >> 2, kq = t & 3;
    const bool lead = kq == 0;
    float *wp = smem;                  // [LPS][8][256] float4: past-tap f|g weights (FG group)
    float *cur = smem + LPS * MAT_F;   // [64] residual stream
    float *zb = cur + 64;              // [64] gated activation
    float *pastb = zb + 64;            // [LPS][64] popped queue entries
    float *skin = pastb + LPS * 64;    // [64] running skip sum as received
    float *ctxb = skin + 64;           // [64] context vector of the step being prepared
    float *ring = a.state + (size_t)b * a.state_per_seq;

    v2f wa[LPS][8], wb[LPS][8];        // FG: f_c | g_c current-tap rows; RS: res_c | skip_c rows
    float bias_r[LPS], bias_s[LPS], pf[LPS], pg[LPS], xs[LPS];
    int doff[LPS], dmask[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
      bias_r[j] = 0.f; bias_s[j] = 0.f; pf[j] = 0.f; pg[j] = 0.f; xs[j] = 0.f;
      doff[j] = 0; dmask[j] = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { wa[j][i] = v2f{0.f, 0.f}; wb[j][i] = v2f{0.f, 0.f}; }
      if (j < nl) {
        const float *lw = a.w + EMB_F + (size_t)(l0 + j) * LAYER_F;
        if (fg_group) {
          load2x16(wa[j], wb[j], (const f4 *)lw, 256, t);
          const f4 *wp4 = (const f4 *)(lw + MAT_F);
#pragma unroll
          for (int i = 0; i < 8; ++i) ((f4 *)wp)[(j * 8 + i) * 256 + t] = wp4[i * 256 + t];
        } else {
          load2x16(wa[j], wb[j], (const f4 *)(lw + 2 * MAT_F), 256, t);
          bias_r[j] = lw[3 * MAT_F + c];
          bias_s[j] = lw[3 * MAT_F + 64 + c];
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
      if (!fg_group && lead) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + c;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j] : ring_load(base + (tn & dmask[j]) * C);
            pastb[j * 64 + c] = pv;
          }
      }
      if (a.ctx_tm && fg_group && t < 64)
        ctxb[t] = a.ctx_tm[(size_t)b * a.ctx_stride_b + (size_t)tn * C + t];
      __syncthreads();
      if (fg_group) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            v2f qa[8], qb[8];
            load2x16(qa, qb, (const f4 *)wp + j * 8 * 256, 256, t);
            const f4 *x4 = (const f4 *)(pastb + j * 64 + 16 * kq);
            const f4 x[4] = {x4[0], x4[1], x4[2], x4[3]};
            pf[j] = quad_sum(dot16(qa, x));
            pg[j] = quad_sum(dot16(qb, x));
            if (a.ctx_tm) {
              // 1x1 context convs (modules.py:58-63, :75-77): their weights are streamed
              // from L2 here, off the critical path (32 KB per layer per step)
              const float *wc = a.wctx + (size_t)(l0 + j) * CTX_LAYER_F;
              load2x16(qa, qb, (const f4 *)wc, 256, t);
              const f4 *c4 = (const f4 *)(ctxb + 16 * kq);
              const f4 cx[4] = {c4[0], c4[1], c4[2], c4[3]};
              pf[j] += quad_sum(dot16(qa, cx)) + wc[MAT_F + c];
              pg[j] += quad_sum(dot16(qb, cx)) + wc[MAT_F + 64 + c];
            }
          }
      }
    };
    __syncthreads();
    precompute(a.t_begin, false);

    for (int ts = a.t_begin; ts < a.t_end; ++ts) {
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      if (wave == 0) {
        float v0, v1;
        const bool ok = wait_inbox(inbox, epoch, err, v0, v1);
        if (ok) {
          // granules 0..63 residual stream, 64..127 running skip sum
          float *dst = lane < 32 ? cur : skin - 64;
          dst[2 * lane] = v0;
          dst[2 * lane + 1] = v1;
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      float skipacc = (!fg_group && lead) ? skin[c] : 0.f;
#pragma unroll
      for (int j = 0; j < LPS; ++j)
        if (j < nl) {
          float old = 0.f;
          if (fg_group) {
            const f4 *x4 = (const f4 *)(cur + 16 * kq);
            const f4 x[4] = {x4[0], x4[1], x4[2], x4[3]};
            const float f = quad_sum(dot16(wa[j], x)) + pf[j];
            const float g = quad_sum(dot16(wb[j], x)) + pg[j];
#if MVN_EXP == 1
            const float z = f + g;
#else
            const float z = gate_fast(f, g);
#endif
            if (lead) zb[c] = z;
          } else if (lead) {
            old = cur[c];  // this layer's input: residual add below, queue push later
          }
          lds_barrier();
          if (!fg_group) {
            const f4 *z4 = (const f4 *)(zb + 16 * kq);
            const f4 x[4] = {z4[0], z4[1], z4[2], z4[3]};
            const float r = quad_sum(dot16(wa[j], x));
            const float k = quad_sum(dot16(wb[j], x));
            if (lead) {
              xs[j] = old;
              const float outv = (r + bias_r[j]) + old;
              cur[c] = outv;
              skipacc += k + bias_s[j];
              if (j == nl - 1) {
                // the stage's last layer: hand the activation on before anything else
                put_granule(outbox + c, epoch, outv, fast_edge);
                put_granule(outbox + 64 + c, epoch, skipacc, fast_edge);
              }
            }
          }
          lds_barrier();
        }
      MVN_STAMP(b, s, ts - a.t_begin, 1);
      if (iflag[0] == 0) break;  // hand-off timed out (checked after the step: off the chain)
      if (ts + 1 < a.t_end) {
        precompute(ts + 1, true);
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
    float *E0 = smem, *E1 = smem + Q * C;   // [Q][C] each
    float *a0 = smem + EMB_F;                // [64]
    float *a1 = a0 + 64;                     // [256]
    float *lgb = a1 + 256;                   // [256] logits
    const float *hw = a.w + EMB_F + (size_t)L * LAYER_F;
    const f4 *W1p = (const f4 *)hw, *W2p = (const f4 *)(hw + W1_F + Q);
    const float *b1 = hw + W1_F, *b2 = hw + W1_F + Q + W2_F;
    int32_t *samples = a.samples + (size_t)b * a.stride;

    {
      const f4 *src = (const f4 *)a.w;
      f4 *dst = (f4 *)E0;
      for (int i = tid; i < EMB_F / 4; i += NT) dst[i] = src[i];
    }
    // conv1: thread (o1 = tid>>1, q1 = tid&1), 32 inputs; conv2: thread (og = tid>>3, q2 = tid&7),
    // 4 outputs x 32 inputs
    const int o1 = tid >> 1, q1 = tid & 1, og = tid >> 3, q2 = tid & 7;
    v2f w1[16], w2[4][16];
    load32(w1, W1p, NT, tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) load32(w2[r], W2p + r * 8 * NT, NT, tid);
    const float b1r = b1[o1];
    const float b2r = b2[4 * og + (q2 & 3)];
    __syncthreads();

    // Wave 0 closes every step alone (softmax, choice) and immediately opens the next
    // one: it gathers the two embedding rows of the causal conv (modules.py:28-30 on a
    // one-hot input) and hands them to stage 0, so no barrier sits between the choice
    // and the next step's first hop.
    int idx_cur = 0, idx_prev = -1;
    auto send_h0 = [&](unsigned ep) {  // wave 0: granule `lane` = residual, 64+lane = skip sum 0
      const int ic = min(max(idx_cur, 0), Q - 1), ip = min(idx_prev, Q - 1);
      float v = E1[ic * C + lane];
      if (ip >= 0) v += E0[ip * C + lane];
      put_granule(outbox + lane, ep, v, fast_edge);
      put_granule(outbox + 64 + lane, ep, 0.f, fast_edge);
    };
    if (wave == 0) {
      idx_cur = samples[a.t_begin];
      idx_prev = a.t_begin > 0 ? samples[a.t_begin - 1] : -1;
      if (a.t_begin < a.t_end) send_h0(1u);
      MVN_STAMP(b, s, 0, 1);
    }

    for (int ts = a.t_begin; ts < a.t_end; ++ts) {
      const unsigned epoch = (unsigned)(ts - a.t_begin + 1);
      const int u = ts + 1;
      const bool want_out = (a.logits_out || a.choices_out) && u >= a.logits_t0;
      const bool do_head = u < a.n_total && (u >= a.n_given || want_out);  // block-uniform
      int next_idx = 0;
      if (wave == 0) {
        if (u < a.n_given) next_idx = samples[u];  // prompt / teacher forcing
        float v0, v1;
        const bool ok = wait_inbox(inbox, epoch, err, v0, v1);
        if (ok && lane >= 32) {
          a0[2 * lane - 64] = leaky(v0);
          a0[2 * lane - 63] = leaky(v1);
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      MVN_STAMP(b, s, ts - a.t_begin, 0);
      if (do_head) {
        {
          float hsum = dot32(w1, a0 + 32 * q1);
          hsum += dpp_mov<DPP_XOR1>(hsum);
          if (q1 == 0) a1[o1] = leaky(hsum + b1r);
        }
        lds_barrier();
        {
          float s0 = dot32(w2[0], a1 + 32 * q2), s1 = dot32(w2[1], a1 + 32 * q2);
          float s2 = dot32(w2[2], a1 + 32 * q2), s3 = dot32(w2[3], a1 + 32 * q2);
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
          if (a.logits_out && u >= a.logits_t0)
            ((f4 *)(a.logits_out +
                    ((size_t)b * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * Q))[lane] = lv;
          const float m = wave_max_dpp(fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3])));
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = __expf(lg[k] - m);
          const float sm = wave_sum_dpp((e[0] + e[1]) + (e[2] + e[3]));
          // v_exp_f32 / v_rcp_f32 forms (1-2 ulp): the choice depends on the ORDER of
          // the probabilities, which these monotone maps preserve
          const float rs = __builtin_amdgcn_rcpf(sm) *
                           (a.temperature > 0.f ? __builtin_amdgcn_rcpf(a.temperature) : 1.0f);
          float p[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) p[k] = e[k] * rs;
          const float m2 = wave_max_dpp(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])));
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = __expf(p[k] - m2);
          const float s2sum = wave_sum_dpp((e[0] + e[1]) + (e[2] + e[3]));
          const float rs2 = __builtin_amdgcn_rcpf(s2sum);
#pragma unroll
          for (int k = 0; k < 4; ++k) p[k] = e[k] * rs2;  // the distribution generate() uses

          int pick;
          if (a.temperature > 0.f) {
            const float lsum = (p[0] + p[1]) + (p[2] + p[3]);
            float incl = lsum;  // inclusive scan of lane totals, class order
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
              const float n = __shfl_up(incl, off, 64);
              if (lane >= off) incl += n;
            }
            const float total = lane_value(incl, 63);
            const float target = philox_uniform(a.seed, (uint32_t)u, (uint32_t)b) * total;
            const float cdf = incl - lsum;
            int cand = Q - 1;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
              // walk down so that the smallest qualifying class wins
              const float c_k = cdf + (k == 0 ? p[0] : k == 1 ? p[0] + p[1]
                                                      : k == 2 ? (p[0] + p[1]) + p[2]
                                                               : ((p[0] + p[1]) + p[2]) + p[3]);
              if (c_k > target) cand = 4 * lane + k;
            }
            cand = min(cand, dpp_movi<DPP_XOR1>(cand));
            cand = min(cand, dpp_movi<DPP_XOR2>(cand));
            cand = min(cand, dpp_movi<DPP_HALF_MIRROR>(cand));
            cand = min(cand, dpp_movi<DPP_MIRROR>(cand));
            pick = min(min(__builtin_amdgcn_readlane(cand, 0), __builtin_amdgcn_readlane(cand, 16)),
                       min(__builtin_amdgcn_readlane(cand, 32), __builtin_amdgcn_readlane(cand, 48)));
          } else {
            float bv = p[0];
            int bi = 4 * lane;
#pragma unroll
            for (int k = 1; k < 4; ++k)
              if (p[k] > bv) {  // strict: first maximum
                bv = p[k];
                bi = 4 * lane + k;
              }
            argmax_take(bv, bi, dpp_mov<DPP_XOR1>(bv), dpp_movi<DPP_XOR1>(bi));
            argmax_take(bv, bi, dpp_mov<DPP_XOR2>(bv), dpp_movi<DPP_XOR2>(bi));
            argmax_take(bv, bi, dpp_mov<DPP_HALF_MIRROR>(bv), dpp_movi<DPP_HALF_MIRROR>(bi));
            argmax_take(bv, bi, dpp_mov<DPP_MIRROR>(bv), dpp_movi<DPP_MIRROR>(bi));
            float rv = lane_value(bv, 0);
            pick = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
            for (int row = 16; row < 64; row += 16)
              argmax_take(rv, pick, lane_value(bv, row), __builtin_amdgcn_readlane(bi, row));
          }
          if (u >= a.n_given) next_idx = pick;
          idx_prev = idx_cur;
          idx_cur = next_idx;
          if (ts + 1 < a.t_end) send_h0(epoch + 1);
          MVN_STAMP(b, s, ts + 1 - a.t_begin, 1);
          if (lane == 0) {
            if (a.choices_out && u >= a.logits_t0) a.choices_out[(size_t)b * a.n_total + u] = pick;
            if (u >= a.n_given) samples[u] = pick;
          }
        } else {
          idx_prev = idx_cur;
          idx_cur = next_idx;
          if (ts + 1 < a.t_end) send_h0(epoch + 1);
          MVN_STAMP(b, s, ts + 1 - a.t_begin, 1);
        }
      }
      if (iflag[0] == 0) break;  // hand-off timed out
    }
  }
}

// ---- packing: state_dict layouts -> per-thread register order -----------------
__global__ void pack_layer_p64_kernel(const float *fw, const float *gw, const float *rw,
                                      const float *rb, const float *sw, const float *sb,
                                      float *__restrict__ dst) {
  using namespace p64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LAYER_F) return;
  if (i >= 3 * MAT_F) {
    const int o = i - 3 * MAT_F;
    dst[i] = o < C ? rb[o] : sb[o - C];
    return;
  }
  // each matrix: [i8 (8)][t (256)] float4; thread t = 4*c + kq owns rows (c, 64+c) x inputs
  // k = 16*kq + 4*(i8 & 3) + e: float4 0..3 belong to row c, 4..7 to row 64+c
  const int region = i / MAT_F, r = i - region * MAT_F;
  const int e = r & 3, v = r >> 2, t = v & 255, i8 = v >> 8;
  const int row = (i8 >> 2) * 64 + (t >> 2), k = 16 * (t & 3) + 4 * (i8 & 3) + e;
  if (region < 2)
    dst[i] = fg_elem(fw, gw, C, row, region == 0 ? 64 + k : k);  // WC: current tap, WP: past tap
  else
    dst[i] = rs_elem(rw, sw, C, row, k);
}

__global__ void pack_head_p64_kernel(const float *w1, const float *b1, const float *w2,
                                     const float *b2, float *__restrict__ dst) {
  using namespace p64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W1_F) {
    const int e = i & 3, v = i >> 2, tid = v & (NT - 1), i8 = v >> 9;
    dst[i] = w1[(size_t)(tid >> 1) * C + 32 * (tid & 1) + 4 * i8 + e];
  } else if (i < W1_F + Q) {
    dst[i] = b1[i - W1_F];
  } else if (i < W1_F + Q + W2_F) {
    const int ii = i - W1_F - Q;
    const int e = ii & 3, v = ii >> 2, tid = v & (NT - 1), rest = v >> 9, r = rest >> 3, i8 = rest & 7;
    dst[i] = w2[(size_t)(4 * (tid >> 3) + r) * Q + 32 * (tid & 7) + 4 * i8 + e];
  } else if (i < HEAD_F) {
    dst[i] = b2[i - W1_F - Q - W2_F];
  }
}

__global__ void pack_embed_p64_kernel(const float *__restrict__ causal_w, float *__restrict__ dst) {
  using namespace p64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= EMB_F) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, qq = r / C, c = r - qq * C;
  dst[i] = causal_w[((size_t)c * Q + qq) * 2 + tap];
}

__global__ void pack_ctx_p64_kernel(const float *wcf, const float *bcf, const float *wcg,
                                    const float *bcg, float *__restrict__ dst) {
  using namespace p64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CTX_LAYER_F) return;
  if (i >= MAT_F) {
    const int o = i - MAT_F;
    dst[i] = o < C ? bcf[o] : bcg[o - C];
    return;
  }
  // same thread mapping as the current-tap matrix: thread t = 4*c + kq owns rows (c, 64+c)
  const int e = i & 3, v = i >> 2, t = v & 255, i8 = v >> 8;
  const int c = t >> 2, k = 16 * (t & 3) + 4 * (i8 & 3) + e;
  dst[i] = (i8 >> 2) ? wcg[(size_t)c * C + k] : wcf[(size_t)c * C + k];
}

int pipe_pack_ctx(const mvn_dims *d, const mvn_params *p, float *ctx_section, hipStream_t s) {
  using namespace p64;
  for (int l = 0; l < n_layers(d); ++l)
    hipLaunchKernelGGL(pack_ctx_p64_kernel, dim3((CTX_LAYER_F + 255) / 256), dim3(256), 0, s,
                       p->ctx_filter_w[l], p->ctx_filter_b[l], p->ctx_gate_w[l], p->ctx_gate_b[l],
                       ctx_section + (size_t)l * CTX_LAYER_F);
  return check_hip(hipGetLastError(), "pipe_pack_ctx");
}

bool pipe_ok(const mvn_dims *d) {
  return d->residual_channels == 64 && d->skip_channels == 64 && d->input_channels == 256 &&
         n_layers(d) >= 1;
}
int pipe_stages(const mvn_dims *d) { return (n_layers(d) + p64::LPS - 1) / p64::LPS + 1; }
size_t pipe_hand_floats(const mvn_dims *d, int batch) {
  // batch * NS inboxes of 128 granules (2 floats each), then 16 flag words (error word
  // first) and batch * NS placement words, padded to 64 floats
  const size_t n = (size_t)batch * pipe_stages(d);
  return n * p64::GRAN * 2 + (16 + n + 63) / 64 * 64;
}

int pipe_pack(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s) {
  using namespace p64;
  const int L = n_layers(d);
  hipLaunchKernelGGL(pack_embed_p64_kernel, dim3((EMB_F + 255) / 256), dim3(256), 0, s, p->causal_w,
                     packed);
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL(pack_layer_p64_kernel, dim3((LAYER_F + 255) / 256), dim3(256), 0, s,
                       p->filter_w[l], p->gate_w[l], p->residual_w[l], p->residual_b[l], p->skip_w[l],
                       p->skip_b[l], packed + EMB_F + (size_t)l * LAYER_F);
  hipLaunchKernelGGL(pack_head_p64_kernel, dim3((HEAD_F + 255) / 256), dim3(256), 0, s, p->head1_w,
                     p->head1_b, p->head2_w, p->head2_b, packed + EMB_F + (size_t)L * LAYER_F);
  return check_hip(hipGetLastError(), "pipe_pack");
}

int pipe_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, hipStream_t s) {
  using namespace p64;
  const int NS = pipe_stages(d);
  int dev = 0, cus = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return MVN_ERR_LAUNCH;
  if (check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev),
                "hipDeviceGetAttribute"))
    return MVN_ERR_LAUNCH;
  if (batch * NS > cus) {
    set_error("PIPE variant needs batch*stages = %d*%d workgroups co-resident, device has %d CUs",
              batch, NS, cus);
    return MVN_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void *)gen_pipe64_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                       "hipFuncSetAttribute(gen_pipe64)");
    if (rc) return rc;
    attr_set = true;
  }
  const size_t hand_bytes = pipe_hand_floats(d, batch) * sizeof(float);
  // every polled word is re-initialised by a memset node ahead of each launch
  int rc = check_hip(hipMemsetAsync(hand, 0, hand_bytes, s), "hipMemsetAsync(hand-off area)");
  if (rc) return rc;
  u64 *gran = (u64 *)hand;
  unsigned *err = (unsigned *)(hand + (size_t)batch * NS * GRAN * 2);
  // grid: ceil(batch/8) groups of NS slots, 8 workgroups (one per XCD) per slot
  const int groups = (batch + 7) / 8;
  hipLaunchKernelGGL(gen_pipe64_kernel, dim3(groups * NS * 8), dim3(NT), LDS_FLOATS * sizeof(float),
                     s, a, gran, err, NS, batch);
  return check_hip(hipGetLastError(), "mvn_generate(pipe)");
}

}  // namespace mvn

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
