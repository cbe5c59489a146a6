// PIPE variant of the generator: a layer pipeline across CUs with RESIDENT weights.
//
// The STREAM variant re-reads 3.3 MB of weights from L2 for every generated
// sample (one CU per sequence, ~42 GB/s => 78 us per step).  Here each sequence
// gets a private pipeline of NS = ceil(L/4)+1 workgroups (512 threads, one per
// CU): stage s keeps the weights of layers 4s..4s+3 in registers (current-tap
// and residual/skip matrices, 128 VGPRs per thread) and LDS (past-tap matrix),
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
constexpr int GRAN = 128;                    // granules per inbox
constexpr int LDS_FLOATS = 33280;            // max(layer stage, head stage), see kernel
constexpr unsigned SPIN_LIMIT = 1u << 23;
}  // namespace p64

__device__ __forceinline__ void put_granule(u64 *g, unsigned epoch, float v) {
  __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
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

__device__ __forceinline__ float dot16(const float (&w)[16], const float *x) {
  const f4 *x4 = (const f4 *)x;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i += 2) {
    const f4 xa = x4[i], xb = x4[i + 1];
    a0 = fmaf(w[4 * i + 0], xa.x, a0);
    a0 = fmaf(w[4 * i + 1], xa.y, a0);
    a0 = fmaf(w[4 * i + 2], xa.z, a0);
    a0 = fmaf(w[4 * i + 3], xa.w, a0);
    a1 = fmaf(w[4 * i + 4], xb.x, a1);
    a1 = fmaf(w[4 * i + 5], xb.y, a1);
    a1 = fmaf(w[4 * i + 6], xb.z, a1);
    a1 = fmaf(w[4 * i + 7], xb.w, a1);
  }
  return a0 + a1;
}

__global__ __launch_bounds__(512, 2) void gen_pipe64_kernel(GenArgs a, u64 *hand, unsigned *err,
                                                           int NS) {
  using namespace p64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / NS, s = blockIdx.x - b * NS;
  const int L = a.L;
  u64 *inbox = hand + ((size_t)b * NS + s) * GRAN;
  u64 *outbox = hand + ((size_t)b * NS + (s + 1 == NS ? 0 : s + 1)) * GRAN;
  int *iflag = (int *)(smem + LDS_FLOATS - 16);  // [0] ok flag, [1] idx_cur, [2] idx_prev

  if (s < NS - 1) {
    // ================= layer stage: layers l0 .. l0+nl-1 =================
    const int l0 = s * LPS, nl = min(LPS, L - l0);
    const int q = tid & 3;
    const int fc = tid >> 3, which = (tid >> 2) & 1;  // f/g mapping: channel, f|g, k-quarter
    const int o = tid >> 2;                           // r/s mapping: row (res 0..63 | skip 64..127)
    float *wp = smem;                  // [LPS][4][512] float4: past-tap f|g weights
    float *cur = smem + LPS * MAT_F;   // [64]
    float *zb = cur + 64;              // [64]
    float *pastb = zb + 64;            // [LPS][64]
    float *skin = pastb + LPS * 64;    // [64] running skip sum as received
    float *ring = a.state + (size_t)b * a.state_per_seq;

    float wc[LPS][16], wr[LPS][16], bias[LPS], pj[LPS], xs[LPS];
    int doff[LPS], dmask[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
      bias[j] = 0.f; pj[j] = 0.f; xs[j] = 0.f; doff[j] = 0; dmask[j] = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) { wc[j][i] = 0.f; wr[j][i] = 0.f; }
      if (j < nl) {
        const float *lw = a.w + EMB_F + (size_t)(l0 + j) * LAYER_F;
        const f4 *wc4 = (const f4 *)lw, *wp4 = (const f4 *)(lw + MAT_F),
                 *wr4 = (const f4 *)(lw + 2 * MAT_F);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f4 vc = wc4[i * NT + tid], vr = wr4[i * NT + tid];
          wc[j][4 * i] = vc.x; wc[j][4 * i + 1] = vc.y; wc[j][4 * i + 2] = vc.z; wc[j][4 * i + 3] = vc.w;
          wr[j][4 * i] = vr.x; wr[j][4 * i + 1] = vr.y; wr[j][4 * i + 2] = vr.z; wr[j][4 * i + 3] = vr.w;
          ((f4 *)wp)[(j * 4 + i) * NT + tid] = wp4[i * NT + tid];
        }
        bias[j] = lw[3 * MAT_F + o];
        const int l = l0 + j;
        dmask[j] = (1 << (l % a.layer_size)) - 1;
        doff[j] = ring_offset(l, a.layer_size, C);
      }
    }

    // queue pop for step tn (+ push of this step's inputs) and the past-tap half
    // of the f/g pre-activations; runs while the other stages work
    auto precompute = [&](int tn, bool push) {
      if (q == 0 && o < 64) {
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) {
            float *base = ring + doff[j] + o;
            if (push) base[((tn - 1) & dmask[j]) * C] = xs[j];
            const float pv = (push && dmask[j] == 0) ? xs[j] : ring_load(base + (tn & dmask[j]) * C);
            pastb[j * 64 + o] = pv;
          }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < LPS; ++j)
        if (j < nl) {
          float w[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f4 v = ((const f4 *)wp)[(j * 4 + i) * NT + tid];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
          }
          float p = dot16(w, pastb + j * 64 + 16 * q);
          p += __shfl_xor(p, 1, 64);
          p += __shfl_xor(p, 2, 64);
          pj[j] = p;
        }
    };
    __syncthreads();
    precompute(a.t_begin, false);

    for (int t = a.t_begin; t < a.t_end; ++t) {
      const unsigned epoch = (unsigned)(t - a.t_begin + 1);
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
      if (iflag[0] == 0) break;
      float skipacc = (q == 0 && o >= 64) ? skin[o - 64] : 0.f;
      float outv = 0.f;
#pragma unroll
      for (int j = 0; j < LPS; ++j)
        if (j < nl) {
          // f,g: current-tap product + precomputed past-tap half
          float p = dot16(wc[j], cur + 16 * q);
          p += __shfl_xor(p, 1, 64);
          p += __shfl_xor(p, 2, 64);
          p += pj[j];
          const float other = __shfl_xor(p, 4, 64);
          const float z = gate(which ? other : p, which ? p : other);
          if ((tid & 7) == 0) zb[fc] = z;
          lds_barrier();
          float r = dot16(wr[j], zb + 16 * q);
          r += __shfl_xor(r, 1, 64);
          r += __shfl_xor(r, 2, 64);
          if (q == 0) {
            const float v = r + bias[j];
            if (o < 64) {
              const float old = cur[o];
              xs[j] = old;
              outv = v + old;
              cur[o] = outv;
            } else {
              skipacc += v;
            }
            // the stage's last layer: hand the activation on before anything else
            if (j == nl - 1) put_granule(outbox + o, epoch, o < 64 ? outv : skipacc);
          }
          lds_barrier();
        }
      if (t + 1 < a.t_end) {
        precompute(t + 1, true);
      } else if (q == 0 && o < 64) {
        // last step of the launch: push only (the next launch pops in its prologue)
#pragma unroll
        for (int j = 0; j < LPS; ++j)
          if (j < nl) ring[doff[j] + o + (t & dmask[j]) * C] = xs[j];
      }
    }
    return;
  }

  // ============================ head stage ============================
  {
    float *E0 = smem, *E1 = smem + Q * C;   // [Q][C] each
    float *a0 = smem + EMB_F;                // [64]
    float *a1 = a0 + 64;                     // [256]
    float *red = a1 + 256;                   // [64]
    int *ired = (int *)(red + 64);           // [16]
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
    float w1[32], w2[4][32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f4 v = W1p[i * NT + tid];
      w1[4 * i] = v.x; w1[4 * i + 1] = v.y; w1[4 * i + 2] = v.z; w1[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f4 v = W2p[(r * 8 + i) * NT + tid];
        w2[r][4 * i] = v.x; w2[r][4 * i + 1] = v.y; w2[r][4 * i + 2] = v.z; w2[r][4 * i + 3] = v.w;
      }
    const float b1r = b1[o1];
    const int cls = 4 * og + (q2 & 3);
    const bool active = q2 < 4;
    const float b2r = b2[cls];
    if (tid == 0) {
      iflag[1] = samples[a.t_begin];
      iflag[2] = a.t_begin > 0 ? samples[a.t_begin - 1] : -1;
    }
    __syncthreads();

    for (int t = a.t_begin; t < a.t_end; ++t) {
      const unsigned epoch = (unsigned)(t - a.t_begin + 1);
      const int u = t + 1;
      // causal conv of the one-hot input = two embedding rows; starts the step
      if (tid < 128) {
        float v = 0.f;
        if (tid < 64) {
          const int ic = min(max(iflag[1], 0), Q - 1), ip = min(iflag[2], Q - 1);
          v = E1[ic * C + tid];
          if (ip >= 0) v += E0[ip * C + tid];
        }
        put_granule(outbox + tid, epoch, v);
      }
      int next_given = 0;
      if (tid == 0 && u < a.n_given) next_given = samples[u];
      if (wave == 0) {
        float v0, v1;
        const bool ok = wait_inbox(inbox, epoch, err, v0, v1);
        if (ok && lane >= 32) {
          a0[2 * lane - 64] = leaky(v0);
          a0[2 * lane - 63] = leaky(v1);
        }
        if (lane == 0) iflag[0] = ok ? 1 : 0;
      }
      lds_barrier();
      if (iflag[0] == 0) break;
      const bool want_out = (a.logits_out || a.choices_out) && u >= a.logits_t0;
      const bool do_head = u < a.n_total && (u >= a.n_given || want_out);  // block-uniform
      int choice = next_given;
      if (do_head) {
        {
          const f4 *x4 = (const f4 *)(a0 + 32 * q1);
          float h0 = 0.f, h1 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            const f4 xa = x4[i], xb = x4[i + 1];
            h0 = fmaf(w1[4 * i], xa.x, h0); h0 = fmaf(w1[4 * i + 1], xa.y, h0);
            h0 = fmaf(w1[4 * i + 2], xa.z, h0); h0 = fmaf(w1[4 * i + 3], xa.w, h0);
            h1 = fmaf(w1[4 * i + 4], xb.x, h1); h1 = fmaf(w1[4 * i + 5], xb.y, h1);
            h1 = fmaf(w1[4 * i + 6], xb.z, h1); h1 = fmaf(w1[4 * i + 7], xb.w, h1);
          }
          float h = h0 + h1;
          h += __shfl_xor(h, 1, 64);
          if (q1 == 0) a1[o1] = leaky(h + b1r);
        }
        lds_barrier();
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        {
          const f4 *x4 = (const f4 *)(a1 + 32 * q2);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f4 x = x4[i];
            s0 = fmaf(w2[0][4 * i], x.x, s0); s0 = fmaf(w2[0][4 * i + 1], x.y, s0);
            s0 = fmaf(w2[0][4 * i + 2], x.z, s0); s0 = fmaf(w2[0][4 * i + 3], x.w, s0);
            s1 = fmaf(w2[1][4 * i], x.x, s1); s1 = fmaf(w2[1][4 * i + 1], x.y, s1);
            s1 = fmaf(w2[1][4 * i + 2], x.z, s1); s1 = fmaf(w2[1][4 * i + 3], x.w, s1);
            s2 = fmaf(w2[2][4 * i], x.x, s2); s2 = fmaf(w2[2][4 * i + 1], x.y, s2);
            s2 = fmaf(w2[2][4 * i + 2], x.z, s2); s2 = fmaf(w2[2][4 * i + 3], x.w, s2);
            s3 = fmaf(w2[3][4 * i], x.x, s3); s3 = fmaf(w2[3][4 * i + 1], x.y, s3);
            s3 = fmaf(w2[3][4 * i + 2], x.z, s3); s3 = fmaf(w2[3][4 * i + 3], x.w, s3);
          }
#pragma unroll
          for (int off = 1; off < 8; off <<= 1) {
            s0 += __shfl_xor(s0, off, 64);
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
            s3 += __shfl_xor(s3, off, 64);
          }
        }
        const int sel = q2 & 3;
        float lg = (sel == 0 ? s0 : sel == 1 ? s1 : sel == 2 ? s2 : s3) + b2r;
        if (active && a.logits_out && u >= a.logits_t0)
          a.logits_out[((size_t)b * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * Q + cls] = lg;
        // softmax -> [/T] -> softmax over the 256 active lanes (32 per wave, class order)
        float m = wave_max(active ? lg : -INFINITY);
        if (lane == 0) red[wave] = m;
        lds_barrier();
        m = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])),
                  fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        const float e = active ? expf(lg - m) : 0.f;
        float sm = wave_sum(e);
        if (lane == 0) red[8 + wave] = sm;
        lds_barrier();
        sm = ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
        float p = e / sm;
        if (a.temperature > 0.f) p = p / a.temperature;
        float m2 = wave_max(active ? p : -INFINITY);
        if (lane == 0) red[16 + wave] = m2;
        lds_barrier();
        m2 = fmaxf(fmaxf(fmaxf(red[16], red[17]), fmaxf(red[18], red[19])),
                   fmaxf(fmaxf(red[20], red[21]), fmaxf(red[22], red[23])));
        const float e2 = active ? expf(p - m2) : 0.f;
        float s2sum = wave_sum(e2);
        if (lane == 0) red[24 + wave] = s2sum;
        lds_barrier();
        s2sum = ((red[24] + red[25]) + (red[26] + red[27])) + ((red[28] + red[29]) + (red[30] + red[31]));
        const float p2 = e2 / s2sum;

        int cand;
        if (a.temperature > 0.f) {
          float cdf = p2;  // inactive lanes hold 0: the scan runs in class order
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const float n = __shfl_up(cdf, off, 64);
            if (lane >= off) cdf += n;
          }
          if (lane == 63) red[32 + wave] = cdf;
          lds_barrier();
          float base = 0.f, total = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            if (w < wave) base += red[32 + w];
            total += red[32 + w];
          }
          const float target = philox_uniform(a.seed, (uint32_t)u, (uint32_t)b) * total;
          cand = (active && base + cdf > target) ? cls : Q - 1;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
        } else {
          float bv = active ? p2 : -1.f;
          cand = active ? cls : Q;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(cand, off, 64);
            if (ov > bv || (ov == bv && oi < cand)) {
              bv = ov;
              cand = oi;
            }
          }
          if (lane == 0) red[40 + wave] = bv;
        }
        if (lane == 0) ired[wave] = cand;
        lds_barrier();
        int pick = ired[0];
        if (a.temperature > 0.f) {
#pragma unroll
          for (int w = 1; w < 8; ++w) pick = min(pick, ired[w]);
        } else {
          float bv = red[40];
#pragma unroll
          for (int w = 1; w < 8; ++w)
            if (red[40 + w] > bv) {  // waves hold ascending class ranges: strict > keeps the first
              bv = red[40 + w];
              pick = ired[w];
            }
        }
        if (tid == 0) {
          if (a.choices_out && u >= a.logits_t0) a.choices_out[(size_t)b * a.n_total + u] = pick;
          if (u >= a.n_given) {
            samples[u] = pick;
            choice = pick;
          }
        }
      }
      lds_barrier();
      if (tid == 0) {
        iflag[2] = iflag[1];
        iflag[1] = choice;
      }
      lds_barrier();
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
  const int region = i / MAT_F, r = i - region * MAT_F;
  const int e = r & 3, v = r >> 2, tid = v & (NT - 1), i4 = v >> 9, q = tid & 3;
  const int k = 16 * q + 4 * i4 + e;
  if (region < 2) {
    const int row = ((tid >> 2) & 1) * 64 + (tid >> 3);
    dst[i] = fg_elem(fw, gw, C, row, region == 0 ? 64 + k : k);  // WC: current tap, WP: past tap
  } else {
    dst[i] = rs_elem(rw, sw, C, tid >> 2, k);
  }
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

bool pipe_ok(const mvn_dims *d) {
  return d->residual_channels == 64 && d->skip_channels == 64 && d->input_channels == 256 &&
         n_layers(d) >= 1;
}
int pipe_stages(const mvn_dims *d) { return (n_layers(d) + p64::LPS - 1) / p64::LPS + 1; }
size_t pipe_hand_floats(const mvn_dims *d, int batch) {
  // batch * NS inboxes of 128 granules (2 floats each) + 64 floats of flags
  return (size_t)batch * pipe_stages(d) * p64::GRAN * 2 + 64;
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
  hipLaunchKernelGGL(gen_pipe64_kernel, dim3(batch * NS), dim3(NT), LDS_FLOATS * sizeof(float), s, a,
                     gran, err, NS);
  return check_hip(hipGetLastError(), "mvn_generate(pipe)");
}

}  // namespace mvn
