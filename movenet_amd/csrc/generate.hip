// Autoregressive generation: ring-buffer ("fast WaveNet") kernels for gfx950.
//
// Replaces the loop of /root/reference/movenet/wavenet.py:217-237, which re-runs
// the whole network on an RF-long window per generated sample, by a cached
// formulation: layer l keeps the last d_l inputs it saw (its dilation queue),
// so one new sample costs one pass over the weights.
//
// Per step t (consume sample x_t, predict x_{t+1}), per sequence:
//   h      = E1[:, x_t] + E0[:, x_{t-1}]            causal conv on a one-hot input
//                                                    (movenet/modules.py:19-30)
//   for l: past = ring_l[t mod d_l]; ring_l[t mod d_l] = h
//          f,g  = Wf0.past + Wf1.h , Wg0.past + Wg1.h      (modules.py:36-46, :73)
//          z    = tanh(f) * sigmoid(g)                      (modules.py:80)
//          skip += Ws.z + bs ; h = Wr.z + br + h            (modules.py:83-91)
//   logits = W2.lrelu(W1.lrelu(skip) + b1) + b2             (modules.py:139-142)
//   x_{t+1} = argmax / multinomial of softmax(softmax(logits)[/T])
//                                                    (wavenet.py:189-191, :227-233)
//
// Two variants:
//   GENERIC  any dims; weights stored transposed ([in][out]) so a thread per
//            output reads coalesced; correctness fallback.
//   STREAM   C=K=64, Q=256: one 256-thread workgroup per sequence, the 3.3 MB
//            of weights streamed from L2 once per step through two register
//            buffers (one block of <=16 float4 per thread always in flight),
//            embedding tables resident in LDS, activations exchanged in LDS.
#include <algorithm>

#include "common.h"
#include "gen_common.h"

namespace mvn {

// ======================================================================
// GENERIC variant
// ======================================================================
__device__ __forceinline__ void matvec_parts(const float *__restrict__ Wt, const float *x, int nin,
                                             int nout, float *part, int P, int chunk) {
  for (int w = threadIdx.x; w < P * nout; w += blockDim.x) {
    const int p = w / nout, o = w - p * nout;
    const int k0 = p * chunk, k1 = min(nin, k0 + chunk);
    // 8 independent weight loads in flight per thread (this loop is L2-latency bound),
    // summed in the original k order
    float acc = 0.f;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = Wt[(size_t)(k + i) * nout + o];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(wv[i], x[k + i], acc);
    }
    for (; k < k1; ++k) acc = fmaf(Wt[(size_t)k * nout + o], x[k], acc);
    part[w] = acc;
  }
}
__device__ __forceinline__ float sum_parts(const float *part, int nout, int P, int o) {
  float s = part[o];
  for (int p = 1; p < P; ++p) s += part[p * nout + o];
  return s;
}
__device__ __forceinline__ void split_parts(int nin, int nout, int nt, int &P, int &chunk) {
  P = nt / nout;
  if (P < 1) P = 1;
  if (P > nin) P = nin;
  chunk = (nin + P - 1) / P;
  P = (nin + chunk - 1) / chunk;
}

__device__ float block_max_g(float v, float *red) {
  v = wave_max(v);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < nw; ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}
__device__ float block_sum_g(float v, float *red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < nw; ++w) r += red[w];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(1024) void gen_generic_kernel(GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, NT = blockDim.x, b = blockIdx.x;
  const int C = a.C, K = a.K, Q = a.Q, L = a.L;
  int partsz = NT;
  if (2 * C > partsz) partsz = 2 * C;
  if (C + K > partsz) partsz = C + K;
  if (Q > partsz) partsz = Q;
  float *xcat = smem;                // [2C]  past | cur
  float *pastAll = xcat + 2 * C;     // [L*C]
  float *part = pastAll + L * C;     // [partsz]
  float *zbuf = part + partsz;       // [C]
  float *skip = zbuf + C;            // [K]
  float *a0 = skip + K;              // [K]
  float *a1 = a0 + K;                // [Q]
  float *logits = a1 + Q;            // [Q]
  float *red = logits + Q;           // [16]
  int *ichoice = (int *)(red + 16);  // [4]
  float *ctxv = red + 16 + 4;        // [C] context vector of this step

  const float *E0t = a.w, *E1t = a.w + (size_t)Q * C;
  const float *lw = a.w + 2 * (size_t)Q * C;
  const size_t fg_sz = 4 * (size_t)C * C, rs_sz = (size_t)C * (C + K);
  const size_t layer_stride = fg_sz + rs_sz + (C + K);
  const float *W1t = lw + L * layer_stride, *b1 = W1t + (size_t)K * Q;
  const float *W2t = b1 + Q, *b2 = W2t + (size_t)Q * Q;
  float *ring = a.state + (size_t)b * a.state_per_seq;
  int32_t *samples = a.samples + (size_t)b * a.stride;

  int Pfg, cfg, Prs, crs, P1, c1, P2, c2;
  split_parts(2 * C, 2 * C, NT, Pfg, cfg);
  split_parts(C, C + K, NT, Prs, crs);
  split_parts(K, Q, NT, P1, c1);
  split_parts(Q, Q, NT, P2, c2);

  for (int t = a.t_begin; t < a.t_end; ++t) {
    int idx_t = samples[t];
    int idx_p = t > 0 ? samples[t - 1] : -1;
    idx_t = min(max(idx_t, 0), Q - 1);
    if (idx_p >= Q) idx_p = Q - 1;
    for (int i = tid; i < L * C; i += NT) {
      const int l = i / C, c = i - l * C;
      const int d = 1 << (l % a.layer_size);
      pastAll[i] = ring_load(ring + ring_offset(l, a.layer_size, C) + (t & (d - 1)) * C + c);
    }
    for (int c = tid; c < C; c += NT) {
      float v = E1t[(size_t)idx_t * C + c];
      if (idx_p >= 0) v += E0t[(size_t)idx_p * C + c];
      xcat[C + c] = v;
    }
    for (int k = tid; k < K; k += NT) skip[k] = 0.f;
    if (a.ctx_tm)
      for (int c = tid; c < C; c += NT)
        ctxv[c] = a.ctx_tm[(size_t)b * a.ctx_stride_b + (size_t)t * C + c];
    __syncthreads();

    for (int l = 0; l < L; ++l) {
      const float *Wfg = lw + l * layer_stride, *Wrs = Wfg + fg_sz, *brs = Wrs + rs_sz;
      const int d = 1 << (l % a.layer_size);
      float *slot = ring + ring_offset(l, a.layer_size, C) + (t & (d - 1)) * C;
      for (int c = tid; c < C; c += NT) xcat[c] = pastAll[l * C + c];
      __syncthreads();
      matvec_parts(Wfg, xcat, 2 * C, 2 * C, part, Pfg, cfg);
      __syncthreads();
      for (int c = tid; c < C; c += NT) {
        float f = sum_parts(part, 2 * C, Pfg, c), g = sum_parts(part, 2 * C, Pfg, C + c);
        if (a.ctx_tm) {
          // 1x1 context convs (modules.py:58-63, :75-77; alignment = build definition)
          const float *wc = a.wctx + (size_t)l * (2 * C * C + 2 * C), *bc = wc + 2 * C * C;
          float cf = 0.f, cg = 0.f;
          for (int k = 0; k < C; ++k) {
            cf = fmaf(wc[(size_t)k * 2 * C + c], ctxv[k], cf);
            cg = fmaf(wc[(size_t)k * 2 * C + C + c], ctxv[k], cg);
          }
          f += cf + bc[c];
          g += cg + bc[C + c];
        }
        zbuf[c] = gate(f, g);
        slot[c] = xcat[C + c];
      }
      __syncthreads();
      matvec_parts(Wrs, zbuf, C, C + K, part, Prs, crs);
      __syncthreads();
      for (int o = tid; o < C + K; o += NT) {
        const float v = sum_parts(part, C + K, Prs, o) + brs[o];
        if (o < C)
          xcat[C + o] = v + xcat[C + o];
        else
          skip[o - C] += v;
      }
      __syncthreads();
    }

    const int u = t + 1;
    const bool want_out = (a.logits_out || a.choices_out) && u >= a.logits_t0;
    if (u < a.n_total && (u >= a.n_given || want_out)) {  // block-uniform
      for (int k = tid; k < K; k += NT) a0[k] = leaky(skip[k]);
      __syncthreads();
      matvec_parts(W1t, a0, K, Q, part, P1, c1);
      __syncthreads();
      for (int q = tid; q < Q; q += NT) a1[q] = leaky(sum_parts(part, Q, P1, q) + b1[q]);
      __syncthreads();
      matvec_parts(W2t, a1, Q, Q, part, P2, c2);
      __syncthreads();
      for (int q = tid; q < Q; q += NT) {
        const float v = sum_parts(part, Q, P2, q) + b2[q];
        logits[q] = v;
        if (a.logits_out && u >= a.logits_t0)
          a.logits_out[((size_t)b * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * Q + q] = v;
      }
      __syncthreads();
      // softmax(softmax(x)[/T])  (wavenet.py:189-191 then :227-233)
      float m = -INFINITY;
      for (int q = tid; q < Q; q += NT) m = fmaxf(m, logits[q]);
      m = block_max_g(m, red);
      float s = 0.f;
      for (int q = tid; q < Q; q += NT) {
        const float e = expf(logits[q] - m);
        a1[q] = e;
        s += e;
      }
      s = block_sum_g(s, red);
      float m2 = -INFINITY;
      for (int q = tid; q < Q; q += NT) {
        float p = a1[q] / s;
        if (a.temperature > 0.f) p = p / a.temperature;
        a1[q] = p;
        m2 = fmaxf(m2, p);
      }
      m2 = block_max_g(m2, red);
      float s2 = 0.f;
      for (int q = tid; q < Q; q += NT) {
        const float e = expf(a1[q] - m2);
        a1[q] = e;
        s2 += e;
      }
      s2 = block_sum_g(s2, red);
      for (int q = tid; q < Q; q += NT) a1[q] = a1[q] / s2;
      __syncthreads();
      if (tid == 0) {
        int choice = 0;
        if (a.temperature > 0.f) {
          float total = 0.f;
          for (int q = 0; q < Q; ++q) total += a1[q];
          const float target = philox_uniform(a.seed, (uint32_t)u, (uint32_t)b) * total;
          float cdf = 0.f;
          choice = Q - 1;
          for (int q = 0; q < Q; ++q) {
            cdf += a1[q];
            if (cdf > target) {
              choice = q;
              break;
            }
          }
        } else {
          float best = a1[0];
          for (int q = 1; q < Q; ++q)
            if (a1[q] > best) {
              best = a1[q];
              choice = q;
            }
        }
        if (a.choices_out && u >= a.logits_t0) a.choices_out[(size_t)b * a.n_total + u] = choice;
        if (u >= a.n_given) samples[u] = choice;
        ichoice[0] = choice;
      }
    }
    __syncthreads();  // also makes samples[u] visible to the next step's loads
  }
}

// ======================================================================
// STREAM variant, C = K = 64, Q = 256, 256 threads
// ======================================================================
typedef float4 f4;

namespace s64 {
constexpr int C = 64, Q = 256, NT = 256;
constexpr int FG_F4 = 2 * 16 * 128;            // [half][k4][o] float4
constexpr int RS_F4 = 2 * 8 * 128;             // [half][k4][o] float4
constexpr int LAYER_F4 = FG_F4 + RS_F4 + 32;   // + brs[128]
constexpr int W1_F4 = 16 * 256, W2_F4 = 64 * 256;
constexpr int HEAD_F4 = W1_F4 + 64 + W2_F4 + 64;
constexpr int EMB_FLOATS = 2 * Q * C;
}  // namespace s64

__device__ __forceinline__ float dot4(const f4 w, const f4 x, float acc) {
  acc = fmaf(w.x, x.x, acc);
  acc = fmaf(w.y, x.y, acc);
  acc = fmaf(w.z, x.z, acc);
  return fmaf(w.w, x.w, acc);
}

__global__ __launch_bounds__(256, 1) void gen_stream64_kernel(GenArgs a) {
  using namespace s64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int o = tid & 127, half = tid >> 7, lane = tid & 63, wave = tid >> 6;
  const int L = a.L;

  float *E0 = smem;                 // [Q][C]
  float *E1 = E0 + Q * C;           // [Q][C]
  float *xcat = E1 + Q * C;         // [128] past | cur
  float *pfg = xcat + 128;          // [256]
  float *zbuf = pfg + 256;          // [64]
  float *prs = zbuf + 64;           // [256]
  float *sk = prs + 256;            // [64]
  float *a1 = sk + 64;              // [256]
  float *red = a1 + 256;            // [8][4]
  int *ired = (int *)(red + 32);    // [8]: [0..3] per-wave candidate, [4] prev idx, [5] cur idx
  float *pastAll = red + 32 + 8;    // [L][64]
  float *pctx = pastAll + (size_t)L * 64;  // [L][128] context-conv terms of the step (conditioned runs only)
  float *cvec = pctx + (a.ctx_tm ? (size_t)L * 128 : 0);  // [64] the step's context column

  const f4 *wl = (const f4 *)(a.w + EMB_FLOATS);
  const f4 *head = wl + (size_t)L * LAYER_F4;
  const f4 *W1p = head, *W2p = head + W1_F4 + 64;
  const float *b1 = (const float *)(head + W1_F4), *b2 = (const float *)(W2p + W2_F4);
  float *ring = a.state + (size_t)b * a.state_per_seq;
  int32_t *samples = a.samples + (size_t)b * a.stride;

  // embedding tables -> LDS (once per launch)
  {
    const f4 *src = (const f4 *)a.w;
    f4 *dst = (f4 *)E0;
#pragma unroll 4
    for (int i = tid; i < EMB_FLOATS / 4; i += NT) dst[i] = src[i];
  }
  if (tid == 0) {
    ired[5] = samples[a.t_begin];
    ired[4] = a.t_begin > 0 ? samples[a.t_begin - 1] : -1;
  }
  f4 A[16], B[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) A[i] = wl[(half * 16 + i) * 128 + o];  // fg(0)
  __syncthreads();

  float skipacc = 0.f;
  for (int t = a.t_begin; t < a.t_end; ++t) {
    // ---- step start: dilation-queue pops for every layer, embedding gather
    for (int i = tid; i < L * C; i += NT) {
      const int l = i >> 6, c = i & 63;
      const int d = 1 << (l % a.layer_size);
      pastAll[i] = ring_load(ring + ring_offset(l, a.layer_size, C) + (t & (d - 1)) * C + c);
    }
    int next_given = 0;
    if (tid == 0 && t + 1 < a.n_given) next_given = samples[t + 1];
    if (tid < 64) {
      int idx_t = ired[5], idx_p = ired[4];
      idx_t = min(max(idx_t, 0), a.Q - 1);
      idx_p = min(idx_p, a.Q - 1);
      float v = E1[idx_t * C + tid];
      if (idx_p >= 0) v += E0[idx_p * C + tid];
      xcat[64 + tid] = v;
    }
    skipacc = 0.f;
    if (a.ctx_tm && tid < 64) cvec[tid] = a.ctx_tm[(size_t)b * a.ctx_stride_b + (size_t)t * C + tid];
    __syncthreads();  // pastAll landed (vmcnt drained here once per step)
    if (a.ctx_tm) {
      // r4: local conditioning (build definition): f | g += Wc ctx(t) + bc for every layer.  None of it
      // depends on the chain, so all L x 128 terms are formed HERE, before the layer loop and outside its weight
      // schedule: thread (o = tid & 127, lp = tid >> 7) takes the layers l = lp (mod 2), row o over the 64 context
      // channels (generic context section: per layer Wt[k][o: f | g] then bias[2C]).
      const int lp = tid >> 7;
      for (int l = lp; l < L; l += 2) {
        const float *wc = a.wctx + (size_t)l * (2 * C * C + 2 * C);
        float acc[4] = {wc[2 * C * C + o], 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < C; k += 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(wc[(size_t)(k + e) * 2 * C + o], cvec[k + e], acc[e]);
        }
        pctx[l * 128 + o] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      }
      lds_barrier();
    }
    if (tid < 64) xcat[tid] = pastAll[tid];
    lds_barrier();

    for (int l = 0; l < L; ++l) {
      const f4 *wlay = wl + (size_t)l * LAYER_F4;
      const float *brs = (const float *)(wlay + FG_F4 + RS_F4);
      // B <- rs(l)
#pragma unroll
      for (int i = 0; i < 8; ++i) B[i] = wlay[FG_F4 + (half * 8 + i) * 128 + o];
      const float bias = brs[o];
      // f,g partial sums over this half's 64 inputs (A = fg(l))
      {
        const f4 *x4 = (const f4 *)xcat + half * 16;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          acc0 = dot4(A[i], x4[i], acc0);
          acc1 = dot4(A[i + 1], x4[i + 1], acc1);
        }
        pfg[tid] = acc0 + acc1;
      }
      // A <- fg(l+1), or the first half of head conv1
      if (l + 1 < L) {
        const f4 *nx = wlay + LAYER_F4;
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = nx[(half * 16 + i) * 128 + o];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = W1p[i * 256 + tid];
      }
      lds_barrier();
      if (tid < 64) {
        float f = pfg[tid] + pfg[128 + tid];
        float g = pfg[64 + tid] + pfg[192 + tid];
        if (a.ctx_tm) {
          f += pctx[l * 128 + tid];
          g += pctx[l * 128 + 64 + tid];
        }
        zbuf[tid] = gate(f, g);
        // queue push: this layer's input at time t replaces the one popped
        const int d = 1 << (l % a.layer_size);
        ring[ring_offset(l, a.layer_size, C) + (t & (d - 1)) * C + tid] = xcat[64 + tid];
      }
      lds_barrier();
      {
        const f4 *z4 = (const f4 *)zbuf + half * 8;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          acc0 = dot4(B[i], z4[i], acc0);
          acc1 = dot4(B[i + 1], z4[i + 1], acc1);
        }
        prs[tid] = acc0 + acc1;
      }
      lds_barrier();
      if (tid < 128) {
        const float v = prs[tid] + prs[128 + tid] + bias;
        if (tid < 64) {
          xcat[64 + tid] = v + xcat[64 + tid];
          if (l + 1 < L) xcat[tid] = pastAll[(l + 1) * C + tid];
        } else {
          skipacc += v;
        }
      }
      lds_barrier();
    }

    // ---- head + choice
    const int u = t + 1;
    const bool want_out = (a.logits_out || a.choices_out) && u >= a.logits_t0;
    const bool do_head = u < a.n_total && (u >= a.n_given || want_out);  // block-uniform
    int choice = next_given;
    // conv1, first 32 inputs (A = W1p k4 0..7), B <- k4 8..15
    if (tid >= 64 && tid < 128) sk[tid - 64] = leaky(skipacc);
#pragma unroll
    for (int i = 0; i < 8; ++i) B[i] = W1p[(8 + i) * 256 + tid];
    lds_barrier();
    float h1 = 0.f;
    {
      const f4 *s4 = (const f4 *)sk;
#pragma unroll
      for (int i = 0; i < 8; ++i) h1 = dot4(A[i], s4[i], h1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = W2p[i * 256 + tid];
    {
      const f4 *s4 = (const f4 *)sk + 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) h1 = dot4(B[i], s4[i], h1);
    }
    a1[tid] = leaky(h1 + b1[tid]);
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = W2p[(16 + i) * 256 + tid];
    lds_barrier();
    float lg = 0.f;
    {
      const f4 *x4 = (const f4 *)a1;
#pragma unroll
      for (int i = 0; i < 16; ++i) lg = dot4(A[i], x4[i], lg);
#pragma unroll
      for (int i = 0; i < 16; ++i) A[i] = W2p[(32 + i) * 256 + tid];
#pragma unroll
      for (int i = 0; i < 16; ++i) lg = dot4(B[i], x4[16 + i], lg);
#pragma unroll
      for (int i = 0; i < 16; ++i) B[i] = W2p[(48 + i) * 256 + tid];
#pragma unroll
      for (int i = 0; i < 16; ++i) lg = dot4(A[i], x4[32 + i], lg);
      // A <- fg(0) for the next step
#pragma unroll
      for (int i = 0; i < 16; ++i) A[i] = wl[(half * 16 + i) * 128 + o];
#pragma unroll
      for (int i = 0; i < 16; ++i) lg = dot4(B[i], x4[48 + i], lg);
    }
    lg += b2[tid];

    if (do_head) {
      // (r4: a.Q in {64, 128, 256}; classes >= a.Q are the padding of the 256-wide head: logit -inf by their packed
      // bias, no probability in either softmax, never written out)
      if (a.logits_out && u >= a.logits_t0 && tid < a.Q)
        a.logits_out[((size_t)b * (a.n_total - a.logits_t0) + (u - a.logits_t0)) * a.Q + tid] = lg;
      // softmax -> [/T] -> softmax, one class per thread
      float m = wave_max(lg);
      if (lane == 0) red[0 + wave] = m;
      lds_barrier();
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      const float e = expf(lg - m);
      float s = wave_sum(e);
      if (lane == 0) red[4 + wave] = s;
      lds_barrier();
      s = (red[4] + red[5]) + (red[6] + red[7]);
      float p = e / s;
      if (a.temperature > 0.f) p = p / a.temperature;
      float m2 = wave_max(p);
      if (lane == 0) red[8 + wave] = m2;
      lds_barrier();
      m2 = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
      const float e2 = tid < a.Q ? expf(p - m2) : 0.f;
      float s2 = wave_sum(e2);
      if (lane == 0) red[12 + wave] = s2;
      lds_barrier();
      s2 = (red[12] + red[13]) + (red[14] + red[15]);
      const float p2 = e2 / s2;

      int cand;
      if (a.temperature > 0.f) {
        // inclusive scan of p2 over the 256 classes
        float c = p2;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float n = __shfl_up(c, off, 64);
          if (lane >= off) c += n;
        }
        if (lane == 63) red[16 + wave] = c;
        lds_barrier();
        float base = 0.f;
        if (wave > 0) base += red[16];
        if (wave > 1) base += red[17];
        if (wave > 2) base += red[18];
        const float total = ((red[16] + red[17]) + red[18]) + red[19];
        const float target = philox_uniform(a.seed, (uint32_t)u, (uint32_t)b) * total;
        cand = (base + c > target) ? tid : a.Q - 1;
        // first class whose cdf exceeds the target
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
      } else {
        // first index of the maximum of p2
        float bv = p2;
        cand = tid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float ov = __shfl_xor(bv, off, 64);
          const int oi = __shfl_xor(cand, off, 64);
          if (ov > bv || (ov == bv && oi < cand)) {
            bv = ov;
            cand = oi;
          }
        }
        if (lane == 0) red[20 + wave] = bv;
      }
      if (lane == 0) ired[wave] = cand;
      lds_barrier();
      int pick;
      if (a.temperature > 0.f) {
        pick = min(min(ired[0], ired[1]), min(ired[2], ired[3]));
      } else {
        pick = ired[0];
        float bv = red[20];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          if (red[20 + w] > bv) {
            bv = red[20 + w];
            pick = ired[w];
          }
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
      ired[4] = ired[5];
      ired[5] = choice;
    }
    // ring pushes of this step must be globally performed before the next
    // step's pops: every wave drains its stores, then the block meets.
    __syncthreads();
  }
}

// ======================================================================
// weight packing (state_dict layouts -> streaming layouts)
// ======================================================================
__global__ void pack_embed_kernel(const float *__restrict__ causal_w, float *__restrict__ dst, int Q,
                                  int C) {
  // dst: E0t[Q][C] (tap 0, multiplies x[t-1]) then E1t[Q][C] (tap 1, x[t])
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * Q * C) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, q = r / C, c = r - q * C;
  dst[i] = causal_w[((size_t)c * Q + q) * 2 + tap];
}

__global__ void pack_layer_generic_kernel(const float *fw, const float *gw, const float *rw,
                                          const float *rb, const float *sw, const float *sb,
                                          float *__restrict__ dst, int C, int K) {
  const int n_fg = 4 * C * C, n_rs = C * (C + K), n_b = C + K;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_fg) {
    const int k = i / (2 * C), o = i - k * 2 * C;
    dst[i] = fg_elem(fw, gw, C, o, k);
  } else if (i < n_fg + n_rs) {
    const int j = i - n_fg, k = j / (C + K), o = j - k * (C + K);
    dst[i] = rs_elem(rw, sw, C, o, k);
  } else if (i < n_fg + n_rs + n_b) {
    const int o = i - n_fg - n_rs;
    dst[i] = o < C ? rb[o] : sb[o - C];
  }
}

__global__ void pack_head_generic_kernel(const float *w1, const float *b1, const float *w2,
                                         const float *b2, float *__restrict__ dst, int Q, int K) {
  const int n1 = K * Q, n2 = Q * Q;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) {
    const int k = i / Q, q = i - k * Q;
    dst[i] = w1[(size_t)q * K + k];
  } else if (i < n1 + Q) {
    dst[i] = b1[i - n1];
  } else if (i < n1 + Q + n2) {
    const int j = i - n1 - Q, k = j / Q, q = j - k * Q;
    dst[i] = w2[(size_t)q * Q + k];
  } else if (i < n1 + Q + n2 + Q) {
    dst[i] = b2[i - n1 - Q - n2];
  }
}

__global__ void pack_layer_s64_kernel(const float *fw, const float *gw, const float *rw,
                                      const float *rb, const float *sw, const float *sb,
                                      float *__restrict__ dst) {
  using namespace s64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float index
  if (i < FG_F4 * 4) {
    const int j = i & 3, v = i >> 2, o = v & 127, hk = v >> 7, half = hk >> 4, k4 = hk & 15;
    dst[i] = fg_elem(fw, gw, C, o, half * 64 + k4 * 4 + j);
  } else if (i < (FG_F4 + RS_F4) * 4) {
    const int ii = i - FG_F4 * 4;
    const int j = ii & 3, v = ii >> 2, o = v & 127, hk = v >> 7, half = hk >> 3, k4 = hk & 7;
    dst[i] = rs_elem(rw, sw, C, o, half * 32 + k4 * 4 + j);
  } else if (i < LAYER_F4 * 4) {
    const int o = i - (FG_F4 + RS_F4) * 4;
    dst[i] = o < C ? rb[o] : sb[o - C];
  }
}

// `qm`: the model's class count (64, 128 or 256); the head runs 256 wide, classes >= qm are padding (zero rows and
// columns, conv2 bias -inf)
__global__ void pack_head_s64_kernel(const float *w1, const float *b1, const float *w2,
                                     const float *b2, float *__restrict__ dst, int qm) {
  using namespace s64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n1 = W1_F4 * 4, n2 = W2_F4 * 4;
  if (i < n1) {
    const int j = i & 3, v = i >> 2, q = v & 255, k4 = v >> 8;
    dst[i] = q < qm ? w1[(size_t)q * 64 + k4 * 4 + j] : 0.f;
  } else if (i < n1 + 256) {
    dst[i] = i - n1 < qm ? b1[i - n1] : 0.f;
  } else if (i < n1 + 256 + n2) {
    const int ii = i - n1 - 256;
    const int j = ii & 3, v = ii >> 2, q = v & 255, k4 = v >> 8, k = k4 * 4 + j;
    dst[i] = (q < qm && k < qm) ? w2[(size_t)q * qm + k] : 0.f;
  } else if (i < n1 + 256 + n2 + 256) {
    const int q = i - n1 - 256 - n2;
    dst[i] = q < qm ? b2[q] : -INFINITY;
  }
}
// the embedding tables of a model with qm classes in the STREAM layout's 256 rows (rows >= qm: zero, never gathered)
__global__ void pack_embed_s64_kernel(const float *__restrict__ causal_w, float *__restrict__ dst, int qm) {
  using namespace s64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= EMB_FLOATS) return;
  const int tap = i / (Q * C), r = i - tap * Q * C, q = r / C, c = r - q * C;
  dst[i] = q < qm ? causal_w[((size_t)c * qm + q) * 2 + tap] : 0.f;
}

// wide models keep more weight loads in flight with a 1024-thread workgroup
static int generic_threads(const mvn_dims *d) {
  return (2 * d->residual_channels >= 256 || d->input_channels >= 512) ? 1024 : 256;
}

static size_t generic_lds_bytes(const mvn_dims *d) {
  const int C = d->residual_channels, K = d->skip_channels, Q = d->input_channels;
  const int L = n_layers(d);
  int partsz = generic_threads(d);
  if (2 * C > partsz) partsz = 2 * C;
  if (C + K > partsz) partsz = C + K;
  if (Q > partsz) partsz = Q;
  return sizeof(float) * ((size_t)2 * C + (size_t)L * C + partsz + C + 2 * K + 2 * Q + 16 + 4 + C);
}

size_t hand_status_offset(const mvn_dims *d, int batch) {
  size_t g = 0;
  if (pipe_ok(d)) g = std::max(g, (size_t)batch * pipe_stages(d) * 4 * d->residual_channels);
  if (pipe_h16_ok(d)) g = std::max(g, (size_t)batch * pipe_h16_stages(d) * 4 * d->residual_channels);
  if (fold_ok(d)) g = std::max(g, (size_t)batch * fold_stages(d) * 6 * d->residual_channels);
  return g;
}
size_t hand_total_floats(const mvn_dims *d, int batch) {
  size_t ns = 0;
  if (pipe_ok(d)) ns = std::max(ns, (size_t)pipe_stages(d));
  if (pipe_h16_ok(d)) ns = std::max(ns, (size_t)pipe_h16_stages(d));
  if (fold_ok(d)) ns = std::max(ns, (size_t)fold_stages(d));
  if (ns == 0) return 0;
  return hand_status_offset(d, batch) + (16 + (size_t)batch * ns + 63) / 64 * 64;
}

// packed blob without the trailing context-conv section
static size_t gen_base_floats(const mvn_dims *dims, int variant) {
  const size_t C = dims->residual_channels, K = dims->skip_channels, Q = dims->input_channels;
  const size_t L = n_layers(dims);
  if (variant == MVN_GEN_PIPE) return pipe_weights_floats(dims);
  if (variant == MVN_GEN_PIPE_F16) return pipe_h16_weights_floats(dims);
  if (variant == MVN_GEN_FOLD) return fold_weights_floats(dims);
  if (variant == MVN_GEN_STREAM) return s64::EMB_FLOATS + 4 * (L * s64::LAYER_F4 + s64::HEAD_F4);
  return 2 * Q * C + L * (4 * C * C + C * (C + K) + (C + K)) + K * Q + Q + Q * Q + Q;
}

// generic context section: per layer Wt[k (C)][o (2C): f | g] then bias[2C]
__global__ void pack_ctx_generic_kernel(const float *wcf, const float *bcf, const float *wcg,
                                        const float *bcg, float *__restrict__ dst, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nw = 2 * C * C;
  if (i < nw) {
    const int k = i / (2 * C), o = i - k * 2 * C;
    dst[i] = o < C ? wcf[(size_t)o * C + k] : wcg[(size_t)(o - C) * C + k];
  } else if (i < nw + 2 * C) {
    const int o = i - nw;
    dst[i] = o < C ? bcf[o] : bcg[o - C];
  }
}

static bool stream_ok(const mvn_dims *d) {
  return d->residual_channels == 64 && d->skip_channels == 64 && head_q_ok(d->input_channels) &&
         n_layers(d) <= 80;
}

}  // namespace mvn

namespace mvn {
// (B, C, ld) -> (B, T, C): 32x32 tiles through LDS, both sides coalesced
__global__ void transpose_ctx_kernel(const float *__restrict__ src, int ld, float *__restrict__ dst,
                                     int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    tile[r][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * ld + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < T && c < C) dst[((size_t)b * T + t) * C + c] = tile[tx][r];
  }
}
}  // namespace mvn

extern "C" {

int mvn_transpose_context(const float *ctx, int ctx_ld, int batch, int channels, int t_len,
                          float *context_tm, void *stream) {
  if (!ctx || !context_tm || batch < 0 || channels < 1 || t_len < 1 || ctx_ld < t_len) {
    mvn::set_error("mvn_transpose_context: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0) return MVN_OK;
  dim3 grid((t_len + 31) / 32, (channels + 31) / 32, batch);
  hipLaunchKernelGGL(mvn::transpose_ctx_kernel, grid, dim3(256), 0, (hipStream_t)stream, ctx, ctx_ld,
                     context_tm, channels, t_len);
  return mvn::check_hip(hipGetLastError(), "mvn_transpose_context");
}

static int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 256;  // no device visible (build box): the MI355X figure
  return cus;
}

int mvn_gen_launch_pipelines(const mvn_dims *dims, int variant, int batch) {
  int rc = mvn::validate_dims(dims);
  if (rc) return rc;
  if (batch < 1) return 0;
  if (variant == MVN_GEN_FOLD && mvn::fold_ok(dims)) return mvn::fold_launch_pipelines(dims, batch);
  if (variant == MVN_GEN_PIPE && mvn::pipe_ok(dims)) return std::min(batch, mvn::pipe_pipelines(dims));
  if (variant == MVN_GEN_PIPE_F16 && mvn::pipe_h16_ok(dims)) return std::min(batch, mvn::pipe_h16_pipelines(dims));
  return 0;
}

int mvn_gen_variant(const mvn_dims *dims, int requested, int batch) {
  int rc = mvn::validate_dims(dims);
  if (rc) return rc;
  const bool pipe_fits =
      mvn::pipe_ok(dims) && batch >= 1 && device_cus() >= 256 && batch <= mvn::pipe_max_batch(dims);
  const bool fold_fits =
      mvn::fold_ok(dims) && batch >= 1 && device_cus() >= 256 && batch <= mvn::fold_max_batch(dims);
  if (requested == MVN_GEN_AUTO) {
    // FOLD where it holds the batch (config 2: 14.6 us per step against PIPE's 16.8; 16 pipelines
    // of up to 8 sequences), PIPE above that (24 pipelines of up to 8; config 5: 4 of up to 16);
    // DESIGN.md section 4.1
    if (fold_fits) return MVN_GEN_FOLD;
    if (pipe_fits) return MVN_GEN_PIPE;
    return mvn::stream_ok(dims) ? MVN_GEN_STREAM : MVN_GEN_GENERIC;
  }
  if (requested == MVN_GEN_PIPE) {
    if (!pipe_fits) {
      mvn::set_error("PIPE variant needs C=K in {64,128}, Q in {64,128,256}, 256 CUs and batch <= %d for these "
                     "dims (stages per sequence: ceil(L/4)+1 at C=64, L+1 at C=128; 32 per XCD)",
                     mvn::pipe_ok(dims) ? mvn::pipe_max_batch(dims) : 0);
      return MVN_ERR_UNSUPPORTED;
    }
    return MVN_GEN_PIPE;
  }
  if (requested == MVN_GEN_FOLD) {
    if (!fold_fits) {
      mvn::set_error("FOLD variant needs C=K=64, Q in {64,128,256}, 256 CUs and batch <= %d for these dims "
                     "(ceil(L/3)+1 stages per sequence, 32 per XCD)",
                     mvn::fold_ok(dims) ? mvn::fold_max_batch(dims) : 0);
      return MVN_ERR_UNSUPPORTED;
    }
    return MVN_GEN_FOLD;
  }
  if (requested == MVN_GEN_PIPE_F16) {
    if (!mvn::pipe_h16_ok(dims) || batch < 1 || device_cus() < 256 || batch > mvn::pipe_h16_max_batch(dims)) {
      mvn::set_error("PIPE_F16 variant needs C=K=128, Q=256, 256 CUs and batch <= %d for these dims "
                     "(ceil(L/2)+1 stages per sequence, 32 per XCD)",
                     mvn::pipe_h16_ok(dims) ? mvn::pipe_h16_max_batch(dims) : 0);
      return MVN_ERR_UNSUPPORTED;
    }
    return MVN_GEN_PIPE_F16;
  }
  if (requested == MVN_GEN_STREAM) {
    if (!mvn::stream_ok(dims)) {
      mvn::set_error("STREAM variant needs C=K=64, Q in {64,128,256}, <=80 layers");
      return MVN_ERR_UNSUPPORTED;
    }
    return MVN_GEN_STREAM;
  }
  if (requested == MVN_GEN_GENERIC) {
    if (dims->residual_channels > 256 || dims->skip_channels > 256 || dims->input_channels > 1024 ||
        mvn::generic_lds_bytes(dims) > 160 * 1024) {
      mvn::set_error("GENERIC variant needs C,K<=256, Q<=1024 and <=160 KiB of LDS");
      return MVN_ERR_UNSUPPORTED;
    }
    return MVN_GEN_GENERIC;
  }
  mvn::set_error("unknown generate variant %d", requested);
  return MVN_ERR_BAD_ARG;
}

size_t mvn_gen_weights_floats(const mvn_dims *dims, int variant) {
  variant = mvn_gen_variant(dims, variant, 1);
  if (variant < 0) return 0;
  const size_t C = dims->residual_channels, K = dims->skip_channels, Q = dims->input_channels;
  const size_t L = mvn::n_layers(dims);
  return mvn::gen_base_floats(dims, variant) + L * (2 * C * C + 2 * C);  // + context-conv section
}

size_t mvn_gen_state_floats(const mvn_dims *dims, int batch) {
  if (mvn::validate_dims(dims) || batch < 0) return 0;
  // dilation queues, then (C=K in {64,128}, Q=256 only) the pipelined variants' hand-off area
  size_t n = (size_t)batch * (size_t)mvn::dilation_sum(dims) * dims->residual_channels;
  return n + mvn::hand_total_floats(dims, batch);
}

size_t mvn_gen_status_offset(const mvn_dims *dims, int batch) {
  if (mvn::validate_dims(dims) || batch < 0 || mvn::hand_total_floats(dims, batch) == 0) return (size_t)-1;
  // queues | inboxes (granules) of the largest pipelined variant | status word ...
  return (size_t)batch * (size_t)mvn::dilation_sum(dims) * dims->residual_channels +
         mvn::hand_status_offset(dims, batch);
}

int mvn_gen_pack_weights(const mvn_dims *dims, int variant, const mvn_params *p, float *packed,
                         void *stream_) {
  if (variant == MVN_GEN_AUTO) {
    mvn::set_error("mvn_gen_pack_weights: resolve the variant with mvn_gen_variant first");
    return MVN_ERR_BAD_ARG;
  }
  variant = mvn_gen_variant(dims, variant, 1);
  if (variant < 0) return variant;
  if (!p || !packed || !p->causal_w || !p->filter_w || !p->gate_w || !p->residual_w ||
      !p->residual_b || !p->skip_w || !p->skip_b || !p->head1_w || !p->head1_b || !p->head2_w ||
      !p->head2_b) {
    mvn::set_error("mvn_gen_pack_weights: NULL parameter pointer");
    return MVN_ERR_BAD_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const bool has_ctx = p->ctx_filter_w && p->ctx_filter_b && p->ctx_gate_w && p->ctx_gate_b;
  float *ctx_section = packed + mvn::gen_base_floats(dims, variant);
  if (variant == MVN_GEN_PIPE) {
    int rc = mvn::pipe_pack(dims, p, packed, stream);
    if (rc || !has_ctx) return rc;
    return mvn::pipe_pack_ctx(dims, p, ctx_section, stream);
  }
  if (variant == MVN_GEN_PIPE_F16) return mvn::pipe_h16_pack(dims, p, packed, has_ctx, stream);
  if (variant == MVN_GEN_FOLD) {
    int rc = mvn::fold_pack(dims, p, packed, stream);
    if (rc || !has_ctx) return rc;
    return mvn::pipe_pack_ctx(dims, p, ctx_section, stream);  // same per-layer layout as PIPE
  }
  if (has_ctx && (variant == MVN_GEN_GENERIC || variant == MVN_GEN_STREAM)) {  // (STREAM reads the generic section)
    const int Cc = dims->residual_channels, n = 2 * Cc * Cc + 2 * Cc;
    for (int l = 0; l < mvn::n_layers(dims); ++l)
      hipLaunchKernelGGL(mvn::pack_ctx_generic_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                         p->ctx_filter_w[l], p->ctx_filter_b[l], p->ctx_gate_w[l], p->ctx_gate_b[l],
                         ctx_section + (size_t)l * n, Cc);
  }
  const int C = dims->residual_channels, K = dims->skip_channels, Q = dims->input_channels;
  const int L = mvn::n_layers(dims);
  if (variant == MVN_GEN_STREAM) {  // (its layout is 256 classes wide whatever the model's Q: padded)
    hipLaunchKernelGGL(mvn::pack_embed_s64_kernel, dim3((mvn::s64::EMB_FLOATS + 255) / 256), dim3(256), 0, stream,
                       p->causal_w, packed, Q);
  } else {
    const int n = 2 * Q * C;
    hipLaunchKernelGGL(mvn::pack_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       p->causal_w, packed, Q, C);
  }
  float *lw = packed + (variant == MVN_GEN_STREAM ? (size_t)mvn::s64::EMB_FLOATS : 2 * (size_t)Q * C);
  if (variant == MVN_GEN_STREAM) {
    const int n = mvn::s64::LAYER_F4 * 4;
    for (int l = 0; l < L; ++l)
      hipLaunchKernelGGL(mvn::pack_layer_s64_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                         p->filter_w[l], p->gate_w[l], p->residual_w[l], p->residual_b[l],
                         p->skip_w[l], p->skip_b[l], lw + (size_t)l * n);
    const int nh = mvn::s64::HEAD_F4 * 4;
    hipLaunchKernelGGL(mvn::pack_head_s64_kernel, dim3((nh + 255) / 256), dim3(256), 0, stream,
                       p->head1_w, p->head1_b, p->head2_w, p->head2_b, lw + (size_t)L * n, Q);
  } else {
    const size_t n = 4 * (size_t)C * C + (size_t)C * (C + K) + (C + K);
    for (int l = 0; l < L; ++l)
      hipLaunchKernelGGL(mvn::pack_layer_generic_kernel, dim3((unsigned)((n + 255) / 256)),
                         dim3(256), 0, stream, p->filter_w[l], p->gate_w[l], p->residual_w[l],
                         p->residual_b[l], p->skip_w[l], p->skip_b[l], lw + l * n, C, K);
    const size_t nh = (size_t)K * Q + Q + (size_t)Q * Q + Q;
    hipLaunchKernelGGL(mvn::pack_head_generic_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256),
                       0, stream, p->head1_w, p->head1_b, p->head2_w, p->head2_b, lw + L * n, Q, K);
  }
  return mvn::check_hip(hipGetLastError(), "mvn_gen_pack_weights");
}

int mvn_generate(const mvn_dims *dims, int variant, const float *packed, float *state,
                 int32_t *samples, int batch, int sample_stride, int n_total, int n_given,
                 int t_begin, int t_end, float temperature, uint64_t seed, float *logits_out,
                 int32_t *choices_out, int logits_t0, const float *context_tm, void *stream) {
  if (variant == MVN_GEN_AUTO) {
    mvn::set_error("mvn_generate: resolve the variant with mvn_gen_variant first (the packed "
                   "weight layout depends on it)");
    return MVN_ERR_BAD_ARG;
  }
  variant = mvn_gen_variant(dims, variant, batch > 0 ? batch : 1);
  if (variant < 0) return variant;
  if (!packed || !state || !samples || batch < 0 || n_total < 1 || sample_stride < n_total ||
      n_given < 1 || n_given > n_total || t_begin < 0 || t_end < t_begin || t_end > n_total ||
      logits_t0 < 0 || logits_t0 > n_total) {
    mvn::set_error(
        "mvn_generate: bad argument (batch %d stride %d n_total %d n_given %d t [%d,%d) logits_t0 %d)",
        batch, sample_stride, n_total, n_given, t_begin, t_end, logits_t0);
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || t_begin == t_end) return MVN_OK;
  mvn::GenArgs a;
  a.L = mvn::n_layers(dims);
  a.layer_size = dims->layer_size;
  a.Q = dims->input_channels;
  a.C = dims->residual_channels;
  a.K = dims->skip_channels;
  a.w = packed;
  a.state = state;
  a.state_per_seq = mvn::dilation_sum(dims) * dims->residual_channels;
  a.samples = samples;
  a.stride = sample_stride;
  a.n_total = n_total;
  a.n_given = n_given;
  a.t_begin = t_begin;
  a.t_end = t_end;
  a.temperature = temperature;
  a.seed = seed;
  a.logits_out = logits_out;
  a.choices_out = choices_out;
  a.logits_t0 = logits_t0;
  a.ctx_tm = context_tm;
  a.ctx_stride_b = (long long)n_total * dims->residual_channels;
  a.wctx = packed + mvn::gen_base_floats(dims, variant);
  if (variant == MVN_GEN_PIPE || variant == MVN_GEN_PIPE_F16 || variant == MVN_GEN_FOLD) {
    float *hand = state + (size_t)batch * a.state_per_seq;
    const size_t total = mvn::hand_total_floats(dims, batch), soff = mvn::hand_status_offset(dims, batch);
    if (variant == MVN_GEN_PIPE) return mvn::pipe_launch(a, dims, batch, hand, total, soff, (hipStream_t)stream);
    if (variant == MVN_GEN_FOLD) return mvn::fold_launch(a, dims, batch, hand, total, soff, (hipStream_t)stream);
    return mvn::pipe_h16_launch(a, dims, batch, hand, total, soff, (hipStream_t)stream);
  }
  if (variant == MVN_GEN_STREAM) {
    const size_t lds =
        sizeof(float) * ((size_t)mvn::s64::EMB_FLOATS + 128 + 256 + 64 + 256 + 64 + 256 + 32 + 8 +
                         (size_t)a.L * 64 + (context_tm ? (size_t)a.L * 128 + 64 : 0));
    if (lds > 160 * 1024) {
      mvn::set_error("STREAM variant: %d conditioned layers do not fit a CU's LDS", a.L);
      return MVN_ERR_UNSUPPORTED;
    }
    int rc = mvn::ensure_max_dynamic_lds((const void *)mvn::gen_stream64_kernel,
                                         "hipFuncSetAttribute(gen_stream64)");
    if (rc) return rc;
    hipLaunchKernelGGL(mvn::gen_stream64_kernel, dim3(batch), dim3(256), lds, (hipStream_t)stream, a);
  } else {
    const size_t lds = mvn::generic_lds_bytes(dims);
    int rc = mvn::ensure_max_dynamic_lds((const void *)mvn::gen_generic_kernel,
                                         "hipFuncSetAttribute(gen_generic)");
    if (rc) return rc;
    hipLaunchKernelGGL(mvn::gen_generic_kernel, dim3(batch), dim3(mvn::generic_threads(dims)), lds,
                       (hipStream_t)stream, a);
  }
  return mvn::check_hip(hipGetLastError(), "mvn_generate");
}

}  // extern "C"
