// Full-sequence forward and backward of the WaveNet decoder on gfx950.
//
// Reference arithmetic: /root/reference/movenet/wavenet.py:166-191 (forward),
// movenet/modules.py:28-30 (causal conv), :67-93 (gated residual layer),
// :139-142 (dense head); the backward is the exact adjoint of those lines.
//
// Everything convolution-shaped here is a "weights x time" product
//     Y[b][m][t] = sum_k W[m][k] * X[b][k][t (+shift_k)]
// with tiny M,K (16..256 channels) and long t, so ONE MFMA kernel family
// (gemm_wx_staged_kernel<Op>, LDS-staged float4 epilogue) serves every forward
// conv and every data-gradient, and ONE family serves every weight gradient, a
// product over time split across workgroups: wgrad2_kernel<Op> (16-byte operand
// traffic, per-workgroup slabs reduced in a fixed order) for the layers and the
// head, wgrad_kernel<Op> (64 x 64 tiles, fp32 atomics) for the dense causal conv
// and the video upsampler.  The Op functor supplies operand addressing (taps,
// transposes, f/g row pairing, validity ranges) and the fused epilogue (gating,
// residual add, skip accumulation, leaky-ReLU and its derivative...).  All use
// v_mfma_f32_32x32x2_f32: exact fp32 products and sums (wgrad2 permutes the order
// of the sum over time inside groups of 8).
//
// Activations use an ABSOLUTE time axis (see movenet_hip.h): tensor (B, ch, Tp),
// column t = input time t, layer l's input valid for t >= A_l.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "gemm_family.h"
#include "fused_layer.h"
#include "fused_bwd.h"
#include "fused_fwd.h"
#include "fused_fwd_bf3.h"
#include "fused_bwd_l.h"

namespace mvn {

// ---- forward ops -------------------------------------------------------
// F1: f,g = dilated k=2 conv, z = tanh(f)*sigmoid(g).  Row block mb covers
// channels [32mb, 32mb+32): tile rows 0..31 = filter rows, 32..63 = gate rows,
// so a lane holds f and g of the same (channel, t) in matching registers.
// With local conditioning (ctx.p != NULL) K grows to 3C: columns [2C,3C) are the 1x1
// context convolutions (modules.py:58-63, :75-77), their biases are added in the epilogue.
// BUILD DEFINITION: the context is aligned on the same absolute time as f/g (the
// reference raises a shape error here).
template <bool HAS_CTX>
struct FgOpT {
  int K, t_begin, t_end, C, d;
  const float *wf, *wg;  // (C, C, 2)
  const float *wcf, *wcg, *bcf, *bcg;  // (C, C, 1), (C)
  Act xin;               // layer input
  Act ctx;               // upsampled video or p == NULL
  Act z, th, sg;         // outputs (th/sg.p may be NULL)
  // 64-row block mb = channels [32 mb, 32 mb + 32); inside each 32-row half rows [0,16) are
  // filter rows and [16,32) the gate rows of the same 16 channels (the LDS-staged epilogue
  // of gemm_wx_staged_kernel pairs row r with row 16 + r)
  static constexpr bool FG_PAIRS = true;
  __device__ __forceinline__ float w(int m, int k) const {
    const int r = m & 63, q = r & 31, c = (m >> 6) * 32 + (r >> 5) * 16 + (q & 15);
    const bool gate_row = q >= 16;
    if (c >= C || k >= K) return 0.f;
    if (HAS_CTX && k >= 2 * C) return (gate_row ? wcg : wcf)[(size_t)c * C + (k - 2 * C)];
    const int tap = k >= C, kc = k - tap * C;
    const float *src = gate_row ? wg : wf;
    return src[((size_t)c * C + kc) * 2 + tap];
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= K || t < t_begin || t >= t_end) return 0.f;
    if (HAS_CTX && k >= 2 * C) return *ctx.at(b, k - 2 * C, t);
    return k < C ? *xin.at(b, k, t - d) : *xin.at(b, k - C, t);  // t >= t_begin >= d
  }
  __device__ __forceinline__ void store_fg(int b, int c, int t, const f4 &f, const f4 &g) const {
    if (c >= C) return;
    float fb = 0.f, gb = 0.f;
    if (HAS_CTX) {
      fb = bcf[c];
      gb = bcg[c];
    }
    const f4 tv = f4{tanh_fast(f.x + fb), tanh_fast(f.y + fb), tanh_fast(f.z + fb), tanh_fast(f.w + fb)};
    const f4 sv = f4{sigmoid_fast(g.x + gb), sigmoid_fast(g.y + gb), sigmoid_fast(g.z + gb),
                     sigmoid_fast(g.w + gb)};
    const f4 zv = f4{tv.x * sv.x, tv.y * sv.y, tv.z * sv.z, tv.w * sv.w};
    float *zp = z.at(b, c, t);
    if (cols_full(t, t_begin, t_end)) {
      *(f4 *)zp = zv;
      if (th.p) {
        *(f4 *)th.at(b, c, t) = tv;
        *(f4 *)sg.at(b, c, t) = sv;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_end) {
          zp[e] = f4_get(zv, e);
          if (th.p) {
            th.at(b, c, t)[e] = f4_get(tv, e);
            sg.at(b, c, t)[e] = f4_get(sv, e);
          }
        }
    }
  }
};

// F2: residual 1x1 (+bias +input) and skip 1x1 (+bias, accumulated for t >= t_skip0).
// Rows: [0,C) residual channels, [C, C+Kc) skip channels.
struct RsOp {
  int K, t_begin, t_end, C, Kc, t_skip0, t_base;  // skip column of time t = t - t_base
  const float *wr, *br, *ws, *bs;  // (C,C,1),(C),(Kc,C,1),(Kc)
  Act z, xin, xout, skip;
  int first_layer;  // skip is overwritten instead of accumulated
  __device__ __forceinline__ float w(int m, int k) const {
    if (k >= C) return 0.f;
    if (m < C) return wr[(size_t)m * C + k];
    if (m < C + Kc) return ws[(size_t)(m - C) * C + k];
    return 0.f;
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    return (k < C && t >= t_begin && t < t_end) ? *z.at(b, k, t) : 0.f;
  }
  static constexpr int EPI_BATCH = 8;
  static constexpr bool FG_PAIRS = false;
  typedef Pre1 Pre;  // residual input (m < C) or the skip accumulator (m >= C)
  __device__ __forceinline__ Pre load4(int b, int m, int t) const {
    Pre p{kZero4};
    if (m < C) {
      if (xout.p && cols_full(t, t_begin, t_end)) p.a = *(const f4 *)xin.at(b, m, t);
    } else if (m < C + Kc) {
      if (!first_layer && cols_full(t, max(t_begin, t_skip0), t_end))
        p.a = *(const f4 *)skip.at(b, m - C, t - t_base);
    }
    return p;
  }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &v, const Pre &p) const {
    if (m < C) {
      if (!xout.p) return;
      const float bias = br[m];
      const float *xi = xin.at(b, m, t);
      float *xo = xout.at(b, m, t);
      if (cols_full(t, t_begin, t_end)) {
        const f4 xv = p.a;
        *(f4 *)xo = f4{(v.x + bias) + xv.x, (v.y + bias) + xv.y, (v.z + bias) + xv.z,
                       (v.w + bias) + xv.w};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e >= t_begin && t + e < t_end) xo[e] = (f4_get(v, e) + bias) + xi[e];
      }
    } else if (m < C + Kc) {
      const int k = m - C, lo = max(t_begin, t_skip0);
      const float bias = bs[k];
      float *sp = skip.at(b, k, t - t_base);
      if (cols_full(t, lo, t_end)) {
        f4 o = f4{v.x + bias, v.y + bias, v.z + bias, v.w + bias};
        if (!first_layer) {
          const f4 old = p.a;
          o = f4{old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w};
        }
        *(f4 *)sp = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e >= lo && t + e < t_end) {
            const float add = f4_get(v, e) + bias;
            sp[e] = first_layer ? add : sp[e] + add;
          }
      }
    }
  }
};

// F3/F4 and the head's data-gradients: plain 1x1 with selectable input/output maps.
enum { IN_ID = 0, IN_LRELU = 1 };
enum { OUT_BIAS = 0, OUT_BIAS_LRELU = 1, OUT_MUL_DLRELU = 2 };
template <int IN, int OUT, bool TRANSPOSED>
struct DenseOp {
  int K, t_begin, t_end, M;
  const float *wmat;  // (rows, cols, 1): W[m][k] = TRANSPOSED ? wmat[k*ldw + m] : wmat[m*ldw + k]
  int ldw;            // row length of wmat
  const float *bias;
  Act xin, yout, ref;  // ref: activation whose sign gates the lrelu derivative
  int t_out_end;       // columns >= this are not stored (remove_last)
  int aligned_out;     // yout rows are 16-byte aligned at multiples of 4
  __device__ __forceinline__ float w(int m, int k) const {
    if (m >= M || k >= K) return 0.f;
    return TRANSPOSED ? wmat[(size_t)k * ldw + m] : wmat[(size_t)m * ldw + k];
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= K || t < t_begin || t >= t_end) return 0.f;
    const float v = *xin.at(b, k, t);
    return IN == IN_LRELU ? leaky(v) : v;
  }
  __device__ __forceinline__ float map(float v, float bv, float refv) const {
    if (OUT == OUT_BIAS) return v + bv;
    if (OUT == OUT_BIAS_LRELU) return leaky(v + bv);
    return v * (refv > 0.f ? 1.0f : kLeakySlope);
  }
  static constexpr int EPI_BATCH = 8;
  static constexpr bool FG_PAIRS = false;
  typedef Pre1 Pre;  // the activation whose sign gates the leaky-ReLU derivative
  __device__ __forceinline__ Pre load4(int b, int m, int t) const {
    Pre p{kZero4};
    if (OUT == OUT_MUL_DLRELU && m < M && aligned_out && cols_full(t, t_begin, t_out_end))
      p.a = *(const f4 *)ref.at(b, m, t);
    return p;
  }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &v, const Pre &p) const {
    if (m >= M) return;
    const float bv = OUT == OUT_MUL_DLRELU ? 0.f : bias[m];
    float *yo = yout.at(b, m, t);
    const float *rp = OUT == OUT_MUL_DLRELU ? ref.at(b, m, t) : nullptr;
    if (aligned_out && cols_full(t, t_begin, t_out_end)) {
      const f4 r = p.a;
      *(f4 *)yo = f4{map(v.x, bv, r.x), map(v.y, bv, r.y), map(v.z, bv, r.z), map(v.w, bv, r.w)};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_out_end)
          yo[e] = map(f4_get(v, e), bv, OUT == OUT_MUL_DLRELU ? rp[e] : 0.f);
    }
  }
};

// A0 (dense input): the causal conv on an arbitrary (B,Q,T) float input
// (modules.py:19-30: k=2, padding 1, last column dropped): K = 2Q, column k < Q is
// tap 0 on x[t-1] (zero at t = 0), k >= Q is tap 1 on x[t].  One-hot inputs take the
// gather path (embed_kernel) instead.
struct CausalOp {
  int K, t_begin, t_end, C, Q;
  const float *cw;  // (C, Q, 2)
  Act audio, x0;
  __device__ __forceinline__ float w(int m, int k) const {
    if (m >= C || k >= 2 * Q) return 0.f;
    const int tap = k >= Q, q = k - tap * Q;
    return cw[((size_t)m * Q + q) * 2 + tap];
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= 2 * Q || t < t_begin || t >= t_end) return 0.f;
    if (k < Q) return t > 0 ? *audio.at(b, k, t - 1) : 0.f;
    return *audio.at(b, k - Q, t);
  }
  static constexpr int EPI_BATCH = 8;
  static constexpr bool FG_PAIRS = false;
  typedef Pre1 Pre;  // nothing to preload
  __device__ __forceinline__ Pre load4(int, int, int) const { return Pre{kZero4}; }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &v, const Pre &) const {
    if (m >= C) return;
    float *o = x0.at(b, m, t);
    if (cols_full(t, t_begin, t_end)) {
      *(f4 *)o = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_end) o[e] = f4_get(v, e);
    }
  }
};
struct WgCausalOp {
  int t_begin, t_end, C, Q;
  Act dx0, audio;
  float *dcw;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    return m < C ? *dx0.at(b, m, t) : 0.f;
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    if (n >= 2 * Q) return 0.f;
    if (n < Q) return t > 0 ? *audio.at(b, n, t - 1) : 0.f;
    return *audio.at(b, n - Q, t);
  }
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (m >= C || n >= 2 * Q) return nullptr;
    const int tap = n >= Q, q = n - tap * Q;
    return dcw + ((size_t)m * Q + q) * 2 + tap;
  }
  __device__ __forceinline__ float *db(int m) const { return nullptr; }
};

// ---- backward data-gradient ops ------------------------------------------
// B3: dz = Wr^T dxo + Ws^T dskip ; df = dz*sg*(1-th^2) ; dg = dz*th*sg*(1-sg)
struct DzOp {
  int K, t_begin, t_end, C, Kc, t_skip0, t_base;
  const float *wr, *ws;
  Act dxo, dskip, th, sg, dfg;  // dxo.p == NULL for the last layer (output unused)
  __device__ __forceinline__ float w(int m, int k) const {
    if (m >= C) return 0.f;
    if (k < C) return dxo.p ? wr[(size_t)k * C + m] : 0.f;
    if (k < C + Kc) return ws[(size_t)(k - C) * C + m];
    return 0.f;
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (t < t_begin || t >= t_end) return 0.f;
    if (k < C) return dxo.p ? *dxo.at(b, k, t) : 0.f;
    if (k < C + Kc) return t >= t_skip0 ? *dskip.at(b, k - C, t - t_base) : 0.f;
    return 0.f;
  }
  static constexpr int EPI_BATCH = 4;
  static constexpr bool FG_PAIRS = false;
  typedef Pre2 Pre;  // tanh, sigmoid of the forward pass
  __device__ __forceinline__ Pre load4(int b, int m, int t) const {
    Pre p{kZero4, kZero4};
    if (m < C && cols_full(t, t_begin, t_end)) {
      p.a = *(const f4 *)th.at(b, m, t);
      p.b = *(const f4 *)sg.at(b, m, t);
    }
    return p;
  }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &dz, const Pre &p) const {
    if (m >= C) return;
    const float *tp = th.at(b, m, t), *sp = sg.at(b, m, t);
    float *df = dfg.at(b, m, t), *dg = dfg.at(b, C + m, t);
    if (cols_full(t, t_begin, t_end)) {
      const f4 tv = p.a, sv = p.b;
      *(f4 *)df = f4{dz.x * sv.x * (1.0f - tv.x * tv.x), dz.y * sv.y * (1.0f - tv.y * tv.y),
                     dz.z * sv.z * (1.0f - tv.z * tv.z), dz.w * sv.w * (1.0f - tv.w * tv.w)};
      *(f4 *)dg = f4{dz.x * tv.x * sv.x * (1.0f - sv.x), dz.y * tv.y * sv.y * (1.0f - sv.y),
                     dz.z * tv.z * sv.z * (1.0f - sv.z), dz.w * tv.w * sv.w * (1.0f - sv.w)};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_end) {
          const float d = f4_get(dz, e), tv = tp[e], sv = sp[e];
          df[e] = d * sv * (1.0f - tv * tv);
          dg[e] = d * tv * sv * (1.0f - sv);
        }
    }
  }
};

// B4: dx[u] = [u>=t_lo] (dxo[u] + W1^T dfg[u]) + [u+d<T] W0^T dfg[u+d]
//     k in [0,2C): tap 1 rows of dfg at u ; k in [2C,4C): tap 0 rows at u+d
struct DxOp {
  int K, t_begin, t_end, C, d, t_lo;  // t_lo = A_{l+1}; outputs cover [A_l, T)
  const float *wf, *wg;
  Act dxo, dfg, dxi;
  __device__ __forceinline__ float w(int m, int k) const {
    if (m >= C || k >= 4 * C) return 0.f;
    const int tap = k < 2 * C ? 1 : 0, kk = k - (tap ? 0 : 2 * C);
    const float *src = kk < C ? wf : wg;
    const int o = kk < C ? kk : kk - C;
    return src[((size_t)o * C + m) * 2 + tap];
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= 4 * C || t < t_begin || t >= t_end) return 0.f;
    if (k < 2 * C) return t >= t_lo ? *dfg.at(b, k, t) : 0.f;
    return t + d < t_end ? *dfg.at(b, k - 2 * C, t + d) : 0.f;
  }
  static constexpr int EPI_BATCH = 8;
  static constexpr bool FG_PAIRS = false;
  typedef Pre1 Pre;  // gradient arriving through the residual connection
  __device__ __forceinline__ Pre load4(int b, int m, int t) const {
    Pre p{kZero4};
    if (m < C && dxo.p && cols_full(t, t_begin, t_end) && t >= t_lo) p.a = *(const f4 *)dxo.at(b, m, t);
    return p;
  }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &v, const Pre &p) const {
    if (m >= C) return;
    float *o = dxi.at(b, m, t);
    const float *po = dxo.p ? dxo.at(b, m, t) : nullptr;
    if (cols_full(t, t_begin, t_end) && (!po || t >= t_lo)) {
      f4 r = v;
      if (po) {
        const f4 u = p.a;
        r = f4{v.x + u.x, v.y + u.y, v.z + u.z, v.w + u.w};
      }
      *(f4 *)o = r;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_end)
          o[e] = f4_get(v, e) + ((po && t + e >= t_lo) ? po[e] : 0.f);
    }
  }
};

// dWf/dWg: A = dfg rows (f | g), X = layer input (past | cur) -> (C,C,2) taps
template <bool HAS_CTX>
struct WgFgOpT {
  int t_begin, t_end, C, d;
  Act dfg, xin, ctx;  // ctx.p == NULL: audio only
  float *dwf, *dwg, *dwcf, *dwcg, *dbcf, *dbcg;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    return m < 2 * C ? *dfg.at(b, m, t) : 0.f;
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    if (n >= 2 * C) return (HAS_CTX && n < 3 * C) ? *ctx.at(b, n - 2 * C, t) : 0.f;
    return n < C ? *xin.at(b, n, t - d) : *xin.at(b, n - C, t);
  }
  // row descriptors for wgrad2 (gemm_family.h): p[t] = value at absolute time t
  static constexpr bool X_PRODUCT = false;
  static constexpr bool HAS_BIAS = HAS_CTX;  // only the context convs carry biases
  static constexpr bool X_ABSENT_ROWS = HAS_CTX;  // 3C rows fill one and a half 128-row blocks
  __device__ __forceinline__ float xmap(float v) const { return v; }
  __device__ __forceinline__ const float *a_ptr(int b, int m) const { return dfg.at(b, min(m, 2 * C - 1), 0); }
  __device__ __forceinline__ int a_lo(int m) const { return m < 2 * C ? t_begin : 0; }
  __device__ __forceinline__ int a_hi(int m) const { return m < 2 * C ? t_end : 0; }
  __device__ __forceinline__ const float *x_ptr(int b, int n) const {
    if (HAS_CTX && n >= 2 * C) return ctx.at(b, min(n - 2 * C, C - 1), 0);
    const int nc = min(n, 2 * C - 1);
    // the past tap reads t - d >= t_begin - d >= 0
    return nc < C ? xin.at(b, nc, 0) - d : xin.at(b, nc - C, 0);
  }
  __device__ __forceinline__ const float *x_ptr2(int, int) const { return nullptr; }
  __device__ __forceinline__ int x_lo(int n) const { return n < (HAS_CTX ? 3 : 2) * C ? t_begin : 0; }
  __device__ __forceinline__ int x_hi(int n) const { return n < (HAS_CTX ? 3 : 2) * C ? t_end : 0; }
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (m >= 2 * C) return nullptr;
    const int o = m < C ? m : m - C;
    if (n >= 2 * C) {
      if (!HAS_CTX || n >= 3 * C) return nullptr;
      return (m < C ? dwcf : dwcg) + (size_t)o * C + (n - 2 * C);
    }
    float *base = m < C ? dwf : dwg;
    const int tap = n >= C, c = n - tap * C;
    return base + ((size_t)o * C + c) * 2 + tap;
  }
  __device__ __forceinline__ float *db(int m) const {
    if (!HAS_CTX || m >= 2 * C) return nullptr;
    return m < C ? dbcf + m : dbcg + (m - C);
  }
};

// gradient w.r.t. the context: dctx[t] += Wcf^T df[t] + Wcg^T dg[t]   (t >= t_lo)
struct DctxOp {
  int K, t_begin, t_end, C;
  const float *wcf, *wcg;
  Act dfg, dctx;
  __device__ __forceinline__ float w(int m, int k) const {
    if (m >= C || k >= 2 * C) return 0.f;
    return k < C ? wcf[(size_t)k * C + m] : wcg[(size_t)(k - C) * C + m];
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    return (k < 2 * C && t >= t_begin && t < t_end) ? *dfg.at(b, k, t) : 0.f;
  }
  static constexpr int EPI_BATCH = 8;
  static constexpr bool FG_PAIRS = false;
  typedef Pre1 Pre;  // the context gradient accumulated so far
  __device__ __forceinline__ Pre load4(int b, int m, int t) const {
    Pre p{kZero4};
    if (m < C && cols_full(t, t_begin, t_end)) p.a = *(const f4 *)dctx.at(b, m, t);
    return p;
  }
  __device__ __forceinline__ void store4(int b, int m, int t, const f4 &v, const Pre &p) const {
    if (m >= C) return;
    float *o = dctx.at(b, m, t);
    if (cols_full(t, t_begin, t_end)) {
      const f4 u = p.a;
      *(f4 *)o = f4{u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e >= t_begin && t + e < t_end) o[e] += f4_get(v, e);
    }
  }
};

// dWr,dbr,dWs,dbs: A = [dxo ; dskip], X = z = th*sg
struct WgRsOp {
  int t_begin, t_end, C, Kc, t_skip0, t_base;
  Act dxo, dskip, th, sg;  // dxo.p == NULL for the last layer
  float *dwr, *dbr, *dws, *dbs;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    if (m < C) return dxo.p ? *dxo.at(b, m, t) : 0.f;  // the wgrad kernel passes t < t_end only
    if (m < C + Kc) {
      const int lo = max(t_begin, t_skip0);
      return ld_masked(dskip, b, m - C, t - t_base, lo - t_base, t_end - t_base);
    }
    return 0.f;
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    return n < C ? *th.at(b, n, t) * *sg.at(b, n, t) : 0.f;
  }
  // row descriptors for wgrad2: dskip's column axis is t - t_base, valid from t_skip0
  static constexpr bool X_PRODUCT = true;  // z = th * sg
  static constexpr bool HAS_BIAS = true;
  static constexpr bool X_ABSENT_ROWS = false;
  __device__ __forceinline__ float xmap(float v) const { return v; }
  __device__ __forceinline__ const float *a_ptr(int b, int m) const {
    if (m < C) return dxo.p ? dxo.at(b, m, 0) : dskip.at(b, 0, 0);  // absent row: p[0] must exist
    if (m < C + Kc) return dskip.at(b, m - C, 0) - t_base;
    return dskip.at(b, 0, 0);
  }
  __device__ __forceinline__ int a_lo(int m) const {
    if (m < C) return dxo.p ? t_begin : 0;
    return m < C + Kc ? max(t_begin, t_skip0) : 0;
  }
  __device__ __forceinline__ int a_hi(int m) const {
    if (m < C) return dxo.p ? t_end : 0;
    return m < C + Kc ? t_end : 0;
  }
  __device__ __forceinline__ const float *x_ptr(int b, int n) const { return th.at(b, min(n, C - 1), 0); }
  __device__ __forceinline__ const float *x_ptr2(int b, int n) const { return sg.at(b, min(n, C - 1), 0); }
  __device__ __forceinline__ int x_lo(int n) const { return n < C ? t_begin : 0; }
  __device__ __forceinline__ int x_hi(int n) const { return n < C ? t_end : 0; }
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (n >= C) return nullptr;
    if (m < C) return dxo.p ? dwr + (size_t)m * C + n : nullptr;
    if (m < C + Kc) return dws + (size_t)(m - C) * C + n;
    return nullptr;
  }
  __device__ __forceinline__ float *db(int m) const {
    if (m < C) return dxo.p ? dbr + m : nullptr;
    if (m < C + Kc) return dbs + (m - C);
    return nullptr;
  }
};

// head: dW[m][n] += sum A[m][s] * in[n][s]
template <int IN>
struct WgDenseOp {
  int t_begin, t_end, M, N;
  Act aact, xact;
  float *dwm, *dbv;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    return m < M ? *aact.at(b, m, t) : 0.f;
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    if (n >= N) return 0.f;
    const float v = *xact.at(b, n, t);
    return IN == IN_LRELU ? leaky(v) : v;
  }
  __device__ __forceinline__ float *dw(int m, int n) const {
    return (m < M && n < N) ? dwm + (size_t)m * N + n : nullptr;
  }
  __device__ __forceinline__ float *db(int m) const { return m < M ? dbv + m : nullptr; }
  // row descriptors for wgrad2 (the head's column axis starts at t_begin = pad < 32)
  static constexpr bool X_PRODUCT = false;
  static constexpr bool HAS_BIAS = true;
  static constexpr bool X_ABSENT_ROWS = false;
  __device__ __forceinline__ const float *a_ptr(int b, int m) const { return aact.at(b, min(m, M - 1), 0); }
  __device__ __forceinline__ int a_lo(int m) const { return m < M ? t_begin : 0; }
  __device__ __forceinline__ int a_hi(int m) const { return m < M ? t_end : 0; }
  __device__ __forceinline__ const float *x_ptr(int b, int n) const { return xact.at(b, min(n, N - 1), 0); }
  __device__ __forceinline__ const float *x_ptr2(int, int) const { return nullptr; }
  __device__ __forceinline__ int x_lo(int n) const { return n < N ? t_begin : 0; }
  __device__ __forceinline__ int x_hi(int n) const { return n < N ? t_end : 0; }
  __device__ __forceinline__ float xmap(float v) const { return IN == IN_LRELU ? leaky(v) : v; }
};

// ======================================================================
// element-wise kernels
// ======================================================================
// causal conv on a one-hot input = two embedding rows (modules.py:28-30)
__global__ void embed_kernel(const float *__restrict__ cw, const int32_t *__restrict__ idx,
                             int idx_stride, Act x0, int C, int Q, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const int32_t *ib = idx + (size_t)b * idx_stride;
  const int q1 = min(max(ib[t], 0), Q - 1);
  float v = cw[((size_t)c * Q + q1) * 2 + 1];
  if (t > 0) v += cw[((size_t)c * Q + min(max(ib[t - 1], 0), Q - 1)) * 2 + 0];
  *x0.at(b, c, t) = v;
}

// ... with the channel's two table rows (Q x 2 floats) in LDS and four steps per thread (r3: the global gathers of
// the kernel above were two uncoalesced L2 reads per output -- 42 us for 65 MB at config 2).  Q <= 1024.
__global__ __launch_bounds__(256) void embed4_kernel(const float *__restrict__ cw, const int32_t *__restrict__ idx,
                                                     int idx_stride, Act x0, int Q, int T) {
  __shared__ float tab[2 * 1024];
  const int c = blockIdx.y, b = blockIdx.z, t = 4 * (blockIdx.x * 256 + threadIdx.x);
  for (int i = threadIdx.x; i < 2 * Q; i += 256) tab[i] = cw[(size_t)c * Q * 2 + i];
  __syncthreads();
  if (t >= T) return;
  const int32_t *ib = idx + (size_t)b * idx_stride;
  int qv[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    const int u = t - 1 + e;
    qv[e] = (u >= 0 && u < T) ? min(max(ib[u], 0), Q - 1) : -1;
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = qv[e + 1] >= 0 ? tab[2 * qv[e + 1] + 1] : 0.f;
    if (qv[e] >= 0) v[e] += tab[2 * qv[e]];
  }
  float *dst = x0.at(b, c, t);
  if (t + 3 < T) {
    *(f4 *)dst = f4{v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t + e < T) dst[e] = v[e];
  }
}

// Gradient of the causal conv on a one-hot input = a scatter-add of dx0 columns into the two
// embedding tables (tap 1 at class idx[t], tap 0 at idx[t-1]).  A workgroup owns `cg` channels
// and EG_CHUNK time steps of one sequence and accumulates them in an LDS copy of its table
// slice [cg][Q][2] (ds_add_f32: lanes walk t, so a collision needs two lanes of one wave
// on the same class); the slice is then added to HBM once -- T/EG_CHUNK times fewer
// global atomics than one per (channel, time step).
constexpr int EG_CHUNK = 1024;
__global__ __launch_bounds__(1024) void embed_grad_kernel(float *__restrict__ dcw,
                                                         const int32_t *__restrict__ idx, int idx_stride,
                                                         Act dx0, int C, int Q, int T, int cg,
                                                         float *__restrict__ part) {
  extern __shared__ float tab[];  // [cg][Q][2]
  const int b = blockIdx.z, c0 = blockIdx.y * cg, t0 = blockIdx.x * EG_CHUNK;
  const int nc = min(cg, C - c0), t1 = min(T, t0 + EG_CHUNK);
  for (int i = threadIdx.x; i < cg * Q * 2; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const int32_t *ib = idx + (size_t)b * idx_stride;
  for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
    const int q1 = min(max(ib[t], 0), Q - 1), q0 = t > 0 ? min(max(ib[t - 1], 0), Q - 1) : -1;
    // eight channels' loads in flight per thread (one at a time, each in front of its two LDS
    // atomics, the kernel was a chain of load latencies: 183 us for 65 MB)
    int c = 0;
    for (; c + 8 <= nc; c += 8) {
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = *dx0.at(b, c0 + c + j, t);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&tab[((c + j) * Q + q1) * 2 + 1], g[j]);
        if (q0 >= 0) atomicAdd(&tab[((c + j) * Q + q0) * 2 + 0], g[j]);
      }
    }
    for (; c < nc; ++c) {
      const float g = *dx0.at(b, c0 + c, t);
      atomicAdd(&tab[(c * Q + q1) * 2 + 1], g);
      if (q0 >= 0) atomicAdd(&tab[(c * Q + q0) * 2 + 0], g);
    }
  }
  __syncthreads();
  if (part) {
    // this workgroup's slice of table copy (chunk, sequence); slab_reduce_kernel adds the copies up.
    // (Flushed with atomics, the 128 workgroups per table word queued up at the memory side:
    // 4.2 M device-scope float atomics took ~150 of the kernel's 181 us.)
    float *dst = part + ((size_t)blockIdx.x * gridDim.z + b) * ((size_t)C * Q * 2) + (size_t)c0 * Q * 2;
    for (int i = threadIdx.x; i < nc * Q * 2; i += blockDim.x) dst[i] = tab[i];
    return;
  }
  for (int i = threadIdx.x; i < nc * Q * 2; i += blockDim.x) {
    const float v = tab[i];
    if (v != 0.f) atomicAdd(dcw + (size_t)c0 * Q * 2 + i, v);
  }
}
// The same scatter-add for C = 64 without atomics: LANES ARE CHANNELS, ONE WAVE PER WORKGROUP.
// ds_add_f32 costs ~3 cycles per LANE on this chip (32.8 M lane-atomics = the 171 us of the
// kernel above, whatever the bank pattern: the same time with conflict-free addresses).  Here a
// wave owns the LDS table of one tap, [class][channel] (64 KB): the class of a time step is a
// scalar, the 64 lanes read-modify-write 64 consecutive words with plain ds_read / ds_write, and
// LDS operations of one wave execute in order.  Four steps are in flight at a time; equal
// classes among them are merged first (scalar compares), so no update is lost.  A lane fetches
// four steps of its row with one 16-byte load, eight such groups a block ahead.
constexpr int EG64_CHUNK = 1024, EG64_PARTS = 1;
__global__ __launch_bounds__(64) void embed_grad64_kernel(const int32_t *__restrict__ idx, int idx_stride,
                                                          Act dx0, int Q, int T, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [Q / EG64_PARTS][64]
  // (EG64_PARTS > 1 splits the classes of a tap over several workgroups, each scanning every step:
  // measured slower with 4 -- the per-group instruction stream is the cost, not the LDS latency)
  const int b = blockIdx.y, tap = blockIdx.z / EG64_PARTS, prt = blockIdx.z % EG64_PARTS, lane = threadIdx.x;
  const int QP = Q / EG64_PARTS, qbase = prt * QP;
  const int t0 = blockIdx.x * EG64_CHUNK, t1 = min(T, t0 + EG64_CHUNK);
  for (int i = lane; i < QP * 16; i += 64) ((f4 *)tab)[i] = kZero4;
  __syncthreads();
  const int32_t *ib = idx + (size_t)b * idx_stride;
  const float *row = dx0.at(b, lane, 0);
  // (rows are padded to a multiple of 64 steps: a 16-byte load at a multiple of 4 below the row
  // length is always inside the row; steps at or beyond t1 carry class -1 and are not used)
  const int t_last = dx0.ld - 4;
  auto fetch = [&](int t) -> f4 { return ldg4(row + min(t, t_last)); };
  // class of step t0 + i for this tap (tap 1: idx[t], tap 0: idx[t - 1]; -1 = no contribution), staged
  // in LDS once: fetched from global memory per group the index loads were a chain of 256 L2
  // round trips per wave -- the whole 165 us of the first build of this kernel
  int *sidx = (int *)(tab + (size_t)QP * 64);
  for (int i = lane; i < EG64_CHUNK; i += 64) {
    const int t = t0 + i, u = t - 1 + tap;
    const int q = (t < t1 && u >= 0) ? min(max(ib[u], 0), Q - 1) - qbase : -1;
    sidx[i] = q < QP ? q : -1;
  }
  __syncthreads();
  typedef int i4 __attribute__((ext_vector_type(4)));
  // eight groups of four steps (8 x 16 bytes per lane) are fetched a block ahead
  constexpr int NG = 8;
  f4 nxt[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) nxt[k] = fetch(t0 + 4 * k);
  for (int tb = t0; tb < t1; tb += 4 * NG) {
    f4 cur[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) cur[k] = nxt[k];
    if (tb + 4 * NG < t1) {
#pragma unroll
      for (int k = 0; k < NG; ++k) nxt[k] = fetch(tb + 4 * NG + 4 * k);
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int t = tb + 4 * k;
      if (t >= t1) break;
      const f4 v = cur[k];
      const i4 qv = *(const i4 *)&sidx[t - t0];
      const int q0 = __builtin_amdgcn_readfirstlane(qv.x), q1 = __builtin_amdgcn_readfirstlane(qv.y);
      const int q2 = __builtin_amdgcn_readfirstlane(qv.z), q3 = __builtin_amdgcn_readfirstlane(qv.w);
      float g0 = v.x, g1 = v.y, g2 = v.z, g3 = v.w;
      // merge equal classes into the first step that has them
      const bool d1 = q1 == q0;
      if (d1) g0 += g1;
      const bool d2 = q2 == q0 || q2 == q1;
      if (q2 == q0) g0 += g2; else if (q2 == q1) g1 += g2;
      const bool d3 = q3 == q0 || q3 == q1 || q3 == q2;
      if (q3 == q0) g0 += g3; else if (q3 == q1) g1 += g3; else if (q3 == q2) g2 += g3;
      const bool l0 = q0 >= 0, l1 = q1 >= 0 && !d1, l2 = q2 >= 0 && !d2, l3 = q3 >= 0 && !d3;
      float *p0 = tab + (l0 ? q0 : 0) * 64 + lane, *p1 = tab + (l1 ? q1 : 0) * 64 + lane;
      float *p2 = tab + (l2 ? q2 : 0) * 64 + lane, *p3 = tab + (l3 ? q3 : 0) * 64 + lane;
      const float o0 = *p0, o1 = *p1, o2 = *p2, o3 = *p3;
      if (l0) *p0 = o0 + g0;
      if (l1) *p1 = o1 + g1;
      if (l2) *p2 = o2 + g2;
      if (l3) *p3 = o3 + g3;
    }
  }
  __syncthreads();
  f4 *dst = (f4 *)(part + ((size_t)blockIdx.x * gridDim.y + b) * ((size_t)2 * Q * 64) + ((size_t)tap * Q + qbase) * 64);
  for (int i = lane; i < QP * 16; i += 64) dst[i] = ((const f4 *)tab)[i];
}
// r3: the same gradient as a PRODUCT on the bf16 matrix cores -- dW[tap][q][c] = sum_t onehot[q][t - 1 + tap] dx0[c][t].
// The one-hot operand is built in registers from the tile's class indices (a lane owns class row q: eight compares
// per operand; 1.0 is exact in the top bf16 plane, the other two planes are zero), dx0 is split into its three
// planes (bf3.h): three MFMAs per block, every product exact, fp32 accumulation.  Workgroup = 4 waves on a chunk of
// EG64_CHUNK steps in tiles of 64; wave w owns classes [64 w, +64) of both taps and all 64 channels (128
// accumulator registers); slabs in embed_grad64_kernel's format.  Q = 256 only.
__global__ __launch_bounds__(256, 2) void embed_grad64_mfma_kernel(const int32_t *__restrict__ idx, int idx_stride, Act dx0,
                                                                  int T, float *__restrict__ part) {
  constexpr int Q = 256, LD = W2_LD, TT = W2_T;
  __shared__ __attribute__((aligned(16))) float Xs[64][LD];
  __shared__ __attribute__((aligned(16))) int sidx[TT + 8];  // classes of steps t0 - 1 .. t0 + 63 (then padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, t_begin = blockIdx.x * EG64_CHUNK, t_end = min(T, t_begin + EG64_CHUNK);
  const int32_t *ib = idx + (size_t)b * idx_stride;
  f32x16 acc[2][2][2];  // [tap][class block][channel block]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][j][r] = 0.f;
  const int srow = tid >> 4, st = 4 * (tid & 15);
  // the next tile's dx0 values and classes are fetched into registers under this tile's products
  f4 nv[4];
  int ncls = -1;
  auto fetch = [&](int t0) {
    const int t = t0 + st;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f4 v = ldg4(dx0.at(b, 16 * p + srow, 0) + min(t, dx0.ld - 4));
      if (t + 3 >= t_end) {
        v.x = t < t_end ? v.x : 0.f;
        v.y = t + 1 < t_end ? v.y : 0.f;
        v.z = t + 2 < t_end ? v.z : 0.f;
        v.w = t + 3 < t_end ? v.w : 0.f;
      }
      nv[p] = v;
    }
    if (tid < TT + 1) {
      const int u = t0 - 1 + tid;  // class of step u, -1 where there is none
      ncls = (u >= 0 && u < t_end) ? min(max(ib[u], 0), Q - 1) : -1;
    }
  };
  fetch(t_begin);
  for (int t0 = t_begin; t0 < t_end; t0 += TT) {
    __syncthreads();  // the tile before has been read
#pragma unroll
    for (int p = 0; p < 4; ++p) *(f4 *)&Xs[16 * p + srow][st] = nv[p];
    if (tid < TT + 1) sidx[tid] = ncls;
    if (t0 + TT < t_end) fetch(t0 + TT);
    __syncthreads();
#pragma unroll
    for (int G = 0; G < TT / 16; ++G) {
      // channel operands: eight consecutive steps of row 32 cb + li, three planes
      u32x4 xh[2], xm[2], xl[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const f4 v0 = *(const f4 *)&Xs[32 * cb + li][16 * G + 8 * lh], v1 = *(const f4 *)&Xs[32 * cb + li][16 * G + 8 * lh + 4];
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf3_split8(v, xh[cb], xm[cb], xl[cb]);
      }
      // the nine classes of steps 16 G + 8 lh - 1 .. + 7 (tap 0 reads e, tap 1 reads e + 1)
      int cls[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) cls[e] = sidx[16 * G + 8 * lh + e];
#pragma unroll
      for (int tap = 0; tap < 2; ++tap)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int qrow = 64 * wave + 32 * rb + li;
          u32x4 oh;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            oh[i] = (cls[2 * i + tap] == qrow ? 0x3F80u : 0u) | (cls[2 * i + 1 + tap] == qrow ? 0x3F800000u : 0u);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x16 &c = acc[tap][rb][cb];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, oh), __builtin_bit_cast(bf16x8, xl[cb]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, oh), __builtin_bit_cast(bf16x8, xm[cb]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, oh), __builtin_bit_cast(bf16x8, xh[cb]), c, 0, 0, 0);
          }
        }
    }
  }
  float *dst = part + ((size_t)blockIdx.x * gridDim.y + b) * ((size_t)2 * Q * 64);
#pragma unroll
  for (int tap = 0; tap < 2; ++tap)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = 64 * wave + 32 * rb + acc_row(r, lane), c = 32 * cb + li;
          dst[((size_t)tap * Q + qq) * 64 + c] = acc[tap][rb][cb][r];
        }
}
struct EmbedSlabOp64 {  // slab word (tap * Q + q, c) -> the (C, Q, 2) table
  float *dcw;
  int Q;
  __device__ __forceinline__ float *dw(int m, int n) const {
    const int tap = m >= Q ? 1 : 0, q = m - tap * Q;
    return dcw + ((size_t)n * Q + q) * 2 + tap;
  }
};
struct EmbedSlabOp {  // slab_reduce_kernel's view of the (C, Q, 2) table as rows of 64 words
  float *dcw;
  __device__ __forceinline__ float *dw(int m, int n) const { return dcw + (size_t)m * 64 + n; }
};

// softmax over channels, in place; one thread per (b, column)
__global__ void softmax_kernel(float *__restrict__ y, int Q, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (s >= S) return;
  float *col = y + (size_t)b * Q * S + s;
  float m = -INFINITY;
  for (int q = 0; q < Q; ++q) m = fmaxf(m, col[(size_t)q * S]);
  float sum = 0.f;
  for (int q = 0; q < Q; ++q) {
    const float e = sm_exp(col[(size_t)q * S] - m);
    col[(size_t)q * S] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  for (int q = 0; q < Q; ++q) col[(size_t)q * S] = col[(size_t)q * S] * inv;
}

// dlogit = normalize ? p * (dout - sum_q dout*p) : dout ; zero for columns >= S_out
__global__ void softmax_bwd_kernel(const float *__restrict__ p, const float *__restrict__ dout,
                                   Act dlogit, int Q, int S_out, int S, int normalize, int pad) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (s >= S) return;
  if (s >= S_out) {
    for (int q = 0; q < Q; ++q) *dlogit.at(b, q, s + pad) = 0.f;
    return;
  }
  const float *pc = p + (size_t)b * Q * S_out + s, *dc = dout + (size_t)b * Q * S_out + s;
  float dot = 0.f;
  if (normalize)
    for (int q = 0; q < Q; ++q) dot += dc[(size_t)q * S_out] * pc[(size_t)q * S_out];
  for (int q = 0; q < Q; ++q) {
    const float d = dc[(size_t)q * S_out];
    *dlogit.at(b, q, s + pad) = normalize ? pc[(size_t)q * S_out] * (d - dot) : d;
  }
}

// ---- single-pass column kernels (Q <= 256): a workgroup owns 64 columns, wave w the class
// rows [64w, 64w+64) of them, lane = column.  A thread keeps its quarter of a column in 64
// registers, so the tensor is read once and written once (the thread-per-column forms above
// walk the column two or three times: 636 MB instead of 212 MB at config 2); the column-wide
// maximum / sum go through 1 KB of LDS.
constexpr int CQ = 64;  // rows per wave = registers per thread
__device__ __forceinline__ float col_reduce(float v, float (*part)[64], int wave, int lane, bool is_max) {
  part[wave][lane] = v;
  __syncthreads();
  const float a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
  __syncthreads();
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

__global__ __launch_bounds__(256) void softmax_cols_kernel(float *__restrict__ y, int Q, int S) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x * 64 + lane;
  const bool live = s < S;
  const __amdgpu_buffer_rsrc_t yb = col_rsrc(y + (size_t)b * Q * S);
  const int voff = 4 * (live ? s : 0), row = 4 * S;
  float v[CQ];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int q = CQ * wave + i;
    v[i] = q < Q ? col_ld(yb, voff, q * row) : -INFINITY;
    v[i] = live ? v[i] : -INFINITY;
    m = fmaxf(m, v[i]);
  }
  m = col_reduce(m, part, wave, lane, true);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    v[i] = sm_exp(v[i] - m);  // exp(-inf) = 0 for the padding rows
    sum += v[i];
  }
  sum = col_reduce(sum, part, wave, lane, false);
  if (!live) return;
  const float inv = 1.0f / sum;  // (one division per column; softmax_ce_fwd_cols_kernel forms the same bits)
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int q = CQ * wave + i;
    if (q < Q) col_st(v[i] * inv, yb, voff, q * row);
  }
}

// dlogit = p * (dout - sum_q dout*p), zero for columns >= S_out (normalize != 0 only)
__global__ __launch_bounds__(256) void softmax_bwd_cols_kernel(const float *__restrict__ p,
                                                               const float *__restrict__ dout, Act dlogit,
                                                               int Q, int S_out, int S, int pad) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int s = blockIdx.x * 64 + lane;
  const bool live = s < S_out;
  const size_t off = (size_t)b * Q * S_out + (live ? s : 0);
  float pv[CQ], dv[CQ];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int q = CQ * wave + i;
    const bool ok = live && q < Q;
    pv[i] = ok ? p[off + (size_t)q * S_out] : 0.f;
    dv[i] = ok ? dout[off + (size_t)q * S_out] : 0.f;
    dot += dv[i] * pv[i];
  }
  dot = col_reduce(dot, part, wave, lane, false);
  if (s >= S) return;
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int q = CQ * wave + i;
    if (q < Q) *dlogit.at(b, q, s + pad) = live ? pv[i] * (dv[i] - dot) : 0.f;
  }
}

// dilation queues <- saved layer inputs (state layout: generate.hip ring_offset)
__global__ void ring_fill_kernel(const float *__restrict__ acts, long long act_stride, Act view,
                                 float *__restrict__ state, long long state_per_seq, int C,
                                 int layer_size, int L, int t_last) {
  // one block per (layer, b); thread over (slot, c)
  const int l = blockIdx.x, b = blockIdx.y;
  const int d = 1 << (l % layer_size);
  const int stack = l / layer_size, pos = l - stack * layer_size;
  const int off = C * (stack * ((1 << layer_size) - 1) + ((1 << pos) - 1));
  const float *x = acts + (size_t)l * act_stride;
  for (int i = threadIdx.x; i < d * C; i += blockDim.x) {
    const int c = i / d, j = i - c * d;  // j-th most recent time
    const int t = t_last - j;
    if (t < 0) continue;
    state[(size_t)b * state_per_seq + off + (t & (d - 1)) * C + c] =
        x[(size_t)b * view.sb + (size_t)c * view.ld + t];
  }
}

// ======================================================================
// host orchestration
// ======================================================================
struct Geometry {
  int L, C, Kc, Q, T, Tp, S, Sp, rf;
  int pad, t_base;  // skip/head tensors: column of time t = t - t_base, t_base = (RF-1) & ~31,
                    // so that column == t (mod 32): float4 accesses and cache lines line up on
                    // both axes; the first valid column is pad = (RF-1) & 31
  long long act;   // floats per (B,C,Tp) tensor
  long long skp;   // floats per (B,Kc,Sp)
  long long hid;   // floats per (B,Q,Sp)
};

static int make_geometry(const mvn_dims *d, int batch, int t_len, Geometry &g) {
  int rc = validate_dims(d);
  if (rc) return rc;
  if (batch < 0 || t_len < 1) {
    set_error("batch %d / t_len %d out of range", batch, t_len);
    return MVN_ERR_BAD_ARG;
  }
  g.L = n_layers(d);
  g.C = d->residual_channels;
  g.Kc = d->skip_channels;
  g.Q = d->input_channels;
  g.T = t_len;
  g.rf = (int)(dilation_sum(d) + d->stack_size);
  g.S = mvn_output_size(d, t_len);
  if (g.S < 0) return g.S;
  g.Tp = mvn_padded_len(t_len);
  g.pad = (g.rf - 1) & 31;
  g.t_base = (g.rf - 1) - g.pad;
  g.Sp = mvn_padded_len(g.S + 31);
  g.act = (long long)batch * g.C * g.Tp;
  g.skp = (long long)batch * g.Kc * g.Sp;
  g.hid = (long long)batch * g.Q * g.Sp;
  return MVN_OK;
}

}  // namespace mvn

namespace mvn {
// A second stream per device for the backward pass: the weight gradients of a layer run on
// it, concurrently with the data gradients on the caller's stream (both read the same
// tensors: one pass over HBM instead of two, and the tails of one kernel are filled by the
// other).  Fork/join with events; on return everything is ordered on the caller's stream.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};
static int side_stream(SideStream **out) {
  static SideStream table[64];
  static std::mutex mu;
  int dev = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return MVN_ERR_LAUNCH;
  if (dev < 0 || dev >= 64) {
    set_error("device ordinal %d out of range", dev);
    return MVN_ERR_BAD_ARG;
  }
  std::lock_guard<std::mutex> lock(mu);
  SideStream &e = table[dev];
  if (!e.s) {
    hipStream_t st;
    if (check_hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreateWithFlags"))
      return MVN_ERR_LAUNCH;
    for (int i = 0; i < 4; ++i)
      if (check_hip(hipEventCreateWithFlags(&e.ev[i], hipEventDisableTiming), "hipEventCreateWithFlags"))
        return MVN_ERR_LAUNCH;
    e.s = st;
  }
  *out = &e;
  return MVN_OK;
}
}  // namespace mvn

using namespace mvn;

// The head's four convolution-shaped products as bf16 x 3 strip kernels (fused_fwd_bf3.h) for Q in {64, 128, 256}
// (r4: the reference's own experiments run Q = 128, experiments/03_kinetics_scale_up.mk:7-10): conv1 = K 64 -> Q rows
// in one workgroup, conv2 = Q -> Q in row blocks of 64, their data gradients with the weights transposed.
template <int QQ>
static int head_fwd1_bf3(const DenseStripArgs &da, const mvn_params *p, int Kc, float *img, int batch, hipStream_t s) {
  launch_ds3_pack<64, QQ>(p->head1_w, Kc, p->head1_b, QQ, img, s);
  launch_ds3_pack<QQ, 64>(p->head2_w, QQ, p->head2_b, QQ, img + DS3_IMG1_F, s);
  return launch_dense_strip_bf3<64, QQ, IN_LRELU, OUT_BIAS_LRELU>(da, img, QQ, batch, s);
}
template <int QQ>
static int head_fwd2_bf3(const DenseStripArgs &da, float *img, int batch, hipStream_t s) {
  return launch_dense_strip_bf3<QQ, 64, IN_ID, OUT_BIAS>(da, img + DS3_IMG1_F, QQ, batch, s);
}
template <int QQ>
static int head_bwd2_bf3(const DenseStripArgs &da, const mvn_params *p, float *img, int batch, hipStream_t s) {
  launch_ds3_pack<QQ, 64, true>(p->head2_w, QQ, nullptr, QQ, img, s);
  return launch_dense_strip_bf3<QQ, 64, IN_ID, OUT_MUL_DLRELU, true>(da, img, QQ, batch, s);
}
template <int QQ>
static int head_bwd1_bf3(const DenseStripArgs &da, const mvn_params *p, int Kc, float *img, int batch, hipStream_t s) {
  launch_ds3_pack<QQ, 64, true>(p->head1_w, Kc, nullptr, Kc, img + 4 * DS3_IMG2_F, s);
  return launch_dense_strip_bf3<QQ, 64, IN_ID, OUT_MUL_DLRELU, true>(da, img + 4 * DS3_IMG2_F, Kc, batch, s);
}
#define MVN_HEAD_Q(call_)                               \
  (Q == 256 ? call_<256> : Q == 128 ? call_<128> : call_<64>)
static bool head_q_strip(int Q) { return Q == 256 || Q == 128 || Q == 64; }

extern "C" {

int mvn_padded_len(int n) { return n <= 0 ? 0 : (n + 63) / 64 * 64; }

static int forward_impl(const mvn_dims *dims, const mvn_params *p, const int32_t *index, int index_stride,
                        int batch, int t_len, const mvn_fwd_buffers *buf, float *out, int normalize,
                        int remove_last, int save, void *stream_, bool f16) {
  Geometry g;
  int rc = make_geometry(dims, batch, t_len, g);
  if (rc) return rc;
  if (!p || !buf || !buf->acts || !buf->z || !buf->skip || !buf->a1 ||
      (save && (!buf->th || !buf->sg)) ||
      (buf->dense_audio ? buf->dense_ld < t_len : (!index || index_stride < t_len))) {
    set_error("mvn_forward: NULL buffer or bad index / dense-audio stride");
    return MVN_ERR_BAD_ARG;
  }
  const int S_out = g.S - (remove_last ? 1 : 0);
  if (S_out > 0 && !out) {
    set_error("mvn_forward: out is NULL");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = g.C, Kc = g.Kc, Q = g.Q, T = g.T;
  const int t_skip0 = g.rf - 1;

  Act x0 = act_view(buf->acts, batch, C, g.Tp);
  if (buf->dense_audio) {
    CausalOp c0;
    c0.K = 2 * Q; c0.t_begin = 0; c0.t_end = T; c0.C = C; c0.Q = Q; c0.cw = p->causal_w;
    c0.audio = act_view(const_cast<float *>(buf->dense_audio), batch, Q, buf->dense_ld);
    c0.x0 = x0;
    launch_gemm_staged(c0, C, batch, s);
  } else {
    if (Q <= 1024)
      hipLaunchKernelGGL(embed4_kernel, dim3((T + 1023) / 1024, C, batch), dim3(256), 0, s, p->causal_w, index, index_stride,
                         x0, Q, T);
    else
      hipLaunchKernelGGL(embed_kernel, dim3((T + 255) / 256, C, batch), dim3(256), 0, s, p->causal_w,
                       index, index_stride, x0, C, Q, T);
  }
  Act zv = act_view(buf->z, batch, C, g.Tp);
  Act skipv = act_view(buf->skip, batch, Kc, g.Sp);
  const bool has_ctx = buf->ctx != nullptr;
  if (has_ctx && (!p->ctx_filter_w || !p->ctx_filter_b || !p->ctx_gate_w || !p->ctx_gate_b ||
                  buf->ctx_ld < T)) {
    set_error("mvn_forward: context given without context-conv parameters / ctx_ld < t_len");
    return MVN_ERR_BAD_ARG;
  }
  Act ctxv = act_view(const_cast<float *>(buf->ctx), batch, C, has_ctx ? buf->ctx_ld : 0);
  int A = 0;  // first valid time of the current layer's input
  for (int l = 0; l < g.L; ++l) {
    const int d = dilation_of(dims, l);
    const int src = save ? l : (l & 1), dst = save ? l + 1 : ((l + 1) & 1);
    Act xin = act_view(buf->acts + (size_t)src * g.act, batch, C, g.Tp);
    Act xout = act_view(buf->acts + (size_t)dst * g.act, batch, C, g.Tp);
    auto run_fg = [&](auto f) {
      f.K = has_ctx ? 3 * C : 2 * C; f.t_begin = A + d; f.t_end = T; f.C = C; f.d = d;
      f.wf = p->filter_w[l]; f.wg = p->gate_w[l];
      f.wcf = has_ctx ? p->ctx_filter_w[l] : nullptr; f.wcg = has_ctx ? p->ctx_gate_w[l] : nullptr;
      f.bcf = has_ctx ? p->ctx_filter_b[l] : nullptr; f.bcg = has_ctx ? p->ctx_gate_b[l] : nullptr;
      f.ctx = ctxv;
      f.xin = xin; f.z = zv;
      f.th = act_view(save ? buf->th + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      f.sg = act_view(save ? buf->sg + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      launch_gemm_staged(f, 2 * ((C + 31) / 32 * 32), batch, s, f16);
    };
    // C = K = 64, audio only, fp32: the layer as ONE kernel (fused_layer.h); same bits as the
    // two-kernel form below (MOVENET_HIP_NO_FUSED_FORWARD=1 keeps the latter: A/B and tests)
    const bool no_fused = switches().no_fused_forward;
    // (the layer's weights are re-packed k-major into the z scratch, which this path never
    // touches otherwise: z stays in LDS)
    // ... and its persistent successor (fused_fwd.h: weights in registers, 64-column tiles);
    // MOVENET_HIP_NO_PERSISTENT_FORWARD=1 keeps the per-tile kernel (common.h: Switches)
    const bool no_persistent = switches().no_persistent_forward;
    if (C == FL_C && Kc == FL_C && !f16 && !no_fused && !no_persistent &&
        (!has_ctx || (g.Tp <= (1 << 22) && buf->ctx_ld <= (1 << 22)))) {
      FusedFwdPArgs fp;
      if (has_ctx) {  // conditioned layer: the strip kernel with the context as a third K block
        fp.wcf = p->ctx_filter_w[l]; fp.wcg = p->ctx_gate_w[l]; fp.bcf = p->ctx_filter_b[l]; fp.bcg = p->ctx_gate_b[l];
        fp.ctx = ctxv;
      }
      fp.t_begin = A + d; fp.t_end = T; fp.d = d; fp.t_skip0 = t_skip0; fp.t_base = g.t_base;
      fp.first_layer = (l == 0);
      fp.wf = p->filter_w[l]; fp.wg = p->gate_w[l]; fp.wr = p->residual_w[l]; fp.ws = p->skip_w[l];
      fp.br = p->residual_b[l]; fp.bs = p->skip_b[l];
      fp.xin = xin; fp.xout = xout; fp.skip = skipv;
      if (l == g.L - 1) fp.xout.p = nullptr;  // the last residual output is never used
      fp.th = act_view(save ? buf->th + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      fp.sg = act_view(save ? buf->sg + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      // audio-only layers: the strip kernel on the bf16 matrix cores (fp32 = 3 bf16 planes, fused_fwd_bf3.h)
      // unless a tile form / the fp32-MFMA strip was asked for, or the rows are longer than a buffer resource spans
      const bool bf3 = !fp.ctx.p && forward_bf3_enabled() && !switches().forward_tile &&
                       fp.xin.ld <= (1 << 22) && fp.skip.ld <= (1 << 22);
      if (bf3 && (size_t)g.act >= (size_t)g.L * FS3_PACK_F) {
        // the layers' weights as LDS images (three bf16 planes, the kernel's layout), written once per call into
        // the z scratch, which this path never touches otherwise: z stays in registers
        if (l == 0) {
          const int rc3 = launch_fs3_pack(p, g.L, buf->z, s);
          if (rc3) return rc3;
        }
        fp.wpack = buf->z + (size_t)l * FS3_PACK_F;
      }
      // conditioned layers: the strip kernel with its residual | skip product on the bf16 matrix cores and a packed
      // weight image (same switches; the image needs the z scratch)
      const bool ctxw2 = fp.ctx.p && forward_bf3_enabled() && !switches().forward_tile &&
                         (size_t)g.act >= (size_t)g.L * FSC_PACK_F;
      if (ctxw2) {
        if (l == 0) {
          const int rc3 = launch_fsc_pack(p, g.L, buf->z, s);
          if (rc3) return rc3;
        }
        fp.wpack = buf->z + (size_t)l * FSC_PACK_F;
        const int rc3 = launch_fused_layer64s_ctxw2(fp, batch, s);
        if (rc3) return rc3;
        A += d;
        continue;
      }
      const int rc2 = bf3 ? launch_fused_layer64s_bf3(fp, batch, s) : launch_fused_layer64p(fp, batch, s);
      if (rc2) return rc2;
      A += d;
      continue;
    }
    if (C == FL_C && Kc == FL_C && !has_ctx && !f16 && !no_fused && (size_t)g.act >= (size_t)g.L * FL_PACK_F) {
      if (l == 0)
        for (int ll = 0; ll < g.L; ++ll)
          hipLaunchKernelGGL(fused_pack_kernel, dim3((FL_PACK_F + 255) / 256), dim3(256), 0, s, p->filter_w[ll],
                             p->gate_w[ll], p->residual_w[ll], p->skip_w[ll], buf->z + (size_t)ll * FL_PACK_F);
      FusedLayerArgs fa;
      fa.t_begin = A + d; fa.t_end = T; fa.d = d; fa.t_skip0 = t_skip0; fa.t_base = g.t_base;
      fa.first_layer = (l == 0);
      fa.wpack = buf->z + (size_t)l * FL_PACK_F;
      fa.br = p->residual_b[l]; fa.bs = p->skip_b[l];
      fa.xin = xin; fa.xout = xout; fa.skip = skipv;
      if (l == g.L - 1) fa.xout.p = nullptr;  // the last residual output is never used
      fa.th = act_view(save ? buf->th + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      fa.sg = act_view(save ? buf->sg + (size_t)l * g.act : nullptr, batch, C, g.Tp);
      launch_fused_layer64(fa, batch, s);
      A += d;
      continue;
    }
    if (has_ctx) run_fg(FgOpT<true>()); else run_fg(FgOpT<false>());
    RsOp r;
    r.K = C; r.t_begin = A + d; r.t_end = T; r.C = C; r.Kc = Kc; r.t_skip0 = t_skip0;
    r.t_base = g.t_base;
    r.wr = p->residual_w[l]; r.br = p->residual_b[l]; r.ws = p->skip_w[l]; r.bs = p->skip_b[l];
    r.z = zv; r.xin = xin; r.xout = xout; r.skip = skipv; r.first_layer = (l == 0);
    if (l == g.L - 1) r.xout.p = nullptr;  // the last residual output is never used
    launch_gemm_staged(r, C + Kc, batch, s, f16);
    A += d;
  }
  // head: a1 = lrelu(W1 lrelu(skip) + b1); logits = W2 a1 + b2   (columns s = 0..S-1)
  Act a1v = act_view(buf->a1, batch, Q, g.Sp);
  float *head_img = nullptr;  // LDS images of the two head convolutions (bf16 planes), when the bf16 x 3 strips run
  {
    DenseOp<IN_LRELU, OUT_BIAS_LRELU, false> h1;
    h1.K = Kc; h1.t_begin = g.pad; h1.t_end = g.pad + g.S; h1.M = Q; h1.wmat = p->head1_w;
    h1.ldw = Kc; h1.bias = p->head1_b; h1.xin = skipv; h1.yout = a1v; h1.ref = a1v;
    h1.t_out_end = g.pad + g.S; h1.aligned_out = 1;
    // Q = 256, K = 64: the strip form reads the skip sum once (fused_fwd.h); MOVENET_HIP_NO_DENSE_STRIP=1: A/B
    const bool strip_ok = Kc == 64 && !f16 && !switches().no_dense_strip;
    const bool strip = strip_ok && Q == 256;  // (the fp32 strips: Q = 256 only)
    // r3: both head convolutions as strip kernels on the bf16 matrix cores (fused_fwd_bf3.h), their LDS images
    // packed once per call behind the layers' in the z scratch
    const size_t img_off = (size_t)g.L * std::max(FS3_PACK_F, FSC_PACK_F);
    if (strip_ok && head_q_strip(Q) && forward_bf3_enabled() && head_bf3_enabled() && (size_t)g.act >= img_off + DS3_IMG_F)
      head_img = buf->z + img_off;
    if (head_img) {
      DenseStripArgs da;
      da.t_begin = h1.t_begin; da.t_end = h1.t_end; da.t_out_end = h1.t_out_end;
      da.wmat = p->head1_w; da.ldw = Kc; da.bias = p->head1_b; da.xin = skipv; da.yout = a1v; da.ref = a1v;
      const int rc2 = MVN_HEAD_Q(head_fwd1_bf3)(da, p, Kc, head_img, batch, s);
      if (rc2) return rc2;
    } else if (strip) {
      DenseStripArgs da;
      da.t_begin = g.pad; da.t_end = g.pad + g.S; da.t_out_end = g.pad + g.S;
      da.wmat = p->head1_w; da.ldw = Kc; da.bias = p->head1_b; da.xin = skipv; da.yout = a1v; da.ref = a1v;
      const int rc2 = launch_dense_strip<64, 256, IN_LRELU, OUT_BIAS_LRELU, false>(da, Q, batch, s);
      if (rc2) return rc2;
    } else {
      launch_gemm_staged(h1, Q, batch, s, f16);
    }
  }
  if (S_out > 0) {
    DenseOp<IN_ID, OUT_BIAS, false> h2;
    h2.K = Q; h2.t_begin = g.pad; h2.t_end = g.pad + g.S; h2.M = Q; h2.wmat = p->head2_w; h2.ldw = Q;
    h2.bias = p->head2_b; h2.xin = a1v; h2.ref = a1v;
    // `out` is the caller's contiguous (B, Q, S_out): column s of the head = out column s - pad
    h2.yout = act_view(out - g.pad, batch, Q, S_out);
    h2.t_out_end = g.pad + S_out; h2.aligned_out = 0;
    const bool strip2 = Q == 256 && !f16 && !switches().no_dense_strip;
    if (head_img && S_out > 0) {
      // four row blocks of 64 on the bf16 matrix cores (fused_fwd_bf3.h)
      DenseStripArgs da;
      da.t_begin = h2.t_begin; da.t_end = h2.t_end; da.t_out_end = h2.t_out_end;
      da.wmat = p->head2_w; da.ldw = Q; da.bias = p->head2_b; da.xin = a1v; da.yout = h2.yout; da.ref = a1v;
      const int rc2 = MVN_HEAD_Q(head_fwd2_bf3)(da, head_img, batch, s);
      if (rc2) return rc2;
    } else if (strip2) {
      // two row blocks of 128: a1 is read twice instead of once per 64-row block (four times)
      DenseStripArgs da;
      da.t_begin = h2.t_begin; da.t_end = h2.t_end; da.t_out_end = h2.t_out_end;
      da.wmat = p->head2_w; da.ldw = Q; da.bias = p->head2_b; da.xin = a1v; da.yout = h2.yout; da.ref = a1v;
      const int rc2 = launch_dense_strip<256, 128, IN_ID, OUT_BIAS, false>(da, Q, batch, s);
      if (rc2) return rc2;
    } else {
      launch_gemm_staged(h2, Q, batch, s, f16);
    }
    if (normalize) {
      if (Q <= 4 * CQ)
        hipLaunchKernelGGL(softmax_cols_kernel, dim3((S_out + 63) / 64, batch), dim3(256), 0, s, out, Q,
                           S_out);
      else
        hipLaunchKernelGGL(softmax_kernel, dim3((S_out + 255) / 256, batch), dim3(256), 0, s, out, Q,
                           S_out);
    }
  }
  return check_hip(hipGetLastError(), "mvn_forward");
}

int mvn_forward(const mvn_dims *dims, const mvn_params *p, const int32_t *index, int index_stride,
                int batch, int t_len, const mvn_fwd_buffers *buf, float *out, int normalize,
                int remove_last, int save, void *stream_) {
  return forward_impl(dims, p, index, index_stride, batch, t_len, buf, out, normalize, remove_last, save, stream_,
                      false);
}

int mvn_forward_f16(const mvn_dims *dims, const mvn_params *p, const int32_t *index, int index_stride,
                    int batch, int t_len, const mvn_fwd_buffers *buf, float *out, int normalize,
                    int remove_last, int save, void *stream_) {
  return forward_impl(dims, p, index, index_stride, batch, t_len, buf, out, normalize, remove_last, save, stream_,
                      true);
}

static int g_last_bwd_form = 0;  // (process-wide: autograd runs the backward on a thread of its own)
int mvn_last_backward_form(void) { return g_last_bwd_form; }

int mvn_backward(const mvn_dims *dims, const mvn_params *p, const mvn_param_grads *gr,
                 const int32_t *index, int index_stride, int batch, int t_len,
                 const mvn_fwd_buffers *fwd, const mvn_bwd_buffers *bwd, const float *out,
                 const float *dout, int normalize, int remove_last, void *stream_) {
  Geometry g;
  int rc = make_geometry(dims, batch, t_len, g);
  if (rc) return rc;
  if (fwd && !fwd->dense_audio && !index) {
    set_error("mvn_backward: index is NULL");
    return MVN_ERR_BAD_ARG;
  }
  if (!p || !gr || !fwd || !bwd || !fwd->acts || !fwd->th || !fwd->sg || !fwd->skip ||
      !fwd->a1 || !bwd->dx_a || !bwd->dx_b || !bwd->dfg || !bwd->dskip || !bwd->da1 ||
      !bwd->dlogit || (dout && normalize && !out)) {
    set_error("mvn_backward: NULL buffer");
    return MVN_ERR_BAD_ARG;
  }
  const int S_out = g.S - (remove_last ? 1 : 0);
  if (batch == 0 || S_out <= 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = g.C, Kc = g.Kc, Q = g.Q, T = g.T;
  const int t_skip0 = g.rf - 1;

  Act dlog = act_view(bwd->dlogit, batch, Q, g.Sp);
  Act da1 = act_view(bwd->da1, batch, Q, g.Sp);
  // per-workgroup bias partial sums live in the dfg scratch (free until the layer loop
  // needs it: each use below is followed by its reduce before dfg is written)
  float *bias_scratch = bwd->dfg;
  {
    const size_t need_head = (size_t)((g.S + WG_CHUNK - 1) / WG_CHUNK) * batch * ((Q + 63) / 64 * 64);
    const size_t need_layer = (size_t)((T + WG_CHUNK - 1) / WG_CHUNK) * batch * ((C + Kc + 63) / 64 * 64);
    const size_t have = (size_t)batch * 2 * C * g.Tp;
    if (need_head > have || need_layer > have) bias_scratch = nullptr;
  }
  // scratch of the layer weight gradients (wgrad2): per-workgroup partial tiles and bias
  // partial sums, both in da1, which is dead once the head's backward below has run
  float *bias_scratch2 = nullptr, *slab = bwd->da1;
  size_t slab_floats = (size_t)batch * Q * g.Sp, bias2_floats = 0;
  {
    // room for wgrad2's workgroups (512-column chunks) AND for the fused halves' (one round of two workgroups per
    // CU whatever the length: up to 2 CUs + batch of them, 128 partial sums each).  r3: the second count was
    // missing -- at T x batch < 2^18 (the parity tests' sizes, not the configs') the fused first half and the
    // conditioned pass wrote their bias partials past the end of da1, into whatever tensor came next.
    const size_t wg2 = (size_t)((T + TILE_ALIGN + W2_CHUNK - 1) / W2_CHUNK) * batch;
    const size_t wgf = (size_t)2 * fb_device_cus() + batch;
    const size_t need = std::max(wg2, wgf) * ((C + Kc + 127) / 128 * 128);
    if (need <= slab_floats / 2) {
      slab_floats -= need;
      bias_scratch2 = bwd->da1 + slab_floats;
      bias2_floats = need;
    }
  }
  // head weight gradients (wgrad2): slabs + bias partials in dfg
  float *head_slab = bwd->dfg, *head_bias = nullptr;
  size_t head_slab_floats = (size_t)batch * 2 * C * g.Tp;
  bool head_scratch_ok = false;
  {
    const size_t chunks = (size_t)(g.S + TILE_ALIGN + W2_CHUNK - 1) / W2_CHUNK;
    const size_t mpad = (size_t)(Q + 127) / 128 * 128;
    const size_t need_bias = chunks * batch * mpad, need_slab = chunks * batch * mpad * mpad;
    if (need_bias + need_slab <= head_slab_floats) {
      head_slab_floats -= need_bias;
      head_bias = bwd->dfg + head_slab_floats;
      head_scratch_ok = true;
    }
  }
  Act dskip = act_view(bwd->dskip, batch, Kc, g.Sp);
  Act a1v = act_view(fwd->a1, batch, Q, g.Sp);
  Act skipv = act_view(fwd->skip, batch, Kc, g.Sp);
  // the head's data gradients as bf16 x 3 strip kernels (fused_fwd_bf3.h): their LDS images go behind the forward's
  // in the forward's z scratch, which nothing else touches between the two passes
  float *bwd_head_img = nullptr;
  {
    const size_t off = (size_t)g.L * std::max(FS3_PACK_F, FSC_PACK_F) + DS3_IMG_F;
    if (head_q_strip(Q) && Kc == 64 && fwd->z && forward_bf3_enabled() && head_bf3_enabled() && (size_t)g.act >= off + DS3_BWD_IMG_F)
      bwd_head_img = fwd->z + off;
  }
  if (!dout) {
    // the caller has filled bwd->dlogit itself (mvn_softmax_ce_backward: the trainer's loss and
    // the model's softmax differentiated in one pass)
  } else if (normalize && Q <= 4 * CQ)
    hipLaunchKernelGGL(softmax_bwd_cols_kernel, dim3((g.S + 63) / 64, batch), dim3(256), 0, s, out, dout,
                       dlog, Q, S_out, g.S, g.pad);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((g.S + 255) / 256, batch), dim3(256), 0, s, out, dout,
                     dlog, Q, S_out, g.S, normalize, g.pad);
  {  // head conv2: weight grad, then data grad (x lrelu'(a1))
    WgDenseOp<IN_ID> w2;
    w2.t_begin = g.pad; w2.t_end = g.pad + g.S; w2.M = Q; w2.N = Q; w2.aact = dlog; w2.xact = a1v;
    w2.dwm = gr->head2_w; w2.dbv = gr->head2_b;
    // (slab + bias scratch of wgrad2 live in da1, which conv2's data gradient below writes:
    // the head uses the dfg tensor instead, free until the layer loop)
    if (head_scratch_ok)
      launch_wgrad2<2>(w2, Q, Q, batch, head_bias, head_slab, head_slab_floats, s);
    else
      launch_wgrad(w2, Q, Q, batch, bias_scratch, s);
    DenseOp<IN_ID, OUT_MUL_DLRELU, true> d2;
    d2.K = Q; d2.t_begin = g.pad; d2.t_end = g.pad + g.S; d2.M = Q; d2.wmat = p->head2_w; d2.ldw = Q;
    d2.bias = nullptr; d2.xin = dlog; d2.yout = da1; d2.ref = a1v; d2.t_out_end = g.pad + g.S;
    d2.aligned_out = 1;
    // (the fp32 strip form measured 454 us here against 342: its 64 leaky-ReLU reference loads per strip
    // come after the MFMAs, and there is no register left to fetch them ahead.  r3: the bf16 x 3 strip -- four row
    // blocks of 64, reference values requested before the products, images packed behind the forward's in the z scratch)
    if (bwd_head_img) {
      DenseStripArgs da;
      da.t_begin = d2.t_begin; da.t_end = d2.t_end; da.t_out_end = d2.t_out_end;
      da.wmat = p->head2_w; da.ldw = Q; da.bias = nullptr; da.xin = dlog; da.yout = da1; da.ref = a1v;
      rc = MVN_HEAD_Q(head_bwd2_bf3)(da, p, bwd_head_img, batch, s);
      if (rc) return rc;
    } else {
      launch_gemm_staged(d2, Q, batch, s);
    }
  }
  {  // head conv1
    WgDenseOp<IN_LRELU> w1;
    w1.t_begin = g.pad; w1.t_end = g.pad + g.S; w1.M = Q; w1.N = Kc; w1.aact = da1; w1.xact = skipv;
    w1.dwm = gr->head1_w; w1.dbv = gr->head1_b;
    if (head_scratch_ok)
      launch_wgrad2<1>(w1, Q, Kc, batch, head_bias, head_slab, head_slab_floats, s);
    else
      launch_wgrad(w1, Q, Kc, batch, bias_scratch, s);
    DenseOp<IN_ID, OUT_MUL_DLRELU, true> d1;
    d1.K = Q; d1.t_begin = g.pad; d1.t_end = g.pad + g.S; d1.M = Kc; d1.wmat = p->head1_w; d1.ldw = Kc;
    d1.bias = nullptr; d1.xin = da1; d1.yout = dskip; d1.ref = skipv; d1.t_out_end = g.pad + g.S;
    d1.aligned_out = 1;
    // (conv1's data gradient has ONE 64-row output block: the generic kernel already reads da1 once,
    // and the fp32 strip form measured 143 us against its 113; r3: the bf16 x 3 strip)
    if (bwd_head_img && Kc == 64) {
      DenseStripArgs da;
      da.t_begin = d1.t_begin; da.t_end = d1.t_end; da.t_out_end = d1.t_out_end;
      da.wmat = p->head1_w; da.ldw = Kc; da.bias = nullptr; da.xin = da1; da.yout = dskip; da.ref = skipv;
      rc = MVN_HEAD_Q(head_bwd1_bf3)(da, p, Kc, bwd_head_img, batch, s);
      if (rc) return rc;
    } else {
      launch_gemm_staged(d1, Kc, batch, s);
    }
  }
  // layers, last to first
  int A_lo[4096];
  {
    int A = 0;
    for (int l = 0; l <= g.L && l < 4096; ++l) {
      A_lo[l] = A;
      if (l < g.L) A += dilation_of(dims, l);
    }
  }
  const bool has_ctx = fwd->ctx != nullptr;
  if (has_ctx && (!bwd->dctx || !gr->ctx_filter_w || !gr->ctx_filter_b || !gr->ctx_gate_w ||
                  !gr->ctx_gate_b || !p->ctx_filter_w || !p->ctx_gate_w)) {
    set_error("mvn_backward: context given without dctx buffer / context-conv gradients");
    return MVN_ERR_BAD_ARG;
  }
  Act ctxv = act_view(const_cast<float *>(fwd->ctx), batch, C, has_ctx ? fwd->ctx_ld : 0);
  Act dctxv = act_view(bwd->dctx, batch, C, g.Tp);
  if (has_ctx) {
    rc = check_hip(hipMemsetAsync(bwd->dctx, 0, sizeof(float) * (size_t)g.act, s), "memset dctx");
    if (rc) return rc;
  }
  // bias partials of the context convs: dfg is busy while WgFgOp runs, use dlogit
  float *ctx_bias_scratch = bwd->dlogit;
  {
    const size_t need = (size_t)((T + WG_CHUNK - 1) / WG_CHUNK) * batch * ((2 * C + 63) / 64 * 64);
    if (need > (size_t)g.hid) ctx_bias_scratch = nullptr;
  }
  float *dxo_p = nullptr;  // gradient w.r.t. the layer's residual output
  float *cur = bwd->dx_a, *nxt = bwd->dx_b;
  Act dfg = act_view(bwd->dfg, batch, 2 * C, g.Tp);
  // Two streams: per layer  WgRs || Dz  then  WgFg || (Dctx,) Dx.
  //   ev[0]: dx of the previous layer / the head's dskip ready (s -> s2)
  //   ev[1]: dfg of this layer ready (s -> s2)
  //   ev[2], ev[3]: WgRs / WgFg done (s2 -> s: the buffers they read may be overwritten)
  SideStream *side = nullptr;
  // MOVENET_HIP_NO_SIDE_STREAM=1 keeps everything on the caller's stream (profiling: kernel
  // durations are only comparable when the kernels do not share the machine)
  const bool no_side = switches().no_side_stream;
  // MOVENET_HIP_NO_FUSED_BACKWARD=1: the two-kernel forms (cross-checks, profiling)
    const bool fused_bwd = !switches().no_fused_backward;
  // (with both fused halves every kernel of the layer loop runs on the caller's stream: no fork,
  // and none of the two event records + waits per layer that go with it -- ~60 gaps of ~8 us per step)
  const bool all_fused = fused_bwd && C == 64 && Kc == 64;  // (conditioned layers too: bwd_dctx_wgctx64_kernel)
  const bool fork = bias_scratch2 && !no_side && !all_fused;
  hipStream_t s2 = s;
  if (fork) {
    rc = side_stream(&side);
    if (rc) return rc;
    s2 = side->s;
    if (check_hip(hipEventRecord(side->ev[0], s), "hipEventRecord")) return MVN_ERR_LAUNCH;
  }
  auto signal = [&](int e, hipStream_t from) {
    if (fork) (void)hipEventRecord(side->ev[e], from);
  };
  auto await = [&](int e, hipStream_t on) {
    if (fork) (void)hipStreamWaitEvent(on, side->ev[e], 0);
  };
  // r4: the layer's backward as ONE kernel, input gradients in scatter form (fused_bwd_l.h): df | dg stays on chip.
  // Two pairs (A', P0) of (B, C, Tp) tensors alternate between the layers: dx_a | dx_b and -- audio only -- the two
  // row halves of the dfg tensor, which nothing else uses then; conditioned layers still write dfg for the context
  // pass, their second pair lives in dlogit (dead behind the head's backward) when it is large enough.
  // MOVENET_HIP_BWD_FORM=split keeps the two-half form of r2 / r3 (cross-checks, A/B; common.h: Switches).
  Act spair[2][2];
  bool scatter = false;
  // (tensors too small for the reservation above -- the parity tests' smallest -- get a bias region sized for
  // the workgroups that fit: the plan takes fewer, longer chunks then)
  float *sc_bias = bias_scratch2, *sc_slab = slab;
  size_t sc_bias_floats = bias2_floats, sc_slab_floats = slab_floats;
  if (!sc_bias && !has_ctx && slab) {
    const size_t total = (size_t)batch * Q * g.Sp, n_fit = total / (128 * 64 + 128 * 128 + 128);
    if (n_fit >= (size_t)batch) {
      sc_bias_floats = n_fit * 128;
      sc_slab_floats = total - sc_bias_floats;
      sc_bias = bwd->da1 + sc_slab_floats;
    }
  }
  if (all_fused && sc_bias && g.L < 4095 && g.Tp <= (1 << 21) && !switches().bwd_split) {
    spair[0][0] = act_view(bwd->dx_a, batch, C, g.Tp);
    spair[0][1] = act_view(bwd->dx_b, batch, C, g.Tp);
    bool have = true;
    if (!has_ctx) {
      spair[1][0] = Act{bwd->dfg, (long long)2 * C * g.Tp, g.Tp};
      spair[1][1] = Act{bwd->dfg + (size_t)C * g.Tp, (long long)2 * C * g.Tp, g.Tp};
    } else if (2 * g.act <= g.hid) {
      spair[1][0] = act_view(bwd->dlogit, batch, C, g.Tp);
      spair[1][1] = act_view(bwd->dlogit + (size_t)g.act, batch, C, g.Tp);
    } else {
      have = false;
    }
    FusedBwdLPlan pl0;
    int cc = 0, cct = 0;
    scatter = have && bwd_layer64_plan(A_lo[1], T, batch, sc_bias, sc_bias_floats, sc_slab, sc_slab_floats, &pl0) &&
              (!has_ctx || bwd_dctx_wgctx64_fits(A_lo[1], T, batch, bias_scratch2, bias2_floats, slab, slab_floats, &cc, &cct));
  }
  g_last_bwd_form = scatter ? MVN_BWD_FORM_ONE : MVN_BWD_FORM_GENERIC;
  if (scatter) {
    int po = 1 ^ ((g.L - 1) & 1);  // (so that layer 0 writes pair 1 and the dense dx0 can go to dx_a)
    for (int l = g.L - 1; l >= 0; --l) {
      const int t_lo = A_lo[l + 1];
      FusedBwdLArgs fa;
      fa.t_lo = t_lo; fa.t_end = T; fa.d = dilation_of(dims, l); fa.t_skip0 = t_skip0; fa.t_base = g.t_base;
      fa.up_lo = l + 1 < g.L ? A_lo[l + 2] : 0;
      fa.up_d = l + 1 < g.L ? dilation_of(dims, l + 1) : 0;
      fa.wr = p->residual_w[l]; fa.ws = p->skip_w[l]; fa.wf = p->filter_w[l]; fa.wg = p->gate_w[l];
      fa.ga = spair[po ^ 1][0]; fa.gp = spair[po ^ 1][1];
      if (l == g.L - 1) fa.ga.p = nullptr;  // the last layer's residual output is unused: no dxo
      fa.dskip = dskip;
      fa.th = act_view(fwd->th + (size_t)l * g.act, batch, C, g.Tp);
      fa.sg = act_view(fwd->sg + (size_t)l * g.act, batch, C, g.Tp);
      fa.xin = act_view(fwd->acts + (size_t)l * g.act, batch, C, g.Tp);
      fa.oa = spair[po][0]; fa.op = spair[po][1];
      fa.dfg = has_ctx ? dfg : Act{nullptr, 0, 0};
      WgRsOp wr;
      wr.t_begin = t_lo; wr.t_end = T; wr.C = C; wr.Kc = Kc; wr.t_skip0 = t_skip0; wr.t_base = g.t_base;
      wr.dxo = fa.ga; wr.dskip = dskip; wr.th = fa.th; wr.sg = fa.sg;
      wr.dwr = gr->residual_w[l]; wr.dbr = gr->residual_b[l]; wr.dws = gr->skip_w[l]; wr.dbs = gr->skip_b[l];
      WgFgOpT<false> wf;
      wf.t_begin = t_lo; wf.t_end = T; wf.C = C; wf.d = fa.d; wf.dfg = dfg; wf.xin = fa.xin; wf.ctx = ctxv;
      wf.dwf = gr->filter_w[l]; wf.dwg = gr->gate_w[l];
      wf.dwcf = nullptr; wf.dwcg = nullptr; wf.dbcf = nullptr; wf.dbcg = nullptr;
      FusedBwdLPlan pl;
      if (!bwd_layer64_plan(t_lo, T, batch, sc_bias, sc_bias_floats, sc_slab, sc_slab_floats, &pl)) {
        set_error("mvn_backward: the fused layer backward lost its scratch at layer %d", l);
        return MVN_ERR_BAD_ARG;
      }
      rc = launch_bwd_layer64(fa, wr, wf, batch, pl, s);
      if (rc) return rc;
      if (has_ctx) {  // (behind the layer's reduce: the slab scratch is free again)
        int c_chunks = 0, c_chunk_t = 0;
        if (!bwd_dctx_wgctx64_fits(t_lo, T, batch, bias_scratch2, bias2_floats, slab, slab_floats, &c_chunks, &c_chunk_t)) {
          set_error("mvn_backward: the context pass lost its scratch at layer %d", l);
          return MVN_ERR_BAD_ARG;
        }
        FusedBwdCArgs fc;
        fc.t_begin = t_lo; fc.t_end = T; fc.wcf = p->ctx_filter_w[l]; fc.wcg = p->ctx_gate_w[l];
        fc.dfg = dfg; fc.ctx = ctxv; fc.dctx = dctxv;
        WgCtxOp co;
        co.dwcf = gr->ctx_filter_w[l]; co.dwcg = gr->ctx_gate_w[l]; co.dbcf = gr->ctx_filter_b[l]; co.dbcg = gr->ctx_gate_b[l];
        launch_bwd_dctx_wgctx64(fc, co, batch, bias_scratch2, slab, c_chunks, c_chunk_t, s);
      }
      po ^= 1;
    }
    // the first layer's input gradient in dense form for the embedding / causal-conv gradient below
    hipLaunchKernelGGL(bwd_scatter_combine_kernel, dim3((T + 255) / 256, C, batch), dim3(256), 0, s, spair[po ^ 1][0],
                       spair[po ^ 1][1], spair[0][0], C, A_lo[1], dilation_of(dims, 0), T);
    dxo_p = bwd->dx_a;
  }
  for (int l = scatter ? -1 : g.L - 1; l >= 0; --l) {
    const int d = dilation_of(dims, l);
    const int t_lo = A_lo[l + 1];
    Act th = act_view(fwd->th + (size_t)l * g.act, batch, C, g.Tp);
    Act sg = act_view(fwd->sg + (size_t)l * g.act, batch, C, g.Tp);
    Act xin = act_view(fwd->acts + (size_t)l * g.act, batch, C, g.Tp);
    Act dxo = act_view(dxo_p, batch, C, g.Tp);
    if (l < g.L - 1) {  // the side stream's work on the previous layer read dxo's twin and dfg
      await(2, s);
      await(3, s);
    }
    await(0, s2);
    WgRsOp wr;
    wr.t_begin = t_lo; wr.t_end = T; wr.C = C; wr.Kc = Kc; wr.t_skip0 = t_skip0; wr.t_base = g.t_base;
    wr.dxo = dxo; wr.dskip = dskip; wr.th = th; wr.sg = sg;
    wr.dwr = gr->residual_w[l]; wr.dbr = gr->residual_b[l]; wr.dws = gr->skip_w[l];
    wr.dbs = gr->skip_b[l];
    bool fused_a = false;
    PendingRsReduce<WgRsOp> pend;
    if (fused_bwd && C == 64 && Kc == 64 && bias_scratch2) {
      // dz and the residual/skip weight gradients from ONE pass over dxo, dskip, tanh, sigmoid
      // (fused_bwd.h; conditioned layers too: nothing here touches the context); on the main
      // stream: the side stream only keeps the filter/gate gradient
      FusedBwdAArgs fa;
      fa.t_begin = t_lo; fa.t_end = T; fa.t_skip0 = t_skip0; fa.t_base = g.t_base;
      fa.wr = p->residual_w[l]; fa.ws = p->skip_w[l];
      fa.dxo = dxo; fa.dskip = dskip; fa.th = th; fa.sg = sg; fa.dfg = dfg;
      // (all_fused: its reduction runs in the second half's reduce launch)
      fused_a = launch_bwd_dz_wgrs64(fa, wr, batch, bias_scratch2, bias2_floats, slab, slab_floats, s,
                                     all_fused ? &pend : nullptr);
    }
    if (!fused_a) {
      if (bias_scratch2)
        launch_wgrad2<1>(wr, C + Kc, C, batch, bias_scratch2, slab, slab_floats, s2);
      else
        launch_wgrad(wr, C + Kc, C, batch, bias_scratch, s2);
    }
    signal(2, s2);
    if (!fused_a) {
      DzOp dz;
      dz.K = C + Kc; dz.t_begin = t_lo; dz.t_end = T; dz.C = C; dz.Kc = Kc; dz.t_skip0 = t_skip0;
      dz.t_base = g.t_base;
      dz.wr = p->residual_w[l]; dz.ws = p->skip_w[l];
      dz.dxo = dxo; dz.dskip = dskip; dz.th = th; dz.sg = sg; dz.dfg = dfg;
      launch_gemm_staged(dz, C, batch, s);
    }
    signal(1, s);
    // conditioned layer: dctx and the context-conv gradients as a third fused pass over dfg
    // (fused_bwd.h), which leaves the audio taps to the fused second half; decided up front from
    // the scratch sizes -- the generic conditioned path computes taps and context rows TOGETHER
    int c_chunks = 0, c_chunk_t = 0;
    const bool fused_c = has_ctx && fused_a && all_fused &&
                         bwd_dctx_wgctx64_fits(t_lo, T, batch, bias_scratch2, bias2_floats, slab, slab_floats, &c_chunks,
                                               &c_chunk_t);
    bool fused_b = false;
    if (fused_bwd && C == 64 && (!has_ctx || fused_c)) {
      // dx and the filter/gate weight gradients from ONE pass over dfg (fused_bwd.h)
      FusedBwdBArgs fb;
      fb.t_out0 = A_lo[l]; fb.t_lo = t_lo; fb.t_end = T; fb.d = d;
      fb.wf = p->filter_w[l]; fb.wg = p->gate_w[l];
      fb.dxo = dxo; fb.dfg = dfg; fb.xin = xin; fb.dxi = act_view(cur, batch, C, g.Tp);
      WgFgOpT<false> wf;
      wf.t_begin = t_lo; wf.t_end = T; wf.C = C; wf.d = d; wf.dfg = dfg; wf.xin = xin; wf.ctx = ctxv;
      wf.dwf = gr->filter_w[l]; wf.dwg = gr->gate_w[l];
      wf.dwcf = nullptr; wf.dwcg = nullptr; wf.dbcf = nullptr; wf.dbcg = nullptr;
      rc = launch_bwd_dx_wgfg64(fb, wf, batch, slab, slab_floats, s, &fused_b, &pend);
      if (rc) return rc;
    }
    flush_pending_rs(pend, s);  // (only if the second half did not take it along)
    if (fused_a && fused_b) g_last_bwd_form = MVN_BWD_FORM_HALVES;
    const bool ctx_done = fused_c && fused_b;
    if (ctx_done) {  // (behind the layer's reduce: the slab scratch is free again)
      FusedBwdCArgs fc;
      fc.t_begin = t_lo; fc.t_end = T; fc.wcf = p->ctx_filter_w[l]; fc.wcg = p->ctx_gate_w[l];
      fc.dfg = dfg; fc.ctx = ctxv; fc.dctx = dctxv;
      WgCtxOp co;
      co.dwcf = gr->ctx_filter_w[l]; co.dwcg = gr->ctx_gate_w[l]; co.dbcf = gr->ctx_filter_b[l]; co.dbcg = gr->ctx_gate_b[l];
      launch_bwd_dctx_wgctx64(fc, co, batch, bias_scratch2, slab, c_chunks, c_chunk_t, s);
    }
    await(1, s2);
    auto run_wf = [&](auto wf) {
      wf.t_begin = t_lo; wf.t_end = T; wf.C = C; wf.d = d; wf.dfg = dfg; wf.xin = xin; wf.ctx = ctxv;
      wf.dwf = gr->filter_w[l]; wf.dwg = gr->gate_w[l];
      wf.dwcf = has_ctx ? gr->ctx_filter_w[l] : nullptr; wf.dwcg = has_ctx ? gr->ctx_gate_w[l] : nullptr;
      wf.dbcf = has_ctx ? gr->ctx_filter_b[l] : nullptr; wf.dbcg = has_ctx ? gr->ctx_gate_b[l] : nullptr;
      if (!has_ctx)
        launch_wgrad2<2>(wf, 2 * C, 2 * C, batch, nullptr, slab, slab_floats, s2);
      else if (bias_scratch2)  // (shares the bias scratch with WgRs: same stream, one after the other)
        launch_wgrad2<2>(wf, 2 * C, 3 * C, batch, bias_scratch2, slab, slab_floats, s2);
      else
        launch_wgrad(wf, 2 * C, 3 * C, batch, ctx_bias_scratch, s2);
    };
    if (fused_b) {
    } else if (has_ctx) run_wf(WgFgOpT<true>()); else run_wf(WgFgOpT<false>());
    signal(3, s2);
    if (has_ctx && !ctx_done) {
      DctxOp dc;
      dc.K = 2 * C; dc.t_begin = t_lo; dc.t_end = T; dc.C = C;
      dc.wcf = p->ctx_filter_w[l]; dc.wcg = p->ctx_gate_w[l]; dc.dfg = dfg; dc.dctx = dctxv;
      launch_gemm_staged(dc, C, batch, s);
    }
    if (!fused_b) {
      DxOp dx;
      dx.K = 4 * C; dx.t_begin = A_lo[l]; dx.t_end = T; dx.C = C; dx.d = d; dx.t_lo = t_lo;
      dx.wf = p->filter_w[l]; dx.wg = p->gate_w[l];
      dx.dxo = dxo; dx.dfg = dfg; dx.dxi = act_view(cur, batch, C, g.Tp);
      launch_gemm_staged(dx, C, batch, s);
    }
    signal(0, s);
    dxo_p = cur;
    float *tmp = cur; cur = nxt; nxt = tmp;
  }
  await(2, s);  // join: everything the side stream did is ordered before what follows on s
  await(3, s);
  if (fwd->dense_audio) {
    WgCausalOp wc;
    wc.t_begin = 0; wc.t_end = T; wc.C = C; wc.Q = Q; wc.dx0 = act_view(dxo_p, batch, C, g.Tp);
    wc.audio = act_view(const_cast<float *>(fwd->dense_audio), batch, Q, fwd->dense_ld);
    wc.dcw = gr->causal_w;
    launch_wgrad(wc, C, 2 * Q, batch, nullptr, s);
  } else {
    // channels per workgroup: the table slice [cg][Q][2] fits 64 KB of LDS
    if (Q > 8192) {
      set_error("mvn_backward: input_channels %d > 8192 not supported by the embedding gradient", Q);
      return MVN_ERR_UNSUPPORTED;
    }
    const int chunks64 = (T + EG64_CHUNK - 1) / EG64_CHUNK;
    if (C == 64 && Q % EG64_PARTS == 0 && (size_t)Q / EG64_PARTS * 64 * sizeof(float) <= 64 * 1024 && slab &&
        (size_t)chunks64 * batch * 2 * Q * 64 <= slab_floats) {
      // MOVENET_HIP_EMBED_GRAD=scalar keeps the LDS read-modify-write kernel (A/B, tests); Q = 256: the product form
      const bool mfma_form = Q == 256 && !switches().embed_scalar;
      if (mfma_form) {
        hipLaunchKernelGGL(embed_grad64_mfma_kernel, dim3(chunks64, batch), dim3(256), 0, s, index, index_stride,
                           act_view(dxo_p, batch, C, g.Tp), T, slab);
      } else {
        const size_t lds = (size_t)Q / EG64_PARTS * 64 * sizeof(float) + EG64_CHUNK * sizeof(int);
        rc = ensure_max_dynamic_lds((const void *)embed_grad64_kernel, "hipFuncSetAttribute(embed_grad64)");
        if (rc) return rc;
        hipLaunchKernelGGL(embed_grad64_kernel, dim3(chunks64, batch, 2 * EG64_PARTS), dim3(64), lds, s, index, index_stride,
                           act_view(dxo_p, batch, C, g.Tp), Q, T, slab);
      }
      EmbedSlabOp64 eo;
      eo.dcw = gr->causal_w; eo.Q = Q;
      hipLaunchKernelGGL(slab_reduce_kernel<EmbedSlabOp64>, dim3((unsigned)(2 * Q * 64 / 32)), dim3(32 * RED_SEG), 0, s,
                         eo, slab, chunks64 * batch, 2 * Q, 64);
      return check_hip(hipGetLastError(), "mvn_backward");
    }
    const int cg = std::max(1, std::min(C, 8192 / Q));
    const int chunks = (T + EG_CHUNK - 1) / EG_CHUNK;
    // one table copy per (chunk, sequence) in the layer loop's slab scratch (idle by now), summed by
    // slab_reduce_kernel; without room for them the workgroups add to the table with atomics
    const size_t table = (size_t)C * Q * 2;
    float *part = (table % 64 == 0 && slab && (size_t)chunks * batch * table <= slab_floats) ? slab : nullptr;
    hipLaunchKernelGGL(embed_grad_kernel, dim3(chunks, (C + cg - 1) / cg, batch),
                       dim3(1024), (size_t)cg * Q * 2 * sizeof(float), s, gr->causal_w, index,
                       index_stride, act_view(dxo_p, batch, C, g.Tp), C, Q, T, cg, part);
    if (part) {
      EmbedSlabOp eo;
      eo.dcw = gr->causal_w;
      hipLaunchKernelGGL(slab_reduce_kernel<EmbedSlabOp>, dim3((unsigned)(table / 32)), dim3(32 * RED_SEG), 0, s, eo,
                         part, chunks * batch, (int)(table / 64), 64);
    }
  }
  return check_hip(hipGetLastError(), "mvn_backward");
}

int mvn_gen_prime_from_forward(const mvn_dims *dims, const mvn_fwd_buffers *fwd, int batch,
                               int t_len, float *state, void *stream_) {
  Geometry g;
  int rc = make_geometry(dims, batch, t_len, g);
  if (rc) return rc;
  if (!fwd || !fwd->acts || !state) {
    set_error("mvn_gen_prime_from_forward: NULL buffer");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0) return MVN_OK;
  // steps t = 0 .. t_len-2 have been consumed: queue l holds x_l[t] for the
  // d_l most recent t <= t_len-2, all of them >= A_l because t_len >= RF
  Act view = act_view(fwd->acts, batch, g.C, g.Tp);
  hipLaunchKernelGGL(ring_fill_kernel, dim3(g.L, batch), dim3(256), 0, (hipStream_t)stream_,
                     fwd->acts, g.act, view, state, dilation_sum(dims) * g.C, g.C,
                     dims->layer_size, g.L, t_len - 2);
  return check_hip(hipGetLastError(), "mvn_gen_prime_from_forward");
}

}  // extern "C"
