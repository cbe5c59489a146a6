// The trainer's element-wise tail as single passes (row F3 of SURVEY.md section 8):
//
//   mvn_softmax_ce_forward   head logits -> probabilities (in place) + the loss
//                            cross_entropy(PROBABILITIES, target) + the accuracy,
//                            one read and one write of the (B,Q,S) tensor
//                            (movenet/wavenet.py:189-191 + pytorch_lightning_trainer.py:64-66)
//   mvn_softmax_ce_backward  probabilities + target -> gradient w.r.t. the LOGITS, written
//                            straight into mvn_backward's padded dlogit tensor: the gradient
//                            of the loss through cross_entropy's own log-softmax and through
//                            the model's softmax in one read and one write
//   mvn_adamw_step           torch.optim.AdamW / Adam over ONE flat parameter / gradient /
//                            moment buffer (pytorch_lightning_trainer.py:186-189)
//
// The arithmetic of the first two is that of softmax_cols_kernel (sequence.hip) followed by
// ce_probs_cols_kernel (common.hip), operation for operation: the fused forms return the same
// bits as the two-kernel forms (one exponential for all of them: common.h sm_exp; one division per column).
#include "common.h"

namespace mvn {

constexpr int TQ = 64;  // class rows per wave = registers per thread (Q <= 256)

__device__ __forceinline__ float t_col_reduce(float v, float (*part)[64], int wave, int lane, bool is_max) {
  part[wave][lane] = v;
  __syncthreads();
  const float a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
  __syncthreads();
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

// 64 columns per workgroup, wave w holds class rows [64w, 64w+64) of them, lane = column (r4c: raw-buffer column accesses and
// ONE division per column -- common.h col_ld / sm_exp -- 4100 -> ~1900 vector instructions per wave, four waves per SIMD
// instead of two)
__global__ __launch_bounds__(256) void softmax_ce_fwd_cols_kernel(float *__restrict__ y,
                                                                  const long long *__restrict__ target,
                                                                  int Q, int S, float *__restrict__ loss_part,
                                                                  int32_t *__restrict__ correct_part) {
  __shared__ float part[4][64];
  __shared__ int argp[4][64];
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x * 64 + lane;
  const bool live = s < S;
  const __amdgpu_buffer_rsrc_t yb = col_rsrc(y + (size_t)b * Q * S);
  const int voff = 4 * (live ? s : 0), row = 4 * S;
  float v[TQ];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const int q = TQ * wave + i;
    v[i] = q < Q ? col_ld(yb, voff, q * row) : -INFINITY;
    v[i] = live ? v[i] : -INFINITY;
    m = fmaxf(m, v[i]);
  }
  m = t_col_reduce(m, part, wave, lane, true);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    v[i] = sm_exp(v[i] - m);  // exp(-inf) = 0 for the padding rows
    sum += v[i];
  }
  sum = t_col_reduce(sum, part, wave, lane, false);
  const float inv = 1.0f / sum;
  // probabilities (wavenet.py:189-191), then cross_entropy ON them: a second log-softmax
  const long long tg = live ? target[(size_t)b * S + s] : 0;
  const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
  float m2 = -INFINITY, pt = 0.f;
  int arg = 0;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const int q = TQ * wave + i;
    if (q < Q) {
      v[i] = v[i] * inv;
      if (live) col_st(v[i], yb, voff, q * row);
      if (v[i] > m2) {  // strict: first maximum inside this wave's rows
        m2 = v[i];
        arg = q;
      }
      if (q == tq) pt = v[i];
    } else {
      v[i] = -INFINITY;
    }
  }
  const float wave_m2 = m2;
  m2 = t_col_reduce(m2, part, wave, lane, true);
  float sum2 = 0.f;
#pragma unroll
  for (int i = 0; i < TQ; ++i) sum2 += sm_exp(v[i] - m2);
  sum2 = t_col_reduce(sum2, part, wave, lane, false);
  argp[wave][lane] = wave_m2 == m2 ? arg : 0x7fffffff;
  const float pt_all = t_col_reduce(pt, part, wave, lane, false);  // one wave holds it, the others 0
  float loss = 0.f;
  int ok = 0;
  if (wave == 0 && live) {
    const int a0 = min(min(argp[0][lane], argp[1][lane]), min(argp[2][lane], argp[3][lane]));
    loss = (m2 + logf(sum2)) - pt_all;
    ok = a0 == tq;
  }
  if (wave == 0) {
    loss = wave_sum(loss);
    const float okf = wave_sum((float)ok);
    if (lane == 0) {
      const int wg = blockIdx.y * gridDim.x + blockIdx.x;
      loss_part[wg] = loss;
      correct_part[wg] = (int)okf;
    }
  }
}

// (r4c: the column's probabilities are the only array a thread keeps -- the exponentials are formed twice, the second time
// where the gradient is written, and sum_q dprobs_q p_q comes from two sums of the first pass -- so that four waves fit a
// SIMD: 248 -> ~110 registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void softmax_ce_bwd_cols_kernel(const float *__restrict__ p,
                                                                  const long long *__restrict__ target,
                                                                  int Q, int S, float scale,
                                                                  const float *__restrict__ upstream,
                                                                  float *__restrict__ dlogit, long long d_sb,
                                                                  int d_ld, int col0, int s_cols) {
  __shared__ float part[4][64], part2[4][64], part3[4][64];
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x * 64 + lane;
  const bool live = s < S;
  const __amdgpu_buffer_rsrc_t pb = col_rsrc(p + (size_t)b * Q * S);
  const int voff = 4 * (live ? s : 0), row = 4 * S;
  float pv[TQ];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const int q = TQ * wave + i;
    pv[i] = q < Q ? col_ld(pb, voff, q * row) : -INFINITY;
    pv[i] = live ? pv[i] : -INFINITY;
    m = fmaxf(m, pv[i]);
  }
  m = t_col_reduce(m, part, wave, lane, true);
  const long long tg = live ? target[(size_t)b * S + s] : 0;
  const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
  // one pass: sum_q e_q, sum_q e_q p_q and p_target (e_q = exp(p_q - m); absent rows: e = 0, p taken as 0)
  float sum = 0.f, s2 = 0.f, pt = 0.f;
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const int q = TQ * wave + i;
    const float e = sm_exp(pv[i] - m);
    const float pz = (live && q < Q) ? pv[i] : 0.f;
    sum += e;
    s2 += e * pz;
    pt += q == tq ? pz : 0.f;
  }
  part[wave][lane] = sum;
  part2[wave][lane] = s2;
  part3[wave][lane] = pt;
  __syncthreads();
  sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  s2 = (part2[0][lane] + part2[1][lane]) + (part2[2][lane] + part2[3][lane]);
  pt = (part3[0][lane] + part3[1][lane]) + (part3[2][lane] + part3[3][lane]);
  if (upstream) scale *= *upstream;
  const float inv = 1.0f / sum;
  // dprobs_q = scale (softmax(p)_q - onehot_q); dlogit_q = p_q (dprobs_q - sum_k dprobs_k p_k)
  const float dot = scale * (s2 * inv - pt);
  if (s >= s_cols) return;
  const __amdgpu_buffer_rsrc_t db = col_rsrc(dlogit + (size_t)b * d_sb);
  const int dvoff = 4 * (col0 + s), drow = 4 * d_ld;
  float m_again = m;
  asm volatile("" : "+v"(m_again));  // (or the first pass's 64 exponentials stay in registers for this one)
#pragma unroll
  for (int i = 0; i < TQ; ++i) {
    const int q = TQ * wave + i;
    if (q < Q) {
      const float e = sm_exp(pv[i] - m_again);
      const float g = scale * (e * inv - (q == tq ? 1.0f : 0.0f));
      col_st(live ? pv[i] * (g - dot) : 0.f, db, dvoff, q * drow);
    }
  }
}

// any Q: one thread per column, the column walked several times (slow path)
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(float *__restrict__ y,
                                                             const long long *__restrict__ target, int Q, int S,
                                                             float *__restrict__ loss_part,
                                                             int32_t *__restrict__ correct_part) {
  const int b = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  float loss = 0.f;
  int ok = 0;
  if (s < S) {
    float *col = y + (size_t)b * Q * S + s;
    float m = -INFINITY;
    for (int q = 0; q < Q; ++q) m = fmaxf(m, col[(size_t)q * S]);
    float sum = 0.f;
    for (int q = 0; q < Q; ++q) {
      const float ev = sm_exp(col[(size_t)q * S] - m);
      col[(size_t)q * S] = ev;
      sum += ev;
    }
    float m2 = -INFINITY;
    int arg = 0;
    const float inv = 1.0f / sum;
    for (int q = 0; q < Q; ++q) {
      const float pq = col[(size_t)q * S] * inv;
      col[(size_t)q * S] = pq;
      if (pq > m2) {
        m2 = pq;
        arg = q;
      }
    }
    float sum2 = 0.f;
    for (int q = 0; q < Q; ++q) sum2 += sm_exp(col[(size_t)q * S] - m2);
    const long long tg = target[(size_t)b * S + s];
    const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
    loss = (m2 + logf(sum2)) - col[(size_t)tq * S];
    ok = arg == tq;
  }
  __shared__ float ls[4];
  __shared__ int cs[4];
  loss = wave_sum(loss);
  const float okf = wave_sum((float)ok);
  if ((threadIdx.x & 63) == 0) {
    ls[threadIdx.x >> 6] = loss;
    cs[threadIdx.x >> 6] = (int)okf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    loss_part[wg] = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    correct_part[wg] = (cs[0] + cs[1]) + (cs[2] + cs[3]);
  }
}

__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float *__restrict__ p,
                                                             const long long *__restrict__ target, int Q, int S,
                                                             float scale, const float *__restrict__ upstream,
                                                             float *__restrict__ dlogit, long long d_sb, int d_ld,
                                                             int col0, int s_cols) {
  const int b = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s_cols) return;
  float *dcol = dlogit + (size_t)b * d_sb + col0 + s;
  if (s >= S) {
    for (int q = 0; q < Q; ++q) dcol[(size_t)q * d_ld] = 0.f;
    return;
  }
  if (upstream) scale *= *upstream;
  const float *col = p + (size_t)b * Q * S + s;
  float m = -INFINITY;
  for (int q = 0; q < Q; ++q) m = fmaxf(m, col[(size_t)q * S]);
  float sum = 0.f;
  for (int q = 0; q < Q; ++q) sum += sm_exp(col[(size_t)q * S] - m);
  const float inv = 1.0f / sum;
  const long long tg = target[(size_t)b * S + s];
  const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
  float dot = 0.f;
  for (int q = 0; q < Q; ++q) {
    const float pq = col[(size_t)q * S];
    dot += scale * (sm_exp(pq - m) * inv - (q == tq ? 1.0f : 0.0f)) * pq;
  }
  for (int q = 0; q < Q; ++q) {
    const float pq = col[(size_t)q * S];
    dcol[(size_t)q * d_ld] = pq * (scale * (sm_exp(pq - m) * inv - (q == tq ? 1.0f : 0.0f)) - dot);
  }
}

// ---- AdamW / Adam over a flat buffer ------------------------------------------------------
// torch.optim.AdamW's arithmetic (its _single_tensor_adam, no amsgrad, no maximize):
//   p *= 1 - lr wd  (decoupled)  |  g += wd p  (Adam's L2 form)
//   m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g g
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Elements inside one of the (up to 4) skip ranges are left alone: parameters that received
// no gradient this step (torch skips them too: no decay, no moment update).
struct AdamSkip {
  unsigned long long lo[4], hi[4];
  int n;
};
__global__ __launch_bounds__(256) void adamw_flat_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v,
                                                         unsigned long long n, float lr, float beta1,
                                                         float beta2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, int decoupled, AdamSkip skip) {
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  const float step_size = lr / bc1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned long long i = i0 + e;
    if (i >= n) return;
    bool skipped = false;
    for (int k = 0; k < skip.n; ++k) skipped = skipped || (i >= skip.lo[k] && i < skip.hi[k]);
    if (skipped) continue;
    float pv = p[i], gv = g[i];
    if (decoupled)
      pv = pv * (1.0f - lr * wd);
    else
      gv = gv + wd * pv;
    const float mv = m[i] + (gv - m[i]) * (1.0f - beta1);
    const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - step_size * (mv / denom);
  }
}

}  // namespace mvn

extern "C" {

int mvn_softmax_ce_forward(float *logits_probs, const long long *target, int batch, int classes, int s_len,
                           float *loss_part, int32_t *correct_part, void *stream) {
  if (!logits_probs || !target || !loss_part || !correct_part || batch < 0 || classes < 2 || s_len < 0) {
    mvn::set_error("mvn_softmax_ce_forward: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || s_len == 0) return MVN_OK;
  if (classes <= 4 * mvn::TQ)
    hipLaunchKernelGGL(mvn::softmax_ce_fwd_cols_kernel, dim3((s_len + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, logits_probs, target, classes, s_len, loss_part, correct_part);
  else
    hipLaunchKernelGGL(mvn::softmax_ce_fwd_kernel, dim3((s_len + 255) / 256, batch), dim3(256), 0,
                       (hipStream_t)stream, logits_probs, target, classes, s_len, loss_part, correct_part);
  return mvn::check_hip(hipGetLastError(), "mvn_softmax_ce_forward");
}

int mvn_softmax_ce_backward(const float *probs, const long long *target, int batch, int classes, int s_len,
                            float scale, const float *upstream, float *dlogit, long long dlogit_batch_stride,
                            int dlogit_ld, int dlogit_col0, int dlogit_cols, void *stream) {
  if (!probs || !target || !dlogit || batch < 0 || classes < 2 || s_len < 0 || dlogit_cols < s_len ||
      dlogit_col0 < 0 || dlogit_ld < dlogit_col0 + dlogit_cols) {
    mvn::set_error("mvn_softmax_ce_backward: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || dlogit_cols == 0) return MVN_OK;
  if (classes <= 4 * mvn::TQ)
    hipLaunchKernelGGL(mvn::softmax_ce_bwd_cols_kernel, dim3((dlogit_cols + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, scale, upstream, dlogit,
                       dlogit_batch_stride, dlogit_ld, dlogit_col0, dlogit_cols);
  else
    hipLaunchKernelGGL(mvn::softmax_ce_bwd_kernel, dim3((dlogit_cols + 255) / 256, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, scale, upstream, dlogit,
                       dlogit_batch_stride, dlogit_ld, dlogit_col0, dlogit_cols);
  return mvn::check_hip(hipGetLastError(), "mvn_softmax_ce_backward");
}

int mvn_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, int decoupled,
                   const size_t *skip_ranges, int n_skip, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1 || n_skip < 0 || n_skip > 4 ||
      (n_skip > 0 && !skip_ranges)) {
    mvn::set_error("mvn_adamw_step: bad argument (step >= 1, at most 4 skip ranges)");
    return MVN_ERR_BAD_ARG;
  }
  if (n == 0) return MVN_OK;
  mvn::AdamSkip sk;
  sk.n = n_skip;
  for (int k = 0; k < 4; ++k) {
    sk.lo[k] = k < n_skip ? skip_ranges[2 * k] : 0;
    sk.hi[k] = k < n_skip ? skip_ranges[2 * k + 1] : 0;
  }
  // bias corrections in double on the host, like torch (python floats)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(mvn::adamw_flat_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (unsigned long long)n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2_sqrt, decoupled, sk);
  return mvn::check_hip(hipGetLastError(), "mvn_adamw_step");
}

}  // extern "C"
