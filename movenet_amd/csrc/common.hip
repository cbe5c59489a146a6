// Error plumbing, dimension checks and the one-hot <-> index converters.
#include "common.h"

#include <cstring>
#include <map>
#include <mutex>

namespace mvn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static Switches g_switches;
static std::once_flag g_switches_once;
void parse_switches() {
  auto is = [](const char *name, char c) {
    const char *e = getenv(name);
    return e && e[0] == c;
  };
  Switches w;
  const char *t = getenv("MOVENET_HIP_FORWARD_TILE");
  w.forward_tile = (t && t[0] == '6') ? 64 : (t && t[0] == '3') ? 32 : (t && t[0]) ? 32 : 0;
  w.head_f32 = is("MOVENET_HIP_HEAD_MFMA", 'f');
  w.forward_f32 = is("MOVENET_HIP_FORWARD_MFMA", 'f');
  w.wgrad_f32 = is("MOVENET_HIP_WGRAD_MFMA", 'f');
  w.no_fused_forward = is("MOVENET_HIP_NO_FUSED_FORWARD", '1');
  w.no_persistent_forward = is("MOVENET_HIP_NO_PERSISTENT_FORWARD", '1');
  w.no_dense_strip = is("MOVENET_HIP_NO_DENSE_STRIP", '1');
  w.no_side_stream = is("MOVENET_HIP_NO_SIDE_STREAM", '1');
  w.no_fused_backward = is("MOVENET_HIP_NO_FUSED_BACKWARD", '1');
  w.bwd_split = is("MOVENET_HIP_BWD_FORM", 's');
  w.embed_scalar = is("MOVENET_HIP_EMBED_GRAD", 's');
  g_switches = w;
}
const Switches &switches() {
  std::call_once(g_switches_once, parse_switches);
  return g_switches;
}

int check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return MVN_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return MVN_ERR_LAUNCH;
}

int ensure_max_dynamic_lds(const void *kernel, const char *what) {
  static std::mutex mu;
  static std::map<const void *, uint64_t> done;  // kernel -> bitmask of device ordinals
  int dev = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return MVN_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(mu);
  uint64_t &mask = done[kernel];
  if (dev >= 0 && dev < 64 && ((mask >> dev) & 1u)) return MVN_OK;
  int rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                     what);
  if (rc) return rc;
  if (dev >= 0 && dev < 64) mask |= (uint64_t)1 << dev;  // ordinals >= 64: set it every time
  return MVN_OK;
}

int validate_dims(const mvn_dims *d) {
  if (!d) {
    set_error("dims is NULL");
    return MVN_ERR_BAD_ARG;
  }
  if (d->layer_size < 1 || d->layer_size > 20 || d->stack_size < 1 || d->stack_size > 64) {
    set_error("layer_size %d / stack_size %d out of range", d->layer_size, d->stack_size);
    return MVN_ERR_BAD_DIMS;
  }
  if (d->input_channels < 2 || d->residual_channels < 1 || d->skip_channels < 1) {
    set_error("channel counts must be positive (Q=%d C=%d K=%d)", d->input_channels,
              d->residual_channels, d->skip_channels);
    return MVN_ERR_BAD_DIMS;
  }
  return MVN_OK;
}

// (B,Q,T) one-hot -> (B,T) index; -1 where the column is not exactly one-hot.
__global__ void onehot_to_index_kernel(const float *__restrict__ x, int32_t *__restrict__ idx, int Q,
                                       int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float *col = x + (size_t)b * Q * T + t;
  int found = -1, bad = 0;
  for (int q = 0; q < Q; ++q) {
    const float v = col[(size_t)q * T];  // lanes walk t: coalesced
    if (v == 1.0f) {
      if (found >= 0) bad = 1;
      found = q;
    } else if (v != 0.0f) {
      bad = 1;
    }
  }
  idx[(size_t)b * T + t] = (bad || found < 0) ? -1 : found;
}
// The same for T % 4 == 0 and 16-byte aligned rows: a thread takes four columns with float4
// loads, eight rows in flight, branch-free (count of ones, count of other non-zeros, last index).
// The (B,256,16000) input of a config-2 training step is 262 MB: 130 us with the kernel above.
__global__ __launch_bounds__(256) void onehot_to_index4_kernel(const float *__restrict__ x, int32_t *__restrict__ idx,
                                                               int Q, int T) {
  const int b = blockIdx.y;
  const int t = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= T) return;
  const float *col = x + (size_t)b * Q * T + t;
  int found[4] = {-1, -1, -1, -1}, ones[4] = {0, 0, 0, 0}, other[4] = {0, 0, 0, 0};
  int q = 0;
  for (; q + 8 <= Q; q += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const float4 *)(col + (size_t)(q + j) * T);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool one = e[k] == 1.0f;
        ones[k] += one;
        other[k] += (!one && e[k] != 0.0f);
        found[k] = one ? q + j : found[k];
      }
    }
  }
  for (; q < Q; ++q) {
    const float4 w = *(const float4 *)(col + (size_t)q * T);
    const float e[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool one = e[k] == 1.0f;
      ones[k] += one;
      other[k] += (!one && e[k] != 0.0f);
      found[k] = one ? q : found[k];
    }
  }
  int4 r;
  r.x = (ones[0] == 1 && other[0] == 0) ? found[0] : -1;
  r.y = (ones[1] == 1 && other[1] == 0) ? found[1] : -1;
  r.z = (ones[2] == 1 && other[2] == 0) ? found[2] : -1;
  r.w = (ones[3] == 1 && other[3] == 0) ? found[3] : -1;
  *(int4 *)(idx + (size_t)b * T + t) = r;
}

__global__ void index_to_onehot_kernel(const int32_t *__restrict__ idx, int stride,
                                       float *__restrict__ x, int Q, int T) {
  const int b = blockIdx.z;
  const int q = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  x[((size_t)b * Q + q) * T + t] = (idx[(size_t)b * stride + t] == q) ? 1.0f : 0.0f;
}

// mu-law companding, the formula the project states (RESEARCH.md:156-163) and
// torchaudio.functional.mu_law_encoding/decoding implement (absent offline: UNPINNED):
//   y = sign(x) ln(1 + mu|x|) / ln(1 + mu),  q = int((y + 1)/2 * mu + 0.5),  mu = Q - 1
__global__ void mu_law_encode_kernel(const float *__restrict__ x, int32_t *__restrict__ q, size_t n,
                                     int Q) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mu = (float)(Q - 1), v = x[i];
  const float y = copysignf(log1pf(mu * fabsf(v)) / log1pf(mu), v);
  q[i] = (int32_t)((y + 1.0f) / 2.0f * mu + 0.5f);
}
__global__ void mu_law_decode_kernel(const int32_t *__restrict__ q, float *__restrict__ x, size_t n,
                                     int Q) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mu = (float)(Q - 1);
  const float y = ((float)q[i] / mu) * 2.0f - 1.0f;
  x[i] = copysignf((expf(fabsf(y) * log1pf(mu)) - 1.0f) / mu, y);
}

// ---- F3: cross_entropy on probabilities + accuracy (pytorch_lightning_trainer.py:64-66) ----
// one thread per (b, s) column, lanes walk s (coalesced), two passes over the Q rows
__global__ __launch_bounds__(256) void ce_probs_fwd_kernel(const float *__restrict__ p,
                                                           const long long *__restrict__ target, int Q,
                                                           int S, float *__restrict__ loss_part,
                                                           int32_t *__restrict__ correct_part) {
  const int b = blockIdx.y, sidx = blockIdx.x * blockDim.x + threadIdx.x;
  float loss = 0.f;
  int ok = 0;
  if (sidx < S) {
    const float *col = p + (size_t)b * Q * S + sidx;
    float m = -INFINITY;
    int arg = 0;
    for (int q = 0; q < Q; ++q) {
      const float v = col[(size_t)q * S];
      if (v > m) {  // strict: first maximum, like torch.argmax
        m = v;
        arg = q;
      }
    }
    float sum = 0.f;
    for (int q = 0; q < Q; ++q) sum += sm_exp(col[(size_t)q * S] - m);
    const long long tg = target[(size_t)b * S + sidx];
    const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
    loss = (m + logf(sum)) - col[(size_t)tq * S];
    ok = arg == tq;
  }
  __shared__ float ls[4];
  __shared__ int cs[4];
  loss = wave_sum(loss);
  float okf = wave_sum((float)ok);  // <= 64: exact
  if ((threadIdx.x & 63) == 0) {
    ls[threadIdx.x >> 6] = loss;
    cs[threadIdx.x >> 6] = (int)okf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int part = blockIdx.y * gridDim.x + blockIdx.x;
    loss_part[part] = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    correct_part[part] = (cs[0] + cs[1]) + (cs[2] + cs[3]);
  }
}

__global__ __launch_bounds__(256) void ce_probs_bwd_kernel(const float *__restrict__ p,
                                                           const long long *__restrict__ target, int Q,
                                                           int S, float scale,
                                                           const float *__restrict__ upstream,
                                                           float *__restrict__ dp) {
  const int b = blockIdx.y, sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= S) return;
  if (upstream) scale *= *upstream;
  const float *col = p + (size_t)b * Q * S + sidx;
  float *dcol = dp + (size_t)b * Q * S + sidx;
  float m = -INFINITY;
  for (int q = 0; q < Q; ++q) m = fmaxf(m, col[(size_t)q * S]);
  float sum = 0.f;
  for (int q = 0; q < Q; ++q) sum += sm_exp(col[(size_t)q * S] - m);
  const float inv = 1.0f / sum;
  const long long tg = target[(size_t)b * S + sidx];
  const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
  for (int q = 0; q < Q; ++q) {
    const float sm = sm_exp(col[(size_t)q * S] - m) * inv;
    dcol[(size_t)q * S] = scale * (sm - (q == tq ? 1.0f : 0.0f));
  }
}

// single-pass forms for Q <= 256 (see softmax_cols_kernel in sequence.hip: 64 columns per
// workgroup, wave w holds class rows [64w, 64w+64) of them in registers, lane = column)
constexpr int CEQ = 64;
__device__ __forceinline__ float ce_col_reduce(float v, float (*part)[64], int wave, int lane, bool is_max) {
  part[wave][lane] = v;
  __syncthreads();
  const float a = part[0][lane], b = part[1][lane], c = part[2][lane], d = part[3][lane];
  __syncthreads();
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

// BWD = false: loss / correct partial sums per workgroup; BWD = true: dprobs
template <bool BWD>
__global__ __launch_bounds__(256) void ce_probs_cols_kernel(const float *__restrict__ p,
                                                            const long long *__restrict__ target, int Q,
                                                            int S, float scale,
                                                            const float *__restrict__ upstream,
                                                            float *__restrict__ out_f,
                                                            int32_t *__restrict__ correct_part) {
  __shared__ float part[4][64];
  __shared__ int argp[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int sidx = blockIdx.x * 64 + lane;
  const bool live = sidx < S;
  const float *col = p + (size_t)b * Q * S + (live ? sidx : 0);
  float v[CEQ];
  float m = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int i = 0; i < CEQ; ++i) {
    const int q = CEQ * wave + i;
    v[i] = (live && q < Q) ? col[(size_t)q * S] : -INFINITY;
    if (v[i] > m) {  // strict: first maximum inside this wave's rows
      m = v[i];
      arg = q;
    }
  }
  const float wave_m = m;
  m = ce_col_reduce(m, part, wave, lane, true);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CEQ; ++i) {
    v[i] = sm_exp(v[i] - m);
    sum += v[i];
  }
  sum = ce_col_reduce(sum, part, wave, lane, false);
  const long long tg = live ? target[(size_t)b * S + sidx] : 0;
  const int tq = (int)min(max(tg, 0LL), (long long)(Q - 1));
  if (BWD) {
    if (!live) return;
    if (upstream) scale *= *upstream;
    const float inv = 1.0f / sum;
    float *dcol = out_f + (size_t)b * Q * S + sidx;
#pragma unroll
    for (int i = 0; i < CEQ; ++i) {
      const int q = CEQ * wave + i;
      if (q < Q) dcol[(size_t)q * S] = scale * (v[i] * inv - (q == tq ? 1.0f : 0.0f));
    }
  } else {
    // first global maximum: the lowest wave whose own maximum equals the column maximum
    argp[wave][lane] = wave_m == m ? arg : 0x7fffffff;
    // the target's probability sits in exactly one wave's registers
    float pt = 0.f;
    {
      const float *pc = col + (size_t)tq * S;
      if (live && wave == tq / CEQ) pt = *pc;
    }
    const float pt_all = ce_col_reduce(pt, part, wave, lane, false);
    __syncthreads();
    float loss = 0.f;
    int ok = 0;
    if (wave == 0 && live) {
      const int a0 = min(min(argp[0][lane], argp[1][lane]), min(argp[2][lane], argp[3][lane]));
      loss = (m + logf(sum)) - pt_all;
      ok = a0 == tq;
    }
    if (wave == 0) {
      loss = wave_sum(loss);
      const float okf = wave_sum((float)ok);
      if (lane == 0) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;
        out_f[wg] = loss;
        correct_part[wg] = (int)okf;
      }
    }
  }
}

}  // namespace mvn

namespace mvn {
// one wave: device words to pinned host memory, the values first, then the sequence number the
// host polls for (system scope: visible to the CPU without waiting for the end of the grid)
__global__ void publish_words_kernel(const uint32_t *__restrict__ words, int n, int32_t seq, uint32_t *host_words) {
  const int lane = threadIdx.x;
  if (lane < n) __hip_atomic_store(host_words + lane, words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __builtin_amdgcn_s_barrier();
  if (lane == 0) __hip_atomic_store(host_words + n, (uint32_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace mvn

extern "C" {

int mvn_ce_parts(int batch, int s_len) {
  if (batch < 0 || s_len < 0) return 0;
  return batch * ((s_len + 63) / 64);
}

int mvn_ce_on_probs_forward(const float *probs, const long long *target, int batch, int classes,
                            int s_len, float *loss_part, int32_t *correct_part, void *stream) {
  if (!probs || !target || !loss_part || !correct_part || batch < 0 || classes < 2 || s_len < 0) {
    mvn::set_error("mvn_ce_on_probs_forward: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || s_len == 0) return MVN_OK;
  // partial-sum slots: mvn_ce_parts() = one per 64-column workgroup (the wide-Q form uses a
  // quarter of them: the caller zero-fills the arrays)
  if (classes <= 4 * mvn::CEQ)
    hipLaunchKernelGGL(mvn::ce_probs_cols_kernel<false>, dim3((s_len + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, 0.f, nullptr, loss_part,
                       correct_part);
  else
    hipLaunchKernelGGL(mvn::ce_probs_fwd_kernel, dim3((s_len + 255) / 256, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, loss_part, correct_part);
  return mvn::check_hip(hipGetLastError(), "ce_on_probs_forward");
}

int mvn_ce_on_probs_backward(const float *probs, const long long *target, int batch, int classes,
                             int s_len, float scale, const float *upstream, float *dprobs,
                             void *stream) {
  if (!probs || !target || !dprobs || batch < 0 || classes < 2 || s_len < 0) {
    mvn::set_error("mvn_ce_on_probs_backward: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || s_len == 0) return MVN_OK;
  if (classes <= 4 * mvn::CEQ)
    hipLaunchKernelGGL(mvn::ce_probs_cols_kernel<true>, dim3((s_len + 63) / 64, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, scale, upstream, dprobs, nullptr);
  else
    hipLaunchKernelGGL(mvn::ce_probs_bwd_kernel, dim3((s_len + 255) / 256, batch), dim3(256), 0,
                       (hipStream_t)stream, probs, target, classes, s_len, scale, upstream, dprobs);
  return mvn::check_hip(hipGetLastError(), "ce_on_probs_backward");
}

int mvn_mu_law_encode(const float *x, int32_t *index, size_t n, int classes, void *stream) {
  if (!x || !index || classes < 2) {
    mvn::set_error("mvn_mu_law_encode: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (n == 0) return MVN_OK;
  hipLaunchKernelGGL(mvn::mu_law_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, index, n, classes);
  return mvn::check_hip(hipGetLastError(), "mu_law_encode");
}

int mvn_mu_law_decode(const int32_t *index, float *x, size_t n, int classes, void *stream) {
  if (!x || !index || classes < 2) {
    mvn::set_error("mvn_mu_law_decode: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (n == 0) return MVN_OK;
  hipLaunchKernelGGL(mvn::mu_law_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, index, x, n, classes);
  return mvn::check_hip(hipGetLastError(), "mu_law_decode");
}

int mvn_reload_switches(void) {
  mvn::parse_switches();
  return MVN_OK;
}

int mvn_abi_version(void) { return MVN_ABI_VERSION; }
const char *mvn_last_error(void) { return mvn::g_err; }

int mvn_receptive_fields(const mvn_dims *dims) {
  int rc = mvn::validate_dims(dims);
  if (rc) return rc;
  return (int)(mvn::dilation_sum(dims) + dims->stack_size);
}

int mvn_output_size(const mvn_dims *dims, int t_len) {
  int rf = mvn_receptive_fields(dims);
  if (rf < 0) return rf;
  int s = t_len - rf + 1;
  if (s < 1) {
    mvn::set_error(
        "input time steps must be larger than the number of receptive fields. "
        "Number of input timesteps = %d, receptive fields = %d",
        t_len, rf);
    return MVN_ERR_TOO_SHORT;
  }
  return s;
}

int mvn_onehot_to_index(const float *onehot, int32_t *index, int batch, int classes, int t_len,
                        void *stream) {
  if (!onehot || !index || batch < 0 || classes < 1 || t_len < 0) {
    mvn::set_error("mvn_onehot_to_index: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || t_len == 0) return MVN_OK;
  if (t_len % 4 == 0 && ((uintptr_t)onehot & 15) == 0 && ((uintptr_t)index & 15) == 0) {
    dim3 grid4((t_len / 4 + 255) / 256, batch);
    hipLaunchKernelGGL(mvn::onehot_to_index4_kernel, grid4, dim3(256), 0, (hipStream_t)stream, onehot, index,
                       classes, t_len);
    return mvn::check_hip(hipGetLastError(), "onehot_to_index");
  }
  dim3 grid((t_len + 255) / 256, batch);
  hipLaunchKernelGGL(mvn::onehot_to_index_kernel, grid, dim3(256), 0, (hipStream_t)stream, onehot,
                     index, classes, t_len);
  return mvn::check_hip(hipGetLastError(), "onehot_to_index");
}

int mvn_publish_words(const uint32_t *words, int n, int32_t seq, uint32_t *host_words, void *stream) {
  if (!words || !host_words || n < 1 || n > 64) {
    mvn::set_error("mvn_publish_words: NULL pointer or n outside 1..64");
    return MVN_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(mvn::publish_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, n, seq, host_words);
  return mvn::check_hip(hipGetLastError(), "mvn_publish_words");
}

int mvn_index_to_onehot(const int32_t *index, int index_stride, float *onehot, int batch,
                        int classes, int t_len, void *stream) {
  if (!onehot || !index || batch < 0 || classes < 1 || t_len < 0 || index_stride < t_len) {
    mvn::set_error("mvn_index_to_onehot: bad argument");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0 || t_len == 0) return MVN_OK;
  dim3 grid((t_len + 255) / 256, classes, batch);
  hipLaunchKernelGGL(mvn::index_to_onehot_kernel, grid, dim3(256), 0, (hipStream_t)stream, index,
                     index_stride, onehot, classes, t_len);
  return mvn::check_hip(hipGetLastError(), "index_to_onehot");
}

}  // extern "C"
