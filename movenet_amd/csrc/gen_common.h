// Shared by the generator kernels (generate.hip, generate_pipe.hip).
#pragma once
#include "common.h"

namespace mvn {

struct GenArgs {
  int L, layer_size, Q, C, K;
  const float *w;
  float *state;
  long long state_per_seq;
  int32_t *samples;
  int stride, n_total, n_given, t_begin, t_end;
  float temperature;
  uint64_t seed;
  float *logits_out;
  int32_t *choices_out;
  int logits_t0;
  // local conditioning (NULL = audio only): context (B, n_total, C) time-major and the
  // packed context-conv section of the weight blob
  const float *ctx_tm;
  long long ctx_stride_b;
  const float *wctx;
};

__device__ __forceinline__ int ring_offset(int l, int layer_size, int C) {
  const int stack = l / layer_size, pos = l - stack * layer_size;
  return C * (stack * ((1 << layer_size) - 1) + ((1 << pos) - 1));
}

// LDS-only barrier: outstanding global loads (the weight prefetch) stay in
// flight across it.  __syncthreads() would add a full vmcnt(0) drain whenever a
// global store is pending.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float ring_load(const float *p) {
  // agent-scope relaxed load: served by L2, never by a stale L1 line
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// full f/g matrix element: row o in [0,2C) (filter | gate), column k in [0,2C)
// (tap 0 = past | tap 1 = current)
__device__ __forceinline__ float fg_elem(const float *fw, const float *gw, int C, int o, int k) {
  const int tap = k >= C, kc = k - tap * C;
  const float *w = o < C ? fw : gw;
  const int oc = o < C ? o : o - C;
  return w[((size_t)oc * C + kc) * 2 + tap];
}
__device__ __forceinline__ float rs_elem(const float *rw, const float *sw, int C, int o, int k) {
  return o < C ? rw[(size_t)o * C + k] : sw[(size_t)(o - C) * C + k];
}


// class counts the 256-wide heads of the tuned generators take (padded at pack time: generate_fold.hip)
inline bool head_q_ok(int q) { return q == 64 || q == 128 || q == 256; }

// ---- PIPE variant (generate_pipe.hip) ------------------------------------
bool pipe_ok(const mvn_dims *d);
int pipe_stages(const mvn_dims *d);
size_t pipe_hand_floats(const mvn_dims *d, int batch);  // hand-off area appended to the state
int pipe_pipelines(const mvn_dims *d);                   // pipelines that fit co-resident (256 CUs)
int pipe_max_batch(const mvn_dims *d);                   // ... each serving up to PipeCfg::GMAX sequences in turn
size_t pipe_weights_floats(const mvn_dims *d);          // packed blob without the context section
int pipe_pack(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s);
int pipe_pack_ctx(const mvn_dims *d, const mvn_params *p, float *ctx_section, hipStream_t s);
int pipe_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_floats_total,
                size_t status_offset_floats, hipStream_t s);


// ---- PIPE variant with fp16 operands (generate_pipe_h16.hip), C = K = 128 ----------------
bool pipe_h16_ok(const mvn_dims *d);
int pipe_h16_stages(const mvn_dims *d);
int pipe_h16_pipelines(const mvn_dims *d);  // pipelines co-resident on the chip
int pipe_h16_max_batch(const mvn_dims *d);  // ... each serving up to h16::GMAX sequences in turn
size_t pipe_h16_weights_floats(const mvn_dims *d);  // packed blob without the context section
int pipe_h16_pack(const mvn_dims *d, const mvn_params *p, float *packed, bool has_ctx, hipStream_t s);
int pipe_h16_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_floats_total,
                    size_t status_offset_floats, hipStream_t s);

// ---- FOLD variant (generate_fold.hip), C = K = 64: residual 1x1 folded into the next layer
bool fold_ok(const mvn_dims *d);
int fold_stages(const mvn_dims *d);
int fold_pipelines(const mvn_dims *d);  // pipelines co-resident on the chip (one sequence each: the fastest step)
int fold_pipelines_max(const mvn_dims *d);  // ... plus those the XCDs' left-over CUs form across XCDs (slower hops)
int fold_max_batch(const mvn_dims *d);  // fold_pipelines_max, each serving up to fold::GMAX sequences in turn
int fold_launch_pipelines(const mvn_dims *d, int batch);  // pipelines a launch of `batch` sequences runs on
size_t fold_weights_floats(const mvn_dims *d);  // packed blob without the context section
size_t fold_hand_floats(const mvn_dims *d, int batch);
int fold_pack(const mvn_dims *d, const mvn_params *p, float *packed, hipStream_t s);
int fold_launch(const GenArgs &a, const mvn_dims *d, int batch, float *hand, size_t hand_floats_total,
                size_t status_offset_floats, hipStream_t s);

// Hand-off area of the generator state, shared by the pipelined variants: [granules: the
// largest variant's count][16 flag words, the sticky status word first][placement words]
size_t hand_status_offset(const mvn_dims *d, int batch);  // floats from the area's start
size_t hand_total_floats(const mvn_dims *d, int batch);

}  // namespace mvn
