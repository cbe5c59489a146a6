// Video encoder and learned upsampler of the local conditioning path
// (/root/reference/movenet/wavenet.py:94-118 construction, :149-156 upsample_video):
//
//   video (B, F, 64, 64, Cin) --permute--> Conv3d(Cin -> C, kernel (1,64,64)) -> (B, C, F)
//        --> 3 x ConvTranspose1d(C -> C, kernel 10, stride 10) --> (B, C, 1000 F)
//
// The Conv3d consumes a whole frame per output: per (b, f) a C x 4096*Cin mat-vec
// (video_conv_kernel: frames staged in LDS, weight rows streamed coalesced).  A
// transposed conv with stride == kernel has no overlap, so it IS a 1x1 product
// with 10*C output rows (row = (c_out, tap)) whose epilogue scatters row (co, j),
// column i to out[co][10 i + j]: the shared gemm_wx / wgrad MFMA families
// (gemm_family.h) do forward, data gradient and weight gradient.
#include "common.h"
#include "gemm_family.h"

namespace mvn {

constexpr int kUp = 10;     // kernel == stride of every upsampling layer
constexpr int kPix = 4096;  // 64 x 64 pixels per frame

// ---- Conv3d(k = (1,64,64)) ----------------------------------------------------
// grid (ceil(F / FB), B), 256 threads; dynamic LDS = FB * Cin * 4096 floats
__global__ __launch_bounds__(256) void video_conv_kernel(const float *__restrict__ video,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ bias, Act enc,
                                                         int C, int Cin, int F, int FB) {
  extern __shared__ __attribute__((aligned(16))) float frames[];  // [FB][Cin][4096]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, f0 = blockIdx.x * FB;
  const int nf = min(FB, F - f0);
  for (int i = tid; i < nf * Cin * kPix; i += 256) {
    const int f = i / (Cin * kPix), r = i - f * Cin * kPix, cin = r / kPix, p = r - cin * kPix;
    frames[i] = video[(((size_t)b * F + f0 + f) * kPix + p) * Cin + cin];
  }
  __syncthreads();
  for (int c = wave; c < C; c += 4) {
    for (int f = 0; f < nf; ++f) {
      float acc = 0.f;
      for (int cin = 0; cin < Cin; ++cin) {
        const float *wr = w + ((size_t)c * Cin + cin) * kPix;
        const float *fr = frames + ((size_t)f * Cin + cin) * kPix;
#pragma unroll 8
        for (int p = lane; p < kPix; p += 64) acc = fmaf(wr[p], fr[p], acc);
      }
      acc = wave_sum(acc);
      if (lane == 0) *enc.at(b, c, f0 + f) = acc + bias[c];
    }
  }
}

// dW[c][cin][p] += sum_f denc[b][c][f] * video[b][f][p][cin]; grid (Cin*4096/256, ceil(C/8), B)
__global__ __launch_bounds__(256) void video_conv_wgrad_kernel(const float *__restrict__ video,
                                                               Act denc, float *__restrict__ dw,
                                                               int C, int Cin, int F) {
  const int k = blockIdx.x * 256 + threadIdx.x;  // cin * 4096 + p
  const int c0 = blockIdx.y * 8, b = blockIdx.z;
  if (k >= Cin * kPix) return;
  const int cin = k / kPix, p = k - cin * kPix;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < F; ++f) {
    const float v = video[(((size_t)b * F + f) * kPix + p) * Cin + cin];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) acc[j] = fmaf(*denc.at(b, c0 + j, f), v, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c0 + j < C) atomicAdd(dw + (size_t)(c0 + j) * Cin * kPix + k, acc[j]);
}

// db[c] += sum_{b,f} d[b][c][f]  (one wave per channel)
__global__ void rowsum_kernel(Act d, float *__restrict__ db, int batch, int n) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < batch; ++b)
    for (int i = lane; i < n; i += 64) s += *d.at(b, c, i);
  s = wave_sum(s);
  if (lane == 0) db[c] += s;
}

// ---- ConvTranspose1d(k = 10, stride = 10) as a 10C-row 1x1 product ---------------
// weight (C_in, C_out, 10): W(m = co*10 + j, k = ci) = w[(ci*C + co)*10 + j]
struct UpOp {
  int K, t_begin, t_end, C;  // t = input time i
  const float *wt, *bias;
  Act xin, out;
  __device__ __forceinline__ float w(int m, int k) const {
    return (m < kUp * C && k < C) ? wt[((size_t)k * C + m / kUp) * kUp + m % kUp] : 0.f;
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    return (k < C && t < t_end) ? *xin.at(b, k, t) : 0.f;
  }
  __device__ __forceinline__ void epilogue(int b, int mb, int t, int lane, const f32x16 &a0,
                                           const f32x16 &a1) const {
    // row m = U + 4 (lane >> 5) with U = 64 mb + 32 half + (r & 3) + 8 (r >> 2) uniform: the two
    // divisions by 10 per output run on the scalar unit, a lane only selects (as per-lane divisions
    // plus 64-bit addresses the 32 outputs of an epilogue spilled 66 registers)
    float *base = out.at(b, 0, 0);
    const bool hi = lane >= 32;
    const unsigned tcol = (unsigned)(kUp * t);
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int U = mb * 64 + 32 * half + (r & 3) + 8 * (r >> 2);
        const int co_a = U / kUp, j_a = U - co_a * kUp, co_b = (U + 4) / kUp, j_b = (U + 4) - co_b * kUp;
        const int co = hi ? co_b : co_a, j = hi ? j_b : j_a;
        if (U + (hi ? 4 : 0) < kUp * C)
          base[(unsigned)(co * out.ld + j) + tcol] = (half ? a1[r] : a0[r]) + bias[(unsigned)co];
      }
  }
};

// data gradient: dx[ci][i] = sum_{co,j} w[ci][co][j] * dout[co][10 i + j]
struct UpDxOp {
  int K, t_begin, t_end, C;
  const float *wt;
  Act dout, dx;
  __device__ __forceinline__ float w(int m, int k) const {
    return (m < C && k < kUp * C) ? wt[(size_t)m * C * kUp + k] : 0.f;  // k = co*10 + j
  }
  __device__ __forceinline__ float x(int b, int k, int t) const {
    if (k >= kUp * C || t >= t_end) return 0.f;
    const int co = k / kUp, j = k - co * kUp;
    return *dout.at(b, co, kUp * t + j);
  }
  __device__ __forceinline__ void epilogue(int b, int mb, int t, int lane, const f32x16 &a0,
                                           const f32x16 &a1) const {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m0 = mb * 64 + acc_row(r, lane), m1 = m0 + 32;
      if (m0 < C) *dx.at(b, m0, t) = a0[r];
      if (m1 < C) *dx.at(b, m1, t) = a1[r];
    }
  }
};

// weight gradient: dw[ci][co][j] += sum_{b,i} dout[co][10 i + j] * x[ci][i]; db[co] += sum dout
struct UpWgOp {
  int t_begin, t_end, C;
  Act dout, xin;
  float *dwt, *dbt;
  __device__ __forceinline__ float a(int b, int m, int t) const {
    if (m >= kUp * C) return 0.f;
    const int co = m / kUp, j = m - co * kUp;
    return *dout.at(b, co, kUp * t + j);
  }
  __device__ __forceinline__ float x(int b, int n, int t) const {
    return n < C ? *xin.at(b, n, t) : 0.f;
  }
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (m >= kUp * C || n >= C) return nullptr;
    const int co = m / kUp, j = m - co * kUp;
    return dwt + ((size_t)n * C + co) * kUp + j;
  }
  __device__ __forceinline__ float *db(int m) const { return m < kUp * C ? dbt + m / kUp : nullptr; }
};

static int check_video_args(const mvn_dims *dims, int batch, int frames, int cin) {
  int rc = validate_dims(dims);
  if (rc) return rc;
  if (batch < 0 || frames < 1 || cin < 1 || cin > 8) {
    set_error("video: batch %d / frames %d / channels %d out of range (channels <= 8)", batch, frames,
              cin);
    return MVN_ERR_BAD_ARG;
  }
  return MVN_OK;
}

}  // namespace mvn

using namespace mvn;

extern "C" {

int mvn_upsample_video(const mvn_dims *dims, const mvn_video_params *vp, const float *video, int batch,
                       int frames, int cin, float *enc, float *u1, float *u2, float *ctx,
                       int ctx_ld, void *stream_) {
  int rc = check_video_args(dims, batch, frames, cin);
  if (rc) return rc;
  if (!vp || !vp->conv_w || !vp->conv_b || !video || !enc || !u1 || !u2 || !ctx ||
      ctx_ld < 1000 * frames) {
    set_error("mvn_upsample_video: NULL buffer or ctx_ld < 1000*frames");
    return MVN_ERR_BAD_ARG;
  }
  for (int i = 0; i < 3; ++i)
    if (!vp->up_w[i] || !vp->up_b[i]) {
      set_error("mvn_upsample_video: NULL upsampler parameter");
      return MVN_ERR_BAD_ARG;
    }
  if (batch == 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = dims->residual_channels, F = frames;
  const int FB = cin == 1 ? 4 : 1;
  const size_t lds = sizeof(float) * (size_t)FB * cin * kPix;
  rc = ensure_max_dynamic_lds((const void *)video_conv_kernel, "hipFuncSetAttribute(video_conv)");
  if (rc) return rc;
  Act encv = act_view(enc, batch, C, mvn_padded_len(F));
  hipLaunchKernelGGL(video_conv_kernel, dim3((F + FB - 1) / FB, batch), dim3(256), lds, s, video,
                     vp->conv_w, vp->conv_b, encv, C, cin, F, FB);
  float *stage[4] = {enc, u1, u2, ctx};
  int len = F;
  for (int i = 0; i < 3; ++i) {
    UpOp u;
    u.K = C; u.t_begin = 0; u.t_end = len; u.C = C; u.wt = vp->up_w[i]; u.bias = vp->up_b[i];
    u.xin = act_view(stage[i], batch, C, mvn_padded_len(len));
    u.out = act_view(stage[i + 1], batch, C, i == 2 ? ctx_ld : mvn_padded_len(len * kUp));
    launch_gemm(u, kUp * C, batch, s);
    len *= kUp;
  }
  return check_hip(hipGetLastError(), "mvn_upsample_video");
}

int mvn_upsample_video_backward(const mvn_dims *dims, const mvn_video_params *vp,
                                const mvn_video_grads *vg, const float *video, int batch, int frames,
                                int cin, const float *enc, const float *u1, const float *u2,
                                const float *dctx, int dctx_ld, float *d_u2, float *d_u1, float *d_enc,
                                void *stream_) {
  int rc = check_video_args(dims, batch, frames, cin);
  if (rc) return rc;
  if (!vp || !vg || !vg->conv_w || !vg->conv_b || !video || !enc || !u1 || !u2 || !dctx || !d_u2 ||
      !d_u1 || !d_enc || dctx_ld < 1000 * frames) {
    set_error("mvn_upsample_video_backward: NULL buffer");
    return MVN_ERR_BAD_ARG;
  }
  if (batch == 0) return MVN_OK;
  hipStream_t s = (hipStream_t)stream_;
  const int C = dims->residual_channels, F = frames;
  const float *acts[3] = {enc, u1, u2};
  float *dacts[3] = {d_enc, d_u1, d_u2};
  int lens[4] = {F, F * kUp, F * kUp * kUp, F * kUp * kUp * kUp};
  for (int i = 2; i >= 0; --i) {
    // layer i maps acts[i] (length lens[i]) to its output (length lens[i+1])
    Act dout = act_view(const_cast<float *>(i == 2 ? dctx : dacts[i + 1]), batch, C,
                        i == 2 ? dctx_ld : mvn_padded_len(lens[i + 1]));
    Act xin = act_view(const_cast<float *>(acts[i]), batch, C, mvn_padded_len(lens[i]));
    if (!vg->up_w[i] || !vg->up_b[i]) {
      set_error("mvn_upsample_video_backward: NULL gradient pointer");
      return MVN_ERR_BAD_ARG;
    }
    UpWgOp wg;
    wg.t_begin = 0; wg.t_end = lens[i]; wg.C = C; wg.dout = dout; wg.xin = xin;
    wg.dwt = vg->up_w[i]; wg.dbt = vg->up_b[i];
    launch_wgrad(wg, kUp * C, C, batch, nullptr, s);  // bias: atomics (10 rows share a word)
    UpDxOp dx;
    dx.K = kUp * C; dx.t_begin = 0; dx.t_end = lens[i]; dx.C = C; dx.wt = vp->up_w[i];
    dx.dout = dout; dx.dx = act_view(dacts[i], batch, C, mvn_padded_len(lens[i]));
    launch_gemm(dx, C, batch, s);
  }
  Act denc = act_view(d_enc, batch, C, mvn_padded_len(F));
  hipLaunchKernelGGL(video_conv_wgrad_kernel, dim3((cin * kPix + 255) / 256, (C + 7) / 8, batch),
                     dim3(256), 0, s, video, denc, vg->conv_w, C, cin, F);
  hipLaunchKernelGGL(rowsum_kernel, dim3(C), dim3(64), 0, s, denc, vg->conv_b, batch, F);
  return check_hip(hipGetLastError(), "mvn_upsample_video_backward");
}

}  // extern "C"
