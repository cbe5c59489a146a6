// The MFMA kernel families every convolution-shaped product of the full-sequence path is
// built from (see sequence.hip for the description), shared by sequence.hip (decoder) and
// video.hip (video encoder / learned upsampler): gemm_wx_kernel / gemm_wx_staged_kernel
// (Y = W X over time), wgrad_kernel / wgrad2_kernel (dW = A X^T, K = time) and their reducers.
#pragma once
#include "common.h"

namespace mvn {

// Every tiled kernel starts its tiles at a multiple of 32 columns of the tensors' time axis (rows
// are 256-byte aligned): the 128-byte row segments the loads and stores of a tile touch are whole
// cache lines.  (Columns in front of an op's t_begin are masked as before.)
constexpr int TILE_ALIGN = 31;


typedef float f32x16 __attribute__((ext_vector_type(16)));
}  // namespace mvn
#include "bf3.h"
namespace mvn {

struct Act {  // (B, ch, ld) view
  float *p;
  long long sb;  // batch stride (floats)
  int ld;        // row stride (floats)
  __device__ __forceinline__ float *at(int b, int ch, int t) const {
    return p + (size_t)b * sb + (size_t)ch * ld + t;
  }
};

// Masked operand load WITHOUT a branch: the address is clamped into [lo, hi) and the value
// selected afterwards.  `cond ? *p : 0` makes hipcc wrap every element in its own exec-mask
// region (s_and_saveexec / s_cbranch_execz), which made operand staging 1.5-2x slower.
__device__ __forceinline__ float ld_masked(const Act &a, int b, int ch, int t, int lo, int hi) {
  const int tc = min(max(t, lo), hi - 1);
  const float v = *a.at(b, ch, tc);
  return (t >= lo && t < hi) ? v : 0.f;
}

// accumulator register r of lane `lane` -> row inside a 32x32 MFMA tile
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// ======================================================================
// gemm_wx: Y[m][t] = sum_k W(m,k) X(k,t), block tile 64(m) x 256(t),
// 4 waves, wave w owns t in [64w, 64w+64) as 2x2 MFMA tiles.
// ======================================================================
constexpr int GX_KC = 16;  // k per LDS chunk

template <class Op>
__global__ __launch_bounds__(256, 2) void gemm_wx_kernel(Op op) {
  __shared__ float Ws[2][GX_KC][64];
  __shared__ float Xs[2][GX_KC][256];
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep row/k indices (and the
  // operand accessors' branches on them) on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, mb = blockIdx.y;
  const int t0 = op.t_begin + blockIdx.x * 256;
  const int nchunk = (op.K + GX_KC - 1) / GX_KC;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float wreg[4], xreg[16];
  auto gload = [&](int c) {
    const int k0 = c * GX_KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = op.w(mb * 64 + lane, k0 + wave + 4 * j);
#pragma unroll
    for (int j = 0; j < 16; ++j) xreg[j] = op.x(b, k0 + j, t0 + tid);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) Ws[buf][wave + 4 * j][lane] = wreg[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) Xs[buf][j][tid] = xreg[j];
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(c + 1);
#pragma unroll
    for (int kk = 0; kk < GX_KC / 2; ++kk) {
      const int kr = 2 * kk + (lane >> 5);
      const float a0 = Ws[buf][kr][lane & 31], a1 = Ws[buf][kr][32 + (lane & 31)];
      const float b0 = Xs[buf][kr][64 * wave + (lane & 31)];
      const float b1 = Xs[buf][kr][64 * wave + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: acc[mi][ni][r] is row 32*mi + acc_row(r), column 64*wave + 32*ni + (lane&31)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int t = t0 + 64 * wave + 32 * ni + (lane & 31);
    if (t < op.t_end) op.epilogue(b, mb, t, lane, acc[0][ni], acc[1][ni]);
  }
}

// Same product, epilogue through LDS: the accumulators are written to a 32-row x
// 256-column stage (the idle X buffers), then every thread takes whole float4 column
// groups of whole rows -- 16-byte, row-contiguous global accesses, no 64-value register
// epilogue (which spilled ~100 VGPRs to scratch in the residual/skip and dz ops).  Tiles
// start at a multiple of 4 so that those accesses are aligned; Op::x must return 0 for
// t < t_begin; Op::load4(b, m, t) fetches what the store of those four columns needs from
// HBM (an Op::Pre) and Op::store4(b, m, t, v, pre) masks columns outside [t_begin, t_end).
typedef float4 f4;

template <class Op>
__global__ __launch_bounds__(256, 2) void gemm_wx_staged_kernel(Op op) {
  __shared__ float Ws[2][GX_KC][64];
  __shared__ __attribute__((aligned(16))) float Xs[2][GX_KC][256];
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep row/k indices (and the
  // operand accessors' branches on them) on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, mb = blockIdx.y;
  const int t0 = (op.t_begin & ~TILE_ALIGN) + blockIdx.x * 256;
  const int nchunk = (op.K + GX_KC - 1) / GX_KC;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float wreg[4], xreg[16];
  auto gload = [&](int c) {
    const int k0 = c * GX_KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = op.w(mb * 64 + lane, k0 + wave + 4 * j);
#pragma unroll
    for (int j = 0; j < 16; ++j) xreg[j] = op.x(b, k0 + j, t0 + tid);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) Ws[buf][wave + 4 * j][lane] = wreg[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) Xs[buf][j][tid] = xreg[j];
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(c + 1);
#pragma unroll
    for (int kk = 0; kk < GX_KC / 2; ++kk) {
      const int kr = 2 * kk + (lane >> 5);
      const float a0 = Ws[buf][kr][lane & 31], a1 = Ws[buf][kr][32 + (lane & 31)];
      const float b0 = Xs[buf][kr][64 * wave + (lane & 31)];
      const float b1 = Xs[buf][kr][64 * wave + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  float *stage = &Xs[0][0][0];  // [32][256]
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[acc_row(r, lane) * 256 + 64 * wave + 32 * ni + (lane & 31)] = acc[mi][ni][r];
    __syncthreads();
    const int c4 = tid & 63, t = t0 + 4 * c4;
    if constexpr (Op::FG_PAIRS) {
      // gated layer: rows [0,16) of this half are filter rows, [16,32) the gate rows of the
      // same channels; one thread gates four columns of a channel and writes z, tanh, sigmoid
      if (t < op.t_end) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wave + 4 * j;
          op.store_fg(b, mb * 32 + 16 * mi + row, t, *(const f4 *)&stage[row * 256 + 4 * c4],
                      *(const f4 *)&stage[(16 + row) * 256 + 4 * c4]);
        }
      }
    } else if (t < op.t_end) {
      // two phases: first every global operand of the eight stores (residual input, skip
      // accumulator, tanh/sigmoid ...) is requested, then the stores run.  Fused into one
      // loop each store waited for its own loads: eight HBM round trips in a row per half.
      // (Op::EPI_BATCH stores per batch: 8, or 4 where two preloaded operands per store would
      // push the kernel over 128 registers, i.e. from 4 to 3 waves per SIMD)
      constexpr int EB = Op::EPI_BATCH;
#pragma unroll
      for (int j0 = 0; j0 < 8; j0 += EB) {
        f4 v[EB];
        typename Op::Pre pre[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
          const int row = wave + 4 * (j0 + j);
          v[j] = *(const f4 *)&stage[row * 256 + 4 * c4];
          pre[j] = op.load4(b, mb * 64 + 32 * mi + row, t);
        }
#pragma unroll
        for (int j = 0; j < EB; ++j)
          op.store4(b, mb * 64 + 32 * mi + wave + 4 * (j0 + j), t, v[j], pre[j]);
      }
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------
// The same product with FP16 OPERANDS and FP32 ACCUMULATION (BASELINE configs[4]; precedent:
// torch.autocast, movenet/trainer.py:124): W and X are rounded to fp16 (nearest even) when they
// are stored to LDS -- after Op::x's own element-wise map (leaky-ReLU ...), i.e. exactly the
// operands the fp16 generator kernel rounds -- and multiplied by v_mfma_f32_32x32x16_f16; the
// accumulators, the epilogue (bias, gate, residual add, skip accumulation) and every tensor
// in HBM stay fp32.  One MFMA per 16-deep k-chunk and tile instead of eight: the kernel is
// bound by its load path.  LDS tiles are k-contiguous ([row][16 halves], pitch 48 bytes: a
// lane's 8 operands are one ds_read_b128), X is transposed on the way in (a thread holds the
// 16 k of its own column).
// ----------------------------------------------------------------------------------------
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
constexpr int GXH_PITCH = 24;  // halves per LDS row (16 used): 48 bytes keeps ds_read_b128 aligned and spreads banks

template <class Op>
__global__ __launch_bounds__(256, 2) void gemm_wx_staged_f16_kernel(Op op) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * GXH_PITCH * 2 + 32 * 256 * 4];
  _Float16 *Wh = (_Float16 *)lds;                                   // [2][64][GXH_PITCH]
  _Float16 *Xh = (_Float16 *)(lds + 2 * 64 * GXH_PITCH * 2);       // [2][256][GXH_PITCH] (24 KB)
  float *stage = (float *)(lds + 2 * 64 * GXH_PITCH * 2);           // [32][256] floats (32 KB), epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, mb = blockIdx.y;
  const int t0 = (op.t_begin & ~TILE_ALIGN) + blockIdx.x * 256;
  const int nchunk = (op.K + GX_KC - 1) / GX_KC;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float wreg[4], xreg[16];
  auto gload = [&](int c) {
    const int k0 = c * GX_KC;
    // thread -> W row (tid >> 2), k quarter (tid & 3): four consecutive k
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = op.w(mb * 64 + (tid >> 2), k0 + 4 * (tid & 3) + j);
#pragma unroll
    for (int j = 0; j < 16; ++j) xreg[j] = op.x(b, k0 + j, t0 + tid);
  };
  auto lstore = [&](int buf) {
    typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
    *(hf4 *)&Wh[(buf * 64 + (tid >> 2)) * GXH_PITCH + 4 * (tid & 3)] =
        hf4{(_Float16)wreg[0], (_Float16)wreg[1], (_Float16)wreg[2], (_Float16)wreg[3]};
    hf8 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lo[j] = (_Float16)xreg[j];
      hi[j] = (_Float16)xreg[8 + j];
    }
    *(hf8 *)&Xh[(buf * 256 + tid) * GXH_PITCH] = lo;
    *(hf8 *)&Xh[(buf * 256 + tid) * GXH_PITCH + 8] = hi;
  };

  gload(0);
  lstore(0);
  __syncthreads();
  const int li = lane & 31, kg = 8 * (lane >> 5);
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(c + 1);
    const hf8 a0 = *(const hf8 *)&Wh[(buf * 64 + li) * GXH_PITCH + kg];
    const hf8 a1 = *(const hf8 *)&Wh[(buf * 64 + 32 + li) * GXH_PITCH + kg];
    const hf8 b0 = *(const hf8 *)&Xh[(buf * 256 + 64 * wave + li) * GXH_PITCH + kg];
    const hf8 b1 = *(const hf8 *)&Xh[(buf * 256 + 64 * wave + 32 + li) * GXH_PITCH + kg];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: identical to gemm_wx_staged_kernel (fp32 accumulators through a 32 x 256 stage)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[acc_row(r, lane) * 256 + 64 * wave + 32 * ni + (lane & 31)] = acc[mi][ni][r];
    __syncthreads();
    const int c4 = tid & 63, t = t0 + 4 * c4;
    if constexpr (Op::FG_PAIRS) {
      if (t < op.t_end) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wave + 4 * j;
          op.store_fg(b, mb * 32 + 16 * mi + row, t, *(const f4 *)&stage[row * 256 + 4 * c4],
                      *(const f4 *)&stage[(16 + row) * 256 + 4 * c4]);
        }
      }
    } else if (t < op.t_end) {
      constexpr int EB = Op::EPI_BATCH;
#pragma unroll
      for (int j0 = 0; j0 < 8; j0 += EB) {
        f4 v[EB];
        typename Op::Pre pre[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
          const int row = wave + 4 * (j0 + j);
          v[j] = *(const f4 *)&stage[row * 256 + 4 * c4];
          pre[j] = op.load4(b, mb * 64 + 32 * mi + row, t);
        }
#pragma unroll
        for (int j = 0; j < EB; ++j)
          op.store4(b, mb * 64 + 32 * mi + wave + 4 * (j0 + j), t, v[j], pre[j]);
      }
    }
    __syncthreads();
  }
}

// column mask helpers for store4: all four columns inside [lo, hi)?
__device__ __forceinline__ bool cols_full(int t, int lo, int hi) { return t >= lo && t + 3 < hi; }
struct Pre1 { f4 a; };     // operands preloaded for one store4 (see gemm_wx_staged_kernel)
struct Pre2 { f4 a, b; };
constexpr f4 kZero4 = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ float f4_get(const f4 &v, int e) {
  return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

// ======================================================================
// wgrad: dW(m,n) += sum_{b,t} A(b,m,t) * X(b,n,t); block tile 64 x 64, each
// wave one 32x32 MFMA tile, K = time.  Grid: (time chunks * B, M/64 * N/64).
// ======================================================================
constexpr int WG_T = 64;       // time per LDS tile
constexpr int WG_CHUNK = 512;  // time per workgroup (8 tiles): >= 900 workgroups at config 2

// bias_part: optional scratch [gridDim.x][64 * (M blocks)] -- every workgroup with
// nblk == 0 stores its 64 row sums there and bias_reduce_kernel adds them up in a
// fixed order.  (Atomics straight into the 64..256 bias words serialise: ~800
// workgroups on four cache lines made this kernel 3x slower.)
template <class Op>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(Op op, int nblk_n, int chunks_per_b,
                                                      float *__restrict__ bias_part, int m_rows64) {
  __shared__ float As[2][64][WG_T + 1];
  __shared__ float Xs[2][64][WG_T + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep row/k indices (and the
  // operand accessors' branches on them) on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int mblk = blockIdx.y / nblk_n, nblk = blockIdx.y - mblk * nblk_n;
  const int mi = wave >> 1, ni = wave & 1;
  const int tb = op.t_begin + ch * WG_CHUNK, te = min(op.t_end, tb + WG_CHUNK);
  const bool want_bias = nblk == 0;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // wave w stages rows w, w+4, ..., w+60 (lane = time): 16 + 16 values per thread
  float areg[16], xreg[16], bsum[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bsum[j] = 0.f;
  auto gload = [&](int t0) {
    const int t = min(t0 + lane, te - 1);  // clamped: loads are unconditional, the tail is
    const bool ok = t0 + lane < te;         // zeroed by a select (no exec-mask branches)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = wave + 4 * j;
      const float av = op.a(b, mblk * 64 + row, t), xv = op.x(b, nblk * 64 + row, t);
      areg[j] = ok ? av : 0.f;
      xreg[j] = ok ? xv : 0.f;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      As[buf][wave + 4 * j][lane] = areg[j];
      Xs[buf][wave + 4 * j][lane] = xreg[j];
      bsum[j] += areg[j];  // bias gradient = row sums of A, folded into the staging pass
    }
  };

  gload(tb);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int t0 = tb; t0 < te; t0 += WG_T, buf ^= 1) {
    const bool more = t0 + WG_T < te;
    if (more) gload(t0 + WG_T);  // next tile's global loads fly under this tile's MFMAs
#pragma unroll 8
    for (int kk = 0; kk < WG_T / 2; ++kk) {
      const int tc = 2 * kk + (lane >> 5);
      const float av = As[buf][32 * mi + (lane & 31)][tc];
      const float xv = Xs[buf][32 * ni + (lane & 31)][tc];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xv, acc, 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mblk * 64 + 32 * mi + acc_row(r, lane), n = nblk * 64 + 32 * ni + (lane & 31);
    float *dst = op.dw(m, n);
    if (dst) atomicAdd(dst, acc[r]);
  }
  if (want_bias) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float v = wave_sum(bsum[j]);
      const int m = mblk * 64 + wave + 4 * j;
      if (lane == 0) {
        if (bias_part) {
          bias_part[(size_t)blockIdx.x * m_rows64 + m] = v;
        } else {  // scratch too small for this shape: contended but correct
          float *dst = op.db(m);
          if (dst) atomicAdd(dst, v);
        }
      }
    }
  }
}


// ======================================================================
// wgrad2: the same product with 16-byte operand traffic end to end, for ops that
// provide row-vector accessors (a4/x4: four consecutive time steps of one row).
// Block tile 128(m) x 64*NB(n), four waves as 2 x 2, wave tile 64 x 32*NB (2 x NB MFMA
// tiles): per 64 time steps a workgroup moves (128 + 64*NB) x 256 B for 64 x 2*NB
// MFMAs per wave -- half the bytes per MFMA of the 64 x 64 tiling, and no operand is
// fetched by two workgroups of the same time chunk.  LDS rows keep TIME contiguous
// (as in HBM: no transpose), padded to 68 floats; a lane fetches its MFMA operands
// for four k-steps with one ds_read_b128 -- the k index inside a group of 8 time
// steps is permuted (MFMA j pairs t = 8g+j with t = 8g+4+j), which a sum over time
// does not see.  One LDS buffer: the next tile waits in registers during the MFMAs.
// ======================================================================
constexpr int W2_T = 64;       // time per LDS tile
constexpr int W2_LD = 68;      // row pitch: conflict-free ds_read_b128 (4 i mod 64 distinct over 16 rows)
constexpr int W2_CHUNK = 512;  // time per workgroup (512 workgroups at config 2)

// dword-aligned 16-byte global load (the dilation shift t - d is not a multiple of 4 for d < 4)
__device__ __forceinline__ f4 ldg4(const float *p) {
  typedef float v4 __attribute__((ext_vector_type(4), aligned(4)));
  const v4 v = *(const v4 *)p;
  return f4{v.x, v.y, v.z, v.w};
}
// Operand rows of wgrad2 are described by a pointer `p` with p[t] = the row's value at
// absolute time t (dilation / skip-axis shifts folded into p) and a validity range
// [lo, hi) outside which the operand is zero; a row that does not exist has lo = hi = 0
// and any dereferenceable p.  ld4_edge is the masked form for tiles that touch a range
// end: clamped addresses, values selected afterwards (no exec-mask branches).
__device__ __forceinline__ f4 ld4_edge(const float *p, int t, int lo, int hi) {
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // multiply by 0/1 rather than select: a select lets the compiler sink the load into a
    // conditional block, i.e. one exec-mask branch and one vmcnt(0) per element
    const int tt = t + e, tc = max(min(tt, hi - 1), lo);
    v[e] = p[tc] * ((tt >= lo && tt < hi) ? 1.0f : 0.0f);
  }
  return f4{v[0], v[1], v[2], v[3]};
}

// BF3: the products on the bf16 matrix cores -- both operands are read from the fp32 tiles eight consecutive time
// steps per lane (two ds_read_b128) and split into three bf16 planes in registers (bf3.h): per 16 time steps a
// wave splits its 2 + NB row operands (44 vector instructions each) and issues 12 NB bf16 MFMAs instead of 16 NB
// fp32 ones -- 384 NB matrix cycles instead of 1024 NB, with the vector work running beside them.
template <class Op, int NB, bool BF3 = false>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(Op op, int nblk_n, int chunks_per_b,
                                                       float *__restrict__ bias_part, int m_rows_pad,
                                                       float *__restrict__ part, int n_cols_pad) {
  __shared__ __attribute__((aligned(16))) float As[128][W2_LD];
  __shared__ __attribute__((aligned(16))) float Xs[64 * NB][W2_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int mblk = blockIdx.y / nblk_n, nblk = blockIdx.y - mblk * nblk_n;
  const int wm = wave >> 1, wn = wave & 1;
  const int tb = (op.t_begin & ~TILE_ALIGN) + ch * W2_CHUNK, te = min(op.t_end, tb + W2_CHUNK);

  f32x16 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread -> row (tid >> 4) + 16 p, columns 4 (tid & 15) .. +3
  const int srow = tid >> 4, st = 4 * (tid & 15);
  f4 areg[8], xreg[4 * NB];
  float bsum[8];
  unsigned azero = 0, xzero = 0;  // rows of the staged tile that are zero (see gload)
#pragma unroll
  for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
  auto gload = [&](int t0) {
    // one test per tile: does every row of this wave cover the whole tile?  Then sixteen
    // back-to-back 16-byte loads; otherwise the masked form (first/last tiles, rows that
    // start later such as the skip gradient, padding rows)
    // (row pointers are rebuilt per tile behind an optimisation fence: hoisted out of the
    // loop they would occupy 32-48 registers next to 64 accumulators and 64 staging registers)
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    // A rows whose range misses the tile altogether (the skip gradient before t_skip0, the
    // absent dxo rows of the last layer) still take the unmasked path: they load a row that
    // IS valid here and are zeroed at the LDS store (azero, bit p)
    bool inter = t0 + W2_T <= te;
    unsigned zero_bits = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int m = mblk * 128 + 16 * p + srow_q;
      const int lo = op.a_lo(m), hi = op.a_hi(m);
      const bool full = t0 >= lo && t0 + W2_T <= hi, none = t0 + W2_T <= lo || t0 >= hi;
      inter = inter && (full || none);
      if (none) zero_bits |= 1u << p;
    }
    // Op::X_ABSENT_ROWS: X rows may be missing too (a half-used block: 3C rows of the
    // conditioned filter/gate gradient in two 128-row blocks); they borrow the first A row,
    // which such an op guarantees to be a real row, and are zeroed at the LDS store (xzero)
    unsigned xzero_bits = 0;
#pragma unroll
    for (int p = 0; p < 4 * NB; ++p) {
      const int n = nblk * 64 * NB + 16 * p + srow_q;
      const int lo = op.x_lo(n), hi = op.x_hi(n);
      const bool full = t0 >= lo && t0 + W2_T <= hi;
      const bool none = Op::X_ABSENT_ROWS && (t0 + W2_T <= lo || t0 >= hi);
      inter = inter && (full || none);
      if (none) xzero_bits |= 1u << p;
    }
    const int t = t0 + st;
    azero = 0;
    xzero = 0;
    // Row pointers are formed ONE AT A TIME, each right in front of its load, behind a
    // scheduling fence: formed together (8 + 4 NB (+ 4 NB) 64-bit pointers next to 32 NB
    // accumulators and the staging registers) they pushed the NB = 2 kernels well over the
    // 256 registers a two-workgroup CU allows, and the spills landed in this loop.
    if (__all(inter)) {
      azero = zero_bits;
      xzero = xzero_bits;
      // (x row 0 of this block is valid over this tile -- every X row is; its index 0 need not
      // be readable, so the masked path below keeps the row's own pointer)
      const float *x0p = op.x_ptr(b, nblk * 64 * NB + srow_q) + t;
      const float *a0p = op.a_ptr(b, mblk * 128 + srow_q) + t;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float *q = ((zero_bits >> p) & 1u) ? x0p : op.a_ptr(b, mblk * 128 + 16 * p + srow_q) + t;
        areg[p] = ldg4(q);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < 4 * NB; ++p) {
        const bool absent = Op::X_ABSENT_ROWS && ((xzero_bits >> p) & 1u);
        const float *q = absent ? a0p : op.x_ptr(b, nblk * 64 * NB + 16 * p + srow_q) + t;
        xreg[p] = ldg4(q);
        if (Op::X_PRODUCT) {
          const f4 v = ldg4(op.x_ptr2(b, nblk * 64 * NB + 16 * p + srow_q) + t);
          xreg[p] = f4{xreg[p].x * v.x, xreg[p].y * v.y, xreg[p].z * v.z, xreg[p].w * v.w};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int m = mblk * 128 + 16 * p + srow_q;
        areg[p] = ld4_edge(op.a_ptr(b, mblk * 128 + 16 * p + srow_q), t, op.a_lo(m), min(op.a_hi(m), te));
      }
#pragma unroll
      for (int p = 0; p < 4 * NB; ++p) {
        const int n = nblk * 64 * NB + 16 * p + srow_q;
        const int lo = op.x_lo(n), hi = min(op.x_hi(n), te);
        xreg[p] = ld4_edge(op.x_ptr(b, nblk * 64 * NB + 16 * p + srow_q), t, lo, hi);
        if (Op::X_PRODUCT) {
          const f4 v = ld4_edge(op.x_ptr2(b, nblk * 64 * NB + 16 * p + srow_q), t, lo, hi);
          xreg[p] = f4{xreg[p].x * v.x, xreg[p].y * v.y, xreg[p].z * v.z, xreg[p].w * v.w};
        }
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if ((azero >> p) & 1u) areg[p] = f4{0.f, 0.f, 0.f, 0.f};
      *(f4 *)&As[16 * p + srow][st] = areg[p];
      if (Op::HAS_BIAS)  // bias gradient = row sums of A
        bsum[p] += (areg[p].x + areg[p].y) + (areg[p].z + areg[p].w);
    }
#pragma unroll
    for (int p = 0; p < 4 * NB; ++p) {
      if (Op::X_ABSENT_ROWS && ((xzero >> p) & 1u)) xreg[p] = f4{0.f, 0.f, 0.f, 0.f};
      const f4 v = xreg[p];  // Op::xmap: element-wise input transform (leaky-ReLU of the head)
      *(f4 *)&Xs[16 * p + srow][st] = f4{op.xmap(v.x), op.xmap(v.y), op.xmap(v.z), op.xmap(v.w)};
    }
  };

  gload(tb);
  lstore();
  __syncthreads();
  const int h4 = 4 * (lane >> 5), li = lane & 31;
  for (int t0 = tb; t0 < te; t0 += W2_T) {
    const bool more = t0 + W2_T < te;
    if (more) gload(t0 + W2_T);  // next tile's global loads fly under this tile's MFMAs
    // (fence: the scheduler otherwise hoists the row sums of lstore() above the MFMAs and
    // waits for the loads it has just issued)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BF3) {
#pragma unroll
      for (int G = 0; G < W2_T / 16; ++G) {
        u32x4 ah[2], am[2], al[2], xh[NB], xm[NB], xl[NB];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const f4 v0 = *(const f4 *)&As[64 * wm + 32 * mi + li][16 * G + 2 * h4];
          const f4 v1 = *(const f4 *)&As[64 * wm + 32 * mi + li][16 * G + 2 * h4 + 4];
          const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          bf3_split8(v, ah[mi], am[mi], al[mi]);
        }
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
          const f4 v0 = *(const f4 *)&Xs[32 * NB * wn + 32 * ni + li][16 * G + 2 * h4];
          const f4 v1 = *(const f4 *)&Xs[32 * NB * wn + 32 * ni + li][16 * G + 2 * h4 + 4];
          const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          bf3_split8(v, xh[ni], xm[ni], xl[ni]);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NB; ++ni) bf3_mfma6r(acc[mi][ni], ah[mi], am[mi], al[mi], xh[ni], xm[ni], xl[ni]);
      }
    } else
#pragma unroll
    for (int g = 0; g < W2_T / 8; ++g) {
      f4 a[2], x[NB];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f4 *)&As[64 * wm + 32 * mi + li][8 * g + h4];
#pragma unroll
      for (int ni = 0; ni < NB; ++ni) x[ni] = *(const f4 *)&Xs[32 * NB * wn + 32 * ni + li][8 * g + h4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NB; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(a[mi], j), f4_get(x[ni], j),
                                                               acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave has read this tile
    if (more) {
      lstore();
      __syncthreads();
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NB; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mblk * 128 + 64 * wm + 32 * mi + acc_row(r, lane);
        const int n = nblk * 64 * NB + 32 * NB * wn + 32 * ni + li;
        if (part) {  // this workgroup's slab; slab_reduce_kernel adds the slabs up in a fixed order
          part[((size_t)blockIdx.x * m_rows_pad + m) * n_cols_pad + n] = acc[mi][ni][r];
        } else {
          float *dst = op.dw(m, n);
          if (dst) atomicAdd(dst, acc[mi][ni][r]);
        }
      }
  if (Op::HAS_BIAS && nblk == 0 && bias_part) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float v = bsum[p];  // 16 lanes share a row
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if ((tid & 15) == 0) bias_part[(size_t)blockIdx.x * m_rows_pad + mblk * 128 + 16 * p + srow] = v;
    }
  }
}

// dW(m,n) += sum over the workgroups' slabs, in a fixed order (deterministic, and ~10x cheaper
// than the 8 M float atomics the slabs replace at config 2).  Workgroup = 32 elements x RED_SEG
// slab segments (1024 threads: the kernel is a latency chain of strided loads, so every thread
// keeps its whole share of <= 16 loads in flight when the launch has <= 512 slabs); one thread
// per element does the final read-modify-write.
constexpr int RED_SEG = 32;
__device__ __forceinline__ float slab_segment_sum(const float *__restrict__ src, size_t stride, size_t idx, int nparts,
                                                  int seg) {
  const int per = (nparts + RED_SEG - 1) / RED_SEG, p0 = seg * per, p1 = min(nparts, p0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = p0;
  for (; q + 15 < p1; q += 16) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(q + i) * stride + idx];
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      s0 += v[i];
      s1 += v[i + 1];
      s2 += v[i + 2];
      s3 += v[i + 3];
    }
  }
  for (; q + 3 < p1; q += 4) {
    s0 += src[(size_t)q * stride + idx];
    s1 += src[(size_t)(q + 1) * stride + idx];
    s2 += src[(size_t)(q + 2) * stride + idx];
    s3 += src[(size_t)(q + 3) * stride + idx];
  }
  for (; q < p1; ++q) s0 += src[(size_t)q * stride + idx];
  return (s0 + s1) + (s2 + s3);
}
template <class Op>
__global__ __launch_bounds__(32 * RED_SEG) void slab_reduce_kernel(Op op, const float *__restrict__ part,
                                                                    int nparts, int m_rows_pad, int n_cols_pad) {
  __shared__ float red[RED_SEG][32];
  const int e = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const size_t mn = (size_t)m_rows_pad * n_cols_pad;
  const size_t idx = (size_t)blockIdx.x * 32 + e;
  red[seg][e] = slab_segment_sum(part, mn, idx, nparts, seg);
  __syncthreads();
  if (seg == 0) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < RED_SEG; ++k) t += red[k][e];
    const int m = (int)(idx / n_cols_pad), n = (int)(idx - (size_t)m * n_cols_pad);
    float *dst = op.dw(m, n);
    if (dst) *dst += t;
  }
}

// one wave per bias word: lane-strided partial sums in a fixed order, then a wave sum
template <class Op>
__global__ void bias_reduce_kernel(Op op, const float *__restrict__ bias_part, int nparts,
                                   int m_rows64) {
  const int m = blockIdx.x, lane = threadIdx.x;
  float *dst = op.db(m);
  if (!dst) return;
  float s = 0.f;
  for (int p = lane; p < nparts; p += 64) s += bias_part[(size_t)p * m_rows64 + m];
  s = wave_sum(s);
  if (lane == 0) atomicAdd(dst, s);  // rows may share a word (transposed-conv taps)
}

static Act act_view(float *p, int batch, int ch, int ld) {
  Act a;
  a.p = p;
  a.sb = (long long)ch * ld;
  a.ld = ld;
  return a;
}

template <class Op>
static void launch_gemm(const Op &op, int m_rows, int batch, hipStream_t s) {
  const int nt = op.t_end - op.t_begin;
  if (nt <= 0 || batch <= 0) return;
  dim3 grid((nt + 255) / 256, (m_rows + 63) / 64, batch);
  hipLaunchKernelGGL(gemm_wx_kernel<Op>, grid, dim3(256), 0, s, op);
}

// bias_scratch: >= chunks*batch*64*ceil(m_rows/64) floats, or NULL when the op has no bias
template <class Op>
static void launch_gemm_staged(const Op &op, int m_rows, int batch, hipStream_t s) {
  const int nt = op.t_end - (op.t_begin & ~TILE_ALIGN);
  if (op.t_end <= op.t_begin || batch <= 0) return;
  dim3 grid((nt + 255) / 256, (m_rows + 63) / 64, batch);
  hipLaunchKernelGGL(gemm_wx_staged_kernel<Op>, grid, dim3(256), 0, s, op);
}

// `f16`: the fp16-operand / fp32-accumulate form (inference: mvn_forward_f16)
template <class Op>
static void launch_gemm_staged(const Op &op, int m_rows, int batch, hipStream_t s, bool f16) {
  if (!f16) return launch_gemm_staged(op, m_rows, batch, s);
  const int nt = op.t_end - (op.t_begin & ~TILE_ALIGN);
  if (op.t_end <= op.t_begin || batch <= 0) return;
  dim3 grid((nt + 255) / 256, (m_rows + 63) / 64, batch);
  hipLaunchKernelGGL(gemm_wx_staged_f16_kernel<Op>, grid, dim3(256), 0, s, op);
}

template <class Op>
static void launch_wgrad(const Op &op, int m_rows, int n_rows, int batch, float *bias_scratch,
                         hipStream_t s) {
  const int nt = op.t_end - op.t_begin;
  if (nt <= 0 || batch <= 0) return;
  const int chunks = (nt + WG_CHUNK - 1) / WG_CHUNK;
  const int mb = (m_rows + 63) / 64, nb = (n_rows + 63) / 64;
  dim3 grid(chunks * batch, mb * nb);
  hipLaunchKernelGGL(wgrad_kernel<Op>, grid, dim3(256), 0, s, op, nb, chunks, bias_scratch, mb * 64);
  if (bias_scratch)
    hipLaunchKernelGGL(bias_reduce_kernel<Op>, dim3(mb * 64), dim3(64), 0, s, op, bias_scratch,
                       chunks * batch, mb * 64);
}

// bias_scratch: >= chunks*batch*128*ceil(m_rows/128) floats when the op has biases, else NULL
// (wgrad2 has no atomic bias path: an op with biases must be given the scratch).
// slab_scratch (slab_floats long): room for one m x n slab per workgroup; when it is too
// small the tiles are combined with float atomics instead.  Op::dw must map distinct (m,n)
// to distinct words for the slab path.
template <int NB, class Op>
static void launch_wgrad2(const Op &op, int m_rows, int n_rows, int batch, float *bias_scratch,
                          float *slab_scratch, size_t slab_floats, hipStream_t s) {
  const int nt = op.t_end - (op.t_begin & ~TILE_ALIGN);
  if (op.t_end <= op.t_begin || batch <= 0) return;
  const int chunks = (nt + W2_CHUNK - 1) / W2_CHUNK;
  const int mb = (m_rows + 127) / 128, nb = (n_rows + 64 * NB - 1) / (64 * NB);
  const int mpad = mb * 128, npad = nb * 64 * NB;
  const size_t need = (size_t)chunks * batch * mpad * npad;
  float *part = (slab_scratch && need <= slab_floats) ? slab_scratch : nullptr;
  dim3 grid(chunks * batch, mb * nb);
  // MOVENET_HIP_WGRAD_MFMA=f32 keeps the fp32-MFMA form (A/B, tests; common.h: Switches)
  if (switches().wgrad_f32)
    hipLaunchKernelGGL((wgrad2_kernel<Op, NB, false>), grid, dim3(256), 0, s, op, nb, chunks, bias_scratch,
                       mpad, part, npad);
  else
    hipLaunchKernelGGL((wgrad2_kernel<Op, NB, true>), grid, dim3(256), 0, s, op, nb, chunks, bias_scratch,
                       mpad, part, npad);
  if (part)
    hipLaunchKernelGGL(slab_reduce_kernel<Op>, dim3(mpad * npad / 32), dim3(32 * RED_SEG), 0, s, op, part,
                       chunks * batch, mpad, npad);
  if (bias_scratch)
    hipLaunchKernelGGL(bias_reduce_kernel<Op>, dim3(mpad), dim3(64), 0, s, op, bias_scratch,
                       chunks * batch, mpad);
}

}  // namespace mvn
