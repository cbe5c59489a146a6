// Backward of one gated residual layer, first half, as ONE kernel (C = K = 64, audio only):
//   dz   = Wr^T dxo + Ws^T dskip                       (B3 of sequence.hip, reference arithmetic
//   df   = dz sg (1 - th^2),  dg = dz th sg (1 - sg)    movenet/modules.py:79-93 differentiated)
//   dWr += dxo z^T,  dWs += dskip z^T,  dbr += sum_t dxo,  dbs += sum_t dskip     (z = th sg)
//
// The two-kernel form (gemm_wx_staged_kernel<DzOp> beside wgrad2_kernel<WgRsOp, 1> on a second
// stream) reads dxo, dskip, tanh and sigmoid TWICE: 370 + 306 MB per layer at config 2, both
// kernels at 3.6-3.8 TB/s of HBM traffic and 20-30 % matrix-core utilisation
// (profiles/r02_pmc_summary.json) -- bound by the bytes.  Here a workgroup stages a tile
// [dxo; dskip] (128 rows x 64 time steps), tanh and sigmoid (64 x 64 each) ONCE and runs both
// products on it:
//   * dz in TRANSPOSED form, D'[t][c] = sum_o drs[o][t] W[o][c]: the A operand is the staged
//     tile read along t (one ds_read_b32 per MFMA), the B operand W lives in REGISTERS for the
//     whole launch (wave (wt, wc) owns a 32 x 32 block of D': 64 values per lane);
//   * the weight gradient exactly as wgrad2_kernel does it (row-major tiles of pitch 68, four
//     k-steps per ds_read_b128), z formed on the fly from the tanh and sigmoid tiles;
//   * the gate derivative in registers (a lane holds 4 x 4 consecutive t of one channel), df | dg
//     through the staging tile as whole-row float4 stores.
// Per-workgroup partial tiles and bias sums leave in wgrad2's slab format and are summed in a fixed
// order (deterministic): reduce_layer64_kernel adds up both halves' slabs of a layer in one launch
// behind the second half (reduce_rs64_kernel when the second half runs in its generic form).
// Tiles start at multiples of 32 columns (TILE_ALIGN); interior tiles use raw buffer accesses
// (fb_load16 / fb_store16: no 64-bit address arithmetic per access).
#pragma once
#include "common.h"
#include "gemm_family.h"

namespace mvn {

struct FusedBwdAArgs {
  int t_begin, t_end, t_skip0, t_base;  // t_begin = A_{l+1}: the layer's outputs cover [t_begin, t_end)
  const float *wr, *ws;                 // (64 out, 64 in) each
  Act dxo, dskip, th, sg, dfg;          // dxo.p == NULL: last layer (its residual output is unused)
};

constexpr int FB_C = 64;

static int fb_device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 256;
  return cus;
}
// Chunks per sequence so that the launch is ONE round of `per_cu` workgroups per CU, a chunk
// being a whole number of 64-step tiles (fixed 512-step chunks gave 416-512 workgroups whatever
// the layer's length: the shorter late layers took as long as the first).
// Tiles start at multiples of 32 columns of the absolute time axis: the 256-byte row segments of a
// tile are two whole cache lines (first half 110 -> 103 us per layer, second 172.5 -> 169.8).
// (Non-temporal loads of the saved activations changed nothing.)

static void fb_chunks(int nt, int batch, int per_cu, int *chunks, int *chunk_t, int tile = W2_T) {
  const int tiles = (nt + tile - 1) / tile;
  const int want = std::max(1, per_cu * fb_device_cus() / std::max(batch, 1));
  const int chunk_tiles = std::max(1, (tiles + want - 1) / want);
  *chunks = (tiles + chunk_tiles - 1) / chunk_tiles;
  *chunk_t = chunk_tiles * tile;
}

// 16-byte raw buffer accesses: (resource of a tensor and sequence) + (per-lane byte offset, formed
// once per workgroup) + (scalar byte offset of the row block and tile).  As 64-bit pointers the
// 14 loads of a tile cost ~100 vector instructions of address arithmetic (v_mad_i64_i32,
// v_lshl_add_u64), which in fp32 come straight out of the matrix cores' time.
constexpr int FB_RSRC = 0x00020000;  // raw buffer, 32-bit data format (gfx9)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fb_rsrc(const float *p) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7FFFFFFF, FB_RSRC);
}
__device__ __forceinline__ f4 fb_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ void fb_store16(f4 x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)},
                                         r, voff, soff, 0);
}

__global__ __launch_bounds__(256, 2) void bwd_dz_wgrs64_kernel(FusedBwdAArgs a, int chunks_per_b, int chunk_t,
                                                              float *__restrict__ bias_part,
                                                              float *__restrict__ part) {
  constexpr int C = FB_C, LD = W2_LD, TT = W2_T;
  __shared__ __attribute__((aligned(16))) float As[2 * C][LD];  // dxo rows | dskip rows; later df | dg
  __shared__ __attribute__((aligned(16))) float Th[C][LD];
  __shared__ __attribute__((aligned(16))) float Sg[C][LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5, h4 = 4 * lh;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_begin, a.t_skip0);
  const bool has_dxo = a.dxo.p != nullptr;

  // ---- B operand of the dz product: W[o][32 wc + li] for o = 2 kk + lh, in registers
  const int wt = wave >> 1, wc = wave & 1;  // dz block: t in [32 wt, +32), c in [32 wc, +32)
  float wreg[C];
#pragma unroll
  for (int kk = 0; kk < C; ++kk) {
    const int o = 2 * kk + lh;
    wreg[kk] = o < C ? (has_dxo ? a.wr[(size_t)o * C + 32 * wc + li] : 0.f)
                     : a.ws[(size_t)(o - C) * C + 32 * wc + li];
  }
  // weight-gradient block of this wave (as wgrad2_kernel<.., 1>): rows [64 wm, +64), cols [32 wn, +32)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;

  // ---- staging: thread -> rows (tid >> 4) + 16 p, columns 4 (tid & 15) .. +3
  const int srow = tid >> 4, st = 4 * (tid & 15);
  f4 areg[8], treg[4], sreg[4];
  float bsum[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
  unsigned azero = 0;
  const __amdgpu_buffer_rsrc_t dxob = fb_rsrc(a.dxo.p + (size_t)b * a.dxo.sb);
  const __amdgpu_buffer_rsrc_t dskb = fb_rsrc(a.dskip.p + (size_t)b * a.dskip.sb);
  const __amdgpu_buffer_rsrc_t thb = fb_rsrc(a.th.p + (size_t)b * a.th.sb);
  const __amdgpu_buffer_rsrc_t sgb = fb_rsrc(a.sg.p + (size_t)b * a.sg.sb);
  const __amdgpu_buffer_rsrc_t dfgb = fb_rsrc(a.dfg.p + (size_t)b * a.dfg.sb);
  const int vo_dxo = 4 * (srow * a.dxo.ld + st), vo_dsk = 4 * (srow * a.dskip.ld + st);
  const int vo_th = 4 * (srow * a.th.ld + st), vo_sg = 4 * (srow * a.sg.ld + st), vo_dfg = 4 * (srow * a.dfg.ld + st);
  auto gload = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
    // the tile lies inside every live row's range (one test per tile, workgroup-uniform):
    // sixteen back-to-back 16-byte loads; rows whose range misses the tile altogether (dskip
    // before t_skip0, the absent dxo of the last layer) load a row that IS valid and are zeroed
    // at the LDS store
    const bool x_full = t0 >= a.t_begin && t0 + TT <= te;
    const bool s_full = t0 >= skip_lo && t0 + TT <= te, s_none = t0 + TT <= skip_lo;
    azero = (has_dxo ? 0u : 0x0Fu) | (s_none ? 0xF0u : 0u);
    if (x_full && (s_full || s_none)) {
      // raw buffer loads (fb_load16): per-lane offset of (row srow, column st) + scalar offset of the
      // row block and the tile
      int ld_dxo = 4 * a.dxo.ld, ld_dsk = 4 * a.dskip.ld, ld_th = 4 * a.th.ld, ld_sg = 4 * a.sg.ld;
      asm volatile("" : "+s"(ld_dxo), "+s"(ld_dsk), "+s"(ld_th), "+s"(ld_sg));
      const int c4 = 4 * t0;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if ((azero >> p) & 1u)
          areg[p] = fb_load16(thb, vo_th, c4);
        else if (p < 4)
          areg[p] = fb_load16(dxob, vo_dxo, 16 * p * ld_dxo + c4);
        else
          areg[p] = fb_load16(dskb, vo_dsk, 16 * (p - 4) * ld_dsk + c4 - 4 * a.t_base);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        treg[p] = fb_load16(thb, vo_th, 16 * p * ld_th + c4);
        sreg[p] = fb_load16(sgb, vo_sg, 16 * p * ld_sg + c4);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (p < 4)
          areg[p] = has_dxo ? ld4_edge(a.dxo.at(b, 16 * p + srow_q, 0), t, a.t_begin, te) : kZero4;
        else
          areg[p] = ld4_edge(a.dskip.at(b, 16 * (p - 4) + srow_q, 0) - a.t_base, t, skip_lo, te);
      }
      azero = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        treg[p] = ld4_edge(a.th.at(b, 16 * p + srow_q, 0), t, a.t_begin, te);
        sreg[p] = ld4_edge(a.sg.at(b, 16 * p + srow_q, 0), t, a.t_begin, te);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      if ((azero >> p) & 1u) areg[p] = kZero4;
      *(f4 *)&As[16 * p + srow][st] = areg[p];
      bsum[p] += (areg[p].x + areg[p].y) + (areg[p].z + areg[p].w);  // bias gradient = row sums
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(f4 *)&Th[16 * p + srow][st] = treg[p];
      *(f4 *)&Sg[16 * p + srow][st] = sreg[p];
    }
  };

  gload(tb);
  lstore();
  __syncthreads();
  for (int t0 = tb; t0 < te; t0 += TT) {
    const bool more = t0 + TT < te;
    if (more) gload(t0 + TT);  // the next tile's loads fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    // ---- dz' (32 t x 32 c) = sum over the 128 rows of the tile
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < C; ++kk)
      accd = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + lh][32 * wt + li], wreg[kk], accd, 0, 0, 0);
    // ---- weight gradient: (64 x 32) += A (64 x 64 t) z^T
    // (r3, measured: this product on the bf16 matrix cores as in the second half -- 115.9 against 100.4 us: the
    // planes do not fit beside this kernel's 248 registers, 13 spill in the tile loop.  It stays on fp32 MFMAs.)
    // (... and the dz product above with its K split between wave pairs as in the second half's dx: 111.3 against
    // 104.9 us -- this kernel is bound by its barriers and its bytes, a third barrier per tile costs more than the
    // matrix time it saves.)
#pragma unroll
    for (int g = 0; g < TT / 8; ++g) {
      f4 av[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) av[mi] = *(const f4 *)&As[64 * wm + 32 * mi + li][8 * g + h4];
      const f4 tv = *(const f4 *)&Th[32 * wn + li][8 * g + h4], sv = *(const f4 *)&Sg[32 * wn + li][8 * g + h4];
      const f4 zv = f4{tv.x * sv.x, tv.y * sv.y, tv.z * sv.z, tv.w * sv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          accw[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(av[mi], j), f4_get(zv, j), accw[mi], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // The next tile's loads have had the whole MFMA section to land: wait for them HERE.  On
    // gfx9 stores count in vmcnt too, and the number of stores below depends on the path (edge
    // tiles), so the compiler's own wait in front of lstore() was vmcnt(0) -- behind this tile's
    // global stores, i.e. one full store round trip per tile with the matrix cores idle.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
    __syncthreads();  // every wave has read the tile: As becomes the df | dg staging tile
    // ---- gate derivative: this lane holds dz of channel 32 wc + li at t = 32 wt + 8 q + 4 lh + e
    // (after the barrier, straight into the staging tile: held in registers across it, the 32
    // values spilled next to the next tile's 64 staging registers.  Stored to global memory straight
    // from this layout -- 16 bytes per lane and row, no staging tile, two barriers per tile instead of
    // three -- the kernel took 104.6 against 98.8 us: the write path wants whole cache lines.)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tc = 32 * wt + 8 * q + h4;
      const f4 tv = *(const f4 *)&Th[32 * wc + li][tc], sv = *(const f4 *)&Sg[32 * wc + li][tc];
      const f4 dz = f4{accd[4 * q], accd[4 * q + 1], accd[4 * q + 2], accd[4 * q + 3]};
      *(f4 *)&As[32 * wc + li][tc] =
          f4{dz.x * sv.x * (1.0f - tv.x * tv.x), dz.y * sv.y * (1.0f - tv.y * tv.y),
             dz.z * sv.z * (1.0f - tv.z * tv.z), dz.w * sv.w * (1.0f - tv.w * tv.w)};
      *(f4 *)&As[C + 32 * wc + li][tc] =
          f4{dz.x * tv.x * sv.x * (1.0f - sv.x), dz.y * tv.y * sv.y * (1.0f - sv.y),
             dz.z * tv.z * sv.z * (1.0f - sv.z), dz.w * tv.w * sv.w * (1.0f - sv.w)};
    }
    __syncthreads();
    {
      // whole-row float4 stores: rows 16 p + srow, columns t0 + st .. +3, inside [t_begin, te)
      const int t = t0 + st;
      float *base = a.dfg.p + (size_t)b * a.dfg.sb + t;
      if (t >= a.t_begin && t + 3 < te) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
          fb_store16(*(const f4 *)&As[16 * p + srow][st], dfgb, vo_dfg, 4 * (16 * p * a.dfg.ld + t0));
      } else {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= a.t_begin && t + e < te) base[(size_t)(16 * p + srow) * a.dfg.ld + e] = As[16 * p + srow][st + e];
      }
    }
    // (no barrier here: a thread's lstore() overwrites exactly the staging elements the same thread
    // has just read for its global stores, and Th / Sg were last read before the barrier above)
    if (more) {
      lstore();
      __syncthreads();
    }
  }
  // ---- this workgroup's slab and bias partial sums (wgrad2_kernel's format, m_rows_pad 128, n_cols_pad 64)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 64 * wm + 32 * mi + acc_row(r, lane), n = 32 * wn + li;
      part[((size_t)blockIdx.x * 128 + m) * 64 + n] = accw[mi][r];
    }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    float v = bsum[p];  // 16 lanes share a row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if ((tid & 15) == 0) bias_part[(size_t)blockIdx.x * 128 + 16 * p + srow] = v;
  }
}

// The first half's two reductions in ONE launch (slab_reduce_kernel's scheme: 32 elements x RED_SEG
// slab segments per workgroup, fixed order): workgroups [0, 256) take the 128 x 64 weight-gradient
// elements, [256, 260) the 128 bias sums.
template <class Op>
__global__ __launch_bounds__(32 * RED_SEG) void reduce_rs64_kernel(Op op, const float *__restrict__ part,
                                                                    const float *__restrict__ bias_part, int nparts) {
  __shared__ float red[RED_SEG][32];
  const int e = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const bool is_bias = blockIdx.x >= 256;
  const size_t stride = is_bias ? 128 : (size_t)128 * 64;
  const size_t idx = is_bias ? (size_t)(blockIdx.x - 256) * 32 + e : (size_t)blockIdx.x * 32 + e;
  red[seg][e] = slab_segment_sum(is_bias ? bias_part : part, stride, idx, nparts, seg);
  __syncthreads();
  if (seg == 0) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < RED_SEG; ++k) t += red[k][e];
    float *dst = is_bias ? op.db((int)idx) : op.dw((int)(idx >> 6), (int)(idx & 63));
    if (dst) *dst += t;
  }
}

// `op` describes where the gradients go (WgRsOp::dw / db); slab: chunks * batch * 128 * 64 floats,
// bias_scratch: chunks * batch * 128 floats.  False when the scratch is too small (the caller
// then runs the two-kernel form).
// The first half's reduction, left for the second half's launch to run together with its own: one
// reduce launch per layer instead of two (each costs ~5 us, most of it fixed).  `slab_used` floats
// at the front of the slab scratch stay occupied until then.
template <class WgOp>
struct PendingRsReduce {
  bool on = false;
  WgOp op;
  const float *part = nullptr, *bias = nullptr;
  int nparts = 0;
  size_t slab_used = 0;
};

template <class WgOp>
static bool launch_bwd_dz_wgrs64(const FusedBwdAArgs &a, const WgOp &op, int batch, float *bias_scratch,
                                 size_t bias_floats, float *slab, size_t slab_floats, hipStream_t s,
                                 PendingRsReduce<WgOp> *defer = nullptr) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return true;
  int chunks, chunk_t;
  fb_chunks(nt, batch, 2, &chunks, &chunk_t);
  const size_t need = (size_t)chunks * batch * 128 * 64;
  // (r3: the bias partials -- 128 per workgroup -- are checked too: the scratch was sized for wgrad2's 512-column
  // chunks, and short sequences in small batches launch more workgroups than that has room for)
  if (!bias_scratch || !slab || need > slab_floats || (size_t)chunks * batch * 128 > bias_floats) return false;
  hipLaunchKernelGGL(bwd_dz_wgrs64_kernel, dim3(chunks * batch), dim3(256), 0, s, a, chunks, chunk_t, bias_scratch, slab);
  if (defer) {
    defer->on = true;
    defer->op = op;
    defer->part = slab;
    defer->bias = bias_scratch;
    defer->nparts = chunks * batch;
    defer->slab_used = need;
  } else {
    hipLaunchKernelGGL(reduce_rs64_kernel<WgOp>, dim3(256 + 4), dim3(32 * RED_SEG), 0, s, op, slab, bias_scratch, chunks * batch);
  }
  return true;
}
template <class WgOp>
static void flush_pending_rs(PendingRsReduce<WgOp> &p, hipStream_t s) {
  if (!p.on) return;
  hipLaunchKernelGGL(reduce_rs64_kernel<WgOp>, dim3(256 + 4), dim3(32 * RED_SEG), 0, s, p.op, p.part, p.bias, p.nparts);
  p.on = false;
}

// ----------------------------------------------------------------------------------------
// Conditioned layers (BASELINE configs[2]/[3]; the build definition of movenet/modules.py:58-63,
// :75-77): f | g += Wc ctx + bc.  Its backward, from the SAME dfg tile the first half wrote:
//   dctx[t]  += Wcf^T df[t] + Wcg^T dg[t]                       (accumulated over the layers)
//   dWcf|dWcg += dfg ctx^T,   dbcf|dbcg += sum_t dfg
// This is the first half's shape with other operands -- A = dfg (128 rows), "z" = ctx (64 rows),
// W = [Wcf; Wcg] (128 x 64) in registers for the transposed product, weight gradient as wgrad2 --
// so it is the same structure: one staging pass per 64-step tile, the next tile's loads in
// registers under the MFMAs, slabs in wgrad2's format (reduce_rs64_kernel<WgCtxOp>).  The
// accumulated dctx tile is staged with the operands and updated IN LDS by the lane that owns the
// element (no staging copy of the product: two barriers per tile, not three).
// Replaces wgrad2<WgFgOpT<true>, 2> (192 operand rows: 207 us per layer at config 3) +
// gemm_wx_staged<DctxOp> (68 us) and lets the audio taps run in bwd_dx_wgfg64_kernel (the generic
// Dx took 138 us): r2 434 us per layer of second-half work.
// ----------------------------------------------------------------------------------------
struct FusedBwdCArgs {
  int t_begin, t_end;       // the layer's outputs cover [t_begin = A_{l+1}, t_end)
  const float *wcf, *wcg;   // (64 out, 64 in) each
  Act dfg, ctx, dctx;       // dfg (B, 128, Tp) read; ctx (B, 64, ld) read; dctx (B, 64, Tp) accumulated in place
};

struct WgCtxOp {  // where the context-conv gradients go (rows: filter | gate)
  float *dwcf, *dwcg, *dbcf, *dbcg;
  __device__ __forceinline__ float *dw(int m, int n) const {
    if (m >= 2 * FB_C || n >= FB_C) return nullptr;
    return (m < FB_C ? dwcf + (size_t)m * FB_C : dwcg + (size_t)(m - FB_C) * FB_C) + n;
  }
  __device__ __forceinline__ float *db(int m) const {
    if (m >= 2 * FB_C) return nullptr;
    return m < FB_C ? dbcf + m : dbcg + (m - FB_C);
  }
};

__global__ __launch_bounds__(256, 2) void bwd_dctx_wgctx64_kernel(FusedBwdCArgs a, int chunks_per_b, int chunk_t,
                                                                 float *__restrict__ bias_part,
                                                                 float *__restrict__ part) {
  constexpr int C = FB_C, LD = W2_LD, TT = W2_T;
  __shared__ __attribute__((aligned(16))) float As[2 * C][LD];  // df rows | dg rows
  __shared__ __attribute__((aligned(16))) float Cx[C][LD];      // the context
  __shared__ __attribute__((aligned(16))) float Dc[C][LD];      // dctx so far; updated in place
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5, h4 = 4 * lh;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);

  // ---- B operand of the dctx product: W[o][32 wc + li] for o = 2 kk + lh, in registers
  const int wt = wave >> 1, wc = wave & 1;  // block: t in [32 wt, +32), c in [32 wc, +32)
  float wreg[C];
#pragma unroll
  for (int kk = 0; kk < C; ++kk) {
    const int o = 2 * kk + lh;
    wreg[kk] = o < C ? a.wcf[(size_t)o * C + 32 * wc + li] : a.wcg[(size_t)(o - C) * C + 32 * wc + li];
  }
  // weight-gradient block of this wave: rows [64 wm, +64), cols [32 wn, +32)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;

  // ---- staging: thread -> rows (tid >> 4) + 16 p, columns 4 (tid & 15) .. +3
  const int srow = tid >> 4, st = 4 * (tid & 15);
  f4 areg[8], creg[4], dreg[4];
  float bsum[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
  const __amdgpu_buffer_rsrc_t dfgb = fb_rsrc(a.dfg.p + (size_t)b * a.dfg.sb);
  const __amdgpu_buffer_rsrc_t ctxb = fb_rsrc(a.ctx.p + (size_t)b * a.ctx.sb);
  const __amdgpu_buffer_rsrc_t dcb = fb_rsrc(a.dctx.p + (size_t)b * a.dctx.sb);
  const int vo_dfg = 4 * (srow * a.dfg.ld + st), vo_ctx = 4 * (srow * a.ctx.ld + st), vo_dc = 4 * (srow * a.dctx.ld + st);
  auto gload = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
    if (t0 >= a.t_begin && t0 + TT <= te) {  // interior tile: raw buffer loads (see fb_load16)
      int ld_dfg = 4 * a.dfg.ld, ld_ctx = 4 * a.ctx.ld, ld_dc = 4 * a.dctx.ld;
      asm volatile("" : "+s"(ld_dfg), "+s"(ld_ctx), "+s"(ld_dc));
      const int c4 = 4 * t0;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        areg[p] = fb_load16(dfgb, vo_dfg, 16 * p * ld_dfg + c4);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        creg[p] = fb_load16(ctxb, vo_ctx, 16 * p * ld_ctx + c4);
        dreg[p] = fb_load16(dcb, vo_dc, 16 * p * ld_dc + c4);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) areg[p] = ld4_edge(a.dfg.at(b, 16 * p + srow_q, 0), t, a.t_begin, te);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        creg[p] = ld4_edge(a.ctx.at(b, 16 * p + srow_q, 0), t, a.t_begin, te);
        dreg[p] = ld4_edge(a.dctx.at(b, 16 * p + srow_q, 0), t, a.t_begin, te);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      *(f4 *)&As[16 * p + srow][st] = areg[p];
      bsum[p] += (areg[p].x + areg[p].y) + (areg[p].z + areg[p].w);  // bias gradient = row sums
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(f4 *)&Cx[16 * p + srow][st] = creg[p];
      *(f4 *)&Dc[16 * p + srow][st] = dreg[p];
    }
  };

  gload(tb);
  lstore();
  __syncthreads();
  for (int t0 = tb; t0 < te; t0 += TT) {
    const bool more = t0 + TT < te;
    if (more) gload(t0 + TT);  // the next tile's loads fly under this tile's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    // ---- dctx' (32 t x 32 c) = sum over the 128 rows of the tile
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < C; ++kk)
      accd = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + lh][32 * wt + li], wreg[kk], accd, 0, 0, 0);
    // ---- weight gradient: (64 x 32) += dfg (64 x 64 t) ctx^T
#pragma unroll
    for (int G = 0; G < TT / 16; ++G) {
      u32x4 ah[2], am[2], al[2], ch, cm, cl;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f4 v0 = *(const f4 *)&As[64 * wm + 32 * mi + li][16 * G + 2 * h4];
        const f4 v1 = *(const f4 *)&As[64 * wm + 32 * mi + li][16 * G + 2 * h4 + 4];
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf3_split8(v, ah[mi], am[mi], al[mi]);
      }
      {
        const f4 v0 = *(const f4 *)&Cx[32 * wn + li][16 * G + 2 * h4], v1 = *(const f4 *)&Cx[32 * wn + li][16 * G + 2 * h4 + 4];
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf3_split8(v, ch, cm, cl);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) bf3_mfma6r(accw[mi], ah[mi], am[mi], al[mi], ch, cm, cl);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- this lane owns dctx of channel 32 wc + li at t = 32 wt + 8 q + 4 lh + e: add in LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tc = 32 * wt + 8 * q + h4;
      const f4 o = *(const f4 *)&Dc[32 * wc + li][tc];
      *(f4 *)&Dc[32 * wc + li][tc] =
          f4{o.x + accd[4 * q], o.y + accd[4 * q + 1], o.z + accd[4 * q + 2], o.w + accd[4 * q + 3]};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next tile's loads, ahead of this tile's stores
    __syncthreads();  // the dctx tile is complete AND every wave has read As / Cx
    {
      // whole-row float4 stores: rows 16 p + srow, columns t0 + st .. +3, inside [t_begin, te)
      const int t = t0 + st;
      float *base = a.dctx.p + (size_t)b * a.dctx.sb + t;
      if (t >= a.t_begin && t + 3 < te) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          fb_store16(*(const f4 *)&Dc[16 * p + srow][st], dcb, vo_dc, 4 * (16 * p * a.dctx.ld + t0));
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= a.t_begin && t + e < te) base[(size_t)(16 * p + srow) * a.dctx.ld + e] = Dc[16 * p + srow][st + e];
      }
    }
    // (a thread's lstore() overwrites exactly the Dc elements the same thread has just read for its
    // global stores; As / Cx were last read before the barrier above)
    if (more) {
      lstore();
      __syncthreads();
    }
  }
  // ---- this workgroup's slab and bias partial sums (wgrad2_kernel's format, m_rows_pad 128, n_cols_pad 64)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 64 * wm + 32 * mi + acc_row(r, lane), n = 32 * wn + li;
      part[((size_t)blockIdx.x * 128 + m) * 64 + n] = accw[mi][r];
    }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    float v = bsum[p];  // 16 lanes share a row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if ((tid & 15) == 0) bias_part[(size_t)blockIdx.x * 128 + 16 * p + srow] = v;
  }
}

// Launch geometry of the conditioned pass (the first half's): false when the scratch cannot hold it.
static bool bwd_dctx_wgctx64_fits(int t_begin, int t_end, int batch, const float *bias_scratch, size_t bias_floats,
                                  const float *slab, size_t slab_floats, int *chunks, int *chunk_t) {
  const int nt = t_end - (t_begin & ~TILE_ALIGN);
  *chunks = 0;
  *chunk_t = 0;
  if (t_end <= t_begin || batch <= 0) return true;
  fb_chunks(nt, batch, 2, chunks, chunk_t);
  return bias_scratch && slab && (size_t)*chunks * batch * 128 * 64 <= slab_floats &&
         (size_t)*chunks * batch * 128 <= bias_floats;
}
static void launch_bwd_dctx_wgctx64(const FusedBwdCArgs &a, const WgCtxOp &op, int batch, float *bias_scratch,
                                    float *slab, int chunks, int chunk_t, hipStream_t s) {
  if (chunks <= 0) return;
  hipLaunchKernelGGL(bwd_dctx_wgctx64_kernel, dim3(chunks * batch), dim3(256), 0, s, a, chunks, chunk_t, bias_scratch, slab);
  hipLaunchKernelGGL(reduce_rs64_kernel<WgCtxOp>, dim3(256 + 4), dim3(32 * RED_SEG), 0, s, op, slab, bias_scratch, chunks * batch);
}

// ----------------------------------------------------------------------------------------
// Second half: the gradient w.r.t. the layer input and the filter/gate weight gradients,
//   dx[u]   = [u >= t_lo] (dxo[u] + W1^T dfg[u]) + [u + d < T] W0^T dfg[u + d]     (B4 of sequence.hip)
//   dWf|dWg[o][c][tap 1] += dfg[o][t] x[c][t],   [tap 0] += dfg[o][t] x[c][t - d]
// The two-kernel form (gemm_wx_staged_kernel<DxOp> beside wgrad2_kernel<WgFgOpT, 2>) reads dfg
// three times (389 + 290 MB per layer).  Here a 512-thread workgroup (one per CU: 136 KB of
// LDS) stages dfg[t] (128 x 64), dfg[t + d] (128 x 64) and [x(t - d); x(t)] (128 x 64) once per
// tile; waves 0-3 take the tap-1 half of dx (K = the 128 rows of dfg[t]), waves 4-7 the tap-0
// half (dfg[t + d]), both in the transposed form of the first half with their 64 weights per
// lane in registers, and every wave owns two of the sixteen 32 x 32 blocks of the weight
// gradient.  The two dx halves meet in the staging tiles (the dfg[t + d] buffer) and leave,
// with dxo added, as whole-row float4 stores.
// Tiles run over u from A_l (& ~3): before t_lo = A_l + d the dfg[t] and x operands are zero
// (masked loads), which is exactly the [u >= t_lo] of the formula.
// ----------------------------------------------------------------------------------------
struct FusedBwdBArgs {
  int t_out0, t_lo, t_end, d;   // outputs cover [t_out0 = A_l, t_end); t_lo = A_{l+1} = A_l + d
  const float *wf, *wg;         // (64 out, 64 in, 2 taps) each
  Act dxo, dfg, xin, dxi;       // dxo.p == NULL: last layer
};

constexpr int FBB_LDS_FLOATS = 4 * 128 * W2_LD;  // dfg[t], dfg[t + d], [x(t - d); x(t)], the two dx halves: 136 KB

__global__ __launch_bounds__(512, 1) void bwd_dx_wgfg64_kernel(FusedBwdBArgs a, int chunks_per_b, int chunk_t,
                                                              float *__restrict__ part) {
  constexpr int C = FB_C, LD = W2_LD, TT = W2_T;
  extern __shared__ __attribute__((aligned(16))) float fbb_lds[];
  float (*As)[LD] = (float (*)[LD])fbb_lds;                    // dfg[t]      (df rows | dg rows)
  float (*A2)[LD] = (float (*)[LD])(fbb_lds + 128 * LD);       // dfg[t + d]
  float (*St)[LD] = (float (*)[LD])(fbb_lds + 3 * 128 * LD);   // the two dx halves [2][64]: a buffer of their own,
                                                               // so that a tile costs two barriers, not four
  float (*Xs)[LD] = (float (*)[LD])(fbb_lds + 2 * 128 * LD);   // x(t - d) rows | x(t) rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5, h4 = 4 * lh;
  const int tb = (a.t_out0 & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const bool has_dxo = a.dxo.p != nullptr;

  // ---- dx on the bf16 matrix cores (r3): wave -> (tap half, K half kh, 32-channel block wc); it forms BOTH 32-step
  // blocks of the tile over its 64 of the tap's 128 rows, so that its weights are 4 k-steps x 3 planes = 48
  // registers (all 128 rows would be 96; as fp32 MFMA operands they were 64) and the two K halves meet in the
  // staging tile.  B operand: W_tap[o][32 wc + li] for o = 64 kh + 16 j + 8 lh + e, e < 8, as planes.
  const int half = wave >> 2, kh = (wave >> 1) & 1, wc = wave & 1;
  u32x4 wp[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = 64 * kh + 16 * j + 8 * lh + e;
      const float *src = o < C ? a.wf : a.wg;
      wv[e] = src[((size_t)(o & (C - 1)) * C + 32 * wc + li) * 2 + (half ? 0 : 1)];
    }
    bf3_split8(wv, wp[j][0], wp[j][1], wp[j][2]);
  }
  // ---- weight gradient: wave -> rows [32 (wave >> 1), +32), columns [64 (wave & 1), +64)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;

  // ---- staging: thread -> rows (tid >> 4) + 32 p, columns 4 (tid & 15) .. +3
  const int srow = tid >> 4, st = 4 * (tid & 15);
  f4 areg[4], a2reg[4], xreg[4], oreg[2];
  // interior tile: every operand row covers it (dfg[t], x, dxo from t_lo; dfg[t + d] up to T - d)
  auto interior = [&](int t0) { return t0 >= a.t_lo && t0 + TT + a.d <= a.t_end && t0 + TT <= te; };
  // The 14 loads of an interior tile are issued in seven parts BETWEEN the MFMA steps of the
  // tile before: issued in one burst at the top of the tile, the vector-memory pipe of the CU
  // took ~8 000 cycles to accept them (12 B/clk), the younger wave of every SIMD sat in that
  // queue while the older one ran its MFMAs alone, and then ran its own alone behind it --
  // the two waves of a SIMD took turns on the matrix core (in-kernel timers: 24 k cycles per
  // tile against 16 k of MFMAs).
  // (interior tiles: raw buffer loads, see fb_load16; the per-lane offsets cover row srow and column st)
  const __amdgpu_buffer_rsrc_t dfgb = fb_rsrc(a.dfg.p + (size_t)b * a.dfg.sb);
  const __amdgpu_buffer_rsrc_t xinb = fb_rsrc(a.xin.p + (size_t)b * a.xin.sb);
  const __amdgpu_buffer_rsrc_t dxob = fb_rsrc(a.dxo.p + (size_t)b * a.dxo.sb);
  const __amdgpu_buffer_rsrc_t dxib = fb_rsrc(a.dxi.p + (size_t)b * a.dxi.sb);
  const int vo_dfg = 4 * (srow * a.dfg.ld + st), vo_xin = 4 * (srow * a.xin.ld + st);
  const int vo_dxo = 4 * (srow * a.dxo.ld + st), vo_dxi = 4 * (srow * a.dxi.ld + st);
  auto gload_part = [&](int t0, int part) {
    int ld_dfg = 4 * a.dfg.ld, ld_xin = 4 * a.xin.ld, ld_dxo = 4 * a.dxo.ld;
    asm volatile("" : "+s"(ld_dfg), "+s"(ld_xin), "+s"(ld_dxo));  // (row-block offsets formed here, not hoisted)
    const int c4 = 4 * t0;
    if (part < 4) {
      areg[part] = fb_load16(dfgb, vo_dfg, 32 * part * ld_dfg + c4);
      a2reg[part] = fb_load16(dfgb, vo_dfg, 32 * part * ld_dfg + c4 + 4 * a.d);
    } else if (part < 6) {
#pragma unroll
      for (int p = 2 * (part - 4); p < 2 * (part - 4) + 2; ++p)  // rows [0, 64): x(t - d); [64, 128): x(t)
        xreg[p] = fb_load16(xinb, vo_xin, ((32 * p) & (C - 1)) * ld_xin + c4 - (p < 2 ? 4 * a.d : 0));
    } else if (part == 6) {
      if (has_dxo) {
#pragma unroll
        for (int p = 0; p < 2; ++p) oreg[p] = fb_load16(dxob, vo_dxo, 32 * p * ld_dxo + c4);
      } else {
        oreg[0] = oreg[1] = kZero4;
      }
    }
  };
  auto gload_edge = [&](int t0) {
    int srow_q = srow;
    asm volatile("" : "+v"(srow_q));
    const int t = t0 + st;
    const int hi = min(a.t_end, te);
    // (the two validity ranges one after the other, with a scheduling fence in between: with both
    // sets of lane masks live at once this cold path spilled two registers)
#pragma unroll
    for (int p = 0; p < 4; ++p) areg[p] = ld4_edge(a.dfg.at(b, 32 * p + srow_q, 0), t, a.t_lo, hi);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = 32 * p + srow_q;
      xreg[p] = ld4_edge(a.xin.at(b, row & (C - 1), 0) - (p < 2 ? a.d : 0), t, a.t_lo, hi);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
      oreg[p] = has_dxo ? ld4_edge(a.dxo.at(b, 32 * p + srow_q, 0), t, a.t_lo, hi) : kZero4;
    __builtin_amdgcn_sched_barrier(0);
    // dfg[u + d] for the outputs u of this tile: u < te, u + d < T
#pragma unroll
    for (int p = 0; p < 4; ++p)
      a2reg[p] = ld4_edge(a.dfg.at(b, 32 * p + srow_q, 0) + a.d, t, a.t_lo - a.d, min(a.t_end - a.d, te));
  };
  auto gload = [&](int t0) {
    if (interior(t0)) {
#pragma unroll
      for (int part = 0; part < 7; ++part) gload_part(t0, part);
    } else {
      gload_edge(t0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(f4 *)&As[32 * p + srow][st] = areg[p];
      *(f4 *)&A2[32 * p + srow][st] = a2reg[p];
      *(f4 *)&Xs[32 * p + srow][st] = xreg[p];
    }
  };

  gload(tb);
  lstore();
  f4 ocur[2] = {oreg[0], oreg[1]};  // dxo of the tile in LDS (the registers take the next tile's)
  __syncthreads();
  for (int t0 = tb; t0 < te; t0 += TT) {
    const bool more = t0 + TT < te;
    const bool spread = more && interior(t0 + TT);
    if (more && !spread) gload_edge(t0 + TT);
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's half of dx' (32 u x 32 c) and its two blocks of the weight gradient, eight
    // steps of 8 + 8 MFMAs.  The LDS operands of step g + 1 are requested BEFORE the MFMAs of
    // step g (two statically named register sets): left to the scheduler every pair of MFMAs
    // waited out the ds_read issued right in front of it.
    f32x16 accd2[2];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd2[ub][r] = 0.f;
    {
      // per 16 time steps: the weight-gradient products of those steps (12 bf16 MFMAs, three operand splits) and
      // one of this wave's four dx k-steps for both 32-step blocks (12 bf16 MFMAs, two operand splits: the operand
      // is read ACROSS the tile's rows, eight ds_read_b32 per block); the next tile's loads in parts in between
      float (*src)[LD] = half ? A2 : As;
#pragma unroll
      for (int G = 0; G < TT / 16; ++G) {
        if (spread) {
          gload_part(t0 + TT, 2 * G);
          gload_part(t0 + TT, 2 * G + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const f4 a0 = *(const f4 *)&As[32 * wm + li][16 * G + 2 * h4], a1 = *(const f4 *)&As[32 * wm + li][16 * G + 2 * h4 + 4];
          const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          u32x4 ah, am, al;
          bf3_split8(av, ah, am, al);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f4 x0 = *(const f4 *)&Xs[64 * wn + 32 * ni + li][16 * G + 2 * h4];
            const f4 x1 = *(const f4 *)&Xs[64 * wn + 32 * ni + li][16 * G + 2 * h4 + 4];
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            u32x4 bh, bm, bl;
            bf3_split8(xv, bh, bm, bl);
            bf3_mfma6r(accw[ni], ah, am, al, bh, bm, bl);
          }
        }
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
          float dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) dv[e] = src[64 * kh + 16 * G + 8 * lh + e][32 * ub + li];
          u32x4 dh, dm, dl;
          bf3_split8(dv, dh, dm, dl);
          bf3_mfma6r(accd2[ub], dh, dm, dl, wp[G][0], wp[G][1], wp[G][2]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next tile's loads, ahead of this tile's stores (see the first half)
    // the two K halves of a tap half meet in the staging tile: kh = 0 stores, kh = 1 adds behind a barrier (a lane
    // owns the same elements in both)
    if (kh == 0) {
#pragma unroll
      for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f4 *)&St[64 * half + 32 * wc + li][32 * ub + 8 * q + h4] =
              f4{accd2[ub][4 * q], accd2[ub][4 * q + 1], accd2[ub][4 * q + 2], accd2[ub][4 * q + 3]};
    }
    __syncthreads();
    if (kh == 1) {
#pragma unroll
      for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 *p4 = (f4 *)&St[64 * half + 32 * wc + li][32 * ub + 8 * q + h4];
          const f4 o4 = *p4;
          *p4 = f4{o4.x + accd2[ub][4 * q], o4.y + accd2[ub][4 * q + 1], o4.z + accd2[ub][4 * q + 2], o4.w + accd2[ub][4 * q + 3]};
        }
    }
    __syncthreads();  // the dx halves are staged AND every wave has read the operand tiles
    {
      // rows srow + 32 p (p < 2), columns t0 + st .. +3 inside [t_out0, te)
      const int t = t0 + st;
      float *base = a.dxi.p + (size_t)b * a.dxi.sb + t;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = 32 * p + srow;
        const f4 v1 = *(const f4 *)&St[row][st], v0 = *(const f4 *)&St[64 + row][st], o = ocur[p];
        const f4 r = f4{(v1.x + v0.x) + o.x, (v1.y + v0.y) + o.y, (v1.z + v0.z) + o.z, (v1.w + v0.w) + o.w};
        float *q = base + (size_t)row * a.dxi.ld;
        if (t >= a.t_out0 && t + 3 < te) {
          fb_store16(r, dxib, vo_dxi, 4 * (32 * p * a.dxi.ld + t0));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t + e >= a.t_out0 && t + e < te) q[e] = f4_get(r, e);
        }
      }
    }
    if (more) {
      lstore();
      ocur[0] = oreg[0];
      ocur[1] = oreg[1];
    }
    __syncthreads();  // the next tile is staged; the staging tile has been read
  }
  // ---- this workgroup's slab (wgrad2_kernel's format, 128 x 128)
  {
    // (indices re-derived from the thread id behind a fence: kept live across the tile loop they
    // were the two registers the kernel spilled)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, wm_e = wave_e >> 1, wn_e = wave_e & 1;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * wm_e + acc_row(r, lane_e), n = 64 * wn_e + 32 * ni + (lane_e & 31);
        part[((size_t)blockIdx.x * 128 + m) * 128 + n] = accw[ni][r];
      }
  }
}

// `op`: where the gradients go (WgFgOpT<false>::dw).  slab: chunks * batch * 128 * 128 floats.
// both halves' reductions of a layer: workgroups [0, 260) as reduce_rs64_kernel, [260, 260 + 512) the
// 128 x 128 filter/gate elements as slab_reduce_kernel
template <class RsOp, class FgOp>
__global__ __launch_bounds__(32 * RED_SEG) void reduce_layer64_kernel(RsOp rs, const float *__restrict__ rs_part,
                                                                       const float *__restrict__ rs_bias, int n_rs, FgOp fg,
                                                                       const float *__restrict__ fg_part, int n_fg) {
  __shared__ float red[RED_SEG][32];
  const int e = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const bool is_fg = blockIdx.x >= 260, is_bias = !is_fg && blockIdx.x >= 256;
  const size_t stride = is_fg ? (size_t)128 * 128 : is_bias ? 128 : (size_t)128 * 64;
  const size_t idx = (size_t)(blockIdx.x - (is_fg ? 260 : is_bias ? 256 : 0)) * 32 + e;
  red[seg][e] = slab_segment_sum(is_fg ? fg_part : is_bias ? rs_bias : rs_part, stride, idx, is_fg ? n_fg : n_rs, seg);
  __syncthreads();
  if (seg == 0) {
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < RED_SEG; ++k) t += red[k][e];
    float *dst = is_fg ? fg.dw((int)(idx >> 7), (int)(idx & 127))
                       : is_bias ? rs.db((int)idx) : rs.dw((int)(idx >> 6), (int)(idx & 63));
    if (dst) *dst += t;
  }
}

template <class WgOp, class RsOp>
static int launch_bwd_dx_wgfg64(const FusedBwdBArgs &a, const WgOp &op, int batch, float *slab,
                                size_t slab_floats, hipStream_t s, bool *done, PendingRsReduce<RsOp> *pend) {
  if (pend && pend->on) {  // the first half's slabs sit at the front of the scratch until they are reduced
    if (pend->slab_used > slab_floats) return MVN_OK;
    slab += pend->slab_used;
    slab_floats -= pend->slab_used;
  }
  *done = false;
  const int nt = a.t_end - (a.t_out0 & ~TILE_ALIGN);
  if (a.t_end <= a.t_out0 || batch <= 0) {
    *done = true;
    return MVN_OK;
  }
  int chunks, chunk_t;
  fb_chunks(nt, batch, 1, &chunks, &chunk_t);
  const size_t need = (size_t)chunks * batch * 128 * 128;
  if (!slab || need > slab_floats) return MVN_OK;
  const void *fn = (const void *)bwd_dx_wgfg64_kernel;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(bwd_dx_wgfg64)");
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_dx_wgfg64_kernel, dim3(chunks * batch), dim3(512), FBB_LDS_FLOATS * sizeof(float), s, a,
                     chunks, chunk_t, slab);
  if (pend && pend->on) {
    hipLaunchKernelGGL((reduce_layer64_kernel<RsOp, WgOp>), dim3(260 + 128 * 128 / 32), dim3(32 * RED_SEG), 0, s, pend->op,
                       pend->part, pend->bias, pend->nparts, op, slab, chunks * batch);
    pend->on = false;
  } else {
    hipLaunchKernelGGL(slab_reduce_kernel<WgOp>, dim3(128 * 128 / 32), dim3(32 * RED_SEG), 0, s, op, slab, chunks * batch, 128, 128);
  }
  *done = true;
  return MVN_OK;
}

}  // namespace mvn
