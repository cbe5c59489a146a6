// fp32 products on the bf16 matrix cores: the strip forward of an audio-only layer, form "bf16 x 3".
//
// CDNA4's fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles for 4096 FLOP) is 1/16 of its bf16 MFMA
// (v_mfma_f32_32x32x16_bf16: <= 32 cycles for 32768 FLOP), and an fp32 MFMA holds the SIMD's vector
// lanes while it runs (scripts/probes/mfma_valu_overlap.hip) where a bf16 MFMA leaves 24 of its 32
// cycles to other instructions.  An fp32 number is EXACTLY the sum of three bf16 numbers
//     x = h + m + l,   h = top16(x),  m = top16(x - h),  l = top16(x - h - m)      (8 + 8 + 8 mantissa bits;
// the two subtractions are exact), so a product of two fp32 numbers is the sum of nine bf16 products, each
// exact in the fp32 accumulator; the six of them with weight >= 2^-16 (hh, hm, mh, mm, hl, lh) carry
// everything above 2^-24 of the product -- what an fp32 multiply keeps.  Six bf16 MFMAs replace the eight fp32
// MFMAs of a 32 x 32 x 16 block: 2.7 - 5 x less matrix time (scripts/probes/mfma_bf16_split.hip: 2048 -> 470
// cycles per block and SIMD, the split of the B operand included), with fp32's exponent range (no scaling,
// no overflow: bf16 has fp32's exponent) and fp32-class error (numpy model in DESIGN 4.3b: max error of a
// 256-term product 4.7e-7 of the output range against 5.7e-7 for fp32 sums in numpy's order).
//
// Same structure as fused_layer64s_kernel<false> (fused_fwd.h): one wave owns a strip of 32 columns, no
// barrier, the strip's input goes from global memory into registers in ACCUMULATOR order and is the B operand;
// gate in registers; z in accumulator order is the second product's B operand.  What changes:
//   * weights in LDS as three bf16 planes, [block][k-step of 16][plane][lane][8 bf16]: one ds_read_b128 per
//     plane feeds an MFMA's A operand (lane -> output row, lane half -> which 8 of the step's 16 inputs); 144 KB;
//   * a k-step takes 8 consecutive input registers of the lane (8 channels of its lane half), split into three
//     packed planes (22 vector instructions) that all four row blocks share: 6 MFMAs each;
//   * per strip 288 bf16 MFMAs instead of 384 fp32 ones, 144 ds_read_b128 instead of 96.
#pragma once
#include "bf3.h"

namespace mvn {

// acc += A (three planes at LDS byte address `wa`, 1 KB apart) x B (three planes): smallest terms first
__device__ __forceinline__ void bf3_mfma6(f32x16 &acc, unsigned wa, const u32x4 &bh, const u32x4 &bm, const u32x4 &bl) {
  typedef __attribute__((address_space(3))) u32x4 lds_u4;
  const u32x4 ah = *(const lds_u4 *)(uintptr_t)wa;
  const u32x4 am = *(const lds_u4 *)(uintptr_t)(wa + 1024u);
  const u32x4 al = *(const lds_u4 *)(uintptr_t)(wa + 2048u);
#if MVN_EXP == 74  // (timing build of fused_bwd_l.h: no MFMAs)
#define BF3_MF(a_, b_) acc[0] += __uint_as_float(a_[0] ^ b_[0])
#else
#define BF3_MF(a_, b_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc, 0, 0, 0)
#endif
  BF3_MF(al, bh);
  BF3_MF(ah, bl);
  BF3_MF(am, bm);
  BF3_MF(am, bh);
  BF3_MF(ah, bm);
  BF3_MF(ah, bh);
#undef BF3_MF
}

// acc[0..3] += W (4 row blocks, NKS k-steps, three planes each in LDS) x B, software-pipelined: step (ks, blk)
// requests the NEXT step's three A planes first, then issues its six MFMAs with a pair of the next k-step's B
// values split between them -- a ds_read has six MFMAs (>= 192 cycles) to arrive, the 88 vector instructions of a
// split are spread over the 24 MFMAs of a k-step, which leave 24 of their 32 cycles to them.  The scheduling
// groups pin that order (the compiler's own put each read right in front of its MFMA: 44 % of the wave cycles
// were waits on LDS, SQ_WAIT_ANY of the build without global accesses).
//   BVAL: expression of (ks_, e_) = value e_ of k-step ks_'s eight
#define BF3_PRODUCT(NKS, WA, WB, BVAL)                                                                              \
  {                                                                                                                 \
    typedef __attribute__((address_space(3))) u32x4 lds_u4_;                                                        \
    u32x4 A_[2][3], B_[2][3];                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
      const int ks_ = 0;                                                                                            \
      float p0_, p1_;                                                                                               \
      { const int e_ = 2 * i_; p0_ = (BVAL); }                                                                      \
      { const int e_ = 2 * i_ + 1; p1_ = (BVAL); }                                                                  \
      unsigned h_, m_, l_;                                                                                          \
      bf3_split2(p0_, p1_, h_, m_, l_);                                                                             \
      B_[0][0][i_] = h_; B_[0][1][i_] = m_; B_[0][2][i_] = l_;                                                      \
    }                                                                                                               \
    _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) A_[0][pl_] = *(const lds_u4_ *)(uintptr_t)((WA) + 1024u * pl_); \
    _Pragma("unroll") for (int ksx_ = 0; ksx_ < (NKS); ++ksx_) {                                                    \
      _Pragma("unroll") for (int blk_ = 0; blk_ < 4; ++blk_) {                                                      \
        const int s_ = ksx_ * 4 + blk_;                                                                             \
        if (s_ + 1 < 4 * (NKS)) {                                                                                   \
          const int nk_ = (s_ + 1) >> 2, nb_ = (s_ + 1) & 3;                                                        \
          const unsigned ad_ = (nb_ < 2 ? (WA) : (WB)) + 3072u * (unsigned)((nb_ & 1) * (NKS) + nk_);               \
          _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_)                                                       \
            A_[(s_ + 1) & 1][pl_] = *(const lds_u4_ *)(uintptr_t)(ad_ + 1024u * pl_);                               \
        }                                                                                                           \
        if (ksx_ + 1 < (NKS)) {                                                                                     \
          const int ks_ = ksx_ + 1;                                                                                 \
          float p0_, p1_;                                                                                           \
          { const int e_ = 2 * blk_; p0_ = (BVAL); }                                                                \
          { const int e_ = 2 * blk_ + 1; p1_ = (BVAL); }                                                            \
          unsigned h_, m_, l_;                                                                                      \
          bf3_split2(p0_, p1_, h_, m_, l_);                                                                         \
          B_[(ksx_ + 1) & 1][0][blk_] = h_; B_[(ksx_ + 1) & 1][1][blk_] = m_; B_[(ksx_ + 1) & 1][2][blk_] = l_;     \
        }                                                                                                           \
        const u32x4 ah_ = A_[s_ & 1][0], am_ = A_[s_ & 1][1], al_ = A_[s_ & 1][2];                                  \
        const u32x4 bh_ = B_[ksx_ & 1][0], bm_ = B_[ksx_ & 1][1], bl_ = B_[ksx_ & 1][2];                            \
        BF3_MF1(acc[blk_], al_, bh_);                                                                               \
        BF3_MF1(acc[blk_], ah_, bl_);                                                                               \
        BF3_MF1(acc[blk_], am_, bm_);                                                                               \
        BF3_MF1(acc[blk_], am_, bh_);                                                                               \
        BF3_MF1(acc[blk_], ah_, bm_);                                                                               \
        BF3_MF1(acc[blk_], ah_, bh_);                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                                          \
        _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                                          \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                        \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
  }
#define BF3_MF1(c_, a_, b_) \
  c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)

constexpr int FS3_W1_BYTES = 4 * 8 * 3 * 1024, FS3_W2_BYTES = 4 * 4 * 3 * 1024;  // 96 KB + 48 KB
constexpr int FS3_LDS_BYTES = FS3_W1_BYTES + FS3_W2_BYTES + 128 * 4;

// The layer's weights as the kernel's LDS image: W1 [4 blocks][8 k-steps][3 planes][64 lanes][8 bf16], W2 [4][4][3][64][8],
// then br | bs as floats.  Input register j of a lane of half lh holds channel (j & 3) + 8 (j >> 2) + 4 lh; k-step ks
// takes registers 8 ks' .. 8 ks' + 7: ks 0..3 of x(t) (tap 1), ks 4..7 of x(t - d) (tap 0).  `img` may be LDS or global.
template <class Ptr>
__device__ __forceinline__ void fs3_stage_weights(Ptr img, const float *wf, const float *wg, const float *wr,
                                                  const float *ws, const float *br, const float *bs, int tid, int nthreads) {
  unsigned short *W1 = (unsigned short *)img;
  unsigned short *W2 = (unsigned short *)(img + FS3_W1_BYTES);
  float *BI = (float *)(img + FS3_W1_BYTES + FS3_W2_BYTES);
  for (int sI = tid; sI < 2 * 8192; sI += nthreads) {
    const int g = sI >> 13, r = sI & 8191;        // g: 0 filter, 1 gate; source (out, in, tap)
    const int tap = r & 1, kc = (r >> 1) & 63, cm = r >> 7;
    const int lhs = (kc >> 2) & 1, j = (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (cm >> 5), ln = (cm & 31) + 32 * lhs, ks = (tap ? 0 : 4) + (j >> 3);
    unsigned short h, m, l;
    bf3_split1((g ? wg : wf)[r], h, m, l);
    const int at = (((blk * 8 + ks) * 3) * 64 + ln) * 8 + (j & 7);
    W1[at] = h;
    W1[at + 512] = m;
    W1[at + 1024] = l;
  }
  for (int sI = tid; sI < 2 * 4096; sI += nthreads) {
    const int g = sI >> 12, r = sI & 4095;        // g: 0 residual, 1 skip; source (out, in)
    const int kc = r & 63, m2 = r >> 6;
    const int lhs = (kc >> 2) & 1, j = (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (m2 >> 5), ln = (m2 & 31) + 32 * lhs, ks = j >> 3;
    unsigned short h, m, l;
    bf3_split1((g ? ws : wr)[r], h, m, l);
    const int at = (((blk * 4 + ks) * 3) * 64 + ln) * 8 + (j & 7);
    W2[at] = h;
    W2[at + 512] = m;
    W2[at + 1024] = l;
  }
  for (int i = tid; i < 128; i += nthreads) BI[i] = i < 64 ? br[i] : bs[i - 64];
}

// up to FS3_PACK_LAYERS layers per launch (blockIdx.y = layer): images FS3_PACK_F floats apart
constexpr int FS3_PACK_LAYERS = 32, FS3_PACK_F = FS3_LDS_BYTES / 4;
struct Fs3PackArgs {
  const float *wf[FS3_PACK_LAYERS], *wg[FS3_PACK_LAYERS], *wr[FS3_PACK_LAYERS], *ws[FS3_PACK_LAYERS];
  const float *br[FS3_PACK_LAYERS], *bs[FS3_PACK_LAYERS];
};
__global__ __launch_bounds__(256) void fs3_pack_kernel(Fs3PackArgs p, float *dst) {
  const int l = blockIdx.y;
  fs3_stage_weights((unsigned char *)(dst + (size_t)l * FS3_PACK_F), p.wf[l], p.wg[l], p.wr[l], p.ws[l], p.br[l], p.bs[l],
                    blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

__global__ __launch_bounds__(512, 1) void fused_layer64s_bf3_kernel(FusedFwdPArgs a, int chunks_per_b, int chunk_t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fs3_lds[];
  unsigned short *W1 = (unsigned short *)fs3_lds;                     // [4 blocks][8 k-steps][3 planes][64 lanes][8]
  unsigned short *W2 = (unsigned short *)(fs3_lds + FS3_W1_BYTES);    // [4 blocks][4 k-steps][3 planes][64 lanes][8]
  float *BI = (float *)(fs3_lds + FS3_W1_BYTES + FS3_W2_BYTES);      // br | bs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_begin, a.t_skip0);
  // ---- weights into LDS: the image fs3_pack_kernel wrote once per forward call (a linear 144 KB copy, every load
  // in flight before the first store), or, without one, converted here (17 us per launch: 48 dependent
  // global load -> split -> three 2-byte LDS stores per thread; timing build 26)
  if (a.wpack) {
    typedef float f4_ __attribute__((ext_vector_type(4)));
    constexpr int N16 = FS3_LDS_BYTES / 16, PER = (N16 + 511) / 512;
    const f4_ *src = (const f4_ *)a.wpack;
    f4_ v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) v[i] = src[at];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) ((f4_ *)fs3_lds)[at] = v[i];
    }
  } else {
    fs3_stage_weights(fs3_lds, a.wf, a.wg, a.wr, a.ws, a.br, a.bs, tid, 512);
  }
  __syncthreads();
  const unsigned w1a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)fs3_lds + 16u * lane;
  // (two base registers per matrix: every plane is then within a ds_read's 16-bit offset of one of them)
  unsigned w1b = w1a + 2u * 8u * 3072u, w2a = w1a + (unsigned)FS3_W1_BYTES, w2b = w2a + 2u * 4u * 3072u;
  asm volatile("" : "+v"(w1b), "+v"(w2a), "+v"(w2b));
  const int cbase = 4 * lh;

  constexpr int RSRC = 0x00020000;  // raw buffer, 32-bit data format (gfx9)
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xin.p + (size_t)b * a.xin.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t thb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.th.p + (size_t)b * a.th.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t sgb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.sg.p + (size_t)b * a.sg.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t xob = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xout.p + (size_t)b * a.xout.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t skb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.skip.p + (size_t)b * a.skip.sb - a.t_base), 0, 0x7FFFFFFF, RSRC);
  // timing builds (wrong results; python -m movenet_amd.csrc.build --stamps --exp=N, scripts/exp_fwd.sh): 21 no tanh /
  // sigmoid stores, 22 no stores, 23 no loads after a wave's first strip, 24 = 22 + 23, 25 no MFMAs (nor splits)
  constexpr bool aux_ok = true;
  constexpr bool st_ok = true;
  constexpr bool ld_ok = true;
  const bool save = aux_ok && a.th.p != nullptr, has_out = st_ok && a.xout.p != nullptr;
  int xld4 = 4 * a.xin.ld, thld4 = 4 * a.th.ld, xold4 = 4 * a.xout.ld, skld4 = 4 * a.skip.ld;

  // x(t) of a strip is fetched one strip ahead, under the second product of the strip before, and consumed first,
  // which gives the x(t - d) half, requested at the top of the strip, the first half of the first product to
  // arrive; the skip accumulator's old values arrive under the second product in the x(t - d) registers
  // (as in fused_layer64s_kernel).  (Tried: 8-byte stores -- a lane pair exchanges two registers by DPP so that one
  // store instruction writes 4 row segments of 128 bytes instead of 2, half the store instructions: 91.9 against
  // 91.2 us, the stores cost their bytes, not their count.  Tried: x(t - d) a strip ahead too and the old skip sums at the top of the
  // strip, both as initial values of the second product's accumulators: 17 us per layer SLOWER -- a wave has 64
  // vector-memory instructions in flight at most (6-bit vmcnt), a strip issues 224, and 96 loads queued in front
  // of the 64 stores of the strip before stall the stores.)  Timing builds, us per layer: all 92, no tanh / sigmoid
  // stores 79, no stores 65, no loads 78, neither 62, no MFMAs / splits 70.
  auto column = [&](int t0_, bool &live_, int &tc_) {
    const int t_ = t0_ + li;
    live_ = t_ >= a.t_begin && t_ < te;
    tc_ = live_ ? t_ : a.t_begin;  // (clamped: dead lanes read a valid column, zeroed afterwards)
  };
  float xb1[32];  // x(t) of the current strip: B operand of k-steps 0..3, residual input
  {
    bool lv;
    int tcc;
    column(tb + 32 * wave, lv, tcc);
    const int o1 = 4 * (cbase * a.xin.ld + tcc);
    FS_FENCE(xld4);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      xb1[j] = lv ? v : 0.f;
    }
  }

  constexpr int FS_OOB = (int)0x80000000;
  for (int t0 = tb + 32 * wave; t0 < te; t0 += 32 * 8) {
    const int t = t0 + li;
    bool live;
    int tc;
    column(t0, live, tc);
    const bool skip_live = st_ok && t >= skip_lo && t < te;
    const int ox0 = 4 * (cbase * a.xin.ld + tc - a.d);
    const int oth = (save && live) ? 4 * (cbase * a.th.ld + tc) : FS_OOB, oxo = 4 * (cbase * a.xout.ld + tc);
    const int osk = 4 * (cbase * a.skip.ld + (skip_live ? t : skip_lo));
    float xn1[32];  // x(t) of the NEXT strip
    // ---- x(t - d): B operand of k-steps 4..7; later the skip accumulator's old values
    float xa0[32];
    FS_FENCE(xld4);
    if (ld_ok || t0 == tb + 32 * wave) {
#pragma unroll
      for (int j = 0; j < 32; ++j)
        xa0[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, ox0, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) xa0[j] = xb1[j] * 0.5f;
    }
    // (interior strips -- wave-uniform test -- need no masking)
    if (!(t0 >= a.t_begin && t0 + 32 <= te)) {
#pragma unroll
      for (int j = 0; j < 32; ++j) xa0[j] = live ? xa0[j] : 0.f;
    }
    // ---- f | g: four 32 x 32 blocks (f c<32, f c>=32, g c<32, g c>=32), K = 128 = 8 k-steps, the x(t) half first
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    BF3_PRODUCT(8, w1a, w1b, (ks_ < 4 ? xb1[8 * ks_ + e_] : xa0[8 * (ks_ - 4) + e_]));
    // ---- gate in registers; tanh / sigmoid leave; z in accumulator order = the next B operand
    float z[32];
    FS_FENCE(thld4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float tv = tanh_fast(acc[h][r]);
        const float sv = sigmoid_fast(acc[2 + h][r]);
        z[16 * h + r] = tv * sv;
        const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tv), thb, oth, c0 * thld4, FS_AUX_SAVE);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv), sgb, oth, c0 * thld4, FS_AUX_SAVE);
      }
    // the skip accumulator's old values, into the x(t - d) registers (dead now), and the NEXT strip's x(t), both
    // under the MFMAs below
    FS_FENCE(skld4);
    if (!a.first_layer && ld_ok) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xa0[16 * h + r] = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(skb, osk, (32 * h + (r & 3) + 8 * (r >> 2)) * skld4, 0));
    }
    if (ld_ok && t0 + 32 * 8 < te) {
      bool lv;
      int tcc;
      column(t0 + 32 * 8, lv, tcc);
      const int o1 = 4 * (cbase * a.xin.ld + tcc);
      FS_FENCE(xld4);
#pragma unroll
      for (int j = 0; j < 32; ++j)
        xn1[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      if (!(t0 + 32 * 8 >= a.t_begin && t0 + 32 * 8 + 32 <= te)) {
#pragma unroll
        for (int j = 0; j < 32; ++j) xn1[j] = lv ? xn1[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) xn1[j] = ld_ok ? 0.f : xb1[j];
    }
    // ---- residual | skip: four blocks (res c<32, res c>=32, skip k<32, skip k>=32), K = 64 = 4 k-steps
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    BF3_PRODUCT(4, w2a, w2b, z[8 * ks_ + e_]);
    // ---- x' = (y + br) + x(t): x(t) of this lane's channel is input register 16 h + r;
    // skip (+)= y + bs, columns t - t_base, live from skip_lo
    FS_FENCE(xold4);
    FS_FENCE(skld4);
    {
      const int oxo_m = (has_out && live) ? oxo : FS_OOB, osk_m = skip_live ? osk : FS_OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((acc[h][r] + BI[c0 + cbase]) + xb1[16 * h + r]), xob, oxo_m,
                                                c0 * xold4, 0);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = 32 * h + (r & 3) + 8 * (r >> 2);
          const float v = acc[2 + h][r] + BI[64 + k0 + cbase];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((a.first_layer || !ld_ok) ? v : xa0[16 * h + r] + v), skb, osk_m,
                                                k0 * skld4, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) xb1[j] = xn1[j];
  }
}

// ----------------------------------------------------------------------------------------
// The CONDITIONED layer (fused_layer64s_kernel<true>, fused_fwd.h) with what fits of this on it: its f | g product has
// three K blocks (x(t - d), x(t), context: 96 KB of fp32 MFMA operands, 144 KB as bf16 planes) and does not fit
// as planes beside anything; the residual | skip product does (48 KB instead of 32: 145 KB in all) and runs on the
// bf16 matrix cores -- 96 bf16 MFMAs per strip instead of 128 fp32 ones, a quarter of the kernel's fp32 MFMA time
// gone.  The layer's LDS image (W1 in fp32 MFMA order, W2 as planes, the four bias vectors) is written once per
// forward call by fsc_pack_kernel and copied linearly (the staging loops of the fp32 kernel: ~8 us per launch).
// ----------------------------------------------------------------------------------------
// LDS image (exactly the CU's 160 KB): the x(t - d) block of W1 as planes (48 KB: it too on the bf16 matrix cores), the
// x(t) and context blocks as fp32 MFMA operands (64 KB), W2 as planes (48 KB).  The four bias vectors do not fit any
// more: each lives in ONE register (lane i holds element i) and is read by ds_bpermute where the LDS copy was read.
constexpr int FSC_XD_BYTES = 4 * 4 * 3 * 1024;                               // [4 blocks][4 k-steps][3 planes][64 lanes][8 bf16]
constexpr int FSC_W1_BYTES = 4 * 16 * 1024;                                  // [4 blocks][16 k-step groups][64 lanes][4] floats
constexpr int FSC_LDS_BYTES = FSC_XD_BYTES + FSC_W1_BYTES + FS3_W2_BYTES, FSC_PACK_F = FSC_LDS_BYTES / 4;
static_assert(FSC_LDS_BYTES == 160 * 1024, "the conditioned layer's image fills a CU's LDS");
struct FscPackArgs {
  const float *wf[FS3_PACK_LAYERS], *wg[FS3_PACK_LAYERS], *wr[FS3_PACK_LAYERS], *ws[FS3_PACK_LAYERS];
  const float *br[FS3_PACK_LAYERS], *bs[FS3_PACK_LAYERS];
  const float *wcf[FS3_PACK_LAYERS], *wcg[FS3_PACK_LAYERS], *bcf[FS3_PACK_LAYERS], *bcg[FS3_PACK_LAYERS];
};
__global__ __launch_bounds__(256) void fsc_pack_kernel(FscPackArgs p, float *dst) {
  const int l = blockIdx.y, tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
  unsigned char *img = (unsigned char *)(dst + (size_t)l * FSC_PACK_F);
  unsigned short *XD = (unsigned short *)img;
  float *W1 = (float *)(img + FSC_XD_BYTES);
  unsigned short *W2 = (unsigned short *)(img + FSC_XD_BYTES + FSC_W1_BYTES);
  constexpr int NK1 = 16;
  for (int sI = tid; sI < 2 * 8192; sI += nthreads) {  // filter | gate, source (out, in, tap)
    const int g = sI >> 13, r = sI & 8191;
    const int tap = r & 1, kc = (r >> 1) & 63, cm = r >> 7;
    const int lhs = (kc >> 2) & 1, j = (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (cm >> 5), ln = (cm & 31) + 32 * lhs;
    const float w = (g ? p.wg[l] : p.wf[l])[r];
    if (tap) {  // x(t): fp32 MFMA operand, k-step groups 0..7
      W1[((blk * NK1 + (j >> 2)) * 64 + ln) * 4 + (j & 3)] = w;
    } else {    // x(t - d): three planes, k-step j >> 3
      unsigned short h, m, lo;
      bf3_split1(w, h, m, lo);
      const int at = (((blk * 4 + (j >> 3)) * 3) * 64 + ln) * 8 + (j & 7);
      XD[at] = h;
      XD[at + 512] = m;
      XD[at + 1024] = lo;
    }
  }
  for (int sI = tid; sI < 2 * 4096; sI += nthreads) {  // context convs, source (out, in): k-step groups 8..15
    const int g = sI >> 12, r = sI & 4095;
    const int kc = r & 63, cm = r >> 6;
    const int lhs = (kc >> 2) & 1, kk = 32 + (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (cm >> 5), ln = (cm & 31) + 32 * lhs;
    W1[((blk * NK1 + (kk >> 2)) * 64 + ln) * 4 + (kk & 3)] = (g ? p.wcg[l] : p.wcf[l])[r];
  }
  for (int sI = tid; sI < 2 * 4096; sI += nthreads) {  // residual | skip as planes (fs3_stage_weights' layout)
    const int g = sI >> 12, r = sI & 4095;
    const int kc = r & 63, m2 = r >> 6;
    const int lhs = (kc >> 2) & 1, j = (kc & 3) + 4 * (kc >> 3);
    const int blk = 2 * g + (m2 >> 5), ln = (m2 & 31) + 32 * lhs, ks = j >> 3;
    unsigned short h, m, lo;
    bf3_split1((g ? p.ws[l] : p.wr[l])[r], h, m, lo);
    const int at = (((blk * 4 + ks) * 3) * 64 + ln) * 8 + (j & 7);
    W2[at] = h;
    W2[at + 512] = m;
    W2[at + 1024] = lo;
  }
}

__global__ __launch_bounds__(512, 1) void fused_layer64s_ctxw2_kernel(FusedFwdPArgs a, int chunks_per_b, int chunk_t) {
  constexpr bool HAS_CTX = true;
  constexpr int C = 64, NK1 = 16;
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int li = lane & 31, lh = lane >> 5;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  const int skip_lo = max(a.t_begin, a.t_skip0);
  // ---- weights into LDS: the image fsc_pack_kernel wrote once per forward call (W1 as fp32 MFMA operands,
  // [block][k-step / 4][lane][k-step % 4]; W2 as three bf16 planes; the four bias vectors)
  {
    typedef float f4_ __attribute__((ext_vector_type(4)));
    constexpr int N16 = FSC_LDS_BYTES / 16, PER = (N16 + 511) / 512;
    const f4_ *src = (const f4_ *)a.wpack;
    f4_ v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) v[i] = src[at];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) ((f4_ *)fs_lds)[at] = v[i];
    }
  }
  __syncthreads();
  const unsigned xda = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)fs_lds + 16u * lane;
  const unsigned w1a = xda + (unsigned)FSC_XD_BYTES;
  unsigned w2a = w1a + (unsigned)FSC_W1_BYTES, w2b = w2a + 2u * 4u * 3072u;
  asm volatile("" : "+v"(w2a), "+v"(w2b));
  // the bias vectors, one register each (lane i: element i), read with ds_bpermute
  const int bv_r = __float_as_int(a.br[lane]), bv_s = __float_as_int(a.bs[lane]);
  const int bv_cf = __float_as_int(a.bcf[lane]), bv_cg = __float_as_int(a.bcg[lane]);
#define FSC_BIAS(vec_, ch_) __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (ch_), (vec_)))
  typedef float fsv4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) fsv4 lds_v4;
  // channel of accumulator register r of block h in this lane: 32 h + (r & 3) + 8 (r >> 2) + 4 lh
  const int cbase = 4 * lh;

  // Every global access below is a raw BUFFER access: (a resource per tensor and sequence:
  // four scalar registers) + (row * ld: one scalar offset) + (one of five per-lane byte offsets).
  // Formed as 64-bit vector addresses the ~290 accesses of a strip spilled 443 registers; as
  // scalar row pointers + vector offsets they still cost a 64-bit vector add each and spilled the
  // scalar file (130 v_writelane / v_readlane per strip).
  constexpr int RSRC = 0x00020000;  // raw buffer, 32-bit data format (gfx9)
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xin.p + (size_t)b * a.xin.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t thb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.th.p + (size_t)b * a.th.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t sgb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.sg.p + (size_t)b * a.sg.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t xob = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xout.p + (size_t)b * a.xout.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t skb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.skip.p + (size_t)b * a.skip.sb - a.t_base), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t cb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.ctx.p + (size_t)b * a.ctx.sb), 0, 0x7FFFFFFF, RSRC);
  constexpr bool st_ok = true;
  const bool save = st_ok && a.th.p != nullptr, has_out = st_ok && a.xout.p != nullptr;
  int xld4 = 4 * a.xin.ld, thld4 = 4 * a.th.ld, xold4 = 4 * a.xout.ld, skld4 = 4 * a.skip.ld, cld4 = 4 * a.ctx.ld;
  // (row offsets = row * ld are re-formed where they are used, behind a fence on ld: hoisted out of
  // the strip loop the ~120 products filled the scalar file and were spilled to vector lanes)
#define FS_FENCE(x) asm volatile("" : "+s"(x))

  // x(t) of a strip (the tap-1 half of the input, 32 registers) is fetched one strip AHEAD, under
  // the second product of the strip before; the first product consumes it first, which gives the
  // x(t - d) half, requested at the top of the strip, 128 MFMAs to arrive.  (Without it the wave
  // counters showed 51 % of the wave cycles waiting for memory: a wave that issues MFMAs back to
  // back starves the other wave of its SIMD, so the two fall into step and wait together.)
  auto column = [&](int t0_, bool &live_, int &tc_) {
    const int t_ = t0_ + li;
    live_ = t_ >= a.t_begin && t_ < te;
    tc_ = live_ ? t_ : a.t_begin;  // (clamped: dead lanes read a valid column, zeroed afterwards)
  };
  float xb1[32];  // x(t) of the current strip: B operand of k-steps 32..63, residual input
  if (!HAS_CTX) {
    bool lv;
    int tcc;
    column(tb + 32 * wave, lv, tcc);
    const int o1 = 4 * (cbase * a.xin.ld + tcc);
    FS_FENCE(xld4);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      xb1[j] = lv ? v : 0.f;
    }
  }

  // Lanes that must not store (columns outside [t_begin, te), no output tensor) carry an offset
  // beyond the buffer's num_records: the hardware drops the access -- no exec-mask branch per store.
  // (Tried: holding a strip's x' and skip sums in registers until after the next strip's first
  // product, so that no wait on a prefetched load sits right behind 64 fresh stores -- on gfx9 loads
  // and stores retire through one in-order counter and the compiler's wait at the loop edge was
  // vmcnt(0).  Measured 131.4 against 131.9 us per layer: the second wave of the SIMD covers it.)
  constexpr int FS_OOB = (int)0x80000000;
  for (int t0 = tb + 32 * wave; t0 < te; t0 += 32 * 8) {
    const int t = t0 + li;
    bool live;
    int tc;
    column(t0, live, tc);
    const bool skip_live = st_ok && t >= skip_lo && t < te;
    // per-lane byte offsets (channel part 4 lh of the row + the column)
    const int ox0 = 4 * (cbase * a.xin.ld + tc - a.d);
    const int oth = (save && live) ? 4 * (cbase * a.th.ld + tc) : FS_OOB, oxo = 4 * (cbase * a.xout.ld + tc);
    const int osk = 4 * (cbase * a.skip.ld + (skip_live ? t : skip_lo));
    float xn1[32];  // without context: x(t) of the NEXT strip; with: ctx(t) of this one (k-steps 64..95)
    if (HAS_CTX) {
      const int o1 = 4 * (cbase * a.xin.ld + tc), oc = 4 * (cbase * a.ctx.ld + tc);
      FS_FENCE(xld4);
      FS_FENCE(cld4);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
        xb1[j] = live ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cb, oc, ((j & 3) + 8 * (j >> 2)) * cld4, 0));
        xn1[j] = live ? v : 0.f;
      }
    }
    // ---- x(t - d) of channel kc(j) + 4 lh: B operand of k-steps 0..31; later the skip accumulator's old values
    float xa0[32];
    FS_FENCE(xld4);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, ox0, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
      xa0[j] = v;
    }
    // (interior strips -- wave-uniform test -- need no masking: 32 selects less per strip)
    if (!(t0 >= a.t_begin && t0 + 32 <= te)) {
#pragma unroll
      for (int j = 0; j < 32; ++j) xa0[j] = live ? xa0[j] : 0.f;
    }
    // ---- f | g: four 32 x 32 blocks (f c<32, f c>=32, g c<32, g c>=32), K = 128, the x(t) half first
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // x(t) (prefetched: consumed first, fp32 MFMAs), x(t - d) (bf16 planes), context (fp32 MFMAs)
#pragma unroll
    for (int kq = 0; kq < NK1; ++kq) {
      if (kq == 8) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u32x4 bh, bm, bl;
          bf3_split8(&xa0[8 * ks], bh, bm, bl);
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) bf3_mfma6(acc[blk], xda + 3072u * (unsigned)(blk * 4 + ks), bh, bm, bl);
        }
      }
      fsv4 aw[4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) aw[blk] = *(const lds_v4 *)(uintptr_t)(w1a + 4u * (unsigned)((blk * NK1 + kq) * 256));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[blk][e], kq >= 8 ? xn1[4 * (kq - 8) + e] : xb1[4 * kq + e], acc[blk], 0, 0,
                                                          0);
    }
    // ---- gate in registers; tanh / sigmoid leave; z in accumulator order = the next B operand
    float z[32];
    FS_FENCE(thld4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cg_ = 32 * h + (r & 3) + 8 * (r >> 2) + cbase;
        const float tv = tanh_fast(acc[h][r] + FSC_BIAS(bv_cf, cg_));
        const float sv = sigmoid_fast(acc[2 + h][r] + FSC_BIAS(bv_cg, cg_));
        z[16 * h + r] = tv * sv;
        const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tv), thb, oth, c0 * thld4, FS_AUX_SAVE);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv), sgb, oth, c0 * thld4, FS_AUX_SAVE);
      }
    // the skip accumulator's old values, into the x(t - d) registers (dead now), and the NEXT strip's
    // x(t), both under the MFMAs below
    FS_FENCE(skld4);
    if (!a.first_layer) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xa0[16 * h + r] = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(skb, osk, (32 * h + (r & 3) + 8 * (r >> 2)) * skld4, 0));
    }
    const bool more = !HAS_CTX && t0 + 32 * 8 < te;
    if (HAS_CTX) {
    } else if (more) {
      bool lv;
      int tcc;
      column(t0 + 32 * 8, lv, tcc);
      const int o1 = 4 * (cbase * a.xin.ld + tcc);
      FS_FENCE(xld4);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, o1, ((j & 3) + 8 * (j >> 2)) * xld4, 0));
        xn1[j] = v;
      }
      if (!(t0 + 32 * 8 >= a.t_begin && t0 + 32 * 8 + 32 <= te)) {
#pragma unroll
        for (int j = 0; j < 32; ++j) xn1[j] = lv ? xn1[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) xn1[j] = 0.f;
    }
    // ---- residual | skip: four blocks (res c<32, res c>=32, skip k<32, skip k>=32), K = 64
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    BF3_PRODUCT(4, w2a, w2b, z[8 * ks_ + e_]);  // (the residual | skip product on the bf16 matrix cores)
    // ---- x' = (y + br) + x(t): x(t) of this lane's channel is input register 32 + 16 h + r;
    // skip (+)= y + bs, columns t - t_base, live from skip_lo
    FS_FENCE(xold4);
    FS_FENCE(skld4);
    {
      const int oxo_m = (has_out && live) ? oxo : FS_OOB, osk_m = skip_live ? osk : FS_OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0 = 32 * h + (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((acc[h][r] + FSC_BIAS(bv_r, c0 + cbase)) + xb1[16 * h + r]), xob, oxo_m,
                                                c0 * xold4, 0);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = 32 * h + (r & 3) + 8 * (r >> 2);
          const float v = acc[2 + h][r] + FSC_BIAS(bv_s, k0 + cbase);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a.first_layer ? v : xa0[16 * h + r] + v), skb, osk_m, k0 * skld4, 0);
        }
    }
    if (!HAS_CTX) {
#pragma unroll
      for (int j = 0; j < 32; ++j) xb1[j] = xn1[j];
    }
  }
}


static int launch_fsc_pack(const mvn_params *p, int L, float *dst, hipStream_t s) {
  for (int l0 = 0; l0 < L; l0 += FS3_PACK_LAYERS) {
    const int n = std::min(FS3_PACK_LAYERS, L - l0);
    FscPackArgs pa;
    for (int i = 0; i < FS3_PACK_LAYERS; ++i) {
      const int l = l0 + std::min(i, n - 1);
      pa.wf[i] = p->filter_w[l]; pa.wg[i] = p->gate_w[l]; pa.wr[i] = p->residual_w[l]; pa.ws[i] = p->skip_w[l];
      pa.br[i] = p->residual_b[l]; pa.bs[i] = p->skip_b[l];
      pa.wcf[i] = p->ctx_filter_w[l]; pa.wcg[i] = p->ctx_gate_w[l]; pa.bcf[i] = p->ctx_filter_b[l]; pa.bcg[i] = p->ctx_gate_b[l];
    }
    hipLaunchKernelGGL(fsc_pack_kernel, dim3(8, n), dim3(256), 0, s, pa, dst + (size_t)l0 * FSC_PACK_F);
  }
  return check_hip(hipGetLastError(), "fsc_pack");
}
static int launch_fused_layer64s_ctxw2(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch, 1, &chunks, &chunk_t, 256);
  const void *fn = (const void *)fused_layer64s_ctxw2_kernel;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(fused_layer64s_ctxw2)");
  if (rc) return rc;
  hipLaunchKernelGGL(fused_layer64s_ctxw2_kernel, dim3(chunks * batch), dim3(512), FSC_LDS_BYTES, s, a, chunks, chunk_t);
  return MVN_OK;
}

// ----------------------------------------------------------------------------------------
// The head's two forward convolutions (dense_strip_kernel, fused_fwd.h) in the same form: weights as three bf16
// planes in LDS -- K x M x 6 bytes, 96 KB for conv1 (64 -> 256 rows, all of them in one workgroup) and for conv2
// in FOUR row blocks of 64 (256 inputs; the fp32 kernel held two blocks of 128: the input is read four times here
// instead of twice, which the matrix cores' time pays for) -- the strip's input in registers, split eight values
// at a time, six MFMAs per block and k-step.  `wimg`: the block's LDS image from ds3_pack_kernel, or NULL
// (converted here).
//   IN: 0 identity, 1 leaky-ReLU on load;  OUT: 0 y + bias, 1 leaky(y + bias), 2 y * leaky'(ref) (the head's data
// gradients, weights TRANSPOSED: their leaky-ReLU reference values are requested BEFORE the products -- the fp32
// strip form had no register left for that and lost to the generic kernel, 454 against 342 us)
// ----------------------------------------------------------------------------------------
template <int K, int M, bool TRANSPOSED>
__device__ __forceinline__ void ds3_stage(unsigned char *img, const float *wmat, int ldw, const float *bias, int m_base,
                                          int tid, int nthreads) {
  constexpr int NKS = K / 16;
  unsigned short *W = (unsigned short *)img;
  float *BI = (float *)(img + K * M * 6);
  for (int r = tid; r < K * M; r += nthreads) {  // source order: coalesced (W[m][k] = TRANSPOSED ? wmat[k][m] : wmat[m][k])
    const int m = TRANSPOSED ? r % M : r / K, k = TRANSPOSED ? r / M : r % K;
    // input register kk of a lane of half lh holds row 2 kk + lh; k-step ks takes registers 8 ks .. 8 ks + 7
    const int lhs = k & 1, kk = k >> 1, ks = kk >> 3, e = kk & 7;
    unsigned short h, mm, l;
    bf3_split1(TRANSPOSED ? wmat[(size_t)k * ldw + m_base + m] : wmat[(size_t)(m_base + m) * ldw + k], h, mm, l);
    const int at = ((((m >> 5) * NKS + ks) * 3) * 64 + (m & 31) + 32 * lhs) * 8 + e;
    W[at] = h;
    W[at + 512] = mm;
    W[at + 1024] = l;
  }
  for (int i = tid; i < M; i += nthreads) BI[i] = bias ? bias[m_base + i] : 0.f;
}
template <int K, int M, bool TRANSPOSED>
__global__ __launch_bounds__(256) void ds3_pack_kernel(const float *wmat, int ldw, const float *bias, float *dst) {
  constexpr int IMG_F = (K * M * 6 + M * 4) / 4;
  ds3_stage<K, M, TRANSPOSED>((unsigned char *)(dst + (size_t)blockIdx.y * IMG_F), wmat, ldw, bias, blockIdx.y * M,
                              blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

template <int K, int M, int IN, int OUT, bool TRANSPOSED>
__global__ __launch_bounds__(512, 1) void dense_strip_bf3_kernel(DenseStripArgs a, const float *wimg, int chunks_per_b,
                                                                 int chunk_t) {
  static_assert(K * M * 6 + M * 4 <= 160 * 1024 - 1024 && K % 16 == 0 && M % 32 == 0, "the planes fit a CU's LDS");
  constexpr int NB = M / 32, NKS = K / 16, IMG_BYTES = K * M * 6 + M * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ds3_lds[];
  float *BI = (float *)(ds3_lds + K * M * 6);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
  const int m_base = blockIdx.y * M;
  const int li = lane & 31, lh = lane >> 5;
  const int tb = (a.t_begin & ~TILE_ALIGN) + ch * chunk_t, te = min(a.t_end, tb + chunk_t);
  if (wimg) {
    typedef float f4_ __attribute__((ext_vector_type(4)));
    constexpr int N16 = IMG_BYTES / 16, PER = (N16 + 511) / 512;
    const f4_ *src = (const f4_ *)(wimg + (size_t)blockIdx.y * (IMG_BYTES / 4));
    f4_ v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) v[i] = src[at];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = tid + 512 * i;
      if (at < N16) ((f4_ *)ds3_lds)[at] = v[i];
    }
  } else {
    ds3_stage<K, M, TRANSPOSED>(ds3_lds, a.wmat, a.ldw, a.bias, m_base, tid, 512);
  }
  __syncthreads();
  const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ds3_lds + 16u * lane;
  constexpr int RSRC = 0x00020000;
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xin.p + (size_t)b * a.xin.sb), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.yout.p + (size_t)b * a.yout.sb + (size_t)m_base * a.yout.ld), 0, 0x7FFFFFFF, RSRC);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.ref.p + (size_t)b * a.ref.sb + (size_t)m_base * a.ref.ld), 0, 0x7FFFFFFF, RSRC);
  int xld4 = 4 * a.xin.ld, yld4 = 4 * a.yout.ld, rld4 = 4 * a.ref.ld;
  for (int t0 = tb + 32 * wave; t0 < te; t0 += 32 * 8) {
    const int t = t0 + li;
    const bool live = t >= a.t_begin && t < te, out_live = live && t < a.t_out_end;
    const int tc = live ? t : a.t_begin;
    const int ox = 4 * (lh * a.xin.ld + tc), oy = 4 * (4 * lh * a.yout.ld + tc), orf = 4 * (4 * lh * a.ref.ld + tc);
    float rv[OUT == 2 ? NB * 16 : 1];
    if (OUT == 2) {
      FS_FENCE(rld4);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[OUT == 2 ? blk * 16 + r : 0] =
              __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, orf, (32 * blk + (r & 3) + 8 * (r >> 2)) * rld4, 0));
    }
    float xr[K / 2];
    FS_FENCE(xld4);
#pragma unroll
    for (int kk = 0; kk < K / 2; ++kk) {
      float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xb, ox, 2 * kk * xld4, 0));
      if (IN == 1) v = leaky(v);
      xr[kk] = live ? v : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 bh, bm, bl;
      bf3_split8(&xr[8 * ks], bh, bm, bl);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        // (base + offset: every plane within a ds_read's 16-bit offset of one of two bases)
        const unsigned off = 3072u * (unsigned)(blk * NKS + ks);
        bf3_mfma6(acc[blk], wa + off, bh, bm, bl);
      }
    }
    FS_FENCE(yld4);
    if (out_live) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m0 = 32 * blk + (r & 3) + 8 * (r >> 2);
          float y = acc[blk][r];
          if (OUT == 2) y *= rv[OUT == 2 ? blk * 16 + r : 0] > 0.f ? 1.0f : kLeakySlope;
          else y += BI[m0 + 4 * lh];
          if (OUT == 1) y = leaky(y);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), yb, oy, m0 * yld4, 0);
        }
    }
  }
}

// `m_total` output rows = m_total / M row blocks; `img`: m_total / M packed images (or NULL)
template <int K, int M, int IN, int OUT, bool TRANSPOSED = false>
static int launch_dense_strip_bf3(const DenseStripArgs &a, const float *img, int m_total, int batch, hipStream_t s) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch * (m_total / M), 1, &chunks, &chunk_t, 256);
  const void *fn = (const void *)dense_strip_bf3_kernel<K, M, IN, OUT, TRANSPOSED>;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(dense_strip_bf3)");
  if (rc) return rc;
  hipLaunchKernelGGL((dense_strip_bf3_kernel<K, M, IN, OUT, TRANSPOSED>), dim3(chunks * batch, m_total / M), dim3(512),
                     K * M * 6 + M * 4, s, a, img, chunks, chunk_t);
  return MVN_OK;
}
template <int K, int M, bool TRANSPOSED = false>
static void launch_ds3_pack(const float *wmat, int ldw, const float *bias, int m_total, float *dst, hipStream_t s) {
  hipLaunchKernelGGL((ds3_pack_kernel<K, M, TRANSPOSED>), dim3(8, m_total / M), dim3(256), 0, s, wmat, ldw, bias, dst);
}
constexpr size_t DS3_IMG1_F = (64 * 256 * 6 + 256 * 4) / 4, DS3_IMG2_F = (256 * 64 * 6 + 64 * 4) / 4;
constexpr size_t DS3_IMG_F = DS3_IMG1_F + 4 * DS3_IMG2_F;      // forward: conv1 + conv2's four blocks
constexpr size_t DS3_BWD_IMG_F = 4 * DS3_IMG2_F + DS3_IMG2_F;  // backward: conv2^T's four blocks + conv1^T

// MOVENET_HIP_FORWARD_MFMA=f32 keeps the fp32-MFMA strip kernel (A/B, tests); read per call
// MOVENET_HIP_HEAD_MFMA=f32 keeps the head's fp32 kernels (strips for the convolutions, the staged kernel for their
// data gradients); common.h: Switches
static bool head_bf3_enabled() { return !switches().head_f32; }
static bool forward_bf3_enabled() { return !switches().forward_f32; }

// the LDS images of layers 0 .. L-1 into `dst` (L x FS3_PACK_F floats), one launch per 32 layers
static int launch_fs3_pack(const mvn_params *p, int L, float *dst, hipStream_t s) {
  for (int l0 = 0; l0 < L; l0 += FS3_PACK_LAYERS) {
    const int n = std::min(FS3_PACK_LAYERS, L - l0);
    Fs3PackArgs pa;
    for (int i = 0; i < FS3_PACK_LAYERS; ++i) {
      const int l = l0 + std::min(i, n - 1);
      pa.wf[i] = p->filter_w[l]; pa.wg[i] = p->gate_w[l]; pa.wr[i] = p->residual_w[l]; pa.ws[i] = p->skip_w[l];
      pa.br[i] = p->residual_b[l]; pa.bs[i] = p->skip_b[l];
    }
    hipLaunchKernelGGL(fs3_pack_kernel, dim3(8, n), dim3(256), 0, s, pa, dst + (size_t)l0 * FS3_PACK_F);
  }
  return check_hip(hipGetLastError(), "fs3_pack");
}

static int launch_fused_layer64s_bf3(const FusedFwdPArgs &a, int batch, hipStream_t s) {
  const int nt = a.t_end - (a.t_begin & ~TILE_ALIGN);
  if (a.t_end <= a.t_begin || batch <= 0) return MVN_OK;
  int chunks, chunk_t;
  fb_chunks(nt, batch, 1, &chunks, &chunk_t, 256);  // a chunk: whole rounds of the 8 waves' strips
  const void *fn = (const void *)fused_layer64s_bf3_kernel;
  const int rc = ensure_max_dynamic_lds(fn, "hipFuncSetAttribute(fused_layer64s_bf3)");
  if (rc) return rc;
  hipLaunchKernelGGL(fused_layer64s_bf3_kernel, dim3(chunks * batch), dim3(512), FS3_LDS_BYTES, s, a, chunks, chunk_t);
  return MVN_OK;
}

}  // namespace mvn
