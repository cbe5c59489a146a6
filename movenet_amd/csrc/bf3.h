// fp32 operands as three bf16 planes (see fused_fwd_bf3.h for the arithmetic): x = h + m + l exactly, with
// h = top16(x), m = top16(x - h), l = top16(x - h - m); a product of two such numbers on the bf16 matrix cores
// is the sum of the six partial products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh), accumulated in fp32.
#pragma once

namespace mvn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf3_top(float x) { return __float_as_uint(x) & 0xffff0000u; }

// three bf16 planes of 8 fp32 values, element e of a plane = value e (two per register, low half first)
__device__ __forceinline__ void bf3_split8(const float *x, u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ah = bf3_top(a), bh = bf3_top(b);
    const float ar = a - __uint_as_float(ah), br = b - __uint_as_float(bh);   // exact
    const unsigned am = bf3_top(ar), bm = bf3_top(br);
    const float ar2 = ar - __uint_as_float(am), br2 = br - __uint_as_float(bm);  // exact
    h[i] = __builtin_amdgcn_perm(bh, ah, 0x07060302);  // {top16(b), top16(a)}
    m[i] = __builtin_amdgcn_perm(bm, am, 0x07060302);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
  }
}
// two values into register i of the three planes (a quarter of bf3_split8: the pipelined products below split
// the NEXT k-step's operand a pair at a time between the MFMAs of the current one)
__device__ __forceinline__ void bf3_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned ah = bf3_top(a), bh = bf3_top(b);
  const float ar = a - __uint_as_float(ah), br = b - __uint_as_float(bh);
  const unsigned am = bf3_top(ar), bm = bf3_top(br);
  const float ar2 = ar - __uint_as_float(am), br2 = br - __uint_as_float(bm);
  h = __builtin_amdgcn_perm(bh, ah, 0x07060302);
  m = __builtin_amdgcn_perm(bm, am, 0x07060302);
  l = __builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302);
}

// one value into its three 16-bit planes (weight staging)
__device__ __forceinline__ void bf3_split1(float w, unsigned short &h, unsigned short &m, unsigned short &l) {
  const unsigned wh = bf3_top(w);
  const float r = w - __uint_as_float(wh);
  const unsigned wm = bf3_top(r);
  const float r2 = r - __uint_as_float(wm);
  h = (unsigned short)(wh >> 16);
  m = (unsigned short)(wm >> 16);
  l = (unsigned short)(__float_as_uint(r2) >> 16);
}

// acc += A x B for one 32 x 32 x 16 block, both operands as planes in registers: smallest terms first
__device__ __forceinline__ void bf3_mfma6r(f32x16 &acc, const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh,
                                           const u32x4 &bm, const u32x4 &bl) {
#if defined(MVN_EXP) && MVN_EXP == 74  // (timing build: no MFMAs)
#define BF3_MFR(a_, b_) acc[0] += __uint_as_float(a_[0] ^ b_[0])
#else
#define BF3_MFR(a_, b_) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc, 0, 0, 0)
#endif
  BF3_MFR(al, bh);
  BF3_MFR(ah, bl);
  BF3_MFR(am, bm);
  BF3_MFR(am, bh);
  BF3_MFR(ah, bm);
  BF3_MFR(ah, bh);
#undef BF3_MFR
}

// A third of a split, in place: the top-16 plane of eight values, the values replaced by their residuals (20 vector
// instructions); applied twice it yields h and m, bf3_pack_l then packs what is left (4)
__device__ __forceinline__ void bf3_peel(float *x, u32x4 &p) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned at = bf3_top(x[2 * i]), bt = bf3_top(x[2 * i + 1]);
    p[i] = __builtin_amdgcn_perm(bt, at, 0x07060302);
    x[2 * i] -= __uint_as_float(at);
    x[2 * i + 1] -= __uint_as_float(bt);
  }
}
__device__ __forceinline__ void bf3_pack_l(const float *r, u32x4 &l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) l[i] = __builtin_amdgcn_perm(__float_as_uint(r[2 * i + 1]), __float_as_uint(r[2 * i]), 0x07060302);
}

// One slot of a software-pipelined bf16 x 3 product (fused_bwd_l.h, phase 2): the six MFMAs of an operand whose h plane is
// ready (f1..f3 need h only, f4, f5 the m plane, f6 the l plane), with the 44 vector instructions that form the m and l
// planes of THIS operand (residuals in xc, in place) and the h plane of the NEXT one (values in xn, replaced by their
// residuals) cut into six pieces behind the MFMAs: 7 + 7 + 6 (m), 8 (l, first of the next h), 8, 8.  A wave issues in
// order and an MFMA holds the SIMD's vector port for 8 of its 32 cycles only: split-then-multiply leaves the matrix pipe idle
// for the split and the vector port idle for the MFMAs, and the SIMD's other wave fills only part of either (scripts/probes/
// simd_pairing.hip).  Scheduling barriers keep the pieces where they are (scheduling GROUPS filled them unevenly: 7 / 1 / 12,
// then two MFMAs back to back).  Terms are summed h-terms first, then m, then l -- a different rounding order than
// bf3_mfma6r's smallest-first, same six terms.
template <class F1, class F2, class F3, class F4, class F5, class F6>
__device__ __forceinline__ void bf3_slot(const bool has_next, float *xc, float *xn, u32x4 &mc, u32x4 &lc, u32x4 &hn, F1 f1, F2 f2, F3 f3,
                                         F4 f4, F5 f5, F6 f6) {
#define BF3_SB __builtin_amdgcn_sched_barrier(0)
  unsigned t0, t1, t2, t3;
  f1(); BF3_SB;
  t0 = bf3_top(xc[0]); t1 = bf3_top(xc[1]); mc[0] = __builtin_amdgcn_perm(t1, t0, 0x07060302);
  xc[0] -= __uint_as_float(t0); xc[1] -= __uint_as_float(t1);
  t2 = bf3_top(xc[2]); t3 = bf3_top(xc[3]);
  BF3_SB; f2(); BF3_SB;
  mc[1] = __builtin_amdgcn_perm(t3, t2, 0x07060302); xc[2] -= __uint_as_float(t2); xc[3] -= __uint_as_float(t3);
  t0 = bf3_top(xc[4]); t1 = bf3_top(xc[5]); mc[2] = __builtin_amdgcn_perm(t1, t0, 0x07060302); xc[4] -= __uint_as_float(t0);
  BF3_SB; f3(); BF3_SB;
  xc[5] -= __uint_as_float(t1);
  t2 = bf3_top(xc[6]); t3 = bf3_top(xc[7]); mc[3] = __builtin_amdgcn_perm(t3, t2, 0x07060302);
  xc[6] -= __uint_as_float(t2); xc[7] -= __uint_as_float(t3);
  BF3_SB; f4(); BF3_SB;
  bf3_pack_l(xc, lc);
  if (has_next) {
    t0 = bf3_top(xn[0]); t1 = bf3_top(xn[1]); hn[0] = __builtin_amdgcn_perm(t1, t0, 0x07060302);
    xn[0] -= __uint_as_float(t0);
  }
  BF3_SB; f5(); BF3_SB;
  if (has_next) {
    xn[1] -= __uint_as_float(t1);
    t2 = bf3_top(xn[2]); t3 = bf3_top(xn[3]); hn[1] = __builtin_amdgcn_perm(t3, t2, 0x07060302);
    xn[2] -= __uint_as_float(t2); xn[3] -= __uint_as_float(t3);
    t0 = bf3_top(xn[4]); t1 = bf3_top(xn[5]);
  }
  BF3_SB; f6(); BF3_SB;
  if (has_next) {
    hn[2] = __builtin_amdgcn_perm(t1, t0, 0x07060302); xn[4] -= __uint_as_float(t0); xn[5] -= __uint_as_float(t1);
    t2 = bf3_top(xn[6]); t3 = bf3_top(xn[7]); hn[3] = __builtin_amdgcn_perm(t3, t2, 0x07060302);
    xn[6] -= __uint_as_float(t2); xn[7] -= __uint_as_float(t3);
  }
  BF3_SB;
#undef BF3_SB
}

}  // namespace mvn
