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

}  // namespace mvn
