// Device helpers shared by the PIPE generator kernels (generate_pipe.hip: fp32, C = 64 / 128;
// generate_pipe_h16.hip: fp16 operands, C = 128): the granule hand-off, cross-lane moves as
// DPP, the gate, and the step-closing choice (double softmax, arg-max / inverse-CDF sample).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "gen_common.h"

namespace mvn {

typedef unsigned long long u64;
typedef float4 f4;

constexpr int PIPE_XCD_CUS = 32;  // CUs per XCD: one workgroup (133 KB of LDS) per CU
constexpr unsigned PIPE_SPIN_LIMIT = 1u << 23;
constexpr int PIPE_MAX_GRAN = 256;

// How the pipelined generators are launched.  Default: hipLaunchCooperativeKernel -- the runtime then guarantees
// what the hand-offs need, every stage of every pipeline co-resident (it refuses the launch otherwise, and does not
// run the grid beside another kernel of the process).  With an ORDINARY launch that guarantee is only the host's
// occupancy check: a second stream or process holding CUs starves a stage, which spins up to PIPE_SPIN_LIMIT polls
// per wait before it raises the sticky status word and the caller reruns on a kernel without hand-offs
// (worst case ~2.3 s per starved wait at ~0.28 us per poll with s_sleep).
// The ordinary launch is taken in exactly two cases:
//   * a profiler's tool library is attached to the process (rocprofv3 / rocprofiler-sdk tool, rocprof v1 / v2): the cooperative
//     queue the HIP runtime creates is torn down at process exit inside libhsa-runtime64 AFTER rocprofiler-sdk has
//     finalised its queue interception -- every rocprofv3 run of a process that had made one cooperative launch
//     ended in SIGSEGV at exit (r3: stack resolved in profiles/r03_exit_crash.md; same step time either way);
//   * MOVENET_PIPE_COOPERATIVE_LAUNCH=0 asks for it (=1 forces the cooperative form even under a profiler).
inline bool pipe_profiler_attached() {
  for (const char *v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "ROCP_METRICS"})
    if (const char *e = getenv(v))
      if (e[0]) return true;
  if (const char *pre = getenv("LD_PRELOAD"))
    if (strstr(pre, "rocprof")) return true;
  bool found = false;
  if (FILE *f = fopen("/proc/self/maps", "r")) {  // a tool library injected any other way
    char line[512];
    while (!found && fgets(line, sizeof line, f))
      // (not libroctracer64 / librocprofiler-register: every PyTorch process maps those)
      found = strstr(line, "librocprofiler-sdk-tool") || strstr(line, "librocprofiler64");
    fclose(f);
  }
  return found;
}
inline bool pipe_cooperative_launch() {
  static const bool on = [] {
    const char *e = getenv("MOVENET_PIPE_COOPERATIVE_LAUNCH");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return !pipe_profiler_attached();
  }();
  return on;
}

#ifdef MVN_PIPE_STAMPS
// Diagnostic build only (python -m movenet_amd.csrc.build --stamps): wall-clock
// (s_memrealtime, 100 MHz) stamps of "inbox complete" and "outbox sent" per stage
// for the first 64 steps of a launch; read back with mvn_debug_read_stamps().
static __device__ unsigned long long g_stamps[16][16][64][4];
#define MVN_STAMP(bb, ss, step, which)                                              \
  do {                                                                              \
    if ((bb) < 16 && (ss) < 16 && (step) < 64 && threadIdx.x == 0) {                \
      g_stamps[bb][ss][step][which] = __builtin_amdgcn_s_memrealtime();             \
      g_stamps[bb][ss][step][2 + (which)] = __builtin_amdgcn_s_memtime();           \
    }                                                                               \
  } while (0)
// finer shader-clock stamps inside the first layer of a stage (thread `who`)
static __device__ unsigned long long g_fine[16][16][64][8];
#define MVN_FINE(bb, ss, step, slot, who)                                           \
  do {                                                                              \
    if ((bb) < 16 && (ss) < 16 && (step) < 64 && threadIdx.x == (who))              \
      g_fine[bb][ss][step][slot] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
#else
#define MVN_STAMP(bb, ss, step, which) do {} while (0)
#define MVN_FINE(bb, ss, step, slot, who) do {} while (0)
#endif

// `same_xcd`: producer and consumer were FOUND (from HW_REG_XCC_ID, exchanged at kernel
// start) to sit on one XCD.  They then share one L2, so a plain store (write-through L1,
// line kept in that L2) is seen by the consumer's L1-bypassing polls: ~0.33 us per hop
// instead of ~0.6.  Otherwise the granule is stored sc1 (write-through to memory), the
// placement-independent form.  Placement only ever selects between two correct forms.
__device__ __forceinline__ void put_granule(u64 *g, unsigned epoch, float v, bool same_xcd) {
  const u64 x = ((u64)epoch << 32) | (u64)__float_as_uint(v);
  // same XCD: a relaxed WORKGROUP-scope atomic store -- on gfx950 the same write-through
  // global_store_dwordx2 as a plain store (line kept in the shared L2), but an atomic in the
  // memory model: never deferred, merged or torn by the compiler
  if (same_xcd)
    __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0 only.  One 16-byte load fetches two granules, and the 64 lanes of a load cover
// 1 KB of the inbox contiguously: lane i owns granules 128*k + 2*i and 128*k + 2*i + 1 for
// k < GL/2 (v[2k], v[2k+1]).  Each granule is still validated by its own epoch word; a
// 16-byte aligned load never tears an 8-byte store.  Returns false on time-out / raised
// error word (wave-uniform).
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Two granules (16 bytes at p, per lane) polled with TWO loads in flight: a poll samples L2
// every half round trip instead of every whole one, which takes ~0.4 round trips off the
// expected detection delay of a hop.  The loop lives in one asm block on fixed registers
// (v240-v247, clobbered): registers with a load in flight must never be copied or renamed by
// the compiler, which a loop-carried C variable cannot promise.  Wave-uniform result: 1 when
// both epochs match in every active lane (v0, v1 = the values), 0 after 64 double polls.
__device__ __forceinline__ int poll16(const u64 *p, unsigned epoch, float &v0, float &v1) {
  int st;
  unsigned r0, r1;
  u64 tmp;  // lane mask scratch
  asm volatile(
      "s_movk_i32 %[st], 64\n\t"
      "global_load_dwordx4 v[240:243], %[p], off sc1\n"
      "1:\n\t"
      "global_load_dwordx4 v[244:247], %[p], off sc1\n\t"
      "s_waitcnt vmcnt(1)\n\t"
      "v_cmp_eq_u32 vcc, %[ep], v241\n\t"
      "v_cmp_eq_u32 %[t], %[ep], v243\n\t"
      "s_and_b64 vcc, vcc, %[t]\n\t"
      "s_cmp_eq_u64 vcc, exec\n\t"
      "s_cbranch_scc1 2f\n\t"
      "global_load_dwordx4 v[240:243], %[p], off sc1\n\t"
      "s_waitcnt vmcnt(1)\n\t"
      "v_cmp_eq_u32 vcc, %[ep], v245\n\t"
      "v_cmp_eq_u32 %[t], %[ep], v247\n\t"
      "s_and_b64 vcc, vcc, %[t]\n\t"
      "s_cmp_eq_u64 vcc, exec\n\t"
      "s_cbranch_scc1 3f\n\t"
      "s_sub_u32 %[st], %[st], 1\n\t"
      "s_cmp_lg_u32 %[st], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b32 %[st], 0\n\t"
      "s_branch 4f\n"
      "2:\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_mov_b32 %[r0], v240\n\t"
      "v_mov_b32 %[r1], v242\n\t"
      "s_mov_b32 %[st], 1\n\t"
      "s_branch 4f\n"
      "3:\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_mov_b32 %[r0], v244\n\t"
      "v_mov_b32 %[r1], v246\n\t"
      "s_mov_b32 %[st], 1\n"
      "4:\n"
      : [st] "=&s"(st), [r0] "=&v"(r0), [r1] "=&v"(r1), [t] "=&s"(tmp)
      : [p] "v"(p), [ep] "s"(epoch)
      : "vcc", "scc", "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
  v0 = __uint_as_float(r0);
  v1 = __uint_as_float(r1);
  return st;
}
// The bounded wait around poll16(): false on time-out / raised error word (wave-uniform).
__device__ __forceinline__ bool wait16(const u64 *p, unsigned epoch, unsigned *err, float &v0, float &v1) {
  for (unsigned calls = 1;; ++calls) {
    if (poll16(p, epoch, v0, v1)) return true;
    const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e != 0 || calls > PIPE_SPIN_LIMIT / 128) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

// The skip granule was sent a stage time ago: its load is issued at the START of the phase
// that ends with its use (the round trip through L2 would otherwise sit on the skip lane, and
// through it on the head); the spin loop is only the fallback.
__device__ __forceinline__ u64 peek_granule(const u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One granule of the skip lane, polled by the lane that owns the channel (off the chain: it was
// sent a whole stage time ago).  A time-out raises the status word; the caller carries on with
// the stale value (the host sees the word).
__device__ __forceinline__ float wait_granule(const u64 *p, unsigned epoch, unsigned *err) {
  for (unsigned spins = 1;; ++spins) {
    unsigned lo, hi;
    {
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      v2u g;
      asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
      lo = g.x;
      hi = g.y;
    }
    if (hi == epoch) return __uint_as_float(lo);
    if ((spins & 255u) == 0) {
      const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || spins > PIPE_SPIN_LIMIT) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __uint_as_float(lo);
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int GL>
__device__ __forceinline__ bool wait_inbox(const u64 *in, unsigned epoch, unsigned *err,
                                           float (&v)[GL]) {
  static_assert(GL == 2 || GL == 4, "one or two 16-byte loads per lane");
  const int lane = threadIdx.x & 63;
  const u64 *p = in + 2 * lane;
  if (GL == 2) return wait16(p, epoch, err, v[0], v[1]);
  for (unsigned spins = 1;; ++spins) {
    v4u g0, g1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g0), "=&v"(g1) : "v"(p) : "memory");
    const bool ok = g0.y == epoch && g0.w == epoch && g1.y == epoch && g1.w == epoch;
    if (__all(ok)) {
      v[0] = __uint_as_float(g0.x);
      v[1] = __uint_as_float(g0.z);
      v[GL - 2] = __uint_as_float(g1.x);
      v[GL - 1] = __uint_as_float(g1.z);
      return true;
    }
    if ((spins & 255u) == 0) {
      const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || spins > PIPE_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// 64 granules (512 bytes) of an inbox: lanes 0-31 take two each with one 16-byte load (lanes
// 32-63 repeat them).  v[0], v[1] = granules 2 (lane & 31), + 1 of `in`.  Wave-uniform result.
__device__ __forceinline__ bool wait_inbox64(const u64 *in, unsigned epoch, unsigned *err, float (&v)[2]) {
  return wait16(in + 2 * (threadIdx.x & 31), epoch, err, v[0], v[1]);
}

// ---- cross-lane moves as DPP (one VALU op) instead of ds_bpermute (an LDS round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each group of 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each row of 16

// sum over each aligned group of 4 lanes, result in all 4
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<DPP_XOR1>(v);
  v += dpp_mov<DPP_XOR2>(v);
  return v;
}
// after quad_sum: the value held by the OTHER quad of the same group of 8
__device__ __forceinline__ float other_quad(float v) { return dpp_mov<DPP_HALF_MIRROR>(v); }

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = quad_sum(v);
  v += dpp_mov<DPP_HALF_MIRROR>(v);
  v += dpp_mov<DPP_MIRROR>(v);  // every lane of a row of 16 now holds the row sum
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// tanh(f) * sigmoid(g) from two v_exp_f32 and one reciprocal-based division:
//   tanh(f) = (1 - e^-2|f|) / (1 + e^-2|f|) * sign(f),  sigmoid(g) = 1 / (1 + e^-g)
// Absolute error ~1e-7 (fp32 rounding of an O(1) value); ~12 VALU ops vs ~100 for
// the libm forms, and this sits on the per-layer critical path.
__device__ __forceinline__ float gate_fast(float f, float g) {
  const float a = __expf(-2.0f * fabsf(f));       // in (0, 1]
  const float e = __expf(-g);                      // may overflow to +inf: 1/inf = 0 is right
  const float num = copysignf(1.0f - a, f);
  const float den = (1.0f + a) * (1.0f + e);
  return num * __builtin_amdgcn_rcpf(den);  // v_rcp_f32: 1 ulp
}

// (value, index) arg-max combine: larger value wins, smaller index on ties
__device__ __forceinline__ void argmax_take(float &bv, int &bi, float ov, int oi) {
  if (ov > bv || (ov == bv && oi < bi)) {
    bv = ov;
    bi = oi;
  }
}
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}


typedef float v2f __attribute__((ext_vector_type(2)));

// N4*4-term dot product as packed FMAs (v_pk_fma_f32) in two (N4 = 4) or four independent
// chains, combined in a fixed order.  w: 2*N4 float2, x: the inputs already fetched from LDS.
template <int N4>
__device__ __forceinline__ float dotn(const v2f (&w)[2 * N4], const f4 (&x)[N4]) {
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < N4; i += 2) {
    a0 = __builtin_elementwise_fma(w[2 * i], v2f{x[i].x, x[i].y}, a0);
    a1 = __builtin_elementwise_fma(w[2 * i + 1], v2f{x[i].z, x[i].w}, a1);
    a2 = __builtin_elementwise_fma(w[2 * i + 2], v2f{x[i + 1].x, x[i + 1].y}, a2);
    a3 = __builtin_elementwise_fma(w[2 * i + 3], v2f{x[i + 1].z, x[i + 1].w}, a3);
  }
  const v2f t = (a0 + a2) + (a1 + a3);
  return t.x + t.y;
}
template <int N4>
__device__ __forceinline__ void ldsn(f4 (&x)[N4], const float *p) {
#pragma unroll
  for (int i = 0; i < N4; ++i) x[i] = ((const f4 *)p)[i];
}
// N4 float4 of a [..][stride] block -> 2*N4 float2 registers
template <int N4>
__device__ __forceinline__ void loadn(v2f (&w)[2 * N4], const f4 *src, int stride, int idx) {
#pragma unroll
  for (int i = 0; i < N4; ++i) {
    const f4 v = src[i * stride + idx];
    w[2 * i] = v2f{v.x, v.y};
    w[2 * i + 1] = v2f{v.z, v.w};
  }
}
// Two rows against the same LDS vector, the vector fetched four float4 at a time with one
// chunk of look-ahead: at N4 = 16 only 32-48 of its 64 registers are live at once (the
// whole vector next to 128 weight registers spills).  Same accumulation order as dotn.
template <int N4>
__device__ __forceinline__ void dot2_lds(const v2f (&wa)[2 * N4], const v2f (&wb)[2 * N4],
                                         const float *xp, float &ra, float &rb) {
  constexpr int CH = 4, NCH = N4 / CH;
  static_assert(N4 % CH == 0, "whole chunks");
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
  v2f b0 = {0.f, 0.f}, b1 = {0.f, 0.f}, b2 = {0.f, 0.f}, b3 = {0.f, 0.f};
  f4 x[N4];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = ((const f4 *)xp)[i];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) {
#pragma unroll
      for (int i = CH * (ch + 1); i < CH * (ch + 2); ++i) x[i] = ((const f4 *)xp)[i];
    }
    if (NCH > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = CH * ch; i < CH * (ch + 1); i += 2) {
      const v2f x0 = {x[i].x, x[i].y}, x1 = {x[i].z, x[i].w};
      const v2f x2 = {x[i + 1].x, x[i + 1].y}, x3 = {x[i + 1].z, x[i + 1].w};
      a0 = __builtin_elementwise_fma(wa[2 * i], x0, a0);
      a1 = __builtin_elementwise_fma(wa[2 * i + 1], x1, a1);
      a2 = __builtin_elementwise_fma(wa[2 * i + 2], x2, a2);
      a3 = __builtin_elementwise_fma(wa[2 * i + 3], x3, a3);
      b0 = __builtin_elementwise_fma(wb[2 * i], x0, b0);
      b1 = __builtin_elementwise_fma(wb[2 * i + 1], x1, b1);
      b2 = __builtin_elementwise_fma(wb[2 * i + 2], x2, b2);
      b3 = __builtin_elementwise_fma(wb[2 * i + 3], x3, b3);
    }
    if (NCH > 1) __builtin_amdgcn_sched_barrier(0);
  }
  const v2f ta = (a0 + a2) + (a1 + a3), tb = (b0 + b2) + (b1 + b3);
  ra = ta.x + ta.y;
  rb = tb.x + tb.y;
}
// one row of dot2_lds (same chunking, same accumulation order as its `a` half)
template <int N4>
__device__ __forceinline__ float dot1_lds(const v2f (&wa)[2 * N4], const float *xp) {
  constexpr int CH = 4, NCH = N4 / CH;
  static_assert(N4 % CH == 0, "whole chunks");
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
  f4 x[N4];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = ((const f4 *)xp)[i];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) {
#pragma unroll
      for (int i = CH * (ch + 1); i < CH * (ch + 2); ++i) x[i] = ((const f4 *)xp)[i];
    }
    if (NCH > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = CH * ch; i < CH * (ch + 1); i += 2) {
      a0 = __builtin_elementwise_fma(wa[2 * i], v2f{x[i].x, x[i].y}, a0);
      a1 = __builtin_elementwise_fma(wa[2 * i + 1], v2f{x[i].z, x[i].w}, a1);
      a2 = __builtin_elementwise_fma(wa[2 * i + 2], v2f{x[i + 1].x, x[i + 1].y}, a2);
      a3 = __builtin_elementwise_fma(wa[2 * i + 3], v2f{x[i + 1].z, x[i + 1].w}, a3);
    }
    if (NCH > 1) __builtin_amdgcn_sched_barrier(0);
  }
  const v2f ta = (a0 + a2) + (a1 + a3);
  return ta.x + ta.y;
}
// dotn() with the weights fetched on the fly (LDS or L2), four float4 at a time so that
// only 32 registers are live; same accumulators and order as dotn: bit-identical to it
template <int N4>
__device__ __forceinline__ float dot_stream(const f4 *wsrc, int stride, int idx, const float *xsrc) {
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
  for (int i0 = 0; i0 < N4; i0 += 4) {
    f4 w[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = wsrc[(i0 + i) * stride + idx];
      x[i] = ((const f4 *)xsrc)[i0 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      a0 = __builtin_elementwise_fma(v2f{w[i].x, w[i].y}, v2f{x[i].x, x[i].y}, a0);
      a1 = __builtin_elementwise_fma(v2f{w[i].z, w[i].w}, v2f{x[i].z, x[i].w}, a1);
      a2 = __builtin_elementwise_fma(v2f{w[i + 1].x, w[i + 1].y}, v2f{x[i + 1].x, x[i + 1].y}, a2);
      a3 = __builtin_elementwise_fma(v2f{w[i + 1].z, w[i + 1].w}, v2f{x[i + 1].z, x[i + 1].w}, a3);
    }
  }
  const v2f t = (a0 + a2) + (a1 + a3);
  return t.x + t.y;
}
// sum over the KQ lanes that share a channel (result in all of them)
template <int KQ>
__device__ __forceinline__ float chan_sum(float v) {
  v += dpp_mov<DPP_XOR1>(v);
  if (KQ == 4) v += dpp_mov<DPP_XOR2>(v);
  return v;
}


// Wave 0 closes a step alone: lane i owns the logits of classes 4i .. 4i+3 (Q = 256); every
// reduction is intra-wave (DPP + readlane), no barrier.  Returns the class WaveNet.generate
// picks for time u: softmax (wavenet.py:189-191), [/ T], softmax again, then multinomial by
// inverse CDF on the Philox uniform of (seed, u, b) or the first arg-max (:227-233).
// v_exp_f32 / v_rcp_f32 forms (1-2 ulp): the choice depends on the ORDER of the
// probabilities, which these monotone maps preserve.
// Greedy decoding, the common case: when the largest logit leads the runner-up by a clear margin,
// the first arg-max of softmax(softmax(logits)) IS the arg-max of the logits, and ONE wave-wide
// reduction (value, index, runner-up) replaces the max, two sums and the arg-max of the full
// form below (~850 -> ~350 cycles of the head stage).  Why 1e-3 is clear: e = exp(l - max) of the
// runner-up is <= e^-0.001 = 0.999 (v_exp_f32 errs by 1-2 ulp of 6e-8), the first softmax keeps the
// ratio and its largest probability is >= 1/Q = 1/256, so the second softmax's inputs differ by
// >= 3.9e-6 and its exponentials by >= 60 ulp below 1.0; the final scaling is monotone.  Ties and
// near-ties (margin below 1e-3, NaNs) return false: the caller runs the full form.
__device__ __forceinline__ bool greedy_pick_clear(const float (&lg)[4], int lane, int &pick) {
  float bv = lg[0], sv = -INFINITY;
  int bi = 4 * lane;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const bool up = lg[k] > bv;
    sv = up ? bv : fmaxf(sv, lg[k]);
    bi = up ? 4 * lane + k : bi;
    bv = up ? lg[k] : bv;
  }
  auto merge = [&](float ov, int oi, float os) {
    const bool up = ov > bv;
    sv = fmaxf(fmaxf(sv, os), up ? bv : ov);  // (equal maxima: the runner-up becomes the maximum, margin 0)
    bi = up ? oi : bi;
    bv = up ? ov : bv;
  };
  merge(dpp_mov<DPP_XOR1>(bv), dpp_movi<DPP_XOR1>(bi), dpp_mov<DPP_XOR1>(sv));
  merge(dpp_mov<DPP_XOR2>(bv), dpp_movi<DPP_XOR2>(bi), dpp_mov<DPP_XOR2>(sv));
  merge(dpp_mov<DPP_HALF_MIRROR>(bv), dpp_movi<DPP_HALF_MIRROR>(bi), dpp_mov<DPP_HALF_MIRROR>(sv));
  merge(dpp_mov<DPP_MIRROR>(bv), dpp_movi<DPP_MIRROR>(bi), dpp_mov<DPP_MIRROR>(sv));
  // every lane of a row of 16 now holds the row's triple; the four rows meet as scalars
  float rv = lane_value(bv, 0), rs = lane_value(sv, 0);
  int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
  for (int row = 16; row < 64; row += 16) {
    const float ov = lane_value(bv, row), os = lane_value(sv, row);
    const int oi = __builtin_amdgcn_readlane(bi, row);
    const bool up = ov > rv;
    rs = fmaxf(fmaxf(rs, os), up ? rv : ov);
    ri = up ? oi : ri;
    rv = up ? ov : rv;
  }
  pick = ri;
  return rv - rs >= 1e-3f;
}

// `uniform`: philox_uniform(seed, u, b), formed by the caller BEFORE it waits for the step's input (ten
// Philox rounds of integer multiplies: ~0.15 us that do not depend on the logits).
__device__ __forceinline__ int choose_class(const float (&lg)[4], float temperature, float uniform, int lane, int Q) {
  if (!(temperature > 0.f)) {
    int fast;
    if (greedy_pick_clear(lg, lane, fast)) return fast;
  }
  const float m = wave_max_dpp(fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3])));
  float e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = __expf(lg[k] - m);
  const float sm = wave_sum_dpp((e[0] + e[1]) + (e[2] + e[3]));
  const float rs = __builtin_amdgcn_rcpf(sm) * (temperature > 0.f ? __builtin_amdgcn_rcpf(temperature) : 1.0f);
  float p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = e[k] * rs;
  // max over p = rs exactly: the largest e is exp(0) = 1.0 (v_exp_f32 of 0 is exact), 1.0 * rs = rs,
  // and rounding is monotone for the others (e <= 1)  -- one wave-wide max reduction less per step
  const float m2 = rs;
  // (classes >= Q are the padding of a head that runs 256 wide for a smaller model: logit -inf, p = 0 -- and no mass
  // in this second softmax either, where exp(0 - m2) would give each of them some)
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = 4 * lane + k < Q ? __expf(p[k] - m2) : 0.f;
  const float s2sum = wave_sum_dpp((e[0] + e[1]) + (e[2] + e[3]));
  const float rs2 = __builtin_amdgcn_rcpf(s2sum);
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = e[k] * rs2;  // the distribution generate() uses

  int pick;
  if (temperature > 0.f) {
    const float lsum = (p[0] + p[1]) + (p[2] + p[3]);
    // inclusive scan of the lane totals in class order: four DPP row shifts inside each row of 16
    // lanes (zeros shifted in), then the totals of the rows below as scalars (six __shfl_up steps
    // = six LDS-crossbar round trips before: ~0.25 us of every sampled step)
    float incl = lsum;
    incl += dpp_mov<0x111>(incl);  // row_shr:1
    incl += dpp_mov<0x112>(incl);  // row_shr:2
    incl += dpp_mov<0x114>(incl);  // row_shr:4
    incl += dpp_mov<0x118>(incl);  // row_shr:8
    {
      const float r0 = lane_value(incl, 15), r1 = lane_value(incl, 31), r2 = lane_value(incl, 47);
      const int row = lane >> 4;
      incl += row == 0 ? 0.f : row == 1 ? r0 : row == 2 ? r0 + r1 : (r0 + r1) + r2;
    }
    const float total = lane_value(incl, 63);
    const float target = uniform * total;
    const float cdf = incl - lsum;
    int cand = Q - 1;
    bool hit = false;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      // walk down so that the smallest qualifying class wins
      const float c_k = cdf + (k == 0 ? p[0] : k == 1 ? p[0] + p[1]
                                              : k == 2 ? (p[0] + p[1]) + p[2]
                                                       : ((p[0] + p[1]) + p[2]) + p[3]);
      if (c_k > target) {
        cand = 4 * lane + k;
        hit = true;
      }
    }
    // the smallest qualifying class over the wave = the candidate of the FIRST lane that has one
    // (classes ascend with the lane): one ballot and one readlane instead of a min-reduction
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(hit);
    pick = lanes ? __builtin_amdgcn_readlane(cand, __builtin_ctzll(lanes)) : Q - 1;
  } else {
    float bv = p[0];
    int bi = 4 * lane;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (p[k] > bv) {  // strict: first maximum
        bv = p[k];
        bi = 4 * lane + k;
      }
    argmax_take(bv, bi, dpp_mov<DPP_XOR1>(bv), dpp_movi<DPP_XOR1>(bi));
    argmax_take(bv, bi, dpp_mov<DPP_XOR2>(bv), dpp_movi<DPP_XOR2>(bi));
    argmax_take(bv, bi, dpp_mov<DPP_HALF_MIRROR>(bv), dpp_movi<DPP_HALF_MIRROR>(bi));
    argmax_take(bv, bi, dpp_mov<DPP_MIRROR>(bv), dpp_movi<DPP_MIRROR>(bi));
    float rv = lane_value(bv, 0);
    pick = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int row = 16; row < 64; row += 16)
      argmax_take(rv, pick, lane_value(bv, row), __builtin_amdgcn_readlane(bi, row));
  }
  return pick;
}

}  // namespace mvn
