// hz_common.h -- shared host/device helpers of libhanabizero_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hz_tiebreak.h"

#define HZ_FLOAT_MAX 1000000.0f  // reference core/ctree/cminimax.h:7
#define HZ_FLOAT_MIN (-HZ_FLOAT_MAX)

// ---- error plumbing: nothing in this library aborts (SURVEY 8b "what the build exports instead") ----
void hz_set_error(const char* fmt, ...);

#define HZ_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      hz_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return -2;                                                                       \
    }                                                                                  \
  } while (0)

#define HZ_REQUIRE(cond, ...)   \
  do {                          \
    if (!(cond)) {              \
      hz_set_error(__VA_ARGS__); \
      return -1;                \
    }                           \
  } while (0)

// ---- wave64 helpers --------------------------------------------------------------------------
#define HZ_WAVE 64

__device__ __forceinline__ float hz_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int hz_readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int hz_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- DPP reductions (no LDS traffic): after the four steps below every lane of a 16-lane row holds the row's result
//      quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
#define HZ_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float hz_row16_max(float v) {
  v = fmaxf(v, HZ_DPP(v, 0xB1));
  v = fmaxf(v, HZ_DPP(v, 0x4E));
  v = fmaxf(v, HZ_DPP(v, 0x141));
  v = fmaxf(v, HZ_DPP(v, 0x140));
  return v;
}
__device__ __forceinline__ float hz_row16_min(float v) {
  v = fminf(v, HZ_DPP(v, 0xB1));
  v = fminf(v, HZ_DPP(v, 0x4E));
  v = fminf(v, HZ_DPP(v, 0x141));
  v = fminf(v, HZ_DPP(v, 0x140));
  return v;
}
__device__ __forceinline__ float hz_row16_sum(float v) {
  v += HZ_DPP(v, 0xB1);
  v += HZ_DPP(v, 0x4E);
  v += HZ_DPP(v, 0x141);
  v += HZ_DPP(v, 0x140);
  return v;
}

// Left-to-right sum of x over lanes 0 .. n-1 (x = +0.0f in lanes that do not take part), seeded with `init`, in exactly
// the order a serial loop over the lanes adds them: a systolic pass -- every step each lane adds its x to what its left
// neighbour held (v_add_f32 with DPP wave_shr:1; lane 0 takes `init`) -- after which lane i holds init + x_0 + ... + x_i.
// n steps of one VALU instruction instead of n trips of a ballot-walking readlane loop.  Returns the total (lane n-1's).
// Adding +0.0f for a skipped lane is exact as long as the running sum is not -0.0f, which a sum seeded with a
// non-negative `init` never is.
__device__ __forceinline__ float hz_ordered_sum(float x, int n, float init) {
  float acc = 0.0f;
  // (steps beyond n change nothing in lanes < n -- each recomputes the value it already holds -- so the loop runs in
  // groups of four without a remainder)
  for (int s = 0; s < n; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(init), __float_as_int(acc), 0x138, 0xf, 0xf, false)) + x;
  }
  return hz_readlane_f(acc, n - 1);
}

// max / min over the 64 lanes (callers mask with +-inf; no NaNs): the classic GCN reduction, one DPP-modified v_max / v_min per
// step -- four butterflies inside the 16-lane rows, row_bcast:15 and row_bcast:31 across them; lane 63 ends up with the result.
// Hand-written because the compiler turns fmaxf(v, dpp(v)) into mov_dpp + a canonicalising max + the max (the tree phases of
// the persistent search kernel are bound by instruction issue: four waves share a SIMD's port).  All 64 lanes must be active.
#define HZ_WAVE_REDUCE(OP)                                                                                            \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"              \
               OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                            \
               OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                \
               OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                     \
               OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                                   \
               OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"                                                   \
               : "+v"(v))
__device__ __forceinline__ float hz_wave_max(float v) {
  HZ_WAVE_REDUCE("v_max_f32_dpp");
  return hz_readlane_f(v, 63);
}
__device__ __forceinline__ float hz_wave_min(float v) {
  HZ_WAVE_REDUCE("v_min_f32_dpp");
  return hz_readlane_f(v, 63);
}

// ---- expf, bit-identical to glibc >= 2.27 expf (sysdeps/ieee754/flt-32/e_expf.c, the x86-64 FMA ifunc
//      variant every FMA-capable host selects).  The reference calls libm expf at core/ctree/cnode.cpp:87.
//      Published algorithm (Szabolcs Nagy, ARM optimized-routines math/expf.c): x*N/ln2 = k + r,
//      exp(x) = 2^(k/N) * (C0 r^3 + C1 r^2 + C2 r + 1) evaluated in double, N = 32.
//      tests/test_hip_math.py checks all 2^32 inputs against the host libm on the GPU box.
__device__ __constant__ const uint64_t hz_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// `tab`: the 32-entry table above, or a copy of it that is cheaper to gather from (LDS in the persistent search kernel)
__device__ __forceinline__ float hz_expf(float x, const uint64_t* tab = hz_exp2f_tab) {
  const uint32_t ux = __float_as_uint(x);
  const uint32_t abstop = (ux >> 20) & 0x7ff;
  if (abstop >= (0x42b00000u >> 20)) {  // |x| >= 88 or NaN
    if (ux == 0xff800000u) return 0.0f;                 // -inf
    if (abstop >= (0x7f800000u >> 20)) return x + x;    // +inf / NaN
    if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                // underflow
  }
  const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
  const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
  const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  const double xd = (double)x;
  double z = InvLn2N * xd;
  double kd = z + Shift;
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd -= Shift;
  const double r = __builtin_fma(InvLn2N, xd, -kd);  // as contracted by the host build
  uint64_t t = tab[ki & 31];
  t += ki << (52 - 5);
  const double s = __longlong_as_double((long long)t);
  double c1 = C1;  // (pinned to scalar registers here: hoisted out of a caller's loop as a vector-register constant it
  asm volatile("" : "+s"(c1));  //  ends up spilled to scratch, a memory round trip in the middle of the expansion)
  z = __builtin_fma(C0, r, c1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

// ---- one row of bytes moved by a whole workgroup, at the widest unit source, destination and length allow
__device__ __forceinline__ void copy_row_block(const uint8_t* a, uint8_t* b, long long n, int tid, int nthreads) {
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)n) & 15) == 0) {
    for (long long off = (long long)tid * 16; off < n; off += (long long)nthreads * 16)
      *reinterpret_cast<uint4*>(b + off) = *reinterpret_cast<const uint4*>(a + off);
  } else if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)n) & 3) == 0) {
    for (long long off = (long long)tid * 4; off < n; off += (long long)nthreads * 4)
      *reinterpret_cast<uint32_t*>(b + off) = *reinterpret_cast<const uint32_t*>(a + off);
  } else {
    for (long long off = tid; off < n; off += nthreads) b[off] = a[off];
  }

}
