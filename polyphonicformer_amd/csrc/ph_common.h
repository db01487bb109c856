// Shared device helpers for libpolyhead (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/polyhead.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

#define PH_LDS __attribute__((address_space(3)))

void ph_set_error(const char* fmt, ...);

#define PH_CHECK_ARG(cond, msg)                                        \
    do {                                                               \
        if (!(cond)) {                                                 \
            ph_set_error("%s: %s", __func__, msg);                     \
            return PH_EINVAL;                                          \
        }                                                              \
    } while (0)

#define PH_CHECK_LAUNCH()                                                          \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ph_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return PH_ELAUNCH;                                                     \
        }                                                                          \
    } while (0)

// ---- the hard mask threshold (kernel_update_head.py:236-238, kernel_head.py:314-317): `sigmoid(z) > 0.5` in fp32.
// Evaluated as the reference does -- 1 / (1 + exp(-z)), every operation rounded to fp32 -- the comparison is true exactly
// for z > 1.5 * 2^-24: below that 1 - z rounds to 1 - 2^-24 or 1, 2 - 2^-24 rounds to 2, and 1 / 2 = 0.5.  (Probe on
// torch's CPU sigmoid: smallest z with sigmoid(z) > 0.5 is 97 * 2^-30.)  `z > 0` would differ on 0 < z <= 1.5 * 2^-24:
// 3.6e-8 of N(0, 1) logits, i.e. about one pixel in four full-size frames -- enough to move a pooled feature by 1e-3.
#define PH_BIN_THR 0x1.8p-24f

// ---- bf16 bit helpers (round to nearest even; inputs are finite in this code base) ----------
// gfx950 has a hardware round-to-nearest-even conversion (v_cvt_pk_bf16_f32); the compiler selects
// it for fp32 -> __bf16 conversions.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)x); }
// two conversions in one instruction: low half = a, high half = b
__device__ __forceinline__ uint32_t f2bf_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
// x ~= hi + lo, |x - hi - lo| <= 2^-17 |x|
__device__ __forceinline__ void f2bf_split(float x, uint32_t& hi, uint32_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return a | (b << 16); }

// ---- 16-bit element formats of feature planes / dynamic kernels / outputs: bf16 (8-bit mantissa, fp32 range) or IEEE
// fp16 (11-bit mantissa, |x| < 65504).  A bf16 value is exactly representable in fp16 when it is inside fp16's normal
// range, so an fp16 plane carries bf16-rounded inputs unchanged and fp32 inputs with 8x finer rounding.
enum { PH_E_BF16 = 0, PH_E_F16 = 1, PH_E_F16_FROM_BF16 = 2 /* conv only: bf16 feature fragments converted to fp16 in registers */ };
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2h(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)x); }
__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
__device__ __forceinline__ uint32_t f2h_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}
template <int E> __device__ __forceinline__ uint32_t f2e(float x) { return E == PH_E_F16 ? f2h(x) : f2bf(x); }
template <int E> __device__ __forceinline__ float e2f(uint32_t h) { return E == PH_E_F16 ? h2f(h) : bf2f(h); }
// x -> (hi, lo) planes of format E: bf16 hi + lo (lo unused by one-plane callers), or ONE fp16 value (lo = 0)
template <int E> __device__ __forceinline__ void f2e_split(float x, uint32_t& hi, uint32_t& lo) {
    if constexpr (E == PH_E_F16) { hi = f2h(x); lo = 0; }
    else f2bf_split(x, hi, lo);
}
template <int E> __device__ __forceinline__ uint32_t f2e_pk(float a, float b) { return E == PH_E_F16 ? f2h_pk(a, b) : f2bf_pk(a, b); }

// ---- streaming (non-temporal) 16-byte accesses for data that is written or read exactly once per kernel: the x2
// upsample's 965 MB of output per launch went from 5.4 to 6.9 TB/s with them (stores no longer allocate in L2 / MALL)
// cache policy of the LDS-DMA feature streams (aux operand of global_load_lds on gfx950: 1 = sc0, 2 = nt, 16 = sc1):
// nt | sc1 measured pool 197 -> 181 us, dynconv bits 102 -> 94 us against the default policy; sc1 alone is slower
#ifndef PH_CPOL_STREAM      // -DPH_CPOL_STREAM=0: default policy (tools/mall_probe.py)
#define PH_CPOL_STREAM 18
#endif
typedef unsigned ph_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt16(void* p, uint4 v) { __builtin_nontemporal_store(ph_u32x4{v.x, v.y, v.z, v.w}, (ph_u32x4*)p); }
__device__ __forceinline__ void st_nt16(void* p, float4 v) {
    __builtin_nontemporal_store(ph_u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, (ph_u32x4*)p);
}
typedef unsigned ph_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 ld_nt8(const void* p) {
    const ph_u32x2 v = __builtin_nontemporal_load((const ph_u32x2*)p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
    const ph_u32x4 v = __builtin_nontemporal_load((const ph_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- MFMA wrappers.  Fragment maps (verified on hardware by ph_selftest_*):
//  16x16x32: A lane l: row l&15, k = (l>>4)*8 + e;  B lane l: col l&15, k = (l>>4)*8 + e;
//            D lane l, reg r: row (l>>4)*4 + r, col l&15.
//  32x32x16: A lane l: row l&31, k = (l>>5)*8 + e;  B lane l: col l&31, k = (l>>5)*8 + e;
//            D lane l, reg r: row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// two bf16 in a dword -> two fp16 (exact whenever the value is inside fp16's normal range; bf16 has 8 significand bits)
__device__ __forceinline__ uint32_t bf2h_pk(uint32_t w) { return f2h_pk(__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)); }
__device__ __forceinline__ uint4 bf2h_x8(uint4 v) { return make_uint4(bf2h_pk(v.x), bf2h_pk(v.y), bf2h_pk(v.z), bf2h_pk(v.w)); }

template <int E> __device__ __forceinline__ f32x4_t mfma16e(uint4 a, uint4 b, f32x4_t c) {
    if constexpr (E == PH_E_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return mfma16(a, b, c);
}
template <int E> __device__ __forceinline__ f32x16_t mfma32e(uint4 a, uint4 b, f32x16_t c) {
    if constexpr (E == PH_E_F16 || E == PH_E_F16_FROM_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return mfma32(a, b, c);
}

// Transposing LDS read: within each 16-lane group the lanes point at 16 8-byte chunks forming a
// [4 rows][16 cols] bf16 block (lane i -> row i>>2, cols (i&3)*4..+3); lane i receives column i,
// i.e. element j = block[j][i].
__device__ __forceinline__ uint2 lds_read_tr16(const uint16_t* p) {
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((PH_LDS bf16x4_t*)(p));
    return __builtin_bit_cast(uint2, v);
}

// 16-lane butterflies on the DPP crossbar (no LDS traffic): quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror.  Every lane of a 16-lane row ends with the same value, and the pairing
// (hence the fp result) is that of an xor-1/2/4/8 butterfly.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_group16_sum(float t) {
    t += dpp_mov<0xB1>(t);
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0x141>(t);
    t += dpp_mov<0x140>(t);
    return t;
}
__device__ __forceinline__ float wave_group16_max(float t) {
    t = fmaxf(t, dpp_mov<0xB1>(t));
    t = fmaxf(t, dpp_mov<0x4E>(t));
    t = fmaxf(t, dpp_mov<0x141>(t));
    t = fmaxf(t, dpp_mov<0x140>(t));
    return t;
}
// 1 ulp hardware transcendentals (v_exp_f32 / v_rcp_f32 / v_rsq_f32): ~2e-7 relative, far inside both
// precision modes' budgets, and an order of magnitude fewer VALU slots than the IEEE-exact library forms.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.f + fast_exp(-x)); }
