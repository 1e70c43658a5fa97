// common.h — shared device helpers for libmpn_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/mpn.h"

typedef unsigned short bf16_t;   // raw bfloat16 storage
typedef _Float16 f16_t;          // IEEE half (dtype code MPN_F16: the inference arithmetic of BASELINE config 5)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even (the rule torch uses for float -> bfloat16); gfx950 has it in hardware
// (v_cvt_pk_bf16_f32, two values per instruction), NaNs come back quiet.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = MPN_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = MPN_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

template <> struct Elem<f16_t> {
    static constexpr int kDtype = MPN_F16;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }      // v_cvt_f16_f32: round to nearest even
    __device__ static __forceinline__ float round(float v) { return (float)(f16_t)v; }
};

// 8-element (bf16 / f16) / 4-element (f32) 16-byte vectors viewed as floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
        v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        uint4 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        t.z = pack_bf16x2(v[4], v[5]);
        t.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = t;
    }
};

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const f16_t* p) {
        const f16x8_t t = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
    __device__ __forceinline__ void store(f16_t* p) const {
        f16x8_t t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (f16_t)v[k];
        *reinterpret_cast<f16x8_t*>(p) = t;
    }
};

// XCD-aware bijective block remap (8 XCDs; block b is observed to run on XCD b % 8): blocks that
// land on one XCD get a contiguous range of logical tile ids, so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, local = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Experiments build (`make experiments` -> libmpn_hip_experiments.so, compiled with -DMPN_EXPERIMENTS; tools/ load it explicitly):
// the s_memtime PROF instantiations of the hot kernels, the timing-only ablation bits (MPN_DEBUG_FLAGS, MPN_WGRAD_ABLATE) and the
// environment overrides of tuned constants.  The production library (`make`, what the package loads) compiles none of them:
// every mpn_tune() is its default, every MPN_DBG() is false.
#ifdef MPN_EXPERIMENTS
#include <stdlib.h>
#define MPN_EXP 1
static inline long mpn_tune(const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; }
#define MPN_DBG(bit) ((dbg & (bit)) != 0)
#else
#define MPN_EXP 0
static inline long mpn_tune(const char*, long dflt) { return dflt; }
#define MPN_DBG(bit) false
#endif

static inline int mpn_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

#define MPN_CHECK_ARG(cond) do { if (!(cond)) return MPN_E_BADARG; } while (0)

static inline bool mpn_dtype_ok(int dtype) { return dtype == MPN_F32 || dtype == MPN_BF16 || dtype == MPN_F16; }

// run `...` with T bound to the element type of dtype code `dtype` (callers validate the code first)
#define MPN_DISPATCH_T(dtype, ...)                                              \
    do {                                                                        \
        if ((dtype) == MPN_F32) { using T = float; __VA_ARGS__; }               \
        else if ((dtype) == MPN_BF16) { using T = bf16_t; __VA_ARGS__; }        \
        else { using T = f16_t; __VA_ARGS__; }                                  \
    } while (0)
