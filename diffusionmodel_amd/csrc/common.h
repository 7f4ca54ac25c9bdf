// Shared device helpers for libdm_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dm_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DM_WAVE 64

void dm_set_error(const char* fmt, ...);
#define DM_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            dm_set_error(__VA_ARGS__);   \
            return DM_EINVAL;            \
        }                                \
    } while (0)
#define DM_LAUNCH_CHECK()                                                   \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            dm_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                \
        }                                                                   \
    } while (0)

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VE = 4;  // elements per 16-byte vector
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16> {
    static constexpr int VE = 8;
    __device__ static inline float ld(const bf16* p) { return (float)*p; }
    __device__ static inline void st(bf16* p, float v) { *p = (bf16)v; }
};

template <> struct Elem<f16> {
    static constexpr int VE = 8;
    __device__ static inline float ld(const f16* p) { return (float)*p; }
    __device__ static inline void st(f16* p, float v) { *p = (f16)v; }
};
// the 2 / 4 / 8-element vector types of a 16-bit storage type (generic epilogues)
template <typename T> struct V16;
template <> struct V16<bf16> { typedef bf16x2 x2; typedef bf16x4 x4; typedef bf16x8 x8; };
template <> struct V16<f16> { typedef f16x2 x2; typedef f16x4 x4; typedef f16x8 x8; };

__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float gelu_grad_f(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ inline float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ inline float act_apply(float x, int act) {
    switch (act) {
        case DM_ACT_GELU: return gelu_f(x);
        case DM_ACT_RELU: return x > 0.f ? x : 0.f;
        case DM_ACT_SIGMOID: return sigmoid_f(x);
        default: return x;
    }
}
// derivative w.r.t. the pre-activation value x
__device__ inline float act_grad(float x, int act) {
    switch (act) {
        case DM_ACT_GELU: return gelu_grad_f(x);
        case DM_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case DM_ACT_SIGMOID: { float s = sigmoid_f(x); return s * (1.f - s); }
        default: return 1.f;
    }
}

// bf16-mode GELU: Abramowitz-Stegun 7.1.26 erf (|err| <= 1.5e-7, far below bf16 rounding) on v_rcp/v_exp —
// ~20 VALU instead of libm erff's ~70, which is what made the BN+GELU streaming kernels VALU-bound, not
// HBM-bound.  The exponential exp(-x^2/2) is shared with the Gaussian pdf of the derivative.  fp32 mode
// keeps libm erff (the 1e-4 parity mode).
__device__ inline void gelu_parts_fast(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    e = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
}
// Two evaluations side by side: the polynomial and the affine steps run on v_pk_fma_f32 / v_pk_mul_f32 (two results per issue
// slot), v_rcp_f32 / v_exp_f32 stay one per element.  The BN + GELU streaming kernels are VALU-bound in 16-bit mode (33.5 M elements
// x ~30 issue slots against 4 SIMDs x 256 CUs = 30 us on a 64x64x128 layer that streams in 25 us): this is what brings them back
// under the HBM time.  Same formula as gelu_parts_fast.
__device__ inline void gelu_parts_fast2(const f32x2 x, f32x2& cdf, f32x2& e) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 d = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 a = (x * x) * (-0.5f * 1.4426950408889634f);               // exp(-x^2 / 2) = exp2(-x^2 / 2 * log2 e)
    e = (f32x2){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x2 q = 1.0f - poly * e;
    const f32x2 sg = {copysignf(q[0], x[0]), copysignf(q[1], x[1])};
    cdf = sg * 0.5f + 0.5f;
}
template <typename T> __device__ inline float act_apply_t(float x, int act) {
    if constexpr (sizeof(T) == 2) {
        if (act == DM_ACT_GELU) { float cdf, e; gelu_parts_fast(x, cdf, e); return x * cdf; }
    }
    return act_apply(x, act);
}
template <typename T> __device__ inline float act_grad_t(float x, int act) {
    if constexpr (sizeof(T) == 2) {
        if (act == DM_ACT_GELU) { float cdf, e; gelu_parts_fast(x, cdf, e); return cdf + x * 0.39894228040143267794f * e; }
    }
    return act_grad(x, act);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS. Result valid in all threads.
__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// 16-byte vector load/store of VE elements converted to/from float
template <typename T> __device__ inline void load_vec(const T* p, float* f);
template <> __device__ inline void load_vec<float>(const float* p, float* f) {
    const f32x4 v = *(const f32x4*)p;
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> __device__ inline void load_vec<bf16>(const bf16* p, float* f) {
    const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <> __device__ inline void load_vec<f16>(const f16* p, float* f) {
    const f16x8 v = *(const f16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ inline void store_vec(T* p, const float* f);
template <> __device__ inline void store_vec<float>(float* p, const float* f) {
    f32x4 v = {f[0], f[1], f[2], f[3]};
    *(f32x4*)p = v;
}
template <> __device__ inline void store_vec<bf16>(bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    *(bf16x8*)p = v;
}

template <> __device__ inline void store_vec<f16>(f16* p, const float* f) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (f16)f[i];
    *(f16x8*)p = v;
}

// caller-owned scratch registered through dm_set_workspace (runtime.hip): split partial sums of the MFMA kernels
extern float* dm_g_ws;
extern int64_t dm_g_ws_bytes;
extern int* dm_g_counters;
#define DM_WS_COUNTERS 16384

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int grid_for(int64_t work_items, int block, int cap = 256 * 16) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

#define DM_DISPATCH_DTYPE(dtype, ...)                          \
    if ((dtype) == DM_F32) { using T = float; __VA_ARGS__; }   \
    else if ((dtype) == DM_BF16) { using T = bf16; __VA_ARGS__; } \
    else if ((dtype) == DM_F16) { using T = f16; __VA_ARGS__; } \
    else { dm_set_error("bad dtype %d", (int)(dtype)); return DM_EINVAL; }
