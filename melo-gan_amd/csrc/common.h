// Shared helpers for libmelogan_hip (gfx950 only; no CUDA path, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdint.h>
#include "../../include/melo_gan_hip.h"

#define MG_OK 0
#define MG_EARG (-1)
#define MG_EUNSUP (-2)
#define MG_EWORK (-3)
#define MG_EHIP (-4)

void mg_set_error(const char* fmt, ...);

#define MG_CHECK_ARG(cond, ...)                    \
    do {                                           \
        if (!(cond)) {                             \
            mg_set_error(__VA_ARGS__);             \
            return MG_EARG;                        \
        }                                          \
    } while (0)

#define MG_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            mg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MG_EHIP;                                                     \
        }                                                                       \
    } while (0)

#define MG_HIP(call)                                                             \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) {                                                 \
            mg_set_error("%s failed: %s", #call, hipGetErrorString(e__));        \
            return MG_EHIP;                                                      \
        }                                                                        \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Exact-erf GELU (nn.GELU() default, reference src/emotion_discriminator/ed_model.py:26-46) without libm's erff: the
// emotion discriminator's convolutions apply GELU / GELU' to 4-17 M elements per launch in their epilogues and ocml's
// erff + expf (~60 VALU instructions per element) cost 5-12 us of a 10-60 us kernel.  Abramowitz-Stegun 7.1.26:
//   erfc(u) = (a1 t + ... + a5 t^5) exp(-u^2),  t = 1 / (1 + p u),  u >= 0,   |error| <= 1.5e-7
// -- measured in fp32 over [-8, 8]: |GELU error| <= 4.2e-7, |GELU' error| <= 3.2e-7, while x * cdf rounded to fp32 is
// itself only good to 1.2e-6 there (torch's fp32 GELU against fp64).  exp(-u^2) = exp(-x^2 / 2) is also the Gaussian
// density GELU' needs, so the derivative costs one v_exp_f32 and one v_rcp_f32 in total.
__device__ __forceinline__ void mg_gauss(float x, float& cdf, float& e) {
    const float u = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
    e = __expf(-u * u);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = 0.5f * p * t * e;          // upper tail of the normal distribution at |x|
    cdf = x >= 0.f ? 1.0f - q : q;
}
__device__ __forceinline__ float mg_gelu(float x) {
    float cdf, e;
    mg_gauss(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float mg_gelu_grad(float x) {
    float cdf, e;
    mg_gauss(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float mg_act(int act, float v) {
    switch (act) {
        case MG_ACT_RELU: return v > 0.f ? v : 0.f;
        case MG_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case MG_ACT_GELU: return mg_gelu(v);
        case MG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative factor given the saved reference value (see mg_epilogue in the public header)
__device__ __forceinline__ float mg_act_grad(int gact, float r) {
    switch (gact) {
        case MG_ACT_RELU: return r > 0.f ? 1.f : 0.f;
        case MG_ACT_LRELU: return r > 0.f ? 1.f : 0.2f;
        case MG_ACT_GELU: return mg_gelu_grad(r);
        case MG_ACT_TANH: return 1.f - r * r;
        default: return 1.f;
    }
}

// The fused epilogue of include/melo_gan_hip.h applied to one accumulator value; di = dense output index,
// n = output channel.  The caller handles `accumulate` and the final store.
__device__ __forceinline__ float mg_apply_epilogue(const mg_epilogue& E, float v, int n, long di) {
    if (E.bias) v += E.bias[n];
    if (E.scale) v = v * E.scale[n] + E.shift[n];
    if (E.zout) E.zout[di] = v;
    v = mg_act(E.act, v);
    if (E.gref) v *= mg_act_grad(E.gact, E.gref[di]);
    if (E.emul) v *= E.emul[di];
    if (E.gscale) v *= E.gscale[n];
    return v;
}

// The same epilogue for a small per-thread set of values, as whole-set passes behind ONE uniform branch each: with the
// activation switch evaluated per element hipcc computed erff and tanhf for every element and selected afterwards.
template <int NV>
__device__ __forceinline__ void mg_apply_epilogue_set(const mg_epilogue& E, float (&v)[NV], const int (&n)[NV],
                                                      const long (&di)[NV], const bool (&ok)[NV]) {
    if (E.bias) {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (ok[q]) v[q] += E.bias[n[q]];
    }
    if (E.scale) {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (ok[q]) v[q] = v[q] * E.scale[n[q]] + E.shift[n[q]];
    }
    if (E.zout) {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (ok[q]) E.zout[di[q]] = v[q];
    }
    if (E.act == MG_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = mg_act(MG_ACT_RELU, v[q]);
    } else if (E.act == MG_ACT_LRELU) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = mg_act(MG_ACT_LRELU, v[q]);
    } else if (E.act == MG_ACT_GELU) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = mg_act(MG_ACT_GELU, v[q]);
    } else if (E.act == MG_ACT_TANH) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = mg_act(MG_ACT_TANH, v[q]);
    }
    // elementwise operands: unconditional clamped loads (all in flight), applied afterwards -- a load behind `if (ok)`
    // compiles to branch + load + wait per element
    if (E.gref) {
        float g[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) g[q] = E.gref[ok[q] ? di[q] : 0];
        if (E.gact == MG_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_RELU, g[q]);
        } else if (E.gact == MG_ACT_LRELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_LRELU, g[q]);
        } else if (E.gact == MG_ACT_GELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_GELU, g[q]);
        } else if (E.gact == MG_ACT_TANH) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_TANH, g[q]);
        }
    }
    if (E.emul) {
        float g[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) g[q] = E.emul[ok[q] ? di[q] : 0];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] *= g[q];
    }
    if (E.gscale) {
#pragma unroll
        for (int q = 0; q < NV; ++q) if (ok[q]) v[q] *= E.gscale[n[q]];
    }
}

// conv_thin.hip: the window GEMMs with a <= 8 channel reduction or output side; MG_EUNSUP = not such a shape
int mg_conv_thin_dispatch(const float* x, const float* w, float* y, int B, int Tin, int Cin, int Tout, int N, int K, int stride,
                          int flip, int transposed, int w_sn, int w_sc, long xbs, long ybs, const mg_epilogue* epi,
                          hipStream_t stream);

// conv_mfma.hip: nn.Linear forward on the 64x64-tile window-GEMM kernel (K = 1), output columns in mg_linear_perm's order
int mg_conv_linear_perm(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, const mg_epilogue* epi,
                        int perm_L, hipStream_t stream);

// XCD-aware block numbering (cdna_hip_programming.md T1): blocks are observed to be dealt round-robin over the 8 XCDs, each
// with a private L2, so blocks with equal id % 8 share an L2.  This bijection hands every such group a CONTIGUOUS range
// of logical ids: blocks that read the same rows get consecutive logical ids and fetch them from memory once per XCD
// instead of once per block.  Placement is not a contract -- a wrong guess is slower, never wrong.
__device__ __forceinline__ int mg_xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

static inline int mg_ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline long mg_cdiv(long a, long b) { return (a + b - 1) / b; }
