// Skinny GEMM for the Linear layers of the hot path (M = batch rows <= a few hundred):
//     y[M, N] = EPI( x[M, K] @ W^T ),   W(n, c) = w[n*w_sn + c*w_sc]
// These layers carry almost no FLOPs (pre.2 aside) -- what matters is latency and the number of
// workgroups.  So: one 32x32 output tile per workgroup, the K range split over the workgroup's EIGHT
// waves (and over blockIdx.z for very deep K) -- the dependent MFMA chain of a wave is the serial part, 64
// cycles per 2 k -- operands streamed global -> registers as float4, all of a wave's loads in flight at
// once (no LDS staging, no barriers in the K loop), fp32 MFMA 32x32x2, and a single LDS pass to add the
// waves' partial tiles.  K-split across workgroups writes partial slabs
// that a finishing kernel sums in fixed order (reproducible) and runs the epilogue on.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct LinP {
    const float* x;
    const float* w;
    float* y;
    float* part;      // partial slabs when ksplit > 1
    int M, K, N;
    int w_sn, w_sc;
    int ksplit;
    int gx, gy;       // row / column tiles
    int kw;           // K range per wave (multiple of 8)
    // perm_L > 0: output column n' is weight row (n' % perm_C) * perm_L + n' / perm_C (perm_C = N / perm_L): the output
    // comes out as (M, perm_L, perm_C) -- the channels-last (B, L, C) tensor the reference reaches by view(B, C, L) +
    // permute (src/gan/models.py:70,73) -- with coalesced stores; a lane reads its own weight row anyway
    int perm_L, perm_C;
    mg_epilogue e;
};
__device__ __forceinline__ int wrow(const LinP& p, int n) { return p.perm_L ? (n % p.perm_C) * p.perm_L + n / p.perm_C : n; }

constexpr int UNR = 8;   // 8-deep k-steps per unrolled iteration: 8 float4 of A + 8 of B in flight
constexpr int NW = 8;    // waves per workgroup (K-split inside the workgroup)

template <bool W_KCONTIG, bool VEC>
__global__ __launch_bounds__(64 * NW) void linear_skinny_kernel(const LinP p) {
    __shared__ float tile[NW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // 1-D launch, XCD remap: the row tiles of one column tile (same weight tile) and neighbouring column tiles get
    // consecutive logical ids -> one L2: pre.2's 16.8 MB of weights cross the fabric once, not once per row tile
    const int id = mg_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int bx = id % p.gx, byz = id / p.gx, by = byz % p.gy, bz = byz / p.gy;
    const int m0 = bx * 32, n0 = by * 32;
    const int row = min(m0 + i, p.M - 1);          // clamped: out-of-range rows/cols are computed but never stored
    const int col = wrow(p, min(n0 + i, p.N - 1));
    const int kbeg = (bz * NW + wave) * p.kw;
    const int kend = min(kbeg + p.kw, p.K);
    const float* xr = p.x + (long)row * p.K;
    const float* wr = p.w + (long)col * p.w_sn;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto load_a = [&](int k) -> float4 {        // x[row][k+4h .. k+4h+3]
        const int kk = k + 4 * h;
        if (VEC) return *reinterpret_cast<const float4*>(xr + kk);
        float4 v;
        v.x = kk + 0 < kend ? xr[kk + 0] : 0.f;
        v.y = kk + 1 < kend ? xr[kk + 1] : 0.f;
        v.z = kk + 2 < kend ? xr[kk + 2] : 0.f;
        v.w = kk + 3 < kend ? xr[kk + 3] : 0.f;
        return v;
    };
    auto load_b = [&](int k) -> float4 {        // W(col, k+4h .. k+4h+3)
        const int kk = k + 4 * h;
        if (W_KCONTIG) {
            if (VEC) return *reinterpret_cast<const float4*>(wr + kk);
            float4 v;
            v.x = kk + 0 < kend ? wr[kk + 0] : 0.f;
            v.y = kk + 1 < kend ? wr[kk + 1] : 0.f;
            v.z = kk + 2 < kend ? wr[kk + 2] : 0.f;
            v.w = kk + 3 < kend ? wr[kk + 3] : 0.f;
            return v;
        }
        float4 v;                               // n contiguous: 4 coalesced scalar loads
        const float* q = p.w + col;
        v.x = (VEC || kk + 0 < kend) ? q[(long)(kk + 0) * p.w_sc] : 0.f;
        v.y = (VEC || kk + 1 < kend) ? q[(long)(kk + 1) * p.w_sc] : 0.f;
        v.z = (VEC || kk + 2 < kend) ? q[(long)(kk + 2) * p.w_sc] : 0.f;
        v.w = (VEC || kk + 3 < kend) ? q[(long)(kk + 3) * p.w_sc] : 0.f;
        return v;
    };
    auto mma4 = [&](const float4& a, const float4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    };

    // a wave's whole range is at most 8*UNR deep in the layers this kernel serves: issue every load first
    auto run = [&](auto ngroups) {
        constexpr int G = decltype(ngroups)::value;
        float4 a[G], b[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            a[u] = load_a(kbeg + 8 * u);
            b[u] = load_b(kbeg + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < G; ++u) mma4(a[u], b[u]);
    };
    const int span = kend - kbeg;
    if (VEC && span == 8 * 8) run(std::integral_constant<int, 8>{});
    else if (VEC && span == 8 * 4) run(std::integral_constant<int, 4>{});
    else if (VEC && span == 8 * 2) run(std::integral_constant<int, 2>{});
    else if (VEC && span == 8 * 1) run(std::integral_constant<int, 1>{});
    else {
        int k = kbeg;
        for (; k + 8 * UNR <= kend; k += 8 * UNR) {
            float4 a[UNR], b[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                a[u] = load_a(k + 8 * u);
                b[u] = load_b(k + 8 * u);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) mma4(a[u], b[u]);
        }
        for (; k < kend; k += 8) mma4(load_a(k), load_b(k));
    }

    // ---- add the waves' tiles ----
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
    __syncthreads();
    constexpr int NV = 1024 / (64 * NW);      // outputs per thread
    float v[NV];
    int nn[NV];
    long di[NV];
    bool ok[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int idx = tid + 64 * NW * q;
        const int rr = idx >> 5, cc = idx & 31;
        const int m = m0 + rr, n = n0 + cc;
        ok[q] = m < p.M && n < p.N;
        nn[q] = ok[q] ? wrow(p, n) : 0;
        di[q] = ok[q] ? (long)m * p.N + n : 0;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 2) t += tile[w][rr][cc] + tile[w + 1][rr][cc];
        v[q] = t;
    }
    if (p.ksplit > 1) {
#pragma unroll
        for (int q = 0; q < NV; ++q)
            if (ok[q]) p.part[(long)bz * p.M * p.N + di[q]] = v[q];
        return;
    }
    mg_apply_epilogue_set<NV>(p.e, v, nn, di, ok);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        if (!ok[q]) continue;
        if (p.e.accumulate) v[q] += p.y[di[q]];
        p.y[di[q]] = v[q];
    }
}

__global__ void linear_finish_kernel(const float* __restrict__ part, float* __restrict__ y, long mn, int N, int ksplit,
                                     const mg_epilogue e, int perm_L, int perm_C) {
    const long di = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (di >= mn) return;
    float v = 0.f;
    for (int z = 0; z < ksplit; ++z) v += part[(long)z * mn + di];
    int n = (int)(di % N);
    if (perm_L) n = (n % perm_C) * perm_L + n / perm_C;
    v = mg_apply_epilogue(e, v, n, di);
    if (e.accumulate) v += y[di];
    y[di] = v;
}

int plan_ksplit(int M, int N, int K) {
    const long tiles = mg_cdiv(M, 32) * mg_cdiv(N, 32);
    int ks = 1;
    // deepen the split while a wave would still run > 64 MFMAs (K/NW/ks/2) and the grid is under ~512 WGs
    while (ks < 32 && K / (NW * ks) > 128 && tiles * ks < 512) ks *= 2;
    return ks;
}

}  // namespace

extern "C" size_t mg_linear_workspace_bytes(int M, int N, int K) {
    const int ks = plan_ksplit(M, N, K);
    return ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
}

extern "C" int mg_linear_perm(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, int w_sc,
                              const mg_epilogue* epi, int perm_L, void* work, size_t work_bytes, mg_stream_t stream);
extern "C" int mg_linear(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, int w_sc,
                         const mg_epilogue* epi, void* work, size_t work_bytes, mg_stream_t stream) {
    return mg_linear_perm(x, w, y, M, K, N, w_sn, w_sc, epi, 0, work, work_bytes, stream);
}

extern "C" int mg_linear_perm(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, int w_sc,
                              const mg_epilogue* epi, int perm_L, void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(x && w && y, "mg_linear: null tensor");
    MG_CHECK_ARG(perm_L >= 0 && (perm_L == 0 || N % perm_L == 0), "mg_linear_perm: perm_L must divide N");
    MG_CHECK_ARG(M > 0 && K > 0 && N > 0 && w_sn > 0 && w_sc > 0, "mg_linear: bad shape");
    MG_CHECK_ARG(w_sn == 1 || w_sc == 1, "mg_linear: one weight stride must be 1");
    // The permuted forward (decoder.pre.2) with enough rows and columns for the 64x64-tile kernel to fill the chip by its
    // output tiling alone: 13.2 us there against 20.3 here at the fused step's 2B = 128 rows (64 rows: 12.8 against 11.1)
    if (perm_L > 1 && w_sc == 1 && M >= 128 && (K & 63) == 0 && (N & 63) == 0 && mg_cdiv(M, 64) * (long)(N / 64) >= 192 &&
        ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0 && (w_sn & 3) == 0 && !getenv("MG_LINEAR_SKINNY_ONLY"))
        return mg_conv_linear_perm(x, w, y, M, K, N, w_sn, epi, perm_L, (hipStream_t)stream);
    LinP p{};
    p.x = x; p.w = w; p.y = y; p.M = M; p.K = K; p.N = N; p.w_sn = w_sn; p.w_sc = w_sc;
    p.e = epi ? *epi : mg_epilogue{};
    p.perm_L = perm_L > 1 ? perm_L : 0;
    p.perm_C = p.perm_L ? N / p.perm_L : 0;
    if (p.e.scale && !p.e.shift) { mg_set_error("mg_linear: scale without shift"); return MG_EARG; }
    p.ksplit = plan_ksplit(M, N, K);
    const bool kcontig = (w_sc == 1);
    // vector path: every wave's K range is a multiple of 8 inside K, rows 16-byte aligned
    const int waves = NW * p.ksplit;
    const bool vec = (K % (8 * waves) == 0) && ((((uintptr_t)x) & 15) == 0) &&
                     (!kcontig || ((((uintptr_t)w) & 15) == 0 && (w_sn & 3) == 0));
    p.kw = vec ? K / waves : (int)(mg_cdiv(mg_cdiv(K, waves), 8) * 8);
    if (p.ksplit > 1) {
        const size_t need = (size_t)p.ksplit * M * N * sizeof(float);
        if (!work || work_bytes < need) { mg_set_error("mg_linear: workspace too small (%zu < %zu)", work_bytes, need); return MG_EWORK; }
        p.part = (float*)work;
    }
    p.gx = (int)mg_cdiv(M, 32); p.gy = (int)mg_cdiv(N, 32);
    dim3 grid((unsigned)(p.gx * p.gy * p.ksplit));
    hipStream_t st = (hipStream_t)stream;
    if (kcontig) {
        if (vec) hipLaunchKernelGGL((linear_skinny_kernel<true, true>), grid, dim3(64 * NW), 0, st, p);
        else hipLaunchKernelGGL((linear_skinny_kernel<true, false>), grid, dim3(64 * NW), 0, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((linear_skinny_kernel<false, true>), grid, dim3(64 * NW), 0, st, p);
        else hipLaunchKernelGGL((linear_skinny_kernel<false, false>), grid, dim3(64 * NW), 0, st, p);
    }
    MG_CHECK_LAUNCH("linear_skinny_kernel");
    if (p.ksplit > 1) {
        const long mn = (long)M * N;
        hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)mg_cdiv(mn, 256)), dim3(256), 0, st,
                           (const float*)work, y, mn, N, p.ksplit, p.e, p.perm_L, p.perm_C);
        MG_CHECK_LAUNCH("linear_finish_kernel");
    }
    return MG_OK;
}
