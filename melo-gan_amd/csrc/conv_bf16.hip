// bf16-storage / fp32-accumulate variant of the frozen emotion discriminator's branch (SECONDARY configuration, never the
// fp32 headline: BASELINE.json configs[1] names bf16 next to fp32).  The branch is a fixed network applied to the generated
// batch -- forward, cross-entropy, input gradient (src/gan/train_gan.py:228-236; src/emotion_discriminator/ed_model.py:24-69)
// -- so nothing here touches optimiser state: activations and the (folded, frozen) weights are STORED in bf16, every
// product is accumulated in fp32 on v_mfma_f32_32x32x16_bf16, the epilogue arithmetic (folded BatchNorm, GELU / GELU',
// channel scales) runs in fp32 on the accumulator.
//
//   y[b, t, n] = EPI( sum_{k < K} sum_c  x[b, t + k - (K-1)/2, c] * Wb[k][n][c] )          stride 1, K in {3, 5}
//
// At 16x the fp32 MFMA rate these layers (0.8-6.4 GFLOP) are no longer compute-bound: a launch moves its activations
// once (8-17 MB in bf16) and re-reads the 0.1-0.4 MB of weights per row tile from L2.  128 x 64 output tiles, 4 waves of
// 32 x 64 (two 32x32 accumulators), 32-channel chunks double-buffered in LDS, one barrier per chunk.  LDS rows are 80 bytes
// (64 of data): the 16 lanes of a ds_read_b128 group read rows r .. r+15 at the same column, 80 r mod 128 walks all eight
// 16-byte bank groups.
// Operand lane maps (cdna_hip_programming.md section 3): lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4b;

struct ConvBP {
    const void* x;        // (B, T, Cin) bf16, or fp32 when XF32
    const __bf16* w;      // [K][N][Cin]
    void* y;              // (B, T, N) bf16, or fp32 when YF32
    int B, T, Cin, N;
    mg_epilogue_bf16 e;
};

constexpr int BM = 128, BN = 64, BKC = 32;
constexpr int PX = 40;                      // LDS row pitch in bf16 elements (80 bytes)

template <int K, bool XF32, bool YF32>
__global__ __launch_bounds__(256, K == 3 ? 3 : 2) void conv_bf16_kernel(const ConvBP p) {
    constexpr int PAD = (K - 1) / 2, XR = BM + K - 1;
    constexpr int NXP = (XR * 4 + 255) / 256;             // 16-byte pieces of the x window per thread
    constexpr int XS = XR * PX, WS = K * BN * PX;          // bf16 elements per buffer
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* xs = lds;                                      // [2][XS]
    __bf16* ws = lds + 2 * XS;                             // [2][WS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b = m0 / p.T, t0 = m0 - b * p.T;             // T % BM == 0: a tile lies inside one sequence

    f32x16 acc[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ni][i] = 0.f;

    // ---- staging plan ----
    int x_off[NXP], x_lds[NXP];      // element offset of the piece's first channel in x (row clamped), LDS element offset
    bool x_ok[NXP];
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int idx = tid + 256 * j, row = idx >> 2, q = idx & 3;
        const int t = t0 - PAD + row;
        x_ok[j] = row < XR && t >= 0 && t < p.T;
        x_off[j] = ((b * p.T + (x_ok[j] ? t : 0)) * p.Cin + 8 * q);
        x_lds[j] = row < XR ? row * PX + 8 * q : -1;
    }
    int w_off[K], w_lds[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int idx = tid + 256 * j, k = idx / (BN * 4), n = (idx >> 2) & (BN - 1), q = idx & 3;
        w_off[j] = ((k * p.N + n0 + n) * p.Cin + 8 * q);
        w_lds[j] = (k * BN + n) * PX + 8 * q;
    }
    // Staging: global -> registers -> LDS with the registers TWO chunks ahead of the MFMAs (ring of two register sets, loop
    // unrolled by two so that the ring index is a compile-time constant): a chunk's 12-20 MFMAs take ~400 cycles, a global
    // load ~2000, so a one-chunk prefetch distance left every chunk waiting for memory.
    bf16x8 xr[2][NXP], wr[2][K];
    auto load_chunk = [&](auto ring, int c0) {
        constexpr int G = decltype(ring)::value;
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            bf16x8 v;
            if (XF32) {
                const float* src = (const float*)p.x + x_off[j] + c0;
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (__bf16)a[e]; v[4 + e] = (__bf16)c[e]; }
            } else {
                v = *reinterpret_cast<const bf16x8*>((const __bf16*)p.x + x_off[j] + c0);
            }
            if (!x_ok[j]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            }
            xr[G][j] = v;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) wr[G][j] = *reinterpret_cast<const bf16x8*>(p.w + w_off[j] + c0);
    };
    auto store_chunk = [&](auto ring, int buf) {
        constexpr int G = decltype(ring)::value;
#pragma unroll
        for (int j = 0; j < NXP; ++j)
            if (x_lds[j] >= 0) *reinterpret_cast<bf16x8*>(xs + buf * XS + x_lds[j]) = xr[G][j];
#pragma unroll
        for (int j = 0; j < K; ++j) *reinterpret_cast<bf16x8*>(ws + buf * WS + w_lds[j]) = wr[G][j];
    };
    auto compute = [&](int buf) {
        const __bf16* xb = xs + buf * XS + (32 * wave + r) * PX + 8 * h;
        const __bf16* wb = ws + buf * WS + r * PX + 8 * h;
#pragma unroll
        for (int s = 0; s < BKC / 16; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(xb + k * PX + 16 * s);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wb + (k * BN + 32 * ni) * PX + 16 * s);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[ni], 0, 0, 0);
                }
            }
    };

    const int nchunks = p.Cin / BKC;
    const std::integral_constant<int, 0> R0{};
    const std::integral_constant<int, 1> R1{};
    load_chunk(R0, 0);
    if (nchunks > 1) load_chunk(R1, BKC);
    store_chunk(R0, 0);
    __syncthreads();
    // iteration c: ring[c & 1] is free (chunk c sits in LDS[c & 1]) and takes chunk c + 2; ring[(c + 1) & 1] holds chunk c + 1
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 2 < nchunks) load_chunk(R0, (c + 2) * BKC);
        compute(0);
        if (c + 1 < nchunks) store_chunk(R1, 1);
        __syncthreads();
        if (c + 1 >= nchunks) break;
        if (c + 3 < nchunks) load_chunk(R1, (c + 3) * BKC);
        compute(1);
        if (c + 2 < nchunks) store_chunk(R0, 0);
        __syncthreads();
    }

    // ---- epilogue (fp32 on the accumulator; every row of the tile exists) ----
    // bf16 tensors cross the LDS on their way out / in: an accumulator lane owns ONE column and 16 rows, so storing it
    // directly is 32 two-byte stores per thread in 64-byte runs (measured: the epilogue, not the MFMA loop, was the launch).
    // The tile is parked in LDS as [128 rows][64 columns] (pitch 72: the two half-waves of a ds_write_b16 land on disjoint
    // banks) and leaves as 16-byte pieces, 8 lanes per 128-byte row; gref comes in the same way.
    const mg_epilogue_bf16& E = p.e;
    constexpr int TP = 72;
    __bf16* tile = lds;                                    // the staging buffers are idle: every wave passed the last barrier
    __bf16* tile2 = lds + BM * TP;                         // second parking area (pre-activation z next to the output)
    auto lane_off = [&](int ni, int i) { return (32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h) * TP + 32 * ni + r; };
    auto tile_to_global = [&](const __bf16* src, __bf16* dst) {
#pragma unroll
        for (int j = 0; j < BM * BN / 8 / 256; ++j) {
            const int pc = tid + 256 * j, row = pc >> 3, q = pc & 7;
            *reinterpret_cast<bf16x8*>(dst + (long)(m0 + row) * p.N + n0 + 8 * q) = *reinterpret_cast<const bf16x8*>(src + row * TP + 8 * q);
        }
    };
    float g[2][16];
    if (E.gref) {
#pragma unroll
        for (int j = 0; j < BM * BN / 8 / 256; ++j) {
            const int pc = tid + 256 * j, row = pc >> 3, q = pc & 7;
            *reinterpret_cast<bf16x8*>(tile + row * TP + 8 * q) =
                *reinterpret_cast<const bf16x8*>((const __bf16*)E.gref + (long)(m0 + row) * p.N + n0 + 8 * q);
        }
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) g[ni][i] = (float)tile[lane_off(ni, i)];
        __syncthreads();
    }
    if (E.scale) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const float scale = E.scale[n0 + 32 * ni + r], shift = E.shift[n0 + 32 * ni + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][i] = acc[ni][i] * scale + shift;
        }
    }
    if (E.zout) {           // parked now, leaves with the output behind ONE barrier (fp32 output: its own barrier)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) tile2[lane_off(ni, i)] = (__bf16)acc[ni][i];
        if (YF32) {
            __syncthreads();
            tile_to_global(tile2, (__bf16*)E.zout);
        }
    }
    if (E.act == MG_ACT_GELU) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][i] = mg_act(MG_ACT_GELU, acc[ni][i]);
    } else if (E.act == MG_ACT_RELU) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][i] = mg_act(MG_ACT_RELU, acc[ni][i]);
    }
    if (E.gref) {
        if (E.gact == MG_ACT_GELU) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ni][i] *= mg_act_grad(MG_ACT_GELU, g[ni][i]);
        } else if (E.gact == MG_ACT_RELU) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ni][i] *= mg_act_grad(MG_ACT_RELU, g[ni][i]);
        }
    }
    if (E.gscale) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const float gs = E.gscale[n0 + 32 * ni + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][i] *= gs;
        }
    }
    if (YF32) {
        float* y = (float*)p.y;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const unsigned d0 = (unsigned)((m0 + 32 * wave + 4 * h) * p.N + n0 + 32 * ni + r);   // host checked: < 2^31
            auto di = [&](int i) { return d0 + (unsigned)(((i & 3) + 8 * (i >> 2)) * p.N); };
            if (E.accumulate) {
                float o[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = y[di(i)];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ni][i] += o[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) y[di(i)] = acc[ni][i];
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) tile[lane_off(ni, i)] = (__bf16)acc[ni][i];
        __syncthreads();
        if (E.zout) tile_to_global(tile2, (__bf16*)E.zout);
        tile_to_global(tile, (__bf16*)p.y);
    }
}

// wb[k][n][c] = bf16( w[n*sn + c*sc + (flip ? K-1-k : k)] )
__global__ void wb_relayout_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, int N, int Cc, int K, int sn, int sc, int flip) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Cc * K) return;
    const int c = (int)(i % Cc);
    const long r2 = i / Cc;
    const int n = (int)(r2 % N), k = (int)(r2 / N);
    wb[i] = (__bf16)w[(long)n * sn + (long)c * sc + (flip ? K - 1 - k : k)];
}

// h[b][c] = mean_t a[b][t][c]   (a bf16, h fp32): one block per (b, 64-channel group), 4 row lanes x 64 channels
__global__ __launch_bounds__(256) void meanT_fwd_bf16_kernel(const __bf16* __restrict__ a, float* __restrict__ hout, int T, int C) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b = blockIdx.x, c = blockIdx.y * 64 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        const __bf16* src = a + (long)b * T * C + c;
        int t = rl;
        for (; t + 12 < T; t += 16) {
            s0 += (float)src[(long)t * C];
            s1 += (float)src[(long)(t + 4) * C];
            s2 += (float)src[(long)(t + 8) * C];
            s3 += (float)src[(long)(t + 12) * C];
        }
        for (; t < T; t += 4) s0 += (float)src[(long)t * C];
    }
    red[rl][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < C) hout[(long)b * C + c] = ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) / (float)T;
}

// dz[b][t][c] = dh[b][c] / T * act'(gref[b][t][c]) * gscale[c]   (dz, gref bf16): 8 channels per thread
__global__ __launch_bounds__(256) void meanT_bwd_bf16_kernel(const float* __restrict__ dh, __bf16* __restrict__ dz,
                                                             const __bf16* __restrict__ gref, int gact,
                                                             const float* __restrict__ gscale, long n8, int T, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const long e0 = i * 8;
    const int c = (int)(e0 % C);
    const long bt = e0 / C;
    const int b = (int)(bt / T);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gref + e0);
    bf16x8 o;
    const float inv = 1.f / (float)T;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = dh[(long)b * C + c + e] * inv;
        v *= mg_act_grad(gact, (float)g[e]);
        if (gscale) v *= gscale[c + e];
        o[e] = (__bf16)v;
    }
    *reinterpret_cast<bf16x8*>(dz + e0) = o;
}

template <int K>
int launch_bf16(const ConvBP& p, int xf32, int yf32, hipStream_t stream) {
    constexpr int XR = BM + K - 1;
    const size_t ldsb = (size_t)2 * (XR * PX + K * BN * PX) * sizeof(__bf16);
    dim3 grid((unsigned)((long)p.B * p.T / BM), (unsigned)(p.N / BN));
    auto go = [&](auto kernel) -> int {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { mg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MG_EHIP; }
            attr_set = true;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), ldsb, stream, p);
        MG_CHECK_LAUNCH("conv_bf16");
        return MG_OK;
    };
    if (xf32) return yf32 ? go(&conv_bf16_kernel<K, true, true>) : go(&conv_bf16_kernel<K, true, false>);
    return yf32 ? go(&conv_bf16_kernel<K, false, true>) : go(&conv_bf16_kernel<K, false, false>);
}

}  // namespace

extern "C" int mg_wb_relayout(const float* w, void* wb, int N, int Cc, int K, int w_sn, int w_sc, int flip, mg_stream_t stream) {
    MG_CHECK_ARG(w && wb && N > 0 && Cc > 0 && K > 0 && w_sn > 0 && w_sc > 0, "mg_wb_relayout: bad args");
    const long total = (long)N * Cc * K;
    hipLaunchKernelGGL(wb_relayout_kernel, dim3((unsigned)mg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wb, N, Cc, K,
                       w_sn, w_sc, flip);
    MG_CHECK_LAUNCH("wb_relayout");
    return MG_OK;
}

extern "C" int mg_conv1d_s1_bf16_supported(int B, int T, int Cin, int N, int K) {
    return B > 0 && T > 0 && (T % BM) == 0 && (Cin % BKC) == 0 && (N % BN) == 0 && (K == 3 || K == 5) &&
           (long)B * T * (Cin > N ? Cin : N) < (1L << 31);
}

extern "C" int mg_conv1d_s1_bf16(const void* x, int x_f32, const void* wb, void* y, int y_f32, int B, int T, int Cin, int N, int K,
                                 const mg_epilogue_bf16* epi, mg_stream_t stream) {
    MG_CHECK_ARG(x && wb && y, "mg_conv1d_s1_bf16: null tensor");
    MG_CHECK_ARG(mg_conv1d_s1_bf16_supported(B, T, Cin, N, K), "mg_conv1d_s1_bf16: unsupported shape B=%d T=%d Cin=%d N=%d K=%d", B, T, Cin, N, K);
    MG_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wb) | ((uintptr_t)y)) & 15) == 0, "mg_conv1d_s1_bf16: tensors must be 16-byte aligned");
    ConvBP p{};
    p.x = x; p.w = (const __bf16*)wb; p.y = y; p.B = B; p.T = T; p.Cin = Cin; p.N = N;
    if (epi) p.e = *epi;
    MG_CHECK_ARG(!(p.e.scale && !p.e.shift), "mg_conv1d_s1_bf16: scale without shift");
    MG_CHECK_ARG(!(p.e.accumulate && !y_f32), "mg_conv1d_s1_bf16: accumulate needs an fp32 output");
    return K == 3 ? launch_bf16<3>(p, x_f32, y_f32, (hipStream_t)stream) : launch_bf16<5>(p, x_f32, y_f32, (hipStream_t)stream);
}

extern "C" int mg_meanT_fwd_bf16(const void* a, float* h, int B, int T, int C, mg_stream_t stream) {
    MG_CHECK_ARG(a && h && B > 0 && T > 0 && C > 0, "mg_meanT_fwd_bf16: bad args");
    hipLaunchKernelGGL(meanT_fwd_bf16_kernel, dim3((unsigned)B, (unsigned)mg_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)a, h, T, C);
    MG_CHECK_LAUNCH("meanT_fwd_bf16");
    return MG_OK;
}

extern "C" int mg_meanT_bwd_bf16(const float* dh, void* dz, const void* gref, int gact, const float* gscale, int B, int T, int C,
                                 mg_stream_t stream) {
    MG_CHECK_ARG(dh && dz && gref && B > 0 && T > 0 && C > 0 && C % 8 == 0, "mg_meanT_bwd_bf16: bad args (C %% 8 == 0)");
    const long n8 = (long)B * T * C / 8;
    hipLaunchKernelGGL(meanT_bwd_bf16_kernel, dim3((unsigned)mg_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, dh, (__bf16*)dz,
                       (const __bf16*)gref, gact, gscale, n8, T, C);
    MG_CHECK_LAUNCH("meanT_bwd_bf16");
    return MG_OK;
}
