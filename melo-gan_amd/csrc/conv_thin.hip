// Window "GEMMs" whose reduction or output is only a few channels wide: the first and last layers at the reference's
// own NOTE_DIM = 4 (config/gan_config.yaml:43-44; critic conv.0, generator deconv.6, emotion discriminator conv0, the VAE's
// first encoder / last decoder layer).  5 taps x 4 channels = 20 multiply-adds per output: no MFMA tile fits (the
// 64x64-tile kernel ran them on its scalar staging path with a 64-wide N tile for 4 output columns), and there is nothing
// to tile -- these layers are bound by reading / writing the WIDE side once.  So: plain VALU kernels, one pass over
// memory, 16-byte accesses on the wide tensor, the thin side's weights in registers / LDS.
//
//   thin_in  (Cin <= 8):  y[b,t,n] = EPI( sum_{k,c} x[b, t*S + k - PAD, c] * W(n,c,k) )      gather form, S in {1,2}
//            a thread owns one output row and 4 consecutive columns; the (K x Cin) window is 80 contiguous bytes.
//   thin_out (N <= 8):    the gather form with S = 1 (incl. flipped taps: Conv1d stride-1 data-gradient) or the stride-2
//            transposed form (ConvTranspose1d forward / Conv1d stride-2 data-gradient).  16 lanes share an output row:
//            lane l multiplies channels 4l.. of the wide input rows by its slice of the weights, partial sums are
//            added across the 16 lanes with DPP-free shuffles, lane 0 applies the epilogue and stores.
#include "common.h"

namespace {

struct ThinP {
    const float* x;
    const float* w;
    float* y;
    int B, Tin, Cin, Tout, N, K, stride, flip, transposed, vec_out;
    long xbs, ybs;
    int w_sn, w_sc;
    mg_epilogue e;
};

constexpr int TK = 5;            // taps (K <= 5)
constexpr int TC = 8;            // thin side <= 8 channels

// ---- thin reduction side -------------------------------------------------------------------------------------------
// block = 16 rows x 16 column quads (N <= 64 per block; blockIdx.y walks wider N)
__global__ __launch_bounds__(256) void thin_in_kernel(const ThinP p) {
    __shared__ float Wl[TK * TC][64];            // [k*Cin + c][n]
    const int tid = threadIdx.x, nq = tid & 15, ry = tid >> 4;
    const int n0 = blockIdx.y * 64;
    const int KC = p.K * p.Cin;
    for (int i = tid; i < KC * 64; i += 256) {
        const int kc = i >> 6, n = i & 63;
        const int k = kc / p.Cin, c = kc - k * p.Cin;
        Wl[kc][n] = n0 + n < p.N ? p.w[(long)(n0 + n) * p.w_sn + (long)c * p.w_sc + (p.flip ? p.K - 1 - k : k)] : 0.f;
    }
    __syncthreads();
    const long m = (long)blockIdx.x * 16 + ry;
    if (m >= (long)p.B * p.Tout) return;
    const int b = (int)(m / p.Tout), t = (int)(m - (long)b * p.Tout);
    const int pad = (p.K - 1) / 2, tin0 = t * p.stride - pad;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xb = p.x + (long)b * p.xbs;
    const bool x4 = p.Cin == 4 && (p.xbs & 3) == 0 && ((((uintptr_t)p.x) & 15) == 0);
    if (x4) {
        // the window's rows as five unconditional row-clamped 16-byte loads, all in flight, zeroed afterwards where the
        // row lies outside the sequence (a load behind a per-lane condition costs a branch and a wait each)
        f32x4 xv[TK];
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            const int tin = tin0 + k;
            const bool ok = k < p.K && tin >= 0 && tin < p.Tin;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)(ok ? tin : 0) * 4);
            xv[k] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            if (k >= p.K) break;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&Wl[k * 4 + c][4 * nq]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += xv[k][c] * w[e];
            }
        }
    } else {
        for (int k = 0; k < p.K; ++k) {
            const int tin = tin0 + k;
            const bool ok = tin >= 0 && tin < p.Tin;
            const float* xr = xb + (long)(ok ? tin : 0) * p.Cin;
            for (int c = 0; c < p.Cin; ++c) {
                const float xl = xr[c];
                const float xv = ok ? xl : 0.f;
                const f32x4 w = *reinterpret_cast<const f32x4*>(&Wl[k * p.Cin + c][4 * nq]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += xv * w[e];
            }
        }
    }
    const int n = n0 + 4 * nq;
    const long d0 = ((long)b * p.Tout + t) * p.N + n;
    const int nn[4] = {n, n + 1, n + 2, n + 3};
    const long di[4] = {d0, d0 + 1, d0 + 2, d0 + 3};
    const bool ok[4] = {n < p.N, n + 1 < p.N, n + 2 < p.N, n + 3 < p.N};
    mg_apply_epilogue_set<4>(p.e, acc, nn, di, ok);
    float* yo = p.y + (long)b * p.ybs + (long)t * p.N + n;
    if (p.vec_out && ok[3]) {
        f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
        if (p.e.accumulate) v += *reinterpret_cast<const f32x4*>(yo);
        *reinterpret_cast<f32x4*>(yo) = v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ok[e]) yo[e] = p.e.accumulate ? yo[e] + acc[e] : acc[e];
    }
}

// ---- thin output side ----------------------------------------------------------------------------------------------
// 16 lanes per output position (transposed: per INPUT position u, which yields outputs 2u and 2u+1), lane l owns input
// channels [4l, 4l+4) (+64, +128, ... for wider inputs); block = 16 slots x RPT consecutive positions.  The weights sit
// in LDS as [k][n][c] so a lane's four channels of one (tap, column) are one 16-byte read.
constexpr int THIN_OUT_CMAX = 256;
template <bool TR2, int NMAX>
__global__ __launch_bounds__(256) void thin_out_kernel(const ThinP p, const int rpt) {
    extern __shared__ float Wl[];                // [K][NMAX][Cin]
    const int tid = threadIdx.x, l = tid & 15, ry = tid >> 4;
    for (int i = tid; i < p.K * NMAX * p.Cin; i += 256) {
        const int c = i % p.Cin, kn = i / p.Cin, n = kn % NMAX, k = kn / NMAX;
        Wl[i] = n < p.N ? p.w[(long)n * p.w_sn + (long)c * p.w_sc + (p.flip ? p.K - 1 - k : k)] : 0.f;
    }
    __syncthreads();
    const int Tm = TR2 ? p.Tin : p.Tout;
    const int pad = (p.K - 1) / 2;
    constexpr int NPH = TR2 ? 2 : 1;
    for (int it = 0; it < rpt; ++it) {
        const long m = ((long)blockIdx.x * rpt + it) * 16 + ry;
        const bool live = m < (long)p.B * Tm;
        const int b = live ? (int)(m / Tm) : 0, t = live ? (int)(m - (long)b * Tm) : 0;
        float acc[NPH][NMAX];
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) acc[ph][n] = 0.f;
        const float* xb = p.x + (long)b * p.xbs;
        for (int c0 = 4 * l; c0 < p.Cin; c0 += 64) {
            // window rows of this position: gather: t - pad .. t + pad (tap k <-> row t + k - pad);
            // transposed: u-1, u, u+1 with  phase 0: k=0 <- u+1, k=2 <- u, k=4 <- u-1;  phase 1: k=1 <- u+1, k=3 <- u
            constexpr int NROW = TR2 ? 3 : TK;
            f32x4 xr[NROW];
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const int tin = TR2 ? t - 1 + r : t + r - pad;
                const bool ok = live && tin >= 0 && tin < p.Tin && (TR2 || r < p.K);
                const int tc = ok ? tin : 0;
                f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)tc * p.Cin + c0);
                xr[r] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                auto wv = [&](int k) { return *reinterpret_cast<const f32x4*>(&Wl[((long)k * NMAX + n) * p.Cin + c0]); };
                if (TR2) {
                    const f32x4 w0 = wv(0), w1 = wv(1), w2 = wv(2), w3 = wv(3), w4 = wv(4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][n] += xr[2][e] * w0[e] + xr[1][e] * w2[e] + xr[0][e] * w4[e];
                        acc[1][n] += xr[2][e] * w1[e] + xr[1][e] * w3[e];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        if (k >= p.K) break;
                        const f32x4 w = wv(k);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[0][n] += xr[k][e] * w[e];
                    }
                }
            }
        }
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                float v = acc[ph][n];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                acc[ph][n] = v;
            }
        // lane n of the 16 finishes column n (phase 1 of the transposed form: lane NMAX + n)
        const int ph = TR2 ? l / NMAX : 0, n = l % NMAX;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NPH; ++q)
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (q == ph && j == n) v = acc[q][j];
        const int tout = TR2 ? 2 * t + ph : t;
        if (!live || l >= NPH * NMAX || n >= p.N || tout >= p.Tout) continue;
        const long di = ((long)b * p.Tout + tout) * p.N + n;
        v = mg_apply_epilogue(p.e, v, n, di);
        float* yo = p.y + (long)b * p.ybs + (long)tout * p.N + n;
        *yo = p.e.accumulate ? *yo + v : v;
    }
}

template <bool TR2, int NMAX>
int launch_thin_out(const ThinP& p, long rows, hipStream_t stream) {
    long blocks = mg_cdiv(rows, 16);
    int rpt = 1;
    while (rpt < 8 && blocks / (rpt * 2) >= 1024) rpt *= 2;
    const size_t lds = (size_t)p.K * NMAX * p.Cin * sizeof(float);
    hipLaunchKernelGGL((thin_out_kernel<TR2, NMAX>), dim3((unsigned)mg_cdiv(blocks, rpt)), dim3(256), lds, stream, p, rpt);
    MG_CHECK_LAUNCH("thin_out");
    return MG_OK;
}

}  // namespace

// Which kernel a shape takes: 0 = none (the MFMA tile kernels), 1 = thin_in, 2 / 3 = thin_out with 4 / 8 column slots.
// One decision point for the dispatcher below and for the host's launch observer (ops.py reports kernel symbols).
extern "C" int mg_conv_thin_route(const float* x, long xbs, int Cin, int N, int K, int stride, int transposed) {
    static const bool on = [] { const char* f = getenv("MG_CONV_THIN"); return !(f && atoi(f) == 0); }();   // A/B switch
    if (!on || K > TK || K < 3) return 0;
    if (!transposed && Cin <= TC) return 1;
    const bool vec = (Cin % 4 == 0) && (xbs % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    if (N <= TC && vec && Cin <= THIN_OUT_CMAX && (transposed ? K == 5 : stride == 1)) return N <= 4 ? 2 : 3;
    return 0;
}

// Called by mg_conv1d_gather / mg_conv1d_scatter2; returns MG_EUNSUP if the shape is not a thin one.
int mg_conv_thin_dispatch(const float* x, const float* w, float* y, int B, int Tin, int Cin, int Tout, int N, int K, int stride,
                          int flip, int transposed, int w_sn, int w_sc, long xbs, long ybs, const mg_epilogue* epi,
                          hipStream_t stream) {
    const int route = mg_conv_thin_route(x, xbs, Cin, N, K, stride, transposed);
    if (!route) return MG_EUNSUP;
    ThinP p{};
    p.x = x; p.w = w; p.y = y;
    p.B = B; p.Tin = Tin; p.Cin = Cin; p.Tout = Tout; p.N = N; p.K = K; p.stride = stride; p.flip = flip; p.transposed = transposed;
    p.xbs = xbs; p.ybs = ybs; p.w_sn = w_sn; p.w_sc = w_sc;
    if (epi) p.e = *epi;
    if (route == 1) {
        p.vec_out = (N % 4 == 0) && (ybs % 4 == 0) && ((((uintptr_t)y) & 15) == 0);
        const long rows = (long)B * Tout;
        hipLaunchKernelGGL(thin_in_kernel, dim3((unsigned)mg_cdiv(rows, 16), (unsigned)mg_cdiv(N, 64)), dim3(256), 0, stream, p);
        MG_CHECK_LAUNCH("thin_in");
        return MG_OK;
    }
    // 16 lanes x 4 channels; with 8 column slots and the transposed form's two phases all 16 lanes finish one output
    const long rows = (long)B * (transposed ? Tin : Tout);
    if (transposed) return route == 2 ? launch_thin_out<true, 4>(p, rows, stream) : launch_thin_out<true, 8>(p, rows, stream);
    return route == 2 ? launch_thin_out<false, 4>(p, rows, stream) : launch_thin_out<false, 8>(p, rows, stream);
}
