// Stride-1 three-tap Conv1d (padding 1) and its data-gradient by the minimal-filtering form F(2,3) along the time axis
// (Winograd 1980; Lavin & Gray 2015 in one dimension): the frozen emotion discriminator's conv1-3
// (src/emotion_discriminator/ed_model.py:24-46: Conv1d(k=3, padding=1) -> BatchNorm -> GELU) and their input gradients,
// 21 of the step's 55 GFLOP as direct convolutions.
//
// Two consecutive outputs of one sequence share four input rows d0..d3 = x[2p-1 .. 2p+2]:
//     m0 = (d0 - d2) g0            m1 = (d1 + d2) (g0 + g1 + g2)/2
//     m2 = (d2 - d1) (g0 - g1 + g2)/2            m3 = (d1 - d3) g2
//     y[2p] = m0 + m1 + m2         y[2p+1] = m1 - m2 - m3
// (g0..g2 = the taps; every product a channel reduction) -- four channel GEMMs per output PAIR instead of six: 2/3 of
// the matrix-pipe time of the direct form, which is what bounds these layers (conv_mfma.hip).  The filter transform is
// done once (mg_wino3_weights; the branch's weights never change), the data transform is 16 VALU instructions per thread
// and 16-channel chunk on the way into LDS, the output transform two adds per element in the epilogue.  Rounding: the
// transformed operands carry one extra rounding each (|error| a few ulp of the direct form's; tests/test_conv_wino_gpu.py
// bounds it against fp64).
//
// Kernel = conv_wgemm_kernel's structure (conv_mfma.hip) with "tap" replaced by "transform position": a workgroup of four
// waves owns 64 output pairs (128 rows of one sequence) x 64 columns; a wave a 32-pair x 32-column tile with FOUR
// accumulators m0..m3 (v_mfma_f32_32x32x2_f32); per 16-channel chunk 8 slots (channel group x position) of 4 MFMAs, two
// LDS buffers, the gap schedule of that kernel (operand reads in gap 0 of a slot, one staging instruction per remaining
// gap, no address arithmetic in the loop, one barrier per chunk).
//   LDS images per buffer: Xt[pos][pair][20]  (16 channels + 4 pad: odd number of 16-B granules per row)
//                          Wt[quad][pos][n][4] = the global layout WT[Cin/4][4][N][4] verbatim: 16 runs of 1 KB per chunk
#include "common.h"
#include <type_traits>

long mg_conv_lds_pad_value();      // conv_mfma.hip: mg_conv_set_lds_pad

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WinoP {
    const float* x;
    const float* wt;
    float* y;
    int B, T, Cin, N;
    int tiles_per_seq;
    int x_bytes, wt_bytes;
    mg_epilogue e;
};

constexpr int WP = 64;                        // output pairs per workgroup tile
constexpr int SX = 20;                        // floats per Xt row
constexpr int XT_FLOATS = 4 * WP * SX;        // 5120
constexpr int WT_FLOATS = 4 * 4 * 64 * 4;     // 4096
constexpr int BUF = XT_FLOATS + WT_FLOATS;    // 36 KB per buffer

__global__ __launch_bounds__(256, 2) void wino3_kernel(const WinoP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.x / p.tiles_per_seq;
    const int t0 = (blockIdx.x - b * p.tiles_per_seq) * (2 * WP);
    const int n0 = blockIdx.y * 64;

    auto lds4 = [&](int off) { return reinterpret_cast<f32x4*>(__builtin_assume_aligned(smem + off, 16)); };

    const int abase = (wm * 32 + (lane & 31)) * SX + 4 * (lane >> 5);
    const int bbase = XT_FLOATS + (lane >> 5) * 4 * 256 + (wn * 32 + (lane & 31)) * 4;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);
    auto bload = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned soff) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };

    // X unit of this thread: output pair pl, channel quad q: the four rows 2(p0+pl)-1 .. +2 of the quad; rows outside the
    // sequence carry an offset beyond num_records and read as 0 (the convolution's zero padding)
    const int pl = tid >> 2, q = tid & 3;
    unsigned xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 - 1 + 2 * pl + i;
        xo[i] = (t >= 0 && t < p.T) ? ((unsigned)(b * p.T + t) * (unsigned)p.Cin + 4u * q) * 4u : 0x80000000u;
    }
    const int xl = pl * SX + 4 * q;                                     // + pos * WP * SX
    // W units: load i = channel quad i of the chunk, position = this wave, column = lane
    unsigned wg[4];
    int wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wg[i] = ((unsigned)(i * 4 + wave) * (unsigned)p.N * 4u + (unsigned)(n0 + lane) * 4u) * 4u;
        wl[i] = XT_FLOATS + (i * 4 + wave) * 256 + lane * 4;
    }
    f32x4 xr[4], ut[4], wr[4];
    auto x_soff = [&](int c0) { return 4u * (unsigned)c0; };
    auto w_soff = [&](int c0) { return (unsigned)c0 * (unsigned)p.N * 16u; };      // (c0/4) quads x 4 positions x N x 16 B
    auto transform = [&]() {
        ut[0] = xr[0] - xr[2];
        ut[1] = xr[1] + xr[2];
        ut[2] = xr[2] - xr[1];
        ut[3] = xr[1] - xr[3];
    };
    auto store_x = [&](int j, int boff) { *lds4(boff + xl + j * WP * SX) = ut[j]; };
    auto store_w = [&](int i, int boff) { *lds4(boff + wl[i]) = wr[i]; };
    auto load_x = [&](int i, unsigned soff) { xr[i] = bload(xrsrc, xo[i], soff); };
    auto load_w = [&](int i, unsigned soff) { wr[i] = bload(wrsrc, wg[i], soff); };

    f32x4 fa[2], fb[2];            // two operand register sets, alternating per slot
    auto frag_read = [&](int boff, int slot, f32x4& A, f32x4& Bv) {
        const int g = slot >> 2, j = slot & 3;
        A = *lds4(boff + abase + j * WP * SX + 8 * g);
        Bv = *lds4(boff + bbase + (8 * g + j) * 256);
    };

    const int c_last = p.Cin - 16;
    auto chunk = [&](auto parity, int c0) {
        constexpr int P = decltype(parity)::value;
        const int cur = P ? BUF : 0, oth = BUF - cur;
        const int cn = min(c0 + 32, c_last);
        const unsigned xs = x_soff(cn), ws = w_soff(cn);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int sl = m >> 2, s4 = m & 3, j = sl & 3;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl & 1][s4], fb[sl & 1][s4], acc[j], 0, 0, 0);
            if (s4 == 0) {
                if (sl == 7) __syncthreads();          // the other buffer is complete, this one is read out
                if (sl + 1 < 8) frag_read(cur, sl + 1, fa[(sl + 1) & 1], fb[(sl + 1) & 1]);
                else frag_read(oth, 0, fa[0], fb[0]);
            } else {
                // staging of the NEXT chunk (registers -> other buffer) and reload with the chunk after it:
                //   gap 0: the data transform (16 VALU; clustered: a VALU instruction behind an MFMA costs ~18 cycles, the
                //          ones right behind it ~4 each -- tools/mfma_gap_fillers.hip);  gaps 1-4: the four Xt stores;
                //   gaps 5-12: X reload i / W store i alternating;  gaps 13-16: the W reloads (never in the gap of their
                //   own store);  everything in front of the barrier (gap 21)
                const int gap = sl * 3 + s4 - 1;
                if (gap == 0) transform();
                else if (gap <= 4) store_x(gap - 1, oth);
                else if (gap <= 12) {
                    if (gap & 1) load_x((gap - 5) >> 1, xs);
                    else store_w((gap - 6) >> 1, oth);
                } else if (gap <= 16) load_w(gap - 13, ws);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: chunk 0 -> buffer 0, chunk 1 into the registers
#pragma unroll
    for (int i = 0; i < 4; ++i) load_x(i, x_soff(0));
#pragma unroll
    for (int i = 0; i < 4; ++i) load_w(i, w_soff(0));
    transform();
#pragma unroll
    for (int j = 0; j < 4; ++j) store_x(j, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_w(i, 0);
    {
        const int c1 = min(16, c_last);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_x(i, x_soff(c1));
#pragma unroll
        for (int i = 0; i < 4; ++i) load_w(i, w_soff(c1));
    }
    __syncthreads();
    frag_read(0, 0, fa[0], fb[0]);
    for (int c0 = 0;;) {
        chunk(std::integral_constant<int, 0>{}, c0);
        c0 += 16;
        if (c0 >= p.Cin) break;
        chunk(std::integral_constant<int, 1>{}, c0);
        c0 += 16;
        if (c0 >= p.Cin) break;
    }

    // ---- output transform + the fused epilogue (whole-tile passes behind uniform branches, as in conv_mfma.hip) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float m1 = acc[1][r], m2 = acc[2][r];
        acc[0][r] = acc[0][r] + m1 + m2;
        acc[3][r] = m1 - m2 - acc[3][r];
    }
    const mg_epilogue& E = p.e;
    const int n = n0 + wn * 32 + (lane & 31);
    const bool lin_tile = t0 + 2 * WP <= p.T;
    auto epilogue_pass = [&](auto lin_tag, auto ph_tag) {
        constexpr bool LIN = decltype(lin_tag)::value;
        constexpr int ph = decltype(ph_tag)::value;
        f32x16& a = acc[ph ? 3 : 0];
        const int ip0 = wm * 32 + 4 * (lane >> 5);
        const unsigned lin0 = (unsigned)((b * p.T + t0 + 2 * ip0 + ph) * p.N + n);
        auto index = [&](int r, unsigned& di) -> bool {
            const int dr = 2 * ((r & 3) + 8 * (r >> 2));
            di = lin0 + (unsigned)(dr * p.N);
            return LIN || (t0 + 2 * ip0 + ph + dr < p.T);
        };
        unsigned di;
        if (E.bias) {
            const float bias = E.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += bias;
        }
        if (E.scale) {
            const float scale = E.scale[n], shift = E.shift[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = a[r] * scale + shift;
        }
        if (E.zout) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (index(r, di)) E.zout[di] = a[r];
        }
        if (E.act == MG_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = mg_act(MG_ACT_RELU, a[r]);
        } else if (E.act == MG_ACT_LRELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = mg_act(MG_ACT_LRELU, a[r]);
        } else if (E.act == MG_ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = mg_act(MG_ACT_GELU, a[r]);
        } else if (E.act == MG_ACT_TANH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = mg_act(MG_ACT_TANH, a[r]);
        }
        if (E.gref) {      // loaded unconditionally (row-clamped), all sixteen in flight, applied afterwards
            float g[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = E.gref[index(r, di) ? di : 0u];
            if (E.gact == MG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] *= mg_act_grad(MG_ACT_RELU, g[r]);
            } else if (E.gact == MG_ACT_LRELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] *= mg_act_grad(MG_ACT_LRELU, g[r]);
            } else if (E.gact == MG_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] *= mg_act_grad(MG_ACT_GELU, g[r]);
            } else if (E.gact == MG_ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] *= mg_act_grad(MG_ACT_TANH, g[r]);
            }
        }
        if (E.emul) {
            float g[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = E.emul[index(r, di) ? di : 0u];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] *= g[r];
        }
        if (E.gscale) {
            const float gscale = E.gscale[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] *= gscale;
        }
        if (E.accumulate) {
            float g[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = p.y[index(r, di) ? di : 0u];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += g[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (index(r, di)) p.y[di] = a[r];
    };
    if (lin_tile) {
        epilogue_pass(std::true_type{}, std::integral_constant<int, 0>{});
        epilogue_pass(std::true_type{}, std::integral_constant<int, 1>{});
    } else {
        epilogue_pass(std::false_type{}, std::integral_constant<int, 0>{});
        epilogue_pass(std::false_type{}, std::integral_constant<int, 1>{});
    }
}

// wt[cq][pos][n][e] = the F(2,3) filter transform of the taps g_k = W(n, 4 cq + e, flip ? 2 - k : k), W(n,c,k) = w[n w_sn + c w_sc + k]
__global__ void wino3_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int Cin, long w_sn, long w_sc,
                                     int flip) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Cin) return;
    const int e = (int)(i & 3);
    const long r = i >> 2;
    const int n = (int)(r % N);
    const int cq = (int)(r / N);
    const float* g = w + (long)n * w_sn + (long)(4 * cq + e) * w_sc;
    const float g0 = g[flip ? 2 : 0], g1 = g[1], g2 = g[flip ? 0 : 2];
    const long o = ((long)cq * 4 * N + n) * 4 + e;
    const long ps = (long)N * 4;
    wt[o] = g0;
    wt[o + ps] = 0.5f * (g0 + g1 + g2);
    wt[o + 2 * ps] = 0.5f * (g0 - g1 + g2);
    wt[o + 3 * ps] = g2;
}

// several filters in one launch (a network whose weights change every step transforms all its layers at the top of the step)
struct WinoWJobs { mg_wino3_wjob j[MG_MAX_WINO_WJOBS]; int n; };
__global__ void wino3_weights_multi_kernel(const WinoWJobs J) {
    const mg_wino3_wjob q = J.j[blockIdx.y];
    const long total = (long)q.N * q.Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const long r = i >> 2;
        const int n = (int)(r % q.N);
        const int cq = (int)(r / q.N);
        const float* g = q.w + (long)n * q.w_sn + (long)(4 * cq + e) * q.w_sc;
        const float g0 = g[q.flip ? 2 : 0], g1 = g[1], g2 = g[q.flip ? 0 : 2];
        const long o = ((long)cq * 4 * q.N + n) * 4 + e;
        const long ps = (long)q.N * 4;
        q.wt[o] = g0;
        q.wt[o + ps] = 0.5f * (g0 + g1 + g2);
        q.wt[o + 2 * ps] = 0.5f * (g0 - g1 + g2);
        q.wt[o + 3 * ps] = g2;
    }
}

}  // namespace

extern "C" {

int mg_wino3_weights_multi(const mg_wino3_wjob* jobs, int n_jobs, mg_stream_t stream) {
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_WINO_WJOBS, "mg_wino3_weights_multi: 1..%d jobs", MG_MAX_WINO_WJOBS);
    WinoWJobs J{};
    J.n = n_jobs;
    long mx = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const mg_wino3_wjob& q = jobs[i];
        MG_CHECK_ARG(q.w && q.wt && q.N > 0 && q.Cin > 0 && (q.Cin & 3) == 0 && q.w_sn > 0 && q.w_sc > 0,
                     "mg_wino3_weights_multi: job %d: bad args (Cin must be a multiple of 4)", i);
        J.j[i] = q;
        const long t = (long)q.N * q.Cin;
        mx = t > mx ? t : mx;
    }
    long gx = mg_cdiv(mx, 256);
    gx = gx > 256 ? 256 : gx;
    hipLaunchKernelGGL(wino3_weights_multi_kernel, dim3((unsigned)gx, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, J);
    MG_CHECK_LAUNCH("wino3_weights_multi");
    return MG_OK;
}

int mg_conv1d_wino3_supported(int B, int T, int Cin, int N) {
    if (B <= 0 || T < 2 || (T & 1) || Cin < 16 || (Cin & 15) || N < 64 || (N & 63)) return 0;
    if ((long)B * T * Cin * 4 >= (1L << 31) || (long)B * T * N >= (1L << 31) || (long)Cin * N * 16 >= (1L << 31)) return 0;
    return 1;
}

int mg_wino3_weights(const float* w, float* wt, int N, int Cin, long w_sn, long w_sc, int flip, mg_stream_t stream) {
    MG_CHECK_ARG(w && wt && N > 0 && Cin > 0 && (Cin & 3) == 0, "mg_wino3_weights: bad args (Cin must be a multiple of 4)");
    const long total = (long)N * Cin;
    hipLaunchKernelGGL(wino3_weights_kernel, dim3((unsigned)mg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, N,
                       Cin, w_sn, w_sc, flip);
    MG_CHECK_LAUNCH("wino3_weights");
    return MG_OK;
}

int mg_conv1d_wino3(const float* x, const float* wt, float* y, int B, int T, int Cin, int N, const mg_epilogue* epi,
                    mg_stream_t stream) {
    MG_CHECK_ARG(x && wt && y, "mg_conv1d_wino3: null tensor");
    if (!mg_conv1d_wino3_supported(B, T, Cin, N)) {
        mg_set_error("mg_conv1d_wino3: unsupported shape B=%d T=%d Cin=%d N=%d (T even, Cin %% 16, N %% 64)", B, T, Cin, N);
        return MG_EUNSUP;
    }
    MG_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wt)) & 15) == 0, "mg_conv1d_wino3: x / wt must be 16-byte aligned");
    WinoP p{};
    p.x = x; p.wt = wt; p.y = y;
    p.B = B; p.T = T; p.Cin = Cin; p.N = N;
    p.tiles_per_seq = (int)mg_cdiv(T, 2 * WP);
    p.x_bytes = (int)((long)B * T * Cin * 4);
    p.wt_bytes = (int)((long)Cin * N * 16);
    if (epi) {
        p.e = *epi;
        MG_CHECK_ARG(!(p.e.scale && !p.e.shift), "epilogue: scale without shift");
    }
    size_t lds = 2 * (size_t)BUF * sizeof(float);
    const long pad = mg_conv_lds_pad_value();      // occupancy cap of launches that run beside another stream's critical path
    if (pad > 0 && lds + (size_t)pad <= 160 * 1024) lds += (size_t)pad;
    static bool attr_set = false;
    if (!attr_set) {
        MG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)(B * p.tiles_per_seq), (unsigned)(N / 64));
    hipLaunchKernelGGL(wino3_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    MG_CHECK_LAUNCH("wino3");
    return MG_OK;
}

}  // extern "C"
