// Window-GEMM kernels for Conv1d / ConvTranspose1d / Linear on channels-last activations.
//
// out[m, n] = sum_{tap, c} Xwin[m, tap, c] * W(n, c, ktap)      (fp32, v_mfma_f32_32x32x2_f32)
//
// With a (B, T, C) layout the im2col row of output position (b, t) is a window of
// consecutive input rows, so nothing is materialised: a workgroup stages the input rows its
// BM output positions need (once per 16-channel chunk) into LDS and every MFMA A-operand is
// a strided ds_read_b32 from that window.  Weights keep the reference's layouts (see the
// public header); a 16-channel x K-tap x BN slab is transposed into LDS as [(c,k)][n].
//
//   gather   (Conv1d fwd, ConvT1d dgrad, Linear fwd/dgrad, stride-1 Conv1d dgrad with flip):
//            window row of tap k for output t is t*S + k - (K-1)/2.
//   scatter2 (ConvT1d fwd stride 2 K=5 pad 2 outpad 1, Conv1d stride-2 dgrad): the two output
//            phases t=2u / t=2u+1 are two stride-1 correlations over taps {0,2,4} / {1,3}
//            sharing one input window [u-1, u+1]  (SURVEY hard part 6).
//
// fp32 MFMA issues one 32x32x2 every 64 cycles per SIMD, i.e. the matrix pipe -- not LDS or
// HBM -- bounds this kernel by a wide margin: 2 A + 2 B ds_read_b32 feed 4 MFMAs (256 cycles).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// -DMG_STAMPS: debug build that records wall-clock stamps (100 MHz) of workgroup phases; see scratch/stamps.py
#ifdef MG_STAMPS
__device__ long long* mg_stamp_buf = nullptr;
#define MG_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (threadIdx.x == 0 && mg_stamp_buf)                                                              \
            mg_stamp_buf[((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (k)] = \
                wall_clock64();                                                                            \
    } while (0)
#else
#define MG_STAMP(k)
#endif

struct ConvP {
    const float* x;
    const float* w;
    float* y;
    int B, Tin, Cin, Tm, Tout, N;
    long xbs, ybs;
    long x_bytes;    // extent of x in bytes if < 2^31 (buffer-descriptor range), else 0 => slow path
    int w_sn, w_sc;
    int flip;
    int tt_log2;
    int n_ttiles;
    int dbuf;         // 1: two LDS buffers, one barrier per chunk; 0: one buffer, two barriers (more WGs per CU)
    int ksplit;       // channel-chunk ranges handled by different workgroups (blockIdx.z); 1 => none
    int cps;          // chunks per split
    float* part;      // [ksplit][B*Tout*N] raw partial sums when ksplit > 1
    void* work;       // caller workspace (may be null)
    size_t work_bytes;
    mg_epilogue e;
};

// channels per LDS chunk: 16 for K=3/5 taps, 64 for K=1 (Linear layers: fewer, fatter chunks)
template <int K> struct ChunkOf { static constexpr int value = (K == 1) ? 64 : 16; };

template <int S, int K, bool TR2, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_wgemm_kernel(const ConvP p) {
    constexpr int BKC = ChunkOf<K>::value;
    constexpr int SX = BKC + 1;    // padded LDS row stride of the input window (odd => conflict-free reads)
    constexpr int BM = 64 * TM, BN = 64 * TN, SW = BN + 1;
    constexpr int SA = TR2 ? 1 : S;
    constexpr int NR = TR2 ? 3 : K;
    constexpr int NPH = TR2 ? 2 : 1;
    constexpr int PAD = (K - 1) / 2;
    constexpr int XQ = BKC / 4;                                       // float4 per window row
    constexpr int MAXX = (((BM - 1) * SA + NR) * XQ + 255) / 256 + 1;  // prefetch registers (float4) for X
    constexpr int NW4 = BN * BKC * K / 4 / 256;                       // prefetch registers (float4) for W
    static_assert((BN * BKC * K) % 1024 == 0, "weight slab must split evenly into float4 per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    MG_STAMP(0);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int TT = 1 << p.tt_log2;
    const int TB = BM >> p.tt_log2;
    const int R = (TT - 1) * SA + NR;
    const int nrows = TB * R;
    const int xs_floats = (nrows * SX + 3) & ~3;
    const int buf_floats = xs_floats + BKC * K * SW;     // one {X window, W slab} buffer; two are allocated
    float* Xs = smem;
    float* Ws = smem + xs_floats;

    const int mtile = blockIdx.x;
    const int b0 = (mtile / p.n_ttiles) * TB;
    const int t0 = (mtile % p.n_ttiles) * TT;
    const int n0 = blockIdx.y * BN;
    const int tin0 = TR2 ? (t0 - 1) : (t0 * S - PAD);

    // per-lane operand bases
    int abase[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int im = wm * 32 * TM + mi * 32 + (lane & 31);
        const int seg = im >> p.tt_log2, tl = im & (TT - 1);
        abase[mi] = (seg * R + tl * SA) * SX + (lane >> 5);
    }
    int bbase[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) bbase[ni] = (lane >> 5) * K * SW + wn * 32 * TN + ni * 32 + (lane & 31);
    const int kstep = p.flip ? -SW : SW;
    const int kbase = p.flip ? (K - 1) * SW : 0;

    f32x16 acc[NPH][TM][TN];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][mi][ni][r] = 0.f;

    const bool vec_ok = ((p.Cin & 3) == 0) && ((p.xbs & 3) == 0) && ((((uintptr_t)p.x) & 15) == 0);
    // split-K: this workgroup reduces over channels [c_begin, c_end) only
    const int c_begin = blockIdx.z * p.cps * BKC;
    const int c_end = min(p.Cin, c_begin + p.cps * BKC);
    const bool w_nck = p.w_sc < p.w_sn;   // (c,k) contiguous for a fixed n

    // one chunk of MFMAs out of LDS
    // One chunk of MFMAs out of LDS.  The operand fragments of step c2+1 are read into a second
    // register set BEFORE the MFMAs of step c2 issue, so the ~100-cycle ds_read latency hides under
    // the matrix pipe even with a single wave per SIMD (without this, PMC showed 40 % of wave time
    // parked in s_waitcnt lgkmcnt and the MFMA pipe only 33 % busy).
    auto frag_load = [&](const float* Xb, const float* Wb, int c2, float (&a)[NR][TM], float (&bw)[K][TN]) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[r][mi] = Xb[abase[mi] + r * SX + 2 * c2];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bw[k][ni] = Wb[bbase[ni] + 2 * c2 * K * SW + kbase + k * kstep];
    };
    auto frag_mma = [&](const float (&a)[NR][TM], const float (&bw)[K][TN]) {
        if constexpr (!TR2) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][mi], bw[k][ni], acc[0][mi][ni], 0, 0, 0);
        } else {
            // phase 0 (t = 2u):   k=0 <- row u+1, k=2 <- row u, k=4 <- row u-1
            // phase 1 (t = 2u+1): k=1 <- row u+1, k=3 <- row u
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][mi], bw[0][ni], acc[0][mi][ni], 0, 0, 0);
                    acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][mi], bw[1][ni], acc[1][mi][ni], 0, 0, 0);
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][mi], bw[2][ni], acc[0][mi][ni], 0, 0, 0);
                    acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][mi], bw[3][ni], acc[1][mi][ni], 0, 0, 0);
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][mi], bw[4][ni], acc[0][mi][ni], 0, 0, 0);
                }
        }
    };
    constexpr bool FRAG_DB = !(K == 5 && TM * TN >= 2);   // the 5-tap wide tiles have no registers left for it
    auto compute = [&](int nc2, int boff) {
        const float* Xb = Xs + boff;
        const float* Wb = Ws + boff;
        if constexpr (FRAG_DB) {
            float a0[NR][TM], b0[K][TN], a1[NR][TM], b1[K][TN];
            frag_load(Xb, Wb, 0, a0, b0);
            int c2 = 0;
            for (; c2 + 2 <= nc2; c2 += 2) {
                frag_load(Xb, Wb, c2 + 1, a1, b1);
                frag_mma(a0, b0);
                if (c2 + 2 < nc2) frag_load(Xb, Wb, c2 + 2, a0, b0);
                frag_mma(a1, b1);
            }
            if (c2 < nc2) frag_mma(a0, b0);
        } else {
            for (int c2 = 0; c2 < nc2; ++c2) {
                float a0[NR][TM], b0[K][TN];
                frag_load(Xb, Wb, c2, a0, b0);
                frag_mma(a0, b0);
            }
        }
    };

    // Fast path: every chunk and this block's weight slab are full and 16-B aligned, so the next chunk's
    // global loads are issued as float4 into registers BEFORE the current chunk's MFMAs and written to LDS
    // after them -- HBM/L2 latency hides under the matrix pipe.
    const bool w_aligned = ((((uintptr_t)p.w) & 15) == 0) &&
                           (w_nck ? (p.w_sc == K && (p.w_sn & 3) == 0) : (p.w_sn == K && (p.w_sc & 3) == 0));
    const bool fast = vec_ok && w_aligned && (p.Cin % BKC == 0) && (n0 + BN <= p.N) && (nrows * XQ <= 256 * MAXX) &&
                      p.x_bytes > 0;

    if (fast) {
        // chunk-invariant addressing of this thread's prefetch slots
        // X loads go through a raw buffer descriptor: slots of out-of-range rows carry an offset beyond
        // num_records, for which the hardware returns 0 -- no predicated load (hipcc branches and waits per
        // slot) and no arithmetic on the loaded registers (hipcc hoists it above the MFMAs and waits there).
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
        unsigned xo[MAXX];   // byte offset of the slot's float4 (without the chunk's channel offset)
        int xl[MAXX];        // LDS offset, -1 => slot unused
#pragma unroll
        for (int j = 0; j < MAXX; ++j) {
            const int idx = tid + 256 * j;
            xo[j] = 0x80000000u;
            xl[j] = -1;
            if (idx < nrows * XQ) {
                const int row = idx / XQ, q = idx - row * XQ;
                const int seg = row / R, r = row - seg * R;
                const int b = b0 + seg, tin = tin0 + r;
                xl[j] = row * SX + 4 * q;
                if (b < p.B && tin >= 0 && tin < p.Tin)
                    xo[j] = (unsigned)(((long)b * p.xbs + (long)tin * p.Cin + 4 * q) * 4);
            }
        }
        long wg[NW4];
        int wl[NW4];     // LDS offset of the slot's first element inside the weight slab
        int wk[NW4];     // CNK layout: tap index of that element (the 4 elements walk (n, k) in k-major order)
#pragma unroll
        for (int j = 0; j < NW4; ++j) {
            const int e4 = tid + 256 * j;
            wk[j] = 0;
            if (w_nck) {
                constexpr int PER = BKC * K / 4;
                const int n = e4 / PER, q = e4 - n * PER;
                wg[j] = (long)(n0 + n) * p.w_sn + 4 * q;     // + c0*K per chunk
                wl[j] = (4 * q) * SW + n;
            } else {
                constexpr int PER = BN * K / 4;
                const int c = e4 / PER, q = e4 - c * PER;
                const int n = (4 * q) / K, k = 4 * q - n * K;
                wg[j] = (long)c * p.w_sc + (long)n0 * K + 4 * q;   // + c0*w_sc per chunk
                wl[j] = (c * K + k) * SW + n;
                wk[j] = k;
            }
        }
        float4 xr[MAXX], wr[NW4];
        auto load_x = [&](int j, int c0) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo[j] + 4u * (unsigned)c0, 0, 0);
            xr[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        };
        auto load_w = [&](int j, long woff) { wr[j] = *reinterpret_cast<const float4*>(p.w + wg[j] + woff); };
        auto w_off = [&](int c0) { return w_nck ? (long)c0 * K : (long)c0 * p.w_sc; };
        auto store_x = [&](int j, float* d) {
            d[0] = xr[j].x; d[1] = xr[j].y; d[2] = xr[j].z; d[3] = xr[j].w;
        };
        // the 4 elements of a weight float4 walk k within a column (NCK: next LDS row each time) or (n, k) pairs
        // in k-major order (CNK: next row until the tap wraps, then the next column); one branch-free form
        const int wrapk = w_nck ? -1 : K - 1;
        auto store_w = [&](int j, float* Wb) {
            const float v[4] = {wr[j].x, wr[j].y, wr[j].z, wr[j].w};
            int d = wl[j], k = wk[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Wb[d] = v[i];
                const bool wrap = (k == wrapk);
                d += wrap ? 1 - (K - 1) * SW : SW;
                k = wrap ? 0 : k + 1;
            }
        };
        auto load_chunk = [&](int c0) {
#pragma unroll
            for (int j = 0; j < MAXX; ++j) load_x(j, c0);
            const long woff = w_off(c0);
#pragma unroll
            for (int j = 0; j < NW4; ++j) load_w(j, woff);
        };
        auto store_chunk = [&](int boff) {
#pragma unroll
            for (int j = 0; j < MAXX; ++j)
                if (xl[j] >= 0) store_x(j, Xs + boff + xl[j]);
#pragma unroll
            for (int j = 0; j < NW4; ++j) store_w(j, Ws + boff);
        };
        load_chunk(c_begin);
        MG_STAMP(1);
        store_chunk(0);
        if (p.dbuf) {
            // Software-pipelined loop over two LDS buffers.  The next chunk's round trip (prefetched registers ->
            // other LDS buffer -> reload the registers with the chunk after it) is cut into per-slot pieces that
            // sit BETWEEN the MFMA steps of the current chunk: a wave issues in order and every MFMA waits for
            // its predecessor (same accumulator), so staging placed after the chunk's MFMAs ran unoverlapped
            // (1.28 us per chunk against 0.64 us of MFMA time, measured with -DMG_STAMPS); placed between them
            // it rides in their shadow.  The body has no branches: past the end the last chunk is re-staged
            // into the idle buffer, and slots without a row write to a per-thread dummy.
            constexpr int NSTEP = BKC / 2;
            const int c_last = c_end - BKC;
            const int dummy = 2 * buf_floats + 4 * tid;
            load_chunk(min(c_begin + BKC, c_last));
            __syncthreads();
            MG_STAMP(2);
            int cur = 0;
            for (int c0 = c_begin; c0 < c_end; c0 += BKC) {
                const int oth = buf_floats - cur;
                const int cn = min(c0 + 2 * BKC, c_last);
                const long wn_off = w_off(cn);
                const float* Xb = Xs + cur;
                const float* Wb = Ws + cur;
                auto pieces = [&](int c2) {
#ifndef MG_EXP_NOPIECES
#pragma unroll
                    for (int j = 0; j < MAXX; ++j)
                        if (j % NSTEP == c2) {
                            store_x(j, smem + (xl[j] >= 0 ? oth + xl[j] : dummy));
#ifndef MG_EXP_NOLOADS
                            load_x(j, cn);
#endif
                        }
#pragma unroll
                    for (int j = 0; j < NW4; ++j)
                        if ((MAXX + j) % NSTEP == c2) {
                            store_w(j, Ws + oth);
#ifndef MG_EXP_NOLOADS
                            load_w(j, wn_off);
#endif
                        }
#endif
                };
                if constexpr (FRAG_DB) {
                    float a0[NR][TM], b0[K][TN], a1[NR][TM], b1[K][TN];
                    frag_load(Xb, Wb, 0, a0, b0);
#pragma unroll
                    for (int c2 = 0; c2 < NSTEP; c2 += 2) {
                        frag_load(Xb, Wb, c2 + 1, a1, b1);
                        frag_mma(a0, b0);
                        pieces(c2);
                        if (c2 + 2 < NSTEP) frag_load(Xb, Wb, c2 + 2, a0, b0);
                        frag_mma(a1, b1);
                        pieces(c2 + 1);
                    }
                } else {
#pragma unroll
                    for (int c2 = 0; c2 < NSTEP; ++c2) {
                        float a0[NR][TM], b0[K][TN];
                        frag_load(Xb, Wb, c2, a0, b0);
                        frag_mma(a0, b0);
                        pieces(c2);
                    }
                }
                __syncthreads();
                cur = oth;
#ifdef MG_STAMPS
                { const int it = (c0 - c_begin) / BKC; if (it < 2) MG_STAMP(8 + 4 * it); else if (it < 4) MG_STAMP(11 + it); }
#endif
            }
        } else {
            __syncthreads();
            MG_STAMP(2);
            for (int c0 = c_begin; c0 < c_end; c0 += BKC) {
                const bool more = c0 + BKC < c_end;
#ifdef MG_STAMPS
                const int it = (c0 - c_begin) / BKC;
#endif
                if (more) load_chunk(c0 + BKC);
                compute(BKC / 2, 0);
#ifdef MG_STAMPS
                if (it < 2) MG_STAMP(5 + 4 * it);
#endif
                __syncthreads();
#ifdef MG_STAMPS
                if (it < 2) MG_STAMP(6 + 4 * it);
#endif
                if (more) {
                    store_chunk(0);
#ifdef MG_STAMPS
                    if (it < 2) MG_STAMP(7 + 4 * it);
#endif
                    __syncthreads();
                }
#ifdef MG_STAMPS
                if (it < 2) MG_STAMP(8 + 4 * it); else if (it < 4) MG_STAMP(11 + it);
#endif
            }
        }
    } else {
    for (int c0 = c_begin; c0 < c_end; c0 += BKC) {
        __syncthreads();
        // ---- stage the input window chunk: rows (seg, r) x BKC channels ----
        if (vec_ok) {
            for (int idx = tid; idx < nrows * XQ; idx += 256) {
                const int row = idx / XQ, q = idx - row * XQ;
                const int seg = row / R, r = row - seg * R;
                const int b = b0 + seg, tin = tin0 + r, c = c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < p.B && tin >= 0 && tin < p.Tin && c < p.Cin)
                    v = *reinterpret_cast<const float4*>(p.x + (long)b * p.xbs + (long)tin * p.Cin + c);
                float* d = Xs + row * SX + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < nrows * BKC; idx += 256) {
                const int row = idx / BKC, cl = idx - row * BKC;
                const int seg = row / R, r = row - seg * R;
                const int b = b0 + seg, tin = tin0 + r, c = c0 + cl;
                float v = 0.f;
                if (b < p.B && tin >= 0 && tin < p.Tin && c < p.Cin)
                    v = p.x[(long)b * p.xbs + (long)tin * p.Cin + c];
                Xs[row * SX + cl] = v;
            }
        }
        // ---- stage the weight slab: [(c_local, k)][n] ----
        if (w_nck) {
            for (int e = tid; e < BN * BKC * K; e += 256) {
                const int n = e / (BKC * K), ck = e - n * (BKC * K);
                const int c = ck / K, k = ck - c * K;
                float v = 0.f;
                if (n0 + n < p.N && c0 + c < p.Cin) v = p.w[(long)(n0 + n) * p.w_sn + (long)(c0 + c) * p.w_sc + k];
                Ws[ck * SW + n] = v;
            }
        } else {
            for (int e = tid; e < BN * BKC * K; e += 256) {
                const int c = e / (BN * K), nk = e - c * (BN * K);
                const int n = nk / K, k = nk - n * K;
                float v = 0.f;
                if (n0 + n < p.N && c0 + c < p.Cin) v = p.w[(long)(n0 + n) * p.w_sn + (long)(c0 + c) * p.w_sc + k];
                Ws[(c * K + k) * SW + n] = v;
            }
        }
        __syncthreads();
        const int crem = c_end - c0;
        compute((crem >= BKC ? BKC : crem + 1) >> 1, 0);
    }
    }

    // ---- epilogue ----
    // Written as whole-tile passes over the accumulator registers, each selected by ONE uniform branch: with the
    // activation switch inside the per-element loop hipcc evaluated erff/tanhf for every element of every launch
    // and selected afterwards (4 us of a 16-us workgroup on the B=64 layers).
    MG_STAMP(3);
    const mg_epilogue& E = p.e;
    const long slab = (long)p.B * p.Tout * p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn * 32 * TN + ni * 32 + (lane & 31);
        if (n >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                f32x16& a = acc[ph][mi][ni];
                // dense / strided output index of accumulator element r; false if the row is outside the tensor
                auto index = [&](int r, unsigned& di, unsigned& yi) -> bool {
                    const int im = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int seg = im >> p.tt_log2, tl = im & (TT - 1);
                    const int b = b0 + seg, t = t0 + tl;
                    const int tout = TR2 ? 2 * t + ph : t;
                    di = (unsigned)((b * p.Tout + tout) * p.N + n);      // host checked: both fit 31 bits
                    yi = (unsigned)(b * (int)p.ybs + tout * p.N + n);
                    return b < p.B && t < p.Tm && tout < p.Tout;
                };
                unsigned di, yi;
                if (p.ksplit > 1) {      // raw partial sums; conv_finish_kernel adds the slabs and runs the epilogue
                    float* dst = p.part + (long)blockIdx.z * slab;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (index(r, di, yi)) dst[di] = a[r];
                    continue;
                }
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
                        if (index(r, di, yi)) E.zout[di] = a[r];
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
                if (E.gref) {
                    if (E.gact == MG_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (index(r, di, yi)) a[r] *= mg_act_grad(MG_ACT_RELU, E.gref[di]);
                    } else if (E.gact == MG_ACT_LRELU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (index(r, di, yi)) a[r] *= mg_act_grad(MG_ACT_LRELU, E.gref[di]);
                    } else if (E.gact == MG_ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (index(r, di, yi)) a[r] *= mg_act_grad(MG_ACT_GELU, E.gref[di]);
                    } else if (E.gact == MG_ACT_TANH) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (index(r, di, yi)) a[r] *= mg_act_grad(MG_ACT_TANH, E.gref[di]);
                    }
                }
                if (E.emul) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (index(r, di, yi)) a[r] *= E.emul[di];
                }
                if (E.gscale) {
                    const float gscale = E.gscale[n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] *= gscale;
                }
                if (E.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (index(r, di, yi)) a[r] += p.y[yi];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (index(r, di, yi)) p.y[yi] = a[r];
            }
        }
    }
    MG_STAMP(4);
}

// sums the split-K slabs in fixed order and applies the fused epilogue
__global__ void conv_finish_kernel(const float* __restrict__ part, float* __restrict__ y, long total, int Tout, int N,
                                   long ybs, int ksplit, const mg_epilogue e) {
    const long di = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (di >= total) return;
    float v = 0.f;
    for (int z = 0; z < ksplit; ++z) v += part[(long)z * total + di];
    const int n = (int)(di % N);
    v = mg_apply_epilogue(e, v, n, di);
    const long bt = di / N;
    const long yi = (bt / Tout) * ybs + (bt % Tout) * N + n;
    if (e.accumulate) v += y[yi];
    y[yi] = v;
}

template <int S, int K, bool TR2, int TM, int TN>
int launch_cfg(const ConvP& p0, hipStream_t stream) {
    ConvP p = p0;
    constexpr int BM = 64 * TM, BN = 64 * TN, SW = BN + 1;
    constexpr int BKC = ChunkOf<K>::value, SX = BKC + 1;
    constexpr int SA = TR2 ? 1 : S;
    constexpr int NR = TR2 ? 3 : K;
    int lg = mg_ilog2_ceil(p.Tm);
    const int lgbm = mg_ilog2_ceil(BM);
    if (lg > lgbm) lg = lgbm;
    p.tt_log2 = lg;
    const int TT = 1 << lg, TB = BM >> lg;
    p.n_ttiles = (int)mg_cdiv(p.Tm, TT);
    const int R = (TT - 1) * SA + NR;
    const size_t lds1 = ((size_t)((TB * R * SX + 3) & ~3) + (size_t)BKC * K * SW) * sizeof(float);
    // single LDS buffer by default: twice the resident workgroups beat saving one barrier per chunk in every
    // shape measured (e.g. 94.9 vs 85.6 TFLOP/s on the ED conv3 shape); MG_FORCE_DBUF=1 re-enables the double buffer
    p.dbuf = 0;
    if (const char* f = getenv("MG_FORCE_DBUF")) p.dbuf = atoi(f) ? 1 : 0;
    const size_t lds = p.dbuf ? 2 * lds1 + 256 * 4 * sizeof(float) : lds1;
    if (lds > 160 * 1024) {
        mg_set_error("conv_wgemm: LDS request %zu too large", lds);
        return MG_EUNSUP;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgemm_kernel<S, K, TR2, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            mg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MG_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.n_ttiles * mg_cdiv(p.B, TB)), (unsigned)mg_cdiv(p.N, BN));
    // split-K over workgroups when the output tiling alone leaves most of the 256 CUs idle
    const int nchunks = (int)mg_cdiv(p.Cin, BKC);
    const long n_wgs = (long)grid.x * grid.y;
    const long total = (long)p.B * p.Tout * p.N;
    p.ksplit = 1;
    p.cps = nchunks;
    long wg_target = 256, wg_max = 192;
    if (const char* f = getenv("MG_SPLITK_TARGET")) { wg_target = atol(f); wg_max = wg_target * 3 / 4; }
    if (p.work && n_wgs < wg_max && nchunks >= 4) {
        int ks = 1;
        while (ks < 8 && n_wgs * ks < wg_target && nchunks / (ks * 2) >= 2) ks *= 2;
        if (ks > 1 && p.work_bytes >= (size_t)ks * total * sizeof(float)) {
            p.ksplit = ks;
            p.cps = (int)mg_cdiv(nchunks, ks);
            p.ksplit = (int)mg_cdiv(nchunks, p.cps);
            p.part = (float*)p.work;
            grid.z = (unsigned)p.ksplit;
        }
    }
    hipLaunchKernelGGL((conv_wgemm_kernel<S, K, TR2, TM, TN>), grid, dim3(256), lds, stream, p);
    MG_CHECK_LAUNCH("conv_wgemm");
    if (p.ksplit > 1) {
        hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)mg_cdiv(total, 256)), dim3(256), 0, stream,
                           (const float*)p.part, p.y, total, p.Tout, p.N, p.ybs, p.ksplit, p.e);
        MG_CHECK_LAUNCH("conv_finish");
    }
    return MG_OK;
}

// Tile choice.  Measured on MI355X (scratch microbenchmarks, ED conv3 shape and the critic's stride-2 layers):
// the fp32 matrix pipe is fed best by MANY waves per SIMD, not by big register tiles -- 64x64 tiles with a single
// LDS buffer (4-5 workgroups per CU) reach 95-107 TFLOP/s where 128x128 (1 workgroup/CU) reaches 68 and
// 64x128 75-90.  So 64x64 is the default; the wider instantiations stay selectable (MG_FORCE_TILE=22|12) for
// tuning on other shapes.  Returns TM*10+TN.
int gather_tile(long m_total, int N) {
    if (const char* f = getenv("MG_FORCE_TILE")) {
        const int t = atoi(f);
        if ((t == 22 || t == 12) && N > 64) return t;
    }
    (void)m_total;
    return 11;
}
int scatter_tile(long m_total, int N) {
    if (const char* f = getenv("MG_FORCE_TILE"))
        if (atoi(f) == 12 && N > 64) return 12;
    (void)m_total;
    return 11;
}

template <int S, int K>
int launch_gather(const ConvP& p, hipStream_t stream) {
    switch (gather_tile((long)p.B * p.Tm, p.N)) {
        case 22: return launch_cfg<S, K, false, 2, 2>(p, stream);
        case 12: return launch_cfg<S, K, false, 1, 2>(p, stream);
        default: return launch_cfg<S, K, false, 1, 1>(p, stream);
    }
}

int fill_epilogue(ConvP& p, const mg_epilogue* epi) {
    if (epi) {
        p.e = *epi;
        if (p.e.scale && !p.e.shift) {
            mg_set_error("epilogue: scale without shift");
            return MG_EARG;
        }
    } else {
        p.e = mg_epilogue{};
    }
    // the kernel indexes outputs with 32-bit element offsets
    const long dense = (long)p.B * p.Tout * p.N, strided = (long)(p.B - 1) * p.ybs + (long)p.Tout * p.N;
    if (dense >= (1L << 31) || strided >= (1L << 31) || p.ybs >= (1L << 31)) {
        mg_set_error("conv_wgemm: output of %ld (strided %ld) elements exceeds the 2^31 element limit", dense, strided);
        return MG_EUNSUP;
    }
    return MG_OK;
}

}  // namespace

extern "C" int mg_conv1d_gather(const float* x, const float* w, float* y, int B, int Tin, int Cin, int N, int K,
                                int stride, int flip, int w_sn, int w_sc, long xbs, long ybs,
                                const mg_epilogue* epi, void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(x && w && y, "mg_conv1d_gather: null tensor");
    MG_CHECK_ARG(B > 0 && Tin > 0 && Cin > 0 && N > 0, "mg_conv1d_gather: bad shape B=%d Tin=%d Cin=%d N=%d", B, Tin, Cin, N);
    MG_CHECK_ARG(K == 1 || K == 3 || K == 5, "mg_conv1d_gather: K=%d unsupported", K);
    MG_CHECK_ARG(stride == 1 || stride == 2, "mg_conv1d_gather: stride=%d unsupported", stride);
    MG_CHECK_ARG(!(flip && stride != 1), "mg_conv1d_gather: flip requires stride 1");
    MG_CHECK_ARG(w_sn > 0 && w_sc > 0, "mg_conv1d_gather: bad weight strides");
    const int pad = (K - 1) / 2;
    const int Tout = (Tin + 2 * pad - K) / stride + 1;
    MG_CHECK_ARG(Tout > 0, "mg_conv1d_gather: Tout=%d", Tout);
    ConvP p{};
    p.x = x; p.w = w; p.y = y;
    p.B = B; p.Tin = Tin; p.Cin = Cin; p.Tm = Tout; p.Tout = Tout; p.N = N;
    p.xbs = xbs ? xbs : (long)Tin * Cin;
    p.ybs = ybs ? ybs : (long)Tout * N;
    p.w_sn = w_sn; p.w_sc = w_sc; p.flip = flip;
    p.work = work; p.work_bytes = work_bytes;
    { const long xb = ((long)(B - 1) * p.xbs + (long)Tin * Cin) * 4; p.x_bytes = xb < (1L << 31) ? xb : 0; }
    if (int rc = fill_epilogue(p, epi)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1) {
        if (K == 1) return launch_gather<1, 1>(p, s);
        if (K == 3) return launch_gather<1, 3>(p, s);
        return launch_gather<1, 5>(p, s);
    }
    if (K == 5) return launch_gather<2, 5>(p, s);
    mg_set_error("mg_conv1d_gather: stride 2 needs K=5");
    return MG_EUNSUP;
}

extern "C" int mg_conv1d_scatter2(const float* x, const float* w, float* y, int B, int Tin, int Cin, int N,
                                  int Tout, int w_sn, int w_sc, long xbs, long ybs, const mg_epilogue* epi,
                                  void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(x && w && y, "mg_conv1d_scatter2: null tensor");
    MG_CHECK_ARG(B > 0 && Tin > 0 && Cin > 0 && N > 0, "mg_conv1d_scatter2: bad shape");
    MG_CHECK_ARG(Tout == 2 * Tin || Tout == 2 * Tin - 1, "mg_conv1d_scatter2: Tout=%d must be 2*Tin or 2*Tin-1", Tout);
    ConvP p{};
    p.x = x; p.w = w; p.y = y;
    p.B = B; p.Tin = Tin; p.Cin = Cin; p.Tm = Tin; p.Tout = Tout; p.N = N;
    p.xbs = xbs ? xbs : (long)Tin * Cin;
    p.ybs = ybs ? ybs : (long)p.Tout * N;
    p.w_sn = w_sn; p.w_sc = w_sc; p.flip = 0;
    p.work = work; p.work_bytes = work_bytes;
    { const long xb = ((long)(B - 1) * p.xbs + (long)Tin * Cin) * 4; p.x_bytes = xb < (1L << 31) ? xb : 0; }
    if (int rc = fill_epilogue(p, epi)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (scatter_tile((long)B * Tin, N) == 12) return launch_cfg<2, 5, true, 1, 2>(p, s);
    return launch_cfg<2, 5, true, 1, 1>(p, s);
}

// Which template instantiation a call would launch: returns TM*10+TN of conv_wgemm_kernel<S,K,TR2,TM,TN>
// (22 = 128x128 tile, 11 = 64x64, 12 = 64x128 two-phase).  m_rows = B*Tout (gather) or B*Tin (scatter2).
// workspace that lets the window GEMMs split the channel reduction over workgroups (<= 8 output-sized slabs)
extern "C" size_t mg_conv_workspace_bytes(int B, int Tout, int N) { return (size_t)8 * B * Tout * N * sizeof(float); }

#ifdef MG_STAMPS
extern "C" int mg_dbg_set_stamps(long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(mg_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mg_conv_tile_config(long m_rows, int N, int scatter2) {
    return scatter2 ? scatter_tile(m_rows, N) : gather_tile(m_rows, N);
}
