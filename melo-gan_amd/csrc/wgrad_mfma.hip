// Weight-gradient kernel (fp32 MFMA) for Conv1d / ConvTranspose1d / Linear on channels-last
// activations:
//
//   out[a][b][k] = sum_{segments, batch, u}  S[batch, u, a] * L[batch, u*stride + k - (K-1)/2, b]
//
// i.e. for every tap k a GEMM  Out_k = S^T (A x r) * L_k (r x Bc)  whose reduction index r runs over
// (batch, time).  Weight matrices are small (A, Bc <= 256) while r is long (up to 24576 rows), so the
// parallelism has to come from the reduction:
//   * a workgroup owns a 32(a) x 32(b) x K output tile and one slice of the batches;
//   * per 64-row r-chunk it stages the S rows and the L window in LDS (raw-buffer float4 prefetch of the
//     next chunk while the current one is multiplied) and its FOUR WAVES SPLIT THE CHUNK'S ROWS, each
//     keeping K accumulators (32x32 per tap) for the same output tile;
//   * at the end the four waves' accumulators are added through LDS, and slices write partial slabs that
//     reduce_slabs_kernel sums in a fixed order -- bitwise reproducible, no float atomics.
// Compared with a 64x64 tile per workgroup this gives 4x the workgroups per slab (e.g. the critic's
// conv.0: 8 tiles x 32 slices = 256 workgroups with 32 slabs instead of 2 x 64 = 128 with 64 slabs).
// The bias gradient (column sums of S or of L) can ride along in the same launch.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct WgradP {
    const float* s[2];
    const float* l[2];
    int nb[2];
    float* part;     // [nsplit][slab] partial slabs, or null: single slice, write out / bias_out directly
    float* out;
    float* bias_out;
    int Ts, Tl, A, Bc;
    int tt_log2, n_ttiles;
    int bps;        // batch groups per split
    int n_bgroups;  // total batch groups (over both segments)
    int nbg0;       // batch groups in segment 0
    long slab;      // A*Bc*K (+ bias entries): stride between partial slabs
    long wslab;     // A*Bc*K
    int bias_from;  // 0: none; 1: column sums of S (Conv1d / Linear bias); 2: column sums of L (ConvTranspose1d bias)
    int nseg_bias;  // how many segments contribute to the bias (penalty segment never does)
    int vec_ok;     // all four tensors 16-byte aligned and < 2 GiB (raw-buffer float4 path)
    int gx, gy, gz; // output tiles along A and Bc, slices; the launch starts with gz * ceil(C / 32) bias workgroups (bias_from != 0)
};

constexpr int RT = 64;            // reduction rows per LDS chunk (16 per wave)
constexpr int BA = 32, BB = 32;   // output tile per workgroup

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// -DMG_STAMPS: debug build that records wall-clock stamps (100 MHz) of workgroup phases; see tools/
#ifdef MG_STAMPS
__device__ long long* mg_wstamp_buf = nullptr;
#define MG_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (threadIdx.x == 0 && mg_wstamp_buf)                                                             \
            mg_wstamp_buf[((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = \
                wall_clock64();                                                                            \
    } while (0)
#else
#define MG_STAMP(k)
#endif

// One workgroup's work: output tile (bx, by) of slice bz of the weight gradient described by p.
template <int S, int K>
__device__ __forceinline__ void wgrad_body(const WgradP& p, const int bx, const int by, const int bz) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int RMAX = (RT - 1) * S + K;                 // L-window rows when one batch fills the chunk
    constexpr int NS4 = RT * (BA / 4) / 256;               // float4 prefetch slots for S  (= 2)
    constexpr int NL4 = (RMAX * (BB / 4) + 255) / 256 + 1; // float4 prefetch slots for L
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MG_STAMP(0);
    const int TT = 1 << p.tt_log2, TB = RT >> p.tt_log2;
    const int R = (TT - 1) * S + K;
    const int lrows = TB * R;
    float* Ss = smem;                    // [RT][BA]
    float* Ls = smem + RT * BA;          // [TB*R][BB]
    const int a0 = bx * BA, b0 = by * BB;
    const int split = bz;

    f32x16 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const int h = lane >> 5;
    const int g_begin = split * p.bps;
    const int g_end = min(g_begin + p.bps, p.n_bgroups);
    const int n_chunks = (g_end - g_begin) * p.n_ttiles;

    // this wave's quarter of the chunk: rows [wave*16, wave*16+16)
    auto compute = [&]() {
#ifdef MG_EXP_NOMMA
        return;
#endif
#pragma unroll
        for (int r2 = 0; r2 < RT / 8; ++r2) {
            const int r = wave * (RT / 4) + 2 * r2 + h;
            const int seg = r >> p.tt_log2, tl = r & (TT - 1);
            const float av = Ss[r * BA + (lane & 31)];
            const float* lrow = Ls + (seg * R + tl * S) * BB + (lane & 31);
#pragma unroll
            for (int k = 0; k < K; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, lrow[k * BB], acc[k], 0, 0, 0);
        }
    };

    const bool fast = ((p.A & 3) == 0) && ((p.Bc & 3) == 0) && p.vec_ok && (lrows * (BB / 4) <= 256 * NL4);
    if (fast && p.tt_log2 >= 4 && n_chunks > 0) {
        // ---- gap-scheduled loop over two LDS buffers (same rules as conv_mfma.hip's: a wave issues in order, so
        //      staging hides under the matrix pipe only as single LDS / VMEM instructions in the gaps between MFMAs;
        //      VALU work costs ~17 cycles per gap and is done once per chunk, ahead of its MFMAs).  With
        //      TT >= 16 a wave's 16 rows lie in one sequence, so every operand address is lane base + immediate. ----
        constexpr int BUF = 0;   // (documentation only)
        (void)BUF;
        const int buf_floats = RT * BA + lrows * BB;
        const int sink = 2 * buf_floats + 4 * tid;      // per-thread 16-B sink for L slots beyond the window
        // prefetch slots: LDS float offset inside buffer 0 and chunk-invariant pieces of the global offset
        int s_seg[NS4], s_tl[NS4], s_q[NS4];
#pragma unroll
        for (int j = 0; j < NS4; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx / (BA / 4);
            s_q[j] = idx - r * (BA / 4);
            s_seg[j] = r >> p.tt_log2;
            s_tl[j] = r & (TT - 1);
        }
        int l_seg[NL4], l_rr[NL4], l_q[NL4], l_lds[NL4], l_ldd[NL4];
#pragma unroll
        for (int j = 0; j < NL4; ++j) {
            const int idx = tid + 256 * j;
            const int row = idx / (BB / 4);
            l_q[j] = idx - row * (BB / 4);
            l_seg[j] = (TB == 1) ? 0 : row / R;
            l_rr[j] = row - l_seg[j] * R;
            const bool live = row < lrows;
            l_lds[j] = live ? RT * BA + 4 * idx : sink;
            l_ldd[j] = live ? buf_floats : 0;
            if (!live) l_seg[j] = 1 << 20;               // never a valid batch
        }
        f32x4 sr[NS4], lr[NL4];
        unsigned so[NS4], lo[NL4];       // byte offsets of the chunk being fetched (0x80000000 = reads as zero)
        __amdgpu_buffer_rsrc_t srs, lrs;
        // chunk-invariant part of every slot's byte offset (batch 0, time 0 of the chunk's window); the chunk adds one
        // wave-uniform term and three compares per slot -- the only VALU work per chunk
        unsigned s_base[NS4], l_base[NL4];
#pragma unroll
        for (int j = 0; j < NS4; ++j) {
            s_base[j] = (unsigned)((((long)s_seg[j] * p.Ts + s_tl[j]) * p.A + a0 + 4 * s_q[j]) * 4);
            if (a0 + 4 * s_q[j] >= p.A) s_seg[j] = 1 << 20;          // never valid
        }
#pragma unroll
        for (int j = 0; j < NL4; ++j) {
            l_base[j] = (unsigned)((((long)l_seg[j] * p.Tl + l_rr[j]) * p.Bc + b0 + 4 * l_q[j]) * 4);
            if (b0 + 4 * l_q[j] >= p.Bc) l_seg[j] = 1 << 20;
        }
        auto chunk_addr = [&](int c) {
            const int g = g_begin + c / p.n_ttiles, tt = c - (c / p.n_ttiles) * p.n_ttiles;
            const int seg_id = g < p.nbg0 ? 0 : 1;
            const int bg = seg_id ? g - p.nbg0 : g;
            const int nb = p.nb[seg_id];
            const int bb0 = bg * TB, t0 = tt * TT, tl0 = t0 * S - PAD;
            srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(seg_id ? p.s[1] : p.s[0]), 0,
                                                    (int)((long)nb * p.Ts * p.A * 4), 0x00020000);
            lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(seg_id ? p.l[1] : p.l[0]), 0,
                                                    (int)((long)nb * p.Tl * p.Bc * 4), 0x00020000);
            const unsigned s_add = (unsigned)((((long)bb0 * p.Ts + t0) * p.A) * 4);
            const unsigned l_add = (unsigned)((((long)bb0 * p.Tl + tl0) * p.Bc) * 4);     // tl0 < 0 wraps; the sum is right
            const int nbv = nb - bb0, tsv = p.Ts - t0;
#pragma unroll
            for (int j = 0; j < NS4; ++j)
                so[j] = (s_seg[j] < nbv && s_tl[j] < tsv) ? s_base[j] + s_add : 0x80000000u;
#pragma unroll
            for (int j = 0; j < NL4; ++j) {
                const int t = tl0 + l_rr[j];
                lo[j] = (l_seg[j] < nbv && t >= 0 && t < p.Tl) ? l_base[j] + l_add : 0x80000000u;
            }
        };
        auto bload = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned off) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        };
        auto lds4 = [&](int off) { return reinterpret_cast<f32x4*>(__builtin_assume_aligned(smem + off, 16)); };
        auto store_s = [&](int j, int buf) { *lds4(4 * (tid + 256 * j) + (buf ? buf_floats : 0)) = sr[j]; };
        auto store_l = [&](int j, int buf) { *lds4(l_lds[j] + (buf ? l_ldd[j] : 0)) = lr[j]; };

        // this lane's operand bases inside a buffer: rows wave*16 + 2*r2 + h of the chunk
        const int r0 = wave * (RT / 4) + h;
        const int abase = r0 * BA + (lane & 31);
        const int lbase = RT * BA + ((r0 >> p.tt_log2) * R + (r0 & (TT - 1)) * S) * BB + (lane & 31);
        constexpr int NR2 = RT / 8;                 // MFMA row-pairs per wave per chunk
        constexpr int NOPS = 2 * (NS4 + NL4);       // store / reload of every prefetch slot
        constexpr int NGAP = (NR2 - 1) * K;         // gaps ahead of the barrier
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 fr[2];        // operand sets of two consecutive row pairs: [0..K-1] = the K taps of L, [7] = S
        auto frag_read = [&](int boff, int r2) {
            f32x8 v;
            v[7] = smem[boff + abase + 2 * r2 * BA];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = smem[boff + lbase + (2 * r2 * S + k) * BB];
            return v;
        };
        auto chunk = [&](auto parity, int c) {
            constexpr int P = decltype(parity)::value;
            const int cur = P ? buf_floats : 0, oth = buf_floats - cur;
            chunk_addr(min(c + 2, n_chunks - 1));    // addresses of the chunk the reloads below fetch
#pragma unroll
            for (int m = 0; m < NR2 * K; ++m) {
                const int r2 = m / K, k = m % K;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[r2 & 1][7], fr[r2 & 1][k], acc[k], 0, 0, 0);
                if (k == 0) {
                    if (r2 == NR2 - 1) __syncthreads();           // other buffer complete, this one read out
                    if (r2 + 1 < NR2) fr[(r2 + 1) & 1] = frag_read(cur, r2 + 1);
                    else fr[0] = frag_read(oth, 0);
                }
                if (r2 < NR2 - 1) {
                    const int gap = r2 * K + k;
#pragma unroll
                    for (int o = 0; o < NOPS; ++o) {
                        if (o * NGAP / NOPS != gap) continue;
                        const int j = o / 2;
                        if (j < NS4) {
                            if (o % 2 == 0) store_s(j, 1 - P);
                            else sr[j] = bload(srs, so[j]);
                        } else {
                            if (o % 2 == 0) store_l(j - NS4, 1 - P);
                            else lr[j - NS4] = bload(lrs, lo[j - NS4]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto load_all = [&]() {
#pragma unroll
            for (int j = 0; j < NS4; ++j) sr[j] = bload(srs, so[j]);
#pragma unroll
            for (int j = 0; j < NL4; ++j) lr[j] = bload(lrs, lo[j]);
        };
        chunk_addr(0);
        load_all();
#pragma unroll
        for (int j = 0; j < NS4; ++j) store_s(j, 0);
#pragma unroll
        for (int j = 0; j < NL4; ++j) store_l(j, 0);
        if (n_chunks > 1) {              // (a single chunk -- the Linear layers' 64-row batches -- has nothing to prefetch)
            chunk_addr(1);
            load_all();
        }
        __syncthreads();
        fr[0] = frag_read(0, 0);
        MG_STAMP(1);
        for (int c = 0;;) {
            chunk(std::integral_constant<int, 0>{}, c);
            if (++c >= n_chunks) break;
            chunk(std::integral_constant<int, 1>{}, c);
            if (++c >= n_chunks) break;
        }
    } else if (fast) {
        // short sequences (TT < 16: a wave's rows straddle sequences): the pre-gap-scheduling loop.  The next chunk's
        // S rows and L window are fetched with raw-buffer float4 loads (out-of-range slots return 0 in hardware) while
        // the current chunk is multiplied; single LDS buffer, two barriers per chunk
        float4 sr[NS4], lr[NL4];
        auto load_chunk = [&](int c) {
            const int g = g_begin + c / p.n_ttiles, tt = c - (c / p.n_ttiles) * p.n_ttiles;
            const int seg_id = g < p.nbg0 ? 0 : 1;
            const int bg = seg_id ? g - p.nbg0 : g;
            const int nb = p.nb[seg_id];
            const int bb0 = bg * TB, t0 = tt * TT, tl0 = t0 * S - PAD;
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(seg_id ? p.s[1] : p.s[0]), 0, (int)((long)nb * p.Ts * p.A * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(seg_id ? p.l[1] : p.l[0]), 0, (int)((long)nb * p.Tl * p.Bc * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < NS4; ++j) {
                const int idx = tid + 256 * j;
                const int r = idx / (BA / 4), q = idx - r * (BA / 4);
                const int seg = r >> p.tt_log2, tl = r & (TT - 1);
                const int b = bb0 + seg, t = t0 + tl, a = a0 + 4 * q;
                unsigned off = 0x80000000u;
                if (b < nb && t < p.Ts && a < p.A) off = (unsigned)((((long)b * p.Ts + t) * p.A + a) * 4);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
                sr[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
#pragma unroll
            for (int j = 0; j < NL4; ++j) {
                const int idx = tid + 256 * j;
                const int row = idx / (BB / 4), q = idx - row * (BB / 4);
                const int seg = row / R, rr = row - seg * R;
                const int b = bb0 + seg, t = tl0 + rr, cc = b0 + 4 * q;
                unsigned off = 0x80000000u;
                if (row < lrows && b < nb && t >= 0 && t < p.Tl && cc < p.Bc)
                    off = (unsigned)((((long)b * p.Tl + t) * p.Bc + cc) * 4);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(lrs, off, 0, 0);
                lr[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int j = 0; j < NS4; ++j) *reinterpret_cast<float4*>(Ss + 4 * (tid + 256 * j)) = sr[j];
#pragma unroll
            for (int j = 0; j < NL4; ++j) {
                const int idx = tid + 256 * j;
                if (idx < lrows * (BB / 4)) *reinterpret_cast<float4*>(Ls + 4 * idx) = lr[j];
            }
        };
        if (n_chunks > 0) {
            load_chunk(0);
            store_chunk();
            __syncthreads();
            for (int c = 0; c < n_chunks; ++c) {
                const bool more = c + 1 < n_chunks;
#ifndef MG_EXP_NOLOADS
                if (more) load_chunk(c + 1);
#endif
                compute();
                __syncthreads();
                if (more) {
                    store_chunk();
                    __syncthreads();
                }
            }
        }
    } else {
        for (int g = g_begin; g < g_end; ++g) {
            const int seg_id = g < p.nbg0 ? 0 : 1;
            const int bg = seg_id ? g - p.nbg0 : g;
            const float* Sp = p.s[seg_id];
            const float* Lp = p.l[seg_id];
            const int nb = p.nb[seg_id];
            const int bb0 = bg * TB;
            for (int tt = 0; tt < p.n_ttiles; ++tt) {
                const int t0 = tt * TT;
                const int tl0 = t0 * S - PAD;
                __syncthreads();
                for (int idx = tid; idx < RT * BA; idx += 256) {
                    const int r = idx / BA, al = idx - r * BA;
                    const int seg = r >> p.tt_log2, tl = r & (TT - 1);
                    const int b = bb0 + seg, t = t0 + tl, a = a0 + al;
                    float v = 0.f;
                    if (b < nb && t < p.Ts && a < p.A) v = Sp[((long)b * p.Ts + t) * p.A + a];
                    Ss[r * BA + al] = v;
                }
                for (int idx = tid; idx < TB * R * BB; idx += 256) {
                    const int row = idx / BB, cl = idx - row * BB;
                    const int seg = row / R, rr = row - seg * R;
                    const int b = bb0 + seg, t = tl0 + rr, c = b0 + cl;
                    float v = 0.f;
                    if (b < nb && t >= 0 && t < p.Tl && c < p.Bc) v = Lp[((long)b * p.Tl + t) * p.Bc + c];
                    Ls[row * BB + cl] = v;
                }
                __syncthreads();
                compute();
            }
        }
    }

    // ---- add the four waves' accumulators through LDS and write this slice's tile: every wave parks all K
    //      accumulators in red[wave][k][a][b] (ONE barrier), then each thread sums the four waves for consecutive
    //      elements of the tile's rows in the layout of `out` (row a = K*32 contiguous floats: coalesced stores).
    //      (Tap-by-tap passes cost 2 barriers per tap, 3.9 us per workgroup; writing each tap straight to
    //      out[(a*Bc + b)*K + k] scattered 4-byte stores 4*K bytes apart.) ----
#ifdef MG_EXP_NOEPI
    if (acc[0][0] != 12345.f) return;
#endif
    MG_STAMP(2);
    float* red = smem;                                   // [4][KP][32][33]
    float* out = p.part ? p.part + (long)split * p.slab : p.out;
    // In passes of KP <= 3 taps: parking all K = 5 taps at once takes 4*5*32*33*4 = 84 KB of LDS -- more than the two
    // staging buffers (54 KB) and the reason only ONE workgroup fitted a CU.  With KP = 3 the epilogue fits the staging
    // allocation and two workgroups share a CU: the bias workgroups of the launch (wgrad_bias_body) run beside the tile
    // workgroups instead of taking whole CUs.  Summation order per element unchanged: (w0 + w1) + (w2 + w3).
    constexpr int KP = K > 3 ? 3 : K;
    const int nb_cols = min(32, p.Bc - b0);
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KP) {
        const int kn = (K - k0) < KP ? (K - k0) : KP;    // taps of this pass (compile-time after unrolling)
        __syncthreads();
#pragma unroll
        for (int k = k0; k < k0 + kn; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[((wave * KP + (k - k0)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 33 + (lane & 31)] = acc[k][r];
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < 32 * kn * 32; idx += 256) {
            const int ar = idx / (kn * 32), j = idx - ar * (kn * 32);
            const int bc = j / kn, kk = j - bc * kn;
            const int a = a0 + ar;
            const int o = (kk * 32 + ar) * 33 + bc;
            const float v = (red[o] + red[o + KP * 32 * 33]) + (red[o + 2 * KP * 32 * 33] + red[o + 3 * KP * 32 * 33]);
            if (a < p.A && bc < nb_cols) out[((long)a * p.Bc + b0 + bc) * K + k0 + kk] = v;
        }
    }
    MG_STAMP(3);
}

// The bias gradient -- column sums of S (Conv1d / Linear: bias_from 1) or of L (ConvTranspose1d: 2) over the segments
// that carry one -- is the work of EXTRA workgroups of the same launch, one per (slice, 32-column block): a streaming
// sum of the slice's rows (a few hundred rows x 128 B, ~1-2 us) into the slab's bias entries, which the slab reduction
// adds like the weight entries.  Round 1 had the tile workgroups of one tile row / column add up the rows they staged,
// at the head of every chunk: 8-16 dependent LDS reads ahead of the chunk's first MFMA in a quarter of the workgroups,
// which then finished last -- 20 us per training step.
__device__ __forceinline__ void wgrad_bias_body(const WgradP& p, const int cb, const int bz) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* bred = reinterpret_cast<f32x4*>(smem);          // 4 KB of the launch's dynamic allocation
    const int tid = threadIdx.x, q = tid & 7, rl = tid >> 3;          // column quad, row lane (32 of them)
    const bool from_s = p.bias_from == 1;
    const int C = from_s ? p.A : p.Bc, T = from_s ? p.Ts : p.Tl;
    const int TB = RT >> p.tt_log2;
    const int g_begin = bz * p.bps, g_end = min(g_begin + p.bps, p.n_bgroups);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int c = cb * 32 + 4 * q;
    for (int seg = 0; seg < p.nseg_bias && seg < 2; ++seg) {
        // batch rows of this slice inside segment `seg`
        const int gs = seg ? p.nbg0 : 0, ge = seg ? p.n_bgroups : p.nbg0;
        const int ga = max(g_begin, gs), gb = min(g_end, ge);
        if (ga >= gb) continue;
        const int b_lo = (ga - gs) * TB, b_hi = min((gb - gs) * TB, p.nb[seg]);
        const float* base = (from_s ? p.s[seg] : p.l[seg]) + (long)b_lo * T * C;
        const long rows = (long)(b_hi - b_lo) * T;
        if (c + 3 < C && (C & 3) == 0 && p.vec_ok) {
            long r = rl;
            for (; r + 32 * 7 < rows; r += 32 * 8) {          // eight rows in flight per thread: the loop is latency-bound
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (r + 32 * u) * C + c);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; r < rows; r += 32) acc += *reinterpret_cast<const f32x4*>(base + r * C + c);
        } else {
            for (long r = rl; r < rows; r += 32)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) acc[e] += base[r * C + c + e];
        }
    }
    bred[tid] = acc;
    __syncthreads();
    if (tid < 32) {
        const int ch = cb * 32 + tid;
        if (ch < C) {
            float v = 0.f;
#pragma unroll 8
            for (int g = 0; g < 32; ++g) v += bred[8 * g + (tid >> 2)][tid & 3];
            if (p.part) p.part[(long)bz * p.slab + p.wslab + ch] = v;
            else p.bias_out[ch] = v;
        }
    }
}

template <int S, int K>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradP p) {
    const int ncb = p.bias_from ? ((p.bias_from == 1 ? p.A : p.Bc) + 31) >> 5 : 0, nbias = ncb * p.gz;
    const int id = (int)blockIdx.x - nbias, gxy = p.gx * p.gy;
    if (id >= 0) {
        const int bz = id / gxy, r = id - bz * gxy;
        wgrad_body<S, K>(p, r % p.gx, r / p.gx, bz);
    } else {
        const int k = (int)blockIdx.x;
        wgrad_bias_body(p, k % ncb, k / ncb);
    }
}

// Several independent weight gradients of one (stride, K) in ONE launch: the small layers' gradients are each a
// handful of workgroups at the launch floor (~4.7 us per launch and dependent boundary), together they fill the chip
// once.  Workgroups are numbered job by job; first[j] is job j's first workgroup.
struct WgradJobs {
    WgradP p[MG_MAX_WGRAD_JOBS];
    int first[MG_MAX_WGRAD_JOBS + 1];
    int n;
};
template <int S, int K>
__global__ __launch_bounds__(256, 2) void wgrad_multi_kernel(const WgradJobs J) {
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;       // uniform
    const int b = (int)blockIdx.x - J.first[j];
    const WgradP& p = J.p[j];          // a reference: the by-value copy landed in scratch memory once two bodies used it
    const int ncb = p.bias_from ? ((p.bias_from == 1 ? p.A : p.Bc) + 31) >> 5 : 0, nbias = ncb * p.gz;
    const int id = b - nbias, gxy = p.gx * p.gy;
    if (id >= 0) {
        const int bz = id / gxy, r = id - bz * gxy;
        wgrad_body<S, K>(p, r % p.gx, r / p.gx, bz);
    } else {
        wgrad_bias_body(p, b % ncb, b / ncb);
    }
}

// out[i] = sum_z part[z][i]; 64 elements x 4 slice-groups per block (fixed summation order => reproducible).
// Elements [0, wn) go to `out` (weight gradient), elements [wn, n) to `bias_out`.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           float* __restrict__ bias_out, long n, long wn, int nsplit) {
    __shared__ float sh[4][64];
    const int ex = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + ex;
    float s = 0.f;
    if (i < n)
        for (int z = g; z < nsplit; z += 4) s += part[(long)z * n + i];
    sh[g][ex] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        const float v = (sh[0][ex] + sh[1][ex]) + (sh[2][ex] + sh[3][ex]);
        if (i < wn) out[i] = v;
        else bias_out[i - wn] = v;
    }
}

struct ReduceJobs {
    const float* part[MG_MAX_WGRAD_JOBS];
    float* out[MG_MAX_WGRAD_JOBS];
    float* bias_out[MG_MAX_WGRAD_JOBS];
    long n[MG_MAX_WGRAD_JOBS], wn[MG_MAX_WGRAD_JOBS];
    int nsplit[MG_MAX_WGRAD_JOBS];
    int first[MG_MAX_WGRAD_JOBS + 1];
    int njobs;
};
// reduce_slabs_kernel for the split jobs of a multi launch (same summation order per element)
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(const ReduceJobs J) {
    __shared__ float sh[4][64];
    int j = 0;
    while (j + 1 < J.njobs && (int)blockIdx.x >= J.first[j + 1]) ++j;
    const float* __restrict__ part = J.part[j];
    const long n = J.n[j], wn = J.wn[j];
    const int nsplit = J.nsplit[j];
    const int ex = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - J.first[j]) * 64 + ex;
    float s = 0.f;
    if (i < n)
        for (int z = g; z < nsplit; z += 4) s += part[(long)z * n + i];
    sh[g][ex] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        const float v = (sh[0][ex] + sh[1][ex]) + (sh[2][ex] + sh[3][ex]);
        if (i < wn) J.out[j][i] = v;
        else J.bias_out[j][i - wn] = v;
    }
}

struct Plan {
    int tt_log2, TB, n_ttiles, nbg0, nbg1, nsplit, bps;
};

constexpr int MAX_SPLITS = 32;

// want_slices > 0: the launch-level planner of mg_wgrad_multi has decided the slice count (clamped to what the job allows)
Plan make_plan(int A, int Bc, int K, int nb0, int nb1, int Ts, long want_slices = 0) {
    Plan pl;
    int lg = mg_ilog2_ceil(Ts);
    const int lgrt = mg_ilog2_ceil(RT);
    if (lg > lgrt) lg = lgrt;
    pl.tt_log2 = lg;
    pl.TB = RT >> lg;
    pl.n_ttiles = (int)mg_cdiv(Ts, 1 << lg);
    pl.nbg0 = (int)mg_cdiv(nb0, pl.TB);
    pl.nbg1 = nb1 > 0 ? (int)mg_cdiv(nb1, pl.TB) : 0;
    const int ngroups = pl.nbg0 + pl.nbg1;
    const long tiles = mg_cdiv(A, BA) * mg_cdiv(Bc, BB);
    // aim for ~256 workgroups, at least 2 r-chunks per split, at most MAX_SPLITS slabs
    long target = 256;      // one workgroup per CU: the gap-scheduled loop overlaps its own staging, and fewer slices
                            // mean fewer prologues / cross-wave epilogues per CU and fewer slabs to reduce
    if (const char* f = getenv("MG_WGRAD_TARGET")) target = atol(f);
    long want = want_slices > 0 ? want_slices : mg_cdiv(target, tiles);
    long max_by_work = ((long)ngroups * pl.n_ttiles) / 2;
    if (max_by_work < 1) max_by_work = 1;
    if (want > max_by_work) want = max_by_work;
    if (want > ngroups) want = ngroups;
    if (want > MAX_SPLITS) want = MAX_SPLITS;
    if (want < 1) want = 1;
    pl.bps = (int)mg_cdiv(ngroups, want);
    pl.nsplit = (int)mg_cdiv(ngroups, pl.bps);
    return pl;
}

}  // namespace

#ifdef MG_STAMPS
extern "C" int mg_dbg_set_wstamps(long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(mg_wstamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" size_t mg_wgrad_workspace_bytes(int A, int Bc, int K, int nb_total, int Ts) {
    // upper bound over any (nb0, nb1) split of nb_total: nsplit <= min(MAX_SPLITS, #batch groups);
    // includes room for a fused bias gradient of max(A, Bc) entries per slab
    int lg = mg_ilog2_ceil(Ts);
    const int lgrt = mg_ilog2_ceil(RT);
    if (lg > lgrt) lg = lgrt;
    const int TB = RT >> lg;
    long ns = mg_cdiv(nb_total, TB) + 1;
    if (ns > MAX_SPLITS) ns = MAX_SPLITS;
    return (size_t)ns * ((size_t)A * (size_t)Bc * (size_t)K + (size_t)(A > Bc ? A : Bc)) * sizeof(float);
}

namespace {
// Validates one weight-gradient description and fills the kernel parameters, the LDS request and the grid.
// `work` is where THIS job's partial slabs go (used only when the plan splits).
int build_wgrad(const float* s0, const float* l0, int nb0, const float* s1, const float* l1, int nb1, float* out,
                float* bias_out, int bias_from, int Ts, int Tl, int A, int Bc, int K, int stride, void* work,
                size_t work_bytes, WgradP& p, Plan& pl, size_t& lds, dim3& grid, long want_slices = 0) {
    MG_CHECK_ARG(bias_from >= 0 && bias_from <= 2 && ((bias_from == 0) == (bias_out == nullptr)),
                 "mg_wgrad: bias_out and bias_from (1: sums of S, 2: sums of L) must be given together");
    MG_CHECK_ARG(s0 && l0 && out && nb0 > 0, "mg_wgrad: null/empty segment 0");
    MG_CHECK_ARG(nb1 == 0 || (s1 && l1), "mg_wgrad: null segment 1");
    MG_CHECK_ARG(Ts > 0 && Tl > 0 && A > 0 && Bc > 0, "mg_wgrad: bad shape");
    MG_CHECK_ARG(K == 1 || K == 3 || K == 5, "mg_wgrad: K=%d unsupported", K);
    MG_CHECK_ARG(stride == 1 || stride == 2, "mg_wgrad: stride=%d unsupported", stride);
    if (stride == 2 && K != 5) {
        mg_set_error("mg_wgrad: stride 2 needs K=5");
        return MG_EUNSUP;
    }
    pl = make_plan(A, Bc, K, nb0, nb1, Ts, want_slices);
    const long wslab = (long)A * Bc * K;
    const long slab = wslab + (bias_from == 1 ? A : bias_from == 2 ? Bc : 0);
    if (pl.nsplit > 1 && (!work || work_bytes < (size_t)pl.nsplit * slab * sizeof(float))) {
        mg_set_error("mg_wgrad: workspace too small (%zu < %zu)", work_bytes, (size_t)pl.nsplit * slab * sizeof(float));
        return MG_EWORK;
    }
    p = WgradP{};
    p.s[0] = s0; p.l[0] = l0; p.nb[0] = nb0;
    p.s[1] = s1; p.l[1] = l1; p.nb[1] = nb1;
    p.part = pl.nsplit > 1 ? (float*)work : nullptr;     // one slice: no slabs, no reduce launch
    p.out = out;
    p.bias_out = bias_out;
    p.Ts = Ts; p.Tl = Tl; p.A = A; p.Bc = Bc;
    p.tt_log2 = pl.tt_log2; p.n_ttiles = pl.n_ttiles;
    p.bps = pl.bps; p.n_bgroups = pl.nbg0 + pl.nbg1; p.nbg0 = pl.nbg0;
    p.slab = slab;
    p.wslab = wslab;
    p.bias_from = bias_from;
    p.nseg_bias = 1;      // only segment 0 (the loss term) carries a bias gradient; the penalty segment has none
    const int TT = 1 << pl.tt_log2;
    const int R = (TT - 1) * stride + K;
    size_t lds_floats = 2 * ((size_t)RT * BA + (size_t)pl.TB * R * BB) + 256 * 4;   // two buffers + the staging sink
    const size_t epi_floats = (size_t)4 * (K > 3 ? 3 : K) * 32 * 33;     // the cross-wave reduction (passes of <= 3 taps) reuses the buffers
    if (lds_floats < epi_floats) lds_floats = epi_floats;
    lds = lds_floats * sizeof(float);
    auto ok = [](const float* q, long elems) { return q == nullptr || (((((uintptr_t)q) & 15) == 0) && elems * 4 < (1L << 31)); };
    p.vec_ok = ok(s0, (long)nb0 * Ts * A) && ok(l0, (long)nb0 * Tl * Bc) && ok(nb1 ? s1 : nullptr, (long)nb1 * Ts * A) &&
               ok(nb1 ? l1 : nullptr, (long)nb1 * Tl * Bc);
    p.gx = (int)mg_cdiv(A, BA); p.gy = (int)mg_cdiv(Bc, BB); p.gz = pl.nsplit;
    const int nbias = bias_from ? pl.nsplit * (int)mg_cdiv(bias_from == 1 ? A : Bc, 32) : 0;
    grid = dim3((unsigned)(p.gx * p.gy * p.gz + nbias));      // 1-D: the bias workgroups, then the tile workgroups
    return MG_OK;
}
}  // namespace

extern "C" int mg_wgrad(const float* s0, const float* l0, int nb0, const float* s1, const float* l1, int nb1,
                        float* out, float* bias_out, int bias_from, int Ts, int Tl, int A, int Bc, int K, int stride,
                        void* work, size_t work_bytes, mg_stream_t stream) {
    WgradP p;
    Plan pl;
    size_t lds;
    dim3 grid;
    const int rc = build_wgrad(s0, l0, nb0, s1, l1, nb1, out, bias_out, bias_from, Ts, Tl, A, Bc, K, stride, work,
                               work_bytes, p, pl, lds, grid);
    if (rc != MG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
#define MG_WG(S_, K_) hipLaunchKernelGGL((wgrad_kernel<S_, K_>), grid, dim3(256), lds, st, p)
    if (stride == 1) {
        if (K == 1) MG_WG(1, 1); else if (K == 3) MG_WG(1, 3); else MG_WG(1, 5);
    } else {
        MG_WG(2, 5);
    }
#undef MG_WG
    MG_CHECK_LAUNCH("wgrad_kernel");
    if (pl.nsplit > 1) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)mg_cdiv(p.slab, 64)), dim3(256), 0, st,
                           (const float*)work, out, bias_out, p.slab, p.wslab, pl.nsplit);
        MG_CHECK_LAUNCH("reduce_slabs_kernel");
    }
    return MG_OK;
}

extern "C" int mg_wgrad_multi(const mg_wgrad_job* jobs, int n_jobs, int K, int stride, void* work, size_t work_bytes,
                              mg_stream_t stream) {
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_WGRAD_JOBS, "mg_wgrad_multi: 1..%d jobs", MG_MAX_WGRAD_JOBS);
    WgradJobs J{};
    ReduceJobs Rj{};
    size_t lds_max = 0, used = 0;
    int nblocks = 0, rblocks = 0;
    // Launch-level plan: the jobs' workgroups share ONE pass over the chip, so the slice counts
    // are chosen together -- every workgroup gets about the same number of 64-row chunks and the total stays within
    // `target` workgroups.  (Planned job by job, ~256 workgroups each, the critic's three gradients were 768 workgroups =
    // three rounds of prologue + loop + epilogue per CU, and the jobs' workgroups differed 2x in length.)
    long want[MG_MAX_WGRAD_JOBS];
    {
        long target = 512;       // two workgroups per CU (54 KB of LDS each), all resident at once.  Measured, critic's three
                                 // 3B-row gradients / generator's three deconvolutions / training step: 256: 101 / 44 us,
                                 // 384: 77 / 38, 512: 72 / 37 (1.031 ms), 640: 78 / 41 (1.041), 768: 83 / 41, 1024: 78 / 44;
                                 // the per-job plan it replaces (~256 each, 768 in all): 82 / 51 (1.058 ms)
        if (const char* f = getenv("MG_WGRAD_TARGET")) target = atol(f);
        long tiles[MG_MAX_WGRAD_JOBS], chunks[MG_MAX_WGRAD_JOBS], total = 0;
        for (int i = 0; i < n_jobs; ++i) {
            const mg_wgrad_job& q = jobs[i];
            MG_CHECK_ARG(q.A > 0 && q.Bc > 0 && q.Ts > 0 && q.Tl > 0 && q.nb0 > 0 && q.nb1 >= 0 && q.s0 && q.l0 && q.out,
                         "mg_wgrad_multi: job %d: null/empty segment 0 or bad shape", i);
            const Plan pl = make_plan(q.A, q.Bc, K, q.nb0, q.nb1, q.Ts);
            tiles[i] = mg_cdiv(q.A, BA) * mg_cdiv(q.Bc, BB);
            chunks[i] = (long)(pl.nbg0 + pl.nbg1) * pl.n_ttiles;        // per tile
            total += tiles[i] * chunks[i];
        }
        long cpw = mg_cdiv(total, target);                               // chunks per workgroup
        for (int it = 0; it < 64; ++it) {
            long wgs = 0;
            for (int i = 0; i < n_jobs; ++i) {
                const mg_wgrad_job& q = jobs[i];
                want[i] = (chunks[i] + cpw / 2) / cpw;
                if (want[i] < 1) want[i] = 1;
                const Plan pl = make_plan(q.A, q.Bc, K, q.nb0, q.nb1, q.Ts, want[i]);
                wgs += tiles[i] * pl.nsplit;
            }
            if (wgs <= target) break;
            cpw += mg_cdiv(cpw, 16);
        }
    }
    for (int i = 0; i < n_jobs; ++i) {
        const mg_wgrad_job& q = jobs[i];
        Plan pl;
        size_t lds;
        dim3 grid;
        // every job's slabs get their own 256-byte aligned piece of the workspace
        const size_t need = mg_wgrad_workspace_bytes(q.A, q.Bc, K, q.nb0 + q.nb1, q.Ts);
        void* w = used < work_bytes && work ? (char*)work + used : nullptr;
        const int rc = build_wgrad(q.s0, q.l0, q.nb0, q.s1, q.l1, q.nb1, q.out, q.bias_out, q.bias_from, q.Ts, q.Tl, q.A,
                                   q.Bc, K, stride, w, w ? work_bytes - used : 0, J.p[i], pl, lds, grid, want[i]);
        if (rc != MG_OK) return rc;
        if (pl.nsplit > 1) {
            used += (need + 255) & ~(size_t)255;
            const int r = Rj.njobs++;
            Rj.part[r] = J.p[i].part; Rj.out[r] = q.out; Rj.bias_out[r] = q.bias_out;
            Rj.n[r] = J.p[i].slab; Rj.wn[r] = J.p[i].wslab; Rj.nsplit[r] = pl.nsplit;
            Rj.first[r] = rblocks;
            rblocks += (int)mg_cdiv(J.p[i].slab, 64);
        }
        lds_max = lds > lds_max ? lds : lds_max;
        J.first[i] = nblocks;
        nblocks += (int)grid.x;
    }
    J.first[n_jobs] = nblocks;
    J.n = n_jobs;
    Rj.first[Rj.njobs] = rblocks;
    hipStream_t st = (hipStream_t)stream;
#define MG_WG(S_, K_) hipLaunchKernelGGL((wgrad_multi_kernel<S_, K_>), dim3((unsigned)nblocks), dim3(256), lds_max, st, J)
    if (stride == 1) {
        if (K == 1) MG_WG(1, 1); else if (K == 3) MG_WG(1, 3); else MG_WG(1, 5);
    } else {
        MG_WG(2, 5);
    }
#undef MG_WG
    MG_CHECK_LAUNCH("wgrad_multi_kernel");
    if (Rj.njobs > 0) {
        hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, Rj);
        MG_CHECK_LAUNCH("reduce_slabs_multi_kernel");
    }
    return MG_OK;
}
