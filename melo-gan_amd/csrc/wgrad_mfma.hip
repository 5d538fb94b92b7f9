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
    int lin;        // K = 1, Ts = 1, aligned: the 32 x 128-tile Linear body (wgrad_lin_body)
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
    // (aligned tensors with sequences of >= 16 positions never get here: wgrad16_body, below, serves them)
    if (fast) {
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

// ---- the same tile on v_mfma_f32_16x16x4_f32 (round 2) --------------------------------------------------------------
// wgrad_body's four waves split the ROWS of a chunk and each keep the whole 32x32xK tile: the tile has to be added across
// the waves through LDS at the end (3.4 us per workgroup, 50-84 KB of LDS) and every wave reads both operands for all 32x32
// outputs.  Here the four waves split the TILE: wave (wa, wb) owns the 16(a) x 16(b) x K sub-tile over ALL rows of every
// chunk -- K accumulators of 4 registers, no cross-wave sum; the epilogue only re-orders the tile through LDS into dW's
// layout for coalesced stores.  Per 4-row block a wave reads S once and the L window K times (1 + K ds_read_b32 for K MFMAs,
// lane (i, kq) = column i, row 4g + kq): MFMA operand maps A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15].  LDS rows are
// padded so that the two row pairs of a half-wave hit disjoint banks: S rows one apart -> pitch 48 floats, L window rows
// S apart -> pitch 40 (stride 2) / 48 (stride 1).  Same staging, same gap-scheduling rules as wgrad_body; used when that
// body's gap-scheduled path applies (aligned tensors, sequences of >= 16 positions), otherwise wgrad_body runs.
constexpr int W16_SP = 48;
template <int S> struct W16LP { static constexpr int value = S == 2 ? 40 : 48; };

template <int S, int K>
__device__ __forceinline__ bool wgrad16_applies(const WgradP& p) {
    constexpr int RMAX = (RT - 1) * S + K;
    constexpr int NL4 = (RMAX * (BB / 4) + 255) / 256 + 1;
    const int TT = 1 << p.tt_log2, TB = RT >> p.tt_log2, R = (TT - 1) * S + K;
    return ((p.A & 3) == 0) && ((p.Bc & 3) == 0) && p.vec_ok && (TB * R * (BB / 4) <= 256 * NL4) && p.tt_log2 >= 4;
}

template <int S, int K>
__device__ __forceinline__ void wgrad16_body(const WgradP& p, const int bx, const int by, const int bz) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int RMAX = (RT - 1) * S + K;
    constexpr int NS4 = RT * (BA / 4) / 256;
    constexpr int NL4 = (RMAX * (BB / 4) + 255) / 256 + 1;
    constexpr int SP = W16_SP, LP = W16LP<S>::value;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4, wa = wave & 1, wb = wave >> 1;
    const int TT = 1 << p.tt_log2, TB = RT >> p.tt_log2;
    const int R = (TT - 1) * S + K;
    const int lrows = TB * R;
    const int a0 = bx * BA, b0 = by * BB;
    const int g_begin = bz * p.bps, g_end = min(g_begin + p.bps, p.n_bgroups);
    const int n_chunks = (g_end - g_begin) * p.n_ttiles;
    float* out = p.part ? p.part + (long)bz * p.slab : p.out;
    MG_STAMP(0);

    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (n_chunks > 0) {
        const int buf_floats = RT * SP + lrows * LP;
        const int sink = 2 * buf_floats + 4 * tid;
        // Staging slots.  Slot j of a thread is the float4 (row idx/8, column quad idx%8) of the S rows / the L window.  Its
        // byte offset has a part fixed for the whole workgroup (s_base / l_base, relative to the chunk's first batch row
        // and, for the L window, to time t0*S - PAD of that row) and a wave-uniform part per chunk.  Nothing is compared per
        // chunk: the chunk's buffer descriptor starts at its first batch row and covers exactly the batch rows that exist
        // (a chunk of long sequences, TB = 1: ONE sequence), so window rows before time 0 (negative offset = huge unsigned),
        // past the sequence end and batch rows past the segment's end are out of range and read as zero in hardware.
        // Chunks of several short sequences (TB > 1) always start at time 0 (one time tile per sequence), so there the
        // time range check is static and folded into the slot.  (Round 1 recomputed three compares per slot and two
        // integer divisions per chunk ahead of the chunk's first MFMA: 7 us of the critic's 72-us launch.)
        constexpr unsigned DEAD = 0xC0000000u;          // stays out of range whatever a chunk adds (< 2^30, host checked)
        int s_lds[NS4];
        unsigned s_base[NS4];
#pragma unroll
        for (int j = 0; j < NS4; ++j) {
            const int idx = tid + 256 * j, r = idx / (BA / 4), q = idx - r * (BA / 4);
            const int seg = r >> p.tt_log2, tl = r & (TT - 1);
            s_lds[j] = r * SP + 4 * q;
            s_base[j] = (unsigned)((((long)seg * p.Ts + tl) * p.A + a0 + 4 * q) * 4);
            if (a0 + 4 * q >= p.A || (TB > 1 && tl >= p.Ts)) s_base[j] = DEAD;
        }
        int l_lds[NL4], l_ldd[NL4];
        unsigned l_base[NL4];
#pragma unroll
        for (int j = 0; j < NL4; ++j) {
            const int idx = tid + 256 * j, row = idx / (BB / 4), q = idx - row * (BB / 4);
            const int seg = (TB == 1) ? 0 : row / R, rr = row - seg * R;
            const bool live = row < lrows;
            l_lds[j] = live ? RT * SP + row * LP + 4 * q : sink;
            l_ldd[j] = live ? buf_floats : 0;
            l_base[j] = (unsigned)(int)((((long)seg * p.Tl + rr - PAD) * p.Bc + b0 + 4 * q) * 4);
            if (!live || b0 + 4 * q >= p.Bc || (TB > 1 && (rr < PAD || rr - PAD >= p.Tl))) l_base[j] = DEAD;
        }
        f32x4 sr[NS4], lr[NL4];
        __amdgpu_buffer_rsrc_t srs, lrs;
        unsigned s_add = 0, l_add = 0;   // the chunk's wave-uniform offset parts (0 when TB > 1)
        int nc = 0, ng = g_begin, ntt = 0;              // the chunk the next loads fetch: index, batch group, time tile
        auto chunk_desc = [&]() {
            const int seg_id = ng < p.nbg0 ? 0 : 1;
            const int bg = seg_id ? ng - p.nbg0 : ng;
            const int bb0 = bg * TB;
            const int nbv = min(TB, p.nb[seg_id] - bb0);
            srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((seg_id ? p.s[1] : p.s[0]) + (long)bb0 * p.Ts * p.A), 0,
                                                    nbv * p.Ts * p.A * 4, 0x00020000);
            lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((seg_id ? p.l[1] : p.l[0]) + (long)bb0 * p.Tl * p.Bc), 0,
                                                    nbv * p.Tl * p.Bc * 4, 0x00020000);
            const int t0 = ntt << p.tt_log2;
            s_add = (unsigned)(t0 * p.A * 4);
            l_add = (unsigned)(t0 * S * p.Bc * 4);
        };
        auto chunk_next = [&]() {        // step to the next chunk unless already at the last one (which is then re-staged)
            if (nc + 1 < n_chunks) {
                ++nc;
                if (++ntt == p.n_ttiles) { ntt = 0; ++ng; }
            }
            chunk_desc();
        };
        auto bload = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned off) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        };
        auto lds4 = [&](int off) { return reinterpret_cast<f32x4*>(__builtin_assume_aligned(smem + off, 16)); };
        auto store_s = [&](int j, int buf) { *lds4(s_lds[j] + (buf ? buf_floats : 0)) = sr[j]; };
        auto store_l = [&](int j, int buf) { *lds4(l_lds[j] + (buf ? l_ldd[j] : 0)) = lr[j]; };

        // operand addresses: row block g covers chunk rows 4g .. 4g+3 (one per kq); block g's L window starts at window row
        // (4g >> tt_log2) * R + (4g & (TT-1)) * S -- a wave-uniform term per block, the lane adds kq * S rows and its column
        const int abase = kq * SP + 16 * wa + li;
        const int lbase = RT * SP + kq * S * LP + 16 * wb + li;
        constexpr int NG = RT / 4;                  // row blocks per chunk
        constexpr int NOPS = 2 * (NS4 + NL4);
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 fr[2];        // operand sets of two consecutive row blocks: [0..K-1] = the K taps of L, [7] = S
        auto block_off = [&](int g) { return (((4 * g) >> p.tt_log2) * R + ((4 * g) & (TT - 1)) * S) * LP; };
        auto read_a = [&](int boff, int g, f32x8& v) { v[7] = smem[boff + abase + 4 * g * SP]; };
        auto read_b = [&](int boff, int g, int k, f32x8& v) { v[k] = smem[boff + lbase + block_off(g) + k * LP]; };
        auto chunk = [&](auto parity, int c) {
            constexpr int P = decltype(parity)::value;
            const int cur = P ? buf_floats : 0, oth = buf_floats - cur;
            chunk_next();                            // descriptors of the chunk the reloads below fetch: c + 2
#pragma unroll
            for (int m = 0; m < NG * K; ++m) {
                const int g = m / K, k = m % K;
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[g & 1][7], fr[g & 1][k], acc[k], 0, 0, 0);
                // gap behind MFMA (g, k): operand k (and the S value with k == 0) of the next row block -- of the next chunk's
                // first block, from the other buffer, behind the barrier in the last block -- and one staging op
                const int gn = g + 1 < NG ? g + 1 : 0, bo = g + 1 < NG ? cur : oth;
                if (g + 1 == NG && k == 0) __syncthreads();       // other buffer complete, this one read out
                if (k == 0) read_a(bo, gn, fr[(g + 1) & 1]);
                read_b(bo, gn, k, fr[(g + 1) & 1]);
                if (g + 1 < NG && k == K - 1) {
#pragma unroll
                    for (int o = 0; o < NOPS; ++o) {
                        if (o * (NG - 1) / NOPS != g) continue;
                        const int j = o / 2;
                        if (j < NS4) {
                            if (o % 2 == 0) store_s(j, 1 - P);
                            else sr[j] = bload(srs, s_base[j] + s_add);
                        } else {
                            if (o % 2 == 0) store_l(j - NS4, 1 - P);
                            else lr[j - NS4] = bload(lrs, l_base[j - NS4] + l_add);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto load_all = [&]() {
#pragma unroll
            for (int j = 0; j < NS4; ++j) sr[j] = bload(srs, s_base[j] + s_add);
#pragma unroll
            for (int j = 0; j < NL4; ++j) lr[j] = bload(lrs, l_base[j] + l_add);
        };
        chunk_desc();
        load_all();
#pragma unroll
        for (int j = 0; j < NS4; ++j) store_s(j, 0);
#pragma unroll
        for (int j = 0; j < NL4; ++j) store_l(j, 0);
        if (n_chunks > 1) {
            chunk_next();
            load_all();
        }
        __syncthreads();
        read_a(0, 0, fr[0]);
#pragma unroll
        for (int k = 0; k < K; ++k) read_b(0, 0, k, fr[0]);
        MG_STAMP(1);
        for (int c = 0;;) {
            chunk(std::integral_constant<int, 0>{}, c);
            if (++c >= n_chunks) break;
            chunk(std::integral_constant<int, 1>{}, c);
            if (++c >= n_chunks) break;
        }
    }
    MG_STAMP(2);

    // ---- epilogue: the tile through LDS into dW's layout (row a = K*32 contiguous floats), coalesced stores ----
    // D layout of the 16x16 MFMA: column (b) = lane & 15, rows (a) = 4 * (lane >> 4) + r
    __syncthreads();
    constexpr int PE = K * 32 + 1;
    float* red = smem;                                   // [32][K*32 (+1)]
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(16 * wa + 4 * kq + r) * PE + (16 * wb + li) * K + k] = acc[k][r];
    __syncthreads();
    const int nb_valid = min(32, p.Bc - b0) * K;         // floats per row of this tile that exist in `out`
#pragma unroll 4
    for (int idx = tid; idx < 32 * K * 32; idx += 256) {
        const int ar = idx / (K * 32), j = idx - ar * (K * 32);
        if (a0 + ar < p.A && j < nb_valid) out[((long)(a0 + ar) * p.Bc + b0) * K + j] = red[ar * PE + j];
    }
    MG_STAMP(3);
}

// ---- Linear layers (K = 1, one row per batch element): 32 x 128 output tiles -------------------------------------------
// dW[a][b] = sum_r dy[r][a] * x[r][b] over <= a few hundred rows: almost no reduction, all output.  With the 32 x 32 tiles
// of the bodies above decoder.pre.2 (8192 x 512, 64 rows) was 4096 workgroups of ONE 64-row chunk each -- prologue and
// epilogue around 16 MFMAs per wave, 16 us for a 16.8-MB write.  Here a workgroup stages its 32 dy-columns once per
// 64-row chunk and each of its four waves multiplies them with its own 32 x-columns: 4x fewer workgroups, the S rows
// fetched once instead of four times, the 32 x 32 accumulator stored straight from registers (a lane's 32-lane row group
// is 128 contiguous bytes of dW).  Same slices / slabs / bias workgroups as the other bodies.
constexpr int LIN_BB = 128;
__device__ __forceinline__ bool wgrad_lin_applies(const WgradP& p) {
    return p.lin != 0;
}
__device__ __forceinline__ void wgrad_lin_body(const WgradP& p, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SPL = 36, LPL = 132;                   // LDS pitches: row pairs of a half-wave on disjoint banks
    float* Ss = smem;                                    // [64][SPL]
    float* Ls = smem + RT * SPL;                         // [64][LPL]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int a0 = bx * BA, b0 = by * LIN_BB;
    const int TB = RT;                                   // Ts = 1: a batch group is 64 rows
    const int g_begin = bz * p.bps, g_end = min(g_begin + p.bps, p.n_bgroups);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int g = g_begin; g < g_end; ++g) {
        const int seg_id = g < p.nbg0 ? 0 : 1;
        const int bb0 = (seg_id ? g - p.nbg0 : g) * TB;
        const int nb = p.nb[seg_id];
        const float* Sp = p.s[seg_id] + (long)bb0 * p.A;
        const float* Lp = p.l[seg_id] + (long)bb0 * p.Bc;
        const int rows = min(TB, nb - bb0);
        __syncthreads();
        // S: 64 rows x 8 quads = 512 float4 (2 per thread); L: 64 rows x 32 quads = 2048 float4 (8 per thread); rows past the
        // segment and columns past the matrix are staged as zeros (row-clamped unconditional loads, selected afterwards)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j, r = idx >> 3, q = idx & 7;
            const bool ok = r < rows && a0 + 4 * q < p.A;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Sp + (long)(ok ? r : 0) * p.A + (ok ? a0 + 4 * q : 0));
            *reinterpret_cast<f32x4*>(Ss + r * SPL + 4 * q) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, r = idx >> 5, q = idx & 31;
            const bool ok = r < rows && b0 + 4 * q < p.Bc;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Lp + (long)(ok ? r : 0) * p.Bc + (ok ? b0 + 4 * q : 0));
            *reinterpret_cast<f32x4*>(Ls + r * LPL + 4 * q) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#pragma unroll 8
        for (int r2 = 0; r2 < RT / 2; ++r2) {
            const int r = 2 * r2 + h;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ss[r * SPL + c], Ls[r * LPL + 32 * wave + c], acc, 0, 0, 0);
        }
    }
    float* out = p.part ? p.part + (long)bz * p.slab : p.out;
    const int b = b0 + 32 * wave + c;
    if (b < p.Bc) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int a = a0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (a < p.A) out[(long)a * p.Bc + b] = acc[i];
        }
    }
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

// Several independent weight gradients of one (stride, K) in ONE launch: the small layers' gradients are each a
// handful of workgroups at the launch floor (~4.7 us per launch and dependent boundary), together they fill the chip
// once.  Workgroups are numbered job by job; first[j] is job j's first workgroup.
struct WgradJobs {
    WgradP p[MG_MAX_WGRAD_JOBS];
    int first[MG_MAX_WGRAD_JOBS + 1];      // tile workgroups, job by job
    int bfirst[MG_MAX_WGRAD_JOBS + 1];     // then every job's bias workgroups (short: they fill the slots the first
    int n;                                 // finished tiles leave; ahead of the tiles they delayed a third of them by 5 us)
};
template <int S, int K>
__global__ __launch_bounds__(256, 2) void wgrad_multi_kernel(const WgradJobs J) {
    const int bid = (int)blockIdx.x;
    const bool tile = bid < J.first[J.n];
    const int* first = tile ? J.first : J.bfirst;
    int j = 0;
    while (j + 1 < J.n && bid >= first[j + 1]) ++j;       // uniform
    const int b = bid - first[j];
    const WgradP& p = J.p[j];          // a reference: the by-value copy landed in scratch memory once two bodies used it
    if (tile) {
        const int gxy = p.gx * p.gy, bz = b / gxy, r = b - bz * gxy;
        if (K == 1 && wgrad_lin_applies(p)) wgrad_lin_body(p, r % p.gx, r / p.gx, bz);
        else if (wgrad16_applies<S, K>(p)) wgrad16_body<S, K>(p, r % p.gx, r / p.gx, bz);
        else wgrad_body<S, K>(p, r % p.gx, r / p.gx, bz);
    } else {
        const int ncb = ((p.bias_from == 1 ? p.A : p.Bc) + 31) >> 5;
        wgrad_bias_body(p, b % ncb, b / ncb);
    }
}

// out[i] = sum_z part[z][i]; 64 elements x 4 slice-groups per block (fixed summation order => reproducible).
// Elements [0, wn) go to `out` (weight gradient), elements [wn, n) to `bias_out`.
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
    auto ok = [](const float* q, long elems) { return q == nullptr || (((((uintptr_t)q) & 15) == 0) && elems * 4 < (1L << 31)); };
    p.vec_ok = ok(s0, (long)nb0 * Ts * A) && ok(l0, (long)nb0 * Tl * Bc) && ok(nb1 ? s1 : nullptr, (long)nb1 * Ts * A) &&
               ok(nb1 ? l1 : nullptr, (long)nb1 * Tl * Bc) && (long)Tl * Bc * 4 < (1L << 30) && (long)Ts * A * 4 < (1L << 30);
    // the 16x16x4 body (same condition as wgrad16_applies): padded images (rows at pitch 48 / 40-48 floats), re-ordering tile
    {
        const int rmax = (RT - 1) * stride + K, nl4 = (rmax * (BB / 4) + 255) / 256 + 1;
        if ((A & 3) == 0 && (Bc & 3) == 0 && p.vec_ok && pl.TB * R * (BB / 4) <= 256 * nl4 && pl.tt_log2 >= 4) {
            const size_t lp16 = stride == 2 ? 40 : 48;
            lds_floats = 2 * ((size_t)RT * W16_SP + (size_t)pl.TB * R * lp16) + 256 * 4;
            const size_t epi16 = (size_t)32 * (K * 32 + 1);
            if (lds_floats < epi16) lds_floats = epi16;
        }
    }
    lds = lds_floats * sizeof(float);
    // (only where the wide tiles still fill the chip: the small Linear layers -- 16-64 tiles of 32 x 128 -- are faster on four
    //  times as many 32 x 32 tiles: critic fc 5.1 against 7.9 us)
    p.lin = (K == 1 && stride == 1 && Ts == 1 && Tl == 1 && (A & 3) == 0 && (Bc & 3) == 0 && p.vec_ok &&
             mg_cdiv(A, BA) * mg_cdiv(Bc, LIN_BB) >= 128) ? 1 : 0;
    if (p.lin) {
        const size_t lin_floats = (size_t)RT * (36 + 132);
        if (lds < lin_floats * sizeof(float)) lds = lin_floats * sizeof(float);
    }
    p.gx = (int)mg_cdiv(A, BA); p.gy = (int)mg_cdiv(Bc, p.lin ? LIN_BB : BB); p.gz = pl.nsplit;
    const int nbias = bias_from ? pl.nsplit * (int)mg_cdiv(bias_from == 1 ? A : Bc, 32) : 0;
    grid = dim3((unsigned)(p.gx * p.gy * p.gz + nbias));      // 1-D: the tile workgroups, then the bias workgroups
    return MG_OK;
}
}  // namespace

extern "C" int mg_wgrad_multi(const mg_wgrad_job* jobs, int n_jobs, int K, int stride, void* work, size_t work_bytes,
                              mg_stream_t stream);

// One weight gradient = a multi-job launch of one job (same kernel, same launch-level plan).
extern "C" int mg_wgrad(const float* s0, const float* l0, int nb0, const float* s1, const float* l1, int nb1,
                        float* out, float* bias_out, int bias_from, int Ts, int Tl, int A, int Bc, int K, int stride,
                        void* work, size_t work_bytes, mg_stream_t stream) {
    mg_wgrad_job j{};
    j.s0 = s0; j.l0 = l0; j.nb0 = nb0;
    j.s1 = s1; j.l1 = l1; j.nb1 = nb1;
    j.out = out; j.bias_out = bias_out; j.bias_from = bias_from;
    j.Ts = Ts; j.Tl = Tl; j.A = A; j.Bc = Bc;
    return mg_wgrad_multi(&j, 1, K, stride, work, work_bytes, stream);
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
        const int ntile = J.p[i].gx * J.p[i].gy * J.p[i].gz;
        J.first[i] = nblocks;
        nblocks += ntile;
        J.bfirst[i] = (int)grid.x - ntile;            // count for now; turned into a start below
    }
    J.first[n_jobs] = nblocks;
    for (int i = 0; i < n_jobs; ++i) {
        const int nb = J.bfirst[i];
        J.bfirst[i] = nblocks;
        nblocks += nb;
    }
    J.bfirst[n_jobs] = nblocks;
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
