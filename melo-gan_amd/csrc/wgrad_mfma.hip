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
};

constexpr int RT = 64;            // reduction rows per LDS chunk (16 per wave)
constexpr int BA = 32, BB = 32;   // output tile per workgroup

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int S, int K>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradP p) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int RMAX = (RT - 1) * S + K;                 // L-window rows when one batch fills the chunk
    constexpr int NS4 = RT * (BA / 4) / 256;               // float4 prefetch slots for S  (= 2)
    constexpr int NL4 = (RMAX * (BB / 4) + 255) / 256 + 1; // float4 prefetch slots for L
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TT = 1 << p.tt_log2, TB = RT >> p.tt_log2;
    const int R = (TT - 1) * S + K;
    const int lrows = TB * R;
    float* Ss = smem;                    // [RT][BA]
    float* Ls = smem + RT * BA;          // [TB*R][BB]
    const int a0 = blockIdx.x * BA, b0 = blockIdx.y * BB;
    const int split = blockIdx.z;

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

    // bias gradient fused in: per chunk every thread adds its share of the staged rows' column sums
    float bsum = 0.f;
    const bool do_bias_s = p.bias_from == 1 && blockIdx.y == 0;
    const bool do_bias_l = p.bias_from == 2 && blockIdx.x == 0;
    auto bias_accum = [&](int seg_id) {
        if (seg_id >= p.nseg_bias) return;
        const int cx = tid & 31, rq = tid >> 5;      // 32 channels x 8 row lanes
        if (do_bias_s) {
#pragma unroll
            for (int r = rq; r < RT; r += 8) bsum += Ss[r * BA + cx];
        } else if (do_bias_l) {
            // each L row belongs to exactly one chunk: window rows [PAD, PAD + TT*S) of every segment
            for (int seg = 0; seg < TB; ++seg)
                for (int rr = PAD + rq; rr < PAD + TT * S; rr += 8) bsum += Ls[(seg * R + rr) * BB + cx];
        }
    };

    const bool fast = ((p.A & 3) == 0) && ((p.Bc & 3) == 0) && p.vec_ok && (lrows * (BB / 4) <= 256 * NL4);
    if (fast) {
        // the next chunk's S rows and L window are fetched with raw-buffer float4 loads (out-of-range slots
        // return 0 in hardware) while the current chunk is multiplied; single LDS buffer, two barriers per chunk
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
                bias_accum((g_begin + c / p.n_ttiles) < p.nbg0 ? 0 : 1);
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
                bias_accum(seg_id);
                compute();
            }
        }
    }

    // ---- add the four waves' accumulators tap by tap through LDS into the output tile ot[a][b*K + k] (the layout
    //      of `out`), then write the tile's rows -- K*32 contiguous floats each -- with coalesced stores.  (Writing
    //      each tap straight to out[(a*Bc + b)*K + k] scattered 4-byte stores 4*K bytes apart and cost 7-8 us per
    //      launch.) ----
#ifdef MG_EXP_NOEPI
    if (acc[0][0] != 12345.f) return;
#endif
    __syncthreads();
    float* red = smem;                                   // [4][32][33]
    float* ot = smem + 4 * 32 * 33;                      // [32][K*32 + 1]
    constexpr int OTS = K * 32 + 1;
    float* out = p.part ? p.part + (long)split * p.slab : p.out;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 33 + (lane & 31)] = acc[k][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            const int ar = idx >> 5, bc = idx & 31;
            ot[ar * OTS + bc * K + k] = (red[(0 * 32 + ar) * 33 + bc] + red[(1 * 32 + ar) * 33 + bc]) +
                                        (red[(2 * 32 + ar) * 33 + bc] + red[(3 * 32 + ar) * 33 + bc]);
        }
        __syncthreads();
    }
    {
        const int nb_valid = min(32, p.Bc - b0) * K;     // floats per row of this tile that exist in `out`
        for (int idx = tid; idx < 32 * K * 32; idx += 256) {
            const int ar = idx / (K * 32), j = idx - ar * (K * 32);
            const int a = a0 + ar;
            if (a < p.A && j < nb_valid) out[((long)a * p.Bc + b0) * K + j] = ot[ar * OTS + j];
        }
    }
    if (do_bias_s || do_bias_l) {        // 8 row-lane partials -> one value per channel of this tile
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            const int ch = (do_bias_s ? a0 : b0) + tid;
            if (ch < (do_bias_s ? p.A : p.Bc)) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) v += red[tid + 32 * g];
                if (p.part) out[p.wslab + ch] = v;
                else p.bias_out[ch] = v;
            }
        }
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

struct Plan {
    int tt_log2, TB, n_ttiles, nbg0, nbg1, nsplit, bps;
};

constexpr int MAX_SPLITS = 32;

Plan make_plan(int A, int Bc, int K, int nb0, int nb1, int Ts) {
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
    // aim for ~768 workgroups (3 per CU), at least 2 r-chunks per split, at most MAX_SPLITS slabs
    long want = mg_cdiv(768, tiles);
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

extern "C" int mg_wgrad(const float* s0, const float* l0, int nb0, const float* s1, const float* l1, int nb1,
                        float* out, float* bias_out, int bias_from, int Ts, int Tl, int A, int Bc, int K, int stride,
                        void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(bias_from >= 0 && bias_from <= 2 && ((bias_from == 0) == (bias_out == nullptr)),
                 "mg_wgrad: bias_out and bias_from (1: sums of S, 2: sums of L) must be given together");
    MG_CHECK_ARG(s0 && l0 && out && nb0 > 0, "mg_wgrad: null/empty segment 0");
    MG_CHECK_ARG(nb1 == 0 || (s1 && l1), "mg_wgrad: null segment 1");
    MG_CHECK_ARG(Ts > 0 && Tl > 0 && A > 0 && Bc > 0, "mg_wgrad: bad shape");
    MG_CHECK_ARG(K == 1 || K == 3 || K == 5, "mg_wgrad: K=%d unsupported", K);
    MG_CHECK_ARG(stride == 1 || stride == 2, "mg_wgrad: stride=%d unsupported", stride);
    const Plan pl = make_plan(A, Bc, K, nb0, nb1, Ts);
    const long wslab = (long)A * Bc * K;
    const long slab = wslab + (bias_from == 1 ? A : bias_from == 2 ? Bc : 0);
    if (pl.nsplit > 1 && (!work || work_bytes < (size_t)pl.nsplit * slab * sizeof(float))) {
        mg_set_error("mg_wgrad: workspace too small (%zu < %zu)", work_bytes, (size_t)pl.nsplit * slab * sizeof(float));
        return MG_EWORK;
    }
    WgradP p{};
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
    size_t lds_floats = (size_t)RT * BA + (size_t)pl.TB * R * BB;
    const size_t epi_floats = 4 * 32 * 33 + 32 * ((size_t)K * 32 + 1);     // the final cross-wave reduction reuses the buffer
    if (lds_floats < epi_floats) lds_floats = epi_floats;
    const size_t lds = lds_floats * sizeof(float);
    {
        auto ok = [](const float* q, long elems) { return q == nullptr || (((((uintptr_t)q) & 15) == 0) && elems * 4 < (1L << 31)); };
        p.vec_ok = ok(s0, (long)nb0 * Ts * A) && ok(l0, (long)nb0 * Tl * Bc) && ok(nb1 ? s1 : nullptr, (long)nb1 * Ts * A) &&
                   ok(nb1 ? l1 : nullptr, (long)nb1 * Tl * Bc);
    }
    dim3 grid((unsigned)mg_cdiv(A, BA), (unsigned)mg_cdiv(Bc, BB), (unsigned)pl.nsplit);
    hipStream_t st = (hipStream_t)stream;
#define MG_WG(S_, K_) hipLaunchKernelGGL((wgrad_kernel<S_, K_>), grid, dim3(256), lds, st, p)
    if (stride == 1) {
        if (K == 1) MG_WG(1, 1); else if (K == 3) MG_WG(1, 3); else MG_WG(1, 5);
    } else {
        if (K == 5) MG_WG(2, 5);
        else { mg_set_error("mg_wgrad: stride 2 needs K=5"); return MG_EUNSUP; }
    }
#undef MG_WG
    MG_CHECK_LAUNCH("wgrad_kernel");
    if (pl.nsplit > 1) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)mg_cdiv(slab, 64)), dim3(256), 0, st,
                           (const float*)work, out, bias_out, slab, wslab, pl.nsplit);
        MG_CHECK_LAUNCH("reduce_slabs_kernel");
    }
    return MG_OK;
}
