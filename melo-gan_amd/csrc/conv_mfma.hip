// Window-GEMM kernels for Conv1d / ConvTranspose1d / Linear on channels-last activations.
//
// out[m, n] = sum_{tap, c} Xwin[m, tap, c] * W(n, c, ktap)      (fp32, v_mfma_f32_32x32x2_f32)
//
// With a (B, T, C) layout the im2col row of output position (b, t) is a window of
// consecutive input rows, so nothing is materialised: a workgroup stages the input rows its
// BM output positions need (once per 16-channel chunk) into LDS and every MFMA A-operand is
// a 16-byte read of 4 channels of one window row.  Weights keep the reference's layouts (see the
// public header); a 16-channel x K-tap x BN slab is transposed into LDS as [tap][channel quad][n][4]
// (LDS images and the loop's schedule: comment above conv_wgemm_kernel).
//
//   gather   (Conv1d fwd, ConvT1d dgrad, Linear fwd/dgrad, stride-1 Conv1d dgrad with flip):
//            window row of tap k for output t is t*S + k - (K-1)/2.
//   scatter2 (ConvT1d fwd stride 2 K=5 pad 2 outpad 1, Conv1d stride-2 dgrad): the two output
//            phases t=2u / t=2u+1 are two stride-1 correlations over taps {0,2,4} / {1,3}
//            sharing one input window [u-1, u+1]  (SURVEY hard part 6).
//
// fp32 MFMA issues one 32x32x2 every 64 cycles per SIMD: the matrix pipe -- not LDS bandwidth or
// HBM -- is the bound, provided nothing else sits on a wave's in-order issue path between MFMAs
// (tools/mfma_gap_fillers.hip measures what a gap tolerates).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

long g_conv_lds_pad = 0;      // mg_conv_set_lds_pad

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// -DMG_STAMPS: debug build that records wall-clock stamps (100 MHz) of workgroup phases; see tools/conv_stamps.py
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
    long w_bytes;    // same for w
    int w_sn, w_sc;
    int flip;
    int tt_log2;
    int n_ttiles;
    unsigned nt_magic;   // floor(2^32 / n_ttiles): mtile / n_ttiles = umulhi(mtile, nt_magic) (+1 correction)
    int ksplit;       // channel-chunk ranges handled by different workgroups (blockIdx.z); 1 => none
    int cps;          // chunks per split
    float* part;      // [ksplit][B*Tout*N] raw partial sums when ksplit > 1
    void* work;       // caller workspace (may be null)
    size_t work_bytes;
    int perm_L, perm_C;   // perm_L > 0 (K = 1, Linear forward): output column n is weight row conv_wrow(n) -- mg_linear_perm's order
    mg_epilogue e;
};

// weight row (and bias / scale / gscale entry) behind output column n
__device__ __forceinline__ int conv_wrow(const ConvP& p, int n) {
    return p.perm_L ? (n % p.perm_C) * p.perm_L + n / p.perm_C : n;
}

// channels per LDS chunk: 16 for K=3/5 taps, 64 for K=1 (Linear layers: fewer, fatter chunks)
template <int K> struct ChunkOf { static constexpr int value = (K == 1) ? 64 : 16; };

// LDS images (one chunk = BKC channels):
//   window  Xs[row][SX]            SX = BKC + 4 floats: 16-B aligned rows whose pitch is an odd number of
//                                  16-B granules, so the 16 lanes of a ds_read_b128 group hit 16 bank quads
//   weights Ws[plane][n][4]        one plane per (quad = channel/4, tap): the 4 channels of a quad are contiguous
//                                  per output column; plane order and pitch depend on the weight layout (below)
// Every MFMA operand is read with ds_read_b128: lane (i, h = lane>>5) takes channel quad 2g+h of its row /
// column, which feeds the 4 MFMAs s = 0..3 of channel group g (MFMA k-slot h <-> channel 8g + 4h + s on both
// operands).  One 16-B read per 4 MFMAs instead of one 4-B read per MFMA: with a single wave per SIMD the 4-B
// reads were latency-bound at ~13 cycles each and, not the matrix pipe, set the chunk time (MG_STAMPS runs).
template <int S, int K, bool TR2, bool NCK, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_wgemm_kernel(const ConvP p) {
    constexpr int BKC = ChunkOf<K>::value;
    constexpr int CG = BKC / 4;          // channel quads per chunk
    constexpr int G8 = BKC / 8;          // MFMA groups per chunk
    constexpr int SX = BKC + 4;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    // weight image: planes of [n][4] quads, one per (quad, tap).  The two weight layouts are staged differently (below)
    // and each gets the image its stores like: NCK plane = tap*CG + quad at pitch BN quads; CNK plane = quad*K + tap at
    // a pitch padded by (K mod 8) quads, so that lanes walking (n, tap) runs store to distinct bank quads.
    constexpr int PP = NCK ? BN * 4 : (BN + (K == 5 ? 5 : K == 3 ? 3 : 1)) * 4;   // floats per plane
    constexpr int PQ = NCK ? 1 : K;              // plane step per quad
    constexpr int PK = NCK ? CG : 1;             // plane step per tap
    constexpr int WSLAB = CG * K * PP;           // floats per chunk
    constexpr int SA = TR2 ? 1 : S;
    constexpr int NR = TR2 ? 3 : K;
    constexpr int NPH = TR2 ? 2 : 1;
    constexpr int PAD = (K - 1) / 2;
    constexpr int XQ = BKC / 4;                                        // float4 per window row
    constexpr int MAXX = (((BM - 1) * SA + NR) * XQ + 255) / 256 + 1;  // prefetch registers (float4) for X
    // weight staging units per thread.
    //   NCK (w[n][c][k]): unit = (column n, quad): 4K contiguous floats = K float4 loads; the quad of tap k sits in 4
    //     different prefetch registers and is stored as two ds_write2_b32 pairs (a register transpose through v_movs
    //     would cost ~17 cycles per v_mov, see below).  WS = 2K stores, WL = K loads.
    //   CNK (w[c][n][k]): unit = one quad (4 channels of one (column, tap)) = 4 dword loads from the 4 channel rows,
    //     lanes walking the contiguous (n, k) run, and ONE 16-B store.  WS = 1, WL = 4.  (The same scheme on NCK, whose
    //     dword loads gather 12-B pieces, measured 10 % slower than the float4 loads; on CNK it replaced float4 row
    //     loads + 8 pair stores that were unbalanced over the threads and 4-way bank-conflicted: 3-6 % faster.)
    constexpr int NW = NCK ? BN * CG / 256 : BN * CG * K / 256;
    constexpr int WS = NCK ? 2 * K : 1;
    constexpr int WL = NCK ? K : 4;
    static_assert((BN * CG) % 256 == 0, "weight units must split evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    MG_STAMP(0);
#ifdef MG_STAMPS
    const long long clk0 = clock64();
#endif
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int TT = 1 << p.tt_log2;
    const int TB = BM >> p.tt_log2;
    const int R = (TT - 1) * SA + NR;
    const int nrows = TB * R;
    const int xs_floats = nrows * SX;
    const int buf_floats = xs_floats + WSLAB;            // one {window, weight slab} buffer
    const int sink = 2 * buf_floats + 4 * tid;           // per-thread 16-B sink for staging slots without data

    const int mtile = blockIdx.x;
    // mtile / n_ttiles by the host's reciprocal (one s_mul_hi instead of a ~40-instruction scalar division)
    int mq = (int)__umulhi((unsigned)mtile, p.nt_magic);
    if ((mq + 1) * p.n_ttiles <= mtile) ++mq;
    const int b0 = mq * TB;
    const int t0 = (mtile - mq * p.n_ttiles) * TT;
    const int n0 = blockIdx.y * BN;
    const int tin0 = TR2 ? (t0 - 1) : (t0 * S - PAD);

    // every LDS offset used below is a multiple of 4 floats: tell hipcc, or it splits the 16-B accesses
    auto lds4 = [&](int off) { return reinterpret_cast<f32x4*>(__builtin_assume_aligned(smem + off, 16)); };

    // per-lane operand bases (float offsets inside a buffer)
    int abase[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int im = wm * 32 * TM + mi * 32 + (lane & 31);
        const int seg = im >> p.tt_log2, tl = im & (TT - 1);
        abase[mi] = (seg * R + tl * SA) * SX + 4 * (lane >> 5);
    }
    int bbase[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
        bbase[ni] = xs_floats + (lane >> 5) * PQ * PP + (wn * 32 * TN + ni * 32 + (lane & 31)) * 4;

    f32x16 acc[NPH][TM][TN];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][mi][ni][r] = 0.f;

    // split-K: this workgroup reduces over channels [c_begin, c_end) only
    const int c_begin = blockIdx.z * p.cps * BKC;
    const int c_end = min(p.Cin, c_begin + p.cps * BKC);

    // the MFMAs of channel group g (8 channels) out of the buffer at float offset boff
    auto group_mma = [&](int boff, int g) {
        f32x4 a[NR][TM], bw[K][TN];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[r][mi] = *lds4(boff + abase[mi] + r * SX + 8 * g);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                bw[k][ni] = *lds4(boff + bbase[ni] + (2 * g * PQ + k * PK) * PP);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (!TR2) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
                            acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][mi][s], bw[k][ni][s], acc[0][mi][ni], 0, 0, 0);
            } else {
                // phase 0 (t = 2u):   k=0 <- row u+1, k=2 <- row u, k=4 <- row u-1
                // phase 1 (t = 2u+1): k=1 <- row u+1, k=3 <- row u
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][mi][s], bw[0][ni][s], acc[0][mi][ni], 0, 0, 0);
                        acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][mi][s], bw[1][ni][s], acc[1][mi][ni], 0, 0, 0);
                        acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][mi][s], bw[2][ni][s], acc[0][mi][ni], 0, 0, 0);
                        acc[1][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][mi][s], bw[3][ni][s], acc[1][mi][ni], 0, 0, 0);
                        acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][mi][s], bw[4][ni][s], acc[0][mi][ni], 0, 0, 0);
                    }
            }
        }
    };

    // Fast path: every chunk and this block's weight slab are full and 16-B aligned: the next chunk travels
    // global -> registers (float4, issued a chunk ahead) -> LDS (16-B stores).
    const bool vec_ok = ((p.Cin & 3) == 0) && ((p.xbs & 3) == 0) && ((((uintptr_t)p.x) & 15) == 0);
    const bool w_aligned = ((((uintptr_t)p.w) & 15) == 0) &&
                           (NCK ? (p.w_sc == K && (p.w_sn & 3) == 0) : (p.w_sn == K && (p.w_sc & 3) == 0));
    const bool fast = vec_ok && w_aligned && (p.Cin % BKC == 0) && (n0 + BN <= p.N) && (nrows * XQ <= 256 * MAXX) &&
                      p.x_bytes > 0 && p.w_bytes > 0;

    if (fast) {
        // X and W loads go through raw buffer descriptors: slots of out-of-range rows carry an offset beyond
        // num_records, for which the hardware returns 0 -- no predicated load (hipcc branches and waits per
        // slot) and no arithmetic on the loaded registers (hipcc hoists it above the MFMAs and waits there);
        // both operands use the same load kind (with global_load next to buffer_load hipcc drained vmcnt(0)
        // right after issuing the prefetch).  The chunk's channel offset rides in the scalar offset operand.
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
        unsigned xo[MAXX];   // byte offset of the slot's float4 (without the chunk's channel offset)
        int xl[MAXX];        // float offset inside buffer 0 (or the sink)
        int xd[MAXX];        // what to add for buffer 1 (0 for the sink)
#pragma unroll
        for (int j = 0; j < MAXX; ++j) {
            const int idx = tid + 256 * j;
            xo[j] = 0x80000000u;
            xl[j] = sink;
            xd[j] = 0;
            if (idx < nrows * XQ) {
                const int row = idx / XQ, q = idx - row * XQ;
                const int seg = (TB == 1) ? 0 : row / R, r = row - seg * R;
                const int b = b0 + seg, tin = tin0 + r;
                xl[j] = row * SX + 4 * q;
                xd[j] = buf_floats;
                if (b < p.B && tin >= 0 && tin < p.Tin)
                    xo[j] = ((unsigned)b * (unsigned)p.xbs + (unsigned)(tin * p.Cin + 4 * q)) * 4u;    // < x_bytes < 2^31
            }
        }
        // weight units.  A flipped correlation (stride-1 dgrad) is flipped HERE, in the tap plane a quad is stored to,
        // so the operand reads of the loop keep compile-time offsets.
        unsigned wg[NW];     // byte offset of the unit's first load (without the chunk term)
        int wl[NW];          // float offset inside buffer 0 of the unit's store (NCK: of its tap-0 quad)
        const int wstep = (p.flip ? -PK : PK) * PP;                            // NCK: next tap's plane
        const unsigned wcc = (unsigned)p.w_sc * 4u;                            // CNK: byte step between the quad's channels
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + 256 * i;
            if constexpr (NCK) {
                // 8 consecutive lanes = 8 consecutive columns, then the quads
                const int n = (u & 7) + 8 * ((u >> 3) / CG), cg = (u >> 3) % CG;
                wg[i] = ((unsigned)conv_wrow(p, n0 + n) * (unsigned)p.w_sn + 4u * cg * K) * 4u;             // + c0*K per chunk
                wl[i] = xs_floats + ((p.flip ? (K - 1) * PK : 0) + cg * PQ) * PP + n * 4;              // + k*wstep per tap
            } else {
                const int cg = u / (BN * K), e = u - cg * (BN * K);
                const int n = e / K, k = e - n * K;
                wg[i] = (4u * cg * (unsigned)p.w_sc + (unsigned)(n0 * K + e)) * 4u;                          // + c0*w_sc per chunk
                wl[i] = xs_floats + (cg * PQ + (p.flip ? K - 1 - k : k) * PK) * PP + n * 4;
            }
        }
        f32x4 xr[MAXX], wr[NCK ? NW * K : NW];
        auto bload = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned soff) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
            return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        };
        auto x_soff = [&](int c0) { return 4u * (unsigned)c0; };
        auto w_soff = [&](int c0) { return (unsigned)((NCK ? (long)c0 * K : (long)c0 * p.w_sc) * 4); };
        auto load_x = [&](int j, unsigned soff) { xr[j] = bload(xrsrc, xo[j], soff); };
        auto load_w = [&](int i, int part, unsigned soff) {      // NCK: float4 `part` of the run; CNK: channel `part`
            if constexpr (NCK) wr[i * K + part] = bload(wrsrc, wg[i] + 16u * (unsigned)part, soff);
            else wr[i][part] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, wg[i], soff + (unsigned)part * wcc, 0));
        };
        auto store_x = [&](int j, int buf) { *lds4(xl[j] + (buf ? xd[j] : 0)) = xr[j]; };
        // NCK: the 4 values of a tap's quad come from 4 different prefetch registers.  They are stored as two pairs
        // (ds_write2_b32 takes two unrelated data registers): a ds_write in an MFMA gap is free, while the v_movs that
        // would gather them for one ds_write_b128 are not -- a VALU instruction behind an MFMA delays the next MFMA by
        // ~17 cycles (tools/mfma_gap_fillers.hip: 64.4 -> 82 cycles per MFMA with one v_mov per gap; one LDS/VMEM/SALU
        // filler per gap: 64-69).  This file is built with -fno-slp-vectorize so that hipcc does not re-vectorise the
        // pairs through v_movs; the pair is floats off and off+2, NOT adjacent, or it is merged into a ds_write_b64.
        auto store_w = [&](int i, int part, int buf) {           // NCK: part = 2*tap + half; CNK: part = 0
            if constexpr (NCK) {
                const int k = part >> 1, half = part & 1;
                float* d = smem + wl[i] + k * wstep + (buf ? buf_floats : 0) + half;
                const int e0 = half * K + k, e1 = (half + 2) * K + k;        // element cc*K + k of the unit's 4K floats
                d[0] = wr[i * K + e0 / 4][e0 % 4];
                d[2] = wr[i * K + e1 / 4][e1 % 4];
            } else {
                *lds4(wl[i] + (buf ? buf_floats : 0)) = wr[i];
            }
        };
        auto load_chunk = [&](int c0) {
#pragma unroll
            for (int j = 0; j < MAXX; ++j) load_x(j, x_soff(c0));
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int part = 0; part < WL; ++part) load_w(i, part, w_soff(c0));
        };
        // Gap-scheduled loop over two LDS buffers.  A chunk is NSLOT slots of 4 MFMAs (one tap x one channel
        // group).  A wave issues in order and each MFMA waits 64 cycles for its predecessor on the same
        // accumulator, so work hides under the matrix pipe only if it sits in the GAP right behind an MFMA in
        // program order, a few instructions per gap (measured with -DMG_STAMPS: staging placed after a chunk's
        // MFMAs, or clumped behind a slot's first MFMA, ran unoverlapped: 1.3 / 1.06 us per chunk against
        // 0.70 us of MFMA time).  Gap 0 of a slot holds the operand reads of the NEXT slot (other register
        // set); gaps 1-3 each hold at most a piece of the next chunk's round trip: prefetched registers ->
        // other LDS buffer (one 16-B store), registers reloaded with the chunk after that (one 16-B load).
        // sched_barrier pins every gap.  One barrier per chunk, in gap 0 of the last slot.  No branches: past
        // the end the last chunk is re-staged into the idle buffer; slots without data write to the sink.
        constexpr int NQ = TR2 ? 5 : K;             // MFMA quads per channel group
        constexpr int NSLOT = G8 * NQ;
        constexpr int NGAP = NSLOT * 3;             // staging gaps per chunk; stores only in the first (NSLOT-1)*3
        // Staging ops, one memory instruction each, over the units X slot 0..MAXX-1 (a 16-B store + a 16-B reload)
        // and weight unit 0..NW-1 (WS stores + WL reloads), in the order
        //     S(0) | S(1) L(0) | S(2) L(1) | ... | L(last)
        // so that a unit's reload never shares a gap with its own store (a load overwriting registers that a store
        // right in front of it still reads cost ~20 cycles in tools/mfma_gap_fillers.hip) and the tail of the list
        // is loads only: those may sit behind the chunk's barrier, in the gaps of the last slot.
        constexpr int NOPS = 2 * MAXX + NW * (WS + WL);
        constexpr int WBASE = 2 * MAXX + WS;        // first op behind {S(w0), L(x last)}
        constexpr int LAST_STORE = NW > 1 ? WBASE + (NW - 2) * (WS + WL) + WS - 1 : WBASE - 2;
        // ops spread evenly over the gaps, unless that would put a store behind the barrier: then the ops up to the
        // last store fill the gaps in front of it and the trailing loads the gaps of the last slot
        constexpr int GPRE = (NSLOT - 1) * 3;
        constexpr bool EVEN = LAST_STORE * NGAP / NOPS < GPRE;
        auto gap_of = [](int o) {
            if (EVEN) return o * NGAP / NOPS;
            return o <= LAST_STORE ? o * GPRE / (LAST_STORE + 1) : GPRE + (o - LAST_STORE - 1) * 3 / (NOPS - LAST_STORE - 1);
        };
        static_assert(NSLOT % 2 == 0, "operand register sets alternate per slot");
        f32x4 fa[2][TM], fb[2][TN];
        auto frag_read = [&](int boff, int slot, f32x4 (&A)[TM], f32x4 (&Bv)[TN]) {
            const int g = slot / NQ, q = slot - g * NQ;
            const int row = TR2 ? (q < 2 ? 2 : (q < 4 ? 1 : 0)) : q;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) A[mi] = *lds4(boff + abase[mi] + row * SX + 8 * g);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) Bv[ni] = *lds4(boff + bbase[ni] + (2 * g * PQ + q * PK) * PP);
        };
        const int c_last = c_end - BKC;
        auto chunk = [&](auto parity, int c0) {
            constexpr int P = decltype(parity)::value;      // buffer being read; the other one is being filled
            const int cur = P ? buf_floats : 0, oth = buf_floats - cur;
#ifdef MG_EXP_SAMECHUNK
            const int cn = c_begin;
#else
            const int cn = min(c0 + 2 * BKC, c_last);
#endif
            const unsigned xs = x_soff(cn), ws = w_soff(cn);
#pragma unroll
            for (int m = 0; m < NSLOT * 4; ++m) {
                const int sl = m / 4, s4 = m % 4;
                const int ph = TR2 ? ((sl % NQ) & 1) : 0;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl & 1][mi][s4], fb[sl & 1][ni][s4],
                                                                               acc[ph][mi][ni], 0, 0, 0);
                if (s4 == 0) {
#ifndef MG_EXP_NOBARRIER
                    if (sl == NSLOT - 1) __syncthreads();      // the other buffer is complete, this one is read out
#endif
                    if (sl + 1 < NSLOT) frag_read(cur, sl + 1, fa[(sl + 1) & 1], fb[(sl + 1) & 1]);
                    else frag_read(oth, 0, fa[0], fb[0]);
                } else {
                    const int gap = sl * 3 + s4 - 1;
#pragma unroll
                    for (int o = 0; o < NOPS; ++o) {
                        if (gap_of(o) != gap) continue;
                        // decode the op list above: unit stored / unit reloaded (X slots first), and which part
                        int su = -1, lu = -1, part = 0;
                        if (o < 2 * MAXX - 1) {
                            if (o == 0) su = 0;
                            else if (o & 1) su = (o + 1) / 2;
                            else lu = o / 2 - 1;
                        } else if (o < WBASE - 1) {
                            su = MAXX;
                            part = o - (2 * MAXX - 1);
                        } else if (o == WBASE - 1) {
                            lu = MAXX - 1;
                        } else {
                            const int i = (o - WBASE) / (WS + WL), r = (o - WBASE) % (WS + WL);
                            if (i < NW - 1 && r < WS) { su = MAXX + i + 1; part = r; }
                            else { lu = MAXX + i; part = i < NW - 1 ? r - WS : r; }
                        }
                        if (su >= 0) {
                            if (su < MAXX) {
#ifndef MG_EXP_NOXSTORE
                                store_x(su, 1 - P);
#endif
                            } else {
#ifndef MG_EXP_NOWSTORE
                                store_w(su - MAXX, part, 1 - P);
#endif
                            }
                        }
#ifndef MG_EXP_NOLOADS
                        if (lu >= 0) {
                            if (lu < MAXX) load_x(lu, xs);
                            else load_w(lu - MAXX, part, ws);
                        }
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load_chunk(c_begin);
        MG_STAMP(1);
#pragma unroll
        for (int j = 0; j < MAXX; ++j) store_x(j, 0);
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int part = 0; part < WS; ++part) store_w(i, part, 0);
        load_chunk(min(c_begin + BKC, c_last));
        __syncthreads();
        MG_STAMP(2);
        frag_read(0, 0, fa[0], fb[0]);
        for (int c0 = c_begin;;) {
            chunk(std::integral_constant<int, 0>{}, c0);
            c0 += BKC;
            if (c0 >= c_end) break;
            chunk(std::integral_constant<int, 1>{}, c0);
            c0 += BKC;
            if (c0 >= c_end) break;
        }
    } else {
        // generic path (ragged channel counts / unaligned tensors): scalar staging into the same LDS images
        for (int c0 = c_begin; c0 < c_end; c0 += BKC) {
            __syncthreads();
            for (int idx = tid; idx < nrows * BKC; idx += 256) {
                const int row = idx / BKC, cl = idx - row * BKC;
                const int seg = row / R, r = row - seg * R;
                const int b = b0 + seg, tin = tin0 + r, c = c0 + cl;
                float v = 0.f;
                if (b < p.B && tin >= 0 && tin < p.Tin && c < c_end)
                    v = p.x[(long)b * p.xbs + (long)tin * p.Cin + c];
                smem[row * SX + cl] = v;
            }
            for (int e = tid; e < BN * BKC * K; e += 256) {
                int n, c, k;
                if (NCK) {
                    n = e / (BKC * K);
                    const int ck = e - n * (BKC * K);
                    c = ck / K; k = ck - c * K;
                } else {
                    c = e / (BN * K);
                    const int nk = e - c * (BN * K);
                    n = nk / K; k = nk - n * K;
                }
                float v = 0.f;
                if (n0 + n < p.N && c0 + c < c_end) v = p.w[(long)conv_wrow(p, n0 + n) * p.w_sn + (long)(c0 + c) * p.w_sc + k];
                smem[xs_floats + ((c >> 2) * PQ + (p.flip ? K - 1 - k : k) * PK) * PP + n * 4 + (c & 3)] = v;
            }
            __syncthreads();
            const int ng = min(G8, (c_end - c0 + 7) >> 3);
            for (int g = 0; g < ng; ++g) group_mma(0, g);
        }
    }

    // ---- epilogue ----
    // Written as whole-tile passes over the accumulator registers, each selected by ONE uniform branch: with the
    // activation switch inside the per-element loop hipcc evaluated erff/tanhf for every element of every launch
    // and selected afterwards (4 us of a 16-us workgroup on the B=64 layers).
    MG_STAMP(3);
    const mg_epilogue& E = p.e;
    const long slab = (long)p.B * p.Tout * p.N;
    // LIN: the tile lies wholly inside the tensor and its rows are consecutive rows of a dense (B*Tout, N) output (gather
    // form; a tile of several sequences covers them whole) -- every row exists and element r sits a fixed number of rows
    // below the lane's first one: no per-element index arithmetic, no predicated stores.  Every emotion-discriminator launch.
    const bool lin_tile = !TR2 && p.ybs == (long)p.Tout * p.N && b0 + TB <= p.B && t0 + TT <= p.Tm && (TB == 1 || TT == p.Tm);
    auto epilogue_pass = [&](auto lin_tag) {
    constexpr bool LIN = decltype(lin_tag)::value;
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
                const unsigned lin0 = (unsigned)(((b0 * p.Tm + t0) + wm * 32 * TM + mi * 32 + 4 * (lane >> 5)) * p.N + n);
                auto index = [&](int r, unsigned& di, unsigned& yi) -> bool {
                    if (LIN) {
                        di = yi = lin0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * p.N);
                        return true;
                    }
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
                const int nw = conv_wrow(p, n);          // per-column vectors follow the weight row
                if (E.bias) {
                    const float bias = E.bias[nw];
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] += bias;
                }
                if (E.scale) {
                    const float scale = E.scale[nw], shift = E.shift[nw];
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
                // Elementwise operands are loaded UNCONDITIONALLY, row-clamped (element 0 always exists), all sixteen in
                // flight, and applied afterwards: behind `if (row is inside)` hipcc emitted branch + load + wait per
                // element -- sixteen dependent memory round trips, 8-12 us of the emotion discriminator's data-gradient
                // launches.  Values of rows outside the tensor are never stored.
                if (E.gref) {
                    float g[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] = E.gref[index(r, di, yi) ? di : 0u];
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
                    for (int r = 0; r < 16; ++r) g[r] = E.emul[index(r, di, yi) ? di : 0u];
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] *= g[r];
                }
                if (E.gscale) {
                    const float gscale = E.gscale[nw];
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] *= gscale;
                }
                if (E.accumulate) {
                    float g[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] = p.y[index(r, di, yi) ? yi : 0u];
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] += g[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (index(r, di, yi)) p.y[yi] = a[r];
            }
        }
    }
    };
    if (lin_tile) epilogue_pass(std::true_type{});
    else epilogue_pass(std::false_type{});
    MG_STAMP(4);
#ifdef MG_STAMPS
    if (threadIdx.x == 0 && mg_stamp_buf)    // shader-clock cycles over the workgroup's life: slot 15
    {
        long long* sb = mg_stamp_buf + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
        sb[15] = clock64() - clk0;
        // slot 14: HW_ID (wave/simd/cu/sh/se) | XCC_ID << 32 -- which CU this workgroup ran on
        sb[14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif
}

// sums the split-K slabs in fixed order and applies the fused epilogue
// VEC: four consecutive output channels per thread with 16-byte loads / stores (N % 4 == 0, aligned tensors)
template <bool VEC>
__global__ void conv_finish_kernel(const float* __restrict__ part, float* __restrict__ y, long total, int Tout, int N,
                                   long ybs, int ksplit, const mg_epilogue e) {
    constexpr int NV = VEC ? 4 : 1;
    const long d0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * NV;
    if (d0 >= total) return;
    float v[NV];
    int nn[NV];
    long di[NV];
    bool ok[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { v[q] = 0.f; di[q] = d0 + q; nn[q] = (int)((d0 + q) % N); ok[q] = true; }
    for (int z = 0; z < ksplit; ++z) {
        if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(part + (long)z * total + d0);
            v[0] += t.x; v[NV > 1 ? 1 : 0] += t.y; v[NV > 2 ? 2 : 0] += t.z; v[NV > 3 ? 3 : 0] += t.w;
        } else {
            v[0] += part[(long)z * total + d0];
        }
    }
    mg_apply_epilogue_set<NV>(e, v, nn, di, ok);
    const long bt = d0 / N;
    const long yi = (bt / Tout) * ybs + (bt % Tout) * N + nn[0];
    if (VEC) {
        float4 o = make_float4(v[0], v[NV > 1 ? 1 : 0], v[NV > 2 ? 2 : 0], v[NV > 3 ? 3 : 0]);
        if (e.accumulate) {
            const float4 t = *reinterpret_cast<const float4*>(y + yi);
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        *reinterpret_cast<float4*>(y + yi) = o;
    } else {
        if (e.accumulate) v[0] += y[yi];
        y[yi] = v[0];
    }
}

template <int S, int K, bool TR2, int TM, int TN>
int launch_cfg(const ConvP& p0, hipStream_t stream) {
    ConvP p = p0;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int BKC = ChunkOf<K>::value, SX = BKC + 4;
    constexpr int SA = TR2 ? 1 : S;
    constexpr int NR = TR2 ? 3 : K;
    int lg = mg_ilog2_ceil(p.Tm);
    const int lgbm = mg_ilog2_ceil(BM);
    if (lg > lgbm) lg = lgbm;
    p.tt_log2 = lg;
    const int TT = 1 << lg, TB = BM >> lg;
    p.n_ttiles = (int)mg_cdiv(p.Tm, TT);
    p.nt_magic = p.n_ttiles > 1 ? (unsigned)((1ULL << 32) / (unsigned)p.n_ttiles) : 0xFFFFFFFFu;
    const int R = (TT - 1) * SA + NR;
    constexpr int PP = (BN + (K == 5 ? 5 : K == 3 ? 3 : 1)) * 4;     // the larger (padded, CNK) weight plane of the kernel
    const size_t lds1 = ((size_t)TB * R * SX + (size_t)(BKC / 4) * K * PP) * sizeof(float);
    size_t lds = 2 * lds1 + 256 * 4 * sizeof(float);   // two buffers + the per-thread staging sink
    // mg_conv_set_lds_pad: extra LDS per workgroup = fewer resident workgroups per CU, for launches that run BESIDE another
    // stream's critical path (the frozen emotion discriminator's branch): they leave wave slots, registers and LDS to it
    if (g_conv_lds_pad > 0 && lds + (size_t)g_conv_lds_pad <= 160 * 1024) lds += (size_t)g_conv_lds_pad;
    if (lds > 160 * 1024) {
        mg_set_error("conv_wgemm: LDS request %zu too large", lds);
        return MG_EUNSUP;
    }
    const bool nck = p.w_sc < p.w_sn;   // (c,k) contiguous for a fixed n
    auto kernel = nck ? &conv_wgemm_kernel<S, K, TR2, true, TM, TN> : &conv_wgemm_kernel<S, K, TR2, false, TM, TN>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[nck]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            mg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MG_EHIP;
        }
        attr_set[nck] = true;
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
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, p);
    MG_CHECK_LAUNCH("conv_wgemm");
    if (p.ksplit > 1) {
        const mg_epilogue& E = p.e;
        auto al16 = [](const void* q) { return q == nullptr || ((((uintptr_t)q) & 15) == 0); };
        const bool vec = (p.N % 4 == 0) && (p.ybs % 4 == 0) && al16(p.part) && al16(p.y) && al16(E.zout) && al16(E.gref) &&
                         al16(E.emul);
        if (vec)
            hipLaunchKernelGGL(conv_finish_kernel<true>, dim3((unsigned)mg_cdiv(total / 4, 256)), dim3(256), 0, stream,
                               (const float*)p.part, p.y, total, p.Tout, p.N, p.ybs, p.ksplit, p.e);
        else
            hipLaunchKernelGGL(conv_finish_kernel<false>, dim3((unsigned)mg_cdiv(total, 256)), dim3(256), 0, stream,
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
    { const long wb = ((long)(p.N - 1) * p.w_sn + (long)(p.Cin - 1) * p.w_sc + (p.w_sn < p.w_sc ? p.w_sn : p.w_sc)) * 4;
      p.w_bytes = wb < (1L << 31) ? wb : 0; }
    if (int rc = fill_epilogue(p, epi)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (K > 1 && (Cin <= 8 || N <= 8)) {
        const int rc = mg_conv_thin_dispatch(x, w, y, B, Tin, Cin, Tout, N, K, stride, flip, 0, w_sn, w_sc, p.xbs, p.ybs, &p.e, s);
        if (rc != MG_EUNSUP) return rc;
    }
    if (stride == 1) {
        if (K == 1) return launch_gather<1, 1>(p, s);
        if (K == 3) return launch_gather<1, 3>(p, s);
        return launch_gather<1, 5>(p, s);
    }
    if (K == 5) return launch_gather<2, 5>(p, s);
    mg_set_error("mg_conv1d_gather: stride 2 needs K=5");
    return MG_EUNSUP;
}

// nn.Linear forward with mg_linear_perm's output order on the 64x64-tile kernel (K = 1): decoder.pre.2 (512 -> 8192) at the
// 2B = 128 rows of the fused step, where the skinny kernel's 32x32 tiles re-read the weights four times (linear_skinny.hip
// routes here; alone 13.2 us against 20.3).  No split-K.
int mg_conv_linear_perm(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, const mg_epilogue* epi,
                        int perm_L, hipStream_t stream) {
    ConvP p{};
    p.x = x; p.w = w; p.y = y;
    p.B = M; p.Tin = 1; p.Cin = K; p.Tm = 1; p.Tout = 1; p.N = N;
    p.xbs = K; p.ybs = N;
    p.w_sn = w_sn; p.w_sc = 1; p.flip = 0;
    p.work = nullptr; p.work_bytes = 0;
    p.perm_L = perm_L > 1 ? perm_L : 0;
    p.perm_C = p.perm_L ? N / p.perm_L : 0;
    { const long xb = (long)M * K * 4; p.x_bytes = xb < (1L << 31) ? xb : 0; }
    { const long wb = ((long)(N - 1) * w_sn + K) * 4; p.w_bytes = wb < (1L << 31) ? wb : 0; }
    if (int rc = fill_epilogue(p, epi)) return rc;
    return launch_gather<1, 1>(p, stream);
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
    { const long wb = ((long)(p.N - 1) * p.w_sn + (long)(p.Cin - 1) * p.w_sc + (p.w_sn < p.w_sc ? p.w_sn : p.w_sc)) * 4;
      p.w_bytes = wb < (1L << 31) ? wb : 0; }
    if (int rc = fill_epilogue(p, epi)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 8) {
        const int rc = mg_conv_thin_dispatch(x, w, y, B, Tin, Cin, Tout, N, 5, 2, 0, 1, w_sn, w_sc, p.xbs, p.ybs, &p.e, s);
        if (rc != MG_EUNSUP) return rc;
    }
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

long mg_conv_lds_pad_value() { return g_conv_lds_pad; }      // conv_wino.hip

extern "C" int mg_conv_set_lds_pad(long bytes) {
    MG_CHECK_ARG(bytes >= 0 && bytes <= 120 * 1024, "mg_conv_set_lds_pad: 0..120 KiB");
    g_conv_lds_pad = bytes;
    return MG_OK;
}
