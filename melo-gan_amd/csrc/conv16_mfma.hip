// Stride-2, 5-tap window GEMMs (Conv1d stride 2 forward, ConvTranspose1d stride 2 forward and both data-gradients --
// every convolution of the critic and of the generator) on v_mfma_f32_16x16x4_f32, in 32*RT x 32 output tiles.
//
// Why a second window-GEMM kernel.  conv_mfma.hip's 64x64 tiles (one 32x32x2 MFMA tile per wave) leave the B = 64
// layers with 64-128 workgroups for 256 CUs: it splits the channel reduction over workgroups and pays a second launch
// that sums the partial slabs (13 such launches per training step, ~5 us each, around a ~13-us kernel whose MFMA time
// is 4.3 us), and the 3B-row critic layers land on 384 workgroups -- 1.5 per CU, i.e. 75 % of the chip at best.
// A 16x16 MFMA tile is a quarter of a 32x32 one at the same FLOP rate (32 cycles for 2048 FLOP against 64 for 4096), so
// a 4-wave workgroup can own a 64x32 (RT = 2) or 32x32 (RT = 1) tile: 256-1024 workgroups for the same layers with the
// WHOLE channel reduction inside each -- no split, no partial slabs, no finish launch, bitwise run-to-run
// reproducible by construction -- and the 3B layers get 768 workgroups = 3 per CU.
//
// Weights come in the "WQ" layout, wq[((c/4)*5 + k)*N + n][c%4]: the four channels of a quad contiguous per (tap,
// output column), which IS the LDS image (20 planes of 32 columns x 16 B per 16-channel chunk), so staging a chunk's
// weights is 640 plain 16-byte copies.  The optimiser writes this layout next to the reference's (mg_adam_flat_wq), once
// per update and per direction a convolution is used in; mg_wq_relayout fills it from a state_dict tensor.
//
// MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D: column = lane & 15, rows 4 * (lane >> 4) + r.  Lane (i, kq) reads channel quad kq of its window row / weight
// column with ONE ds_read_b128 and feeds element s of it to MFMA s of the chunk's tap: k-slot kq <-> channel 4 kq + s on
// both operands.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Conv16P {
    const float* x;
    const float* wq;
    float* y;
    int B, Tin, Cin, Tm, Tout, N;
    long xbs, ybs;
    int x_bytes, w_bytes;     // extents for the buffer descriptors (host checked < 2^31)
    int tt_log2, n_ttiles;
    unsigned nt_magic;
    mg_epilogue e;
    // optional per-column statistics of the stored values v (the BatchNorm that follows needs no reduction pass): every
    // wave writes, for its 16 columns, part[(2*mtile + wm)][0][n] = sum v and [1][n] = sum v*v over its valid rows
    float* part;
    // optional temporal mean of the stored values (AdaptiveAvgPool1d(1) behind the critic's last convolution,
    // src/gan/models.py:148): pool[b][n] = pool_scale * sum_t v[b][t][n], written by the wave whose rows are sample b's whole
    // time axis (host checked: one 16*RT-row wave tile per sample) -- the pooling launch and its dependent boundary disappear
    float* pool;
    float pool_scale;
    // y_perm: store y[b*ybs + n*Tout + tout] instead of [b*ybs + tout*N + n] -- the (B, N*Tout) order a following Linear's
    // input view has (src/gan/models.py:70 `view(B, 256, L)` read backwards by the data-gradient); gref / zout / emul keep
    // the dense (b, tout, n) index.  Replaces a transpose launch.
    int y_perm;
    // mix: for output batch rows b < mix_rows ALSO write mix_out[i] = alpha[b] * mix_real[i] + (1 - alpha[b]) * v at the
    // element's y index i -- the gradient penalty's interpolate (src/gan/utils.py:76-79) riding in the launch that
    // produces the fake batch: x_hat needs no pass of its own
    const float* mix_real;
    const float* mix_alpha;
    float* mix_out;
    int mix_rows;
    // bnb: the launch's output is the gradient dy that reaches a train-mode BatchNorm + ReLU / LeakyReLU from above; with the
    // layer's forward tensors a (activation) and z (BatchNorm input) it also leaves the two per-column sums the BatchNorm's
    // backward needs -- bnb_part[(2*mtile + wm)][0][n] = sum g, [1][n] = sum g * x_hat, g = dy * act'(a), x_hat = (z - mean) * invstd
    // -- so that backward is ONE launch (mg_bn_train_bwd_parts) instead of a reduction pass plus an apply pass
    const float* bnb_a;
    const float* bnb_z;
    const float* bnb_mean;
    const float* bnb_invstd;
    double* bnb_part;
    int bnb_act;
};

constexpr int K5 = 5;
constexpr int BKC = 16;                 // channels per chunk
// window row pitch in floats, chosen so that the 16 lanes of every ds_read_b128 group -- 8 rows x 2 channel quads --
// hit 16 different bank quads: 20 for rows two apart (stride-2 gather), 24 for consecutive rows (transposed form)
template <bool TR2> struct PitchOf { static constexpr int value = TR2 ? 24 : 20; };
constexpr int BN = 32;
constexpr int WPLANE = BN * 4;          // floats of one (quad, tap) plane: 32 columns x 4 channels
constexpr int WSLAB = 4 * K5 * WPLANE;  // 20 planes = 10 KB per chunk
constexpr int MAXX = 3;                 // window float4 slots per thread (<= 192 window rows per tile)
constexpr int NWU = 3;                  // weight float4 slots per thread (640 per chunk over 256 threads)

// RID: the launch carries a rider (statistics, temporal mean, permuted output order, interpolate); the plain instantiation
// has none of that code (the riders cost ~0.8 us per launch in registers and index arithmetic even when unused)
template <bool TR2, int RT, bool RID>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const Conv16P p) {
    constexpr int BM = 32 * RT;
    constexpr int SX = PitchOf<TR2>::value;
    constexpr int SA = TR2 ? 1 : 2;     // window rows between consecutive tile positions
    constexpr int NR = TR2 ? 3 : 5;     // window rows one position touches
    constexpr int NPH = TR2 ? 2 : 1;    // output phases (t = 2u, 2u + 1) of the transposed form
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kq = lane >> 4;
    const int TT = 1 << p.tt_log2, TB = BM >> p.tt_log2;
    const int R = (TT - 1) * SA + NR;
    const int nrows = TB * R;
    const int xs_floats = nrows * SX;
    const int buf_floats = xs_floats + WSLAB;

    const int mtile = blockIdx.x;
    int mq = (int)__umulhi((unsigned)mtile, p.nt_magic);
    if ((mq + 1) * p.n_ttiles <= mtile) ++mq;
    const int b0 = mq * TB;
    const int t0 = (mtile - mq * p.n_ttiles) * TT;
    const int n0 = blockIdx.y * BN;
    const int tin0 = TR2 ? (t0 - 1) : (2 * t0 - 2);

    auto lds4 = [&](int off) { return reinterpret_cast<f32x4*>(__builtin_assume_aligned(smem + off, 16)); };

    int abase[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int im = (wm * RT + rt) * 16 + li;
        const int seg = im >> p.tt_log2, tl = im & (TT - 1);
        abase[rt] = (seg * R + tl * SA) * SX + 4 * kq;
    }
    const int bbase = xs_floats + kq * K5 * WPLANE + (16 * wn + li) * 4;

    // RT = 1 leaves one accumulator per phase: consecutive MFMAs would wait 40 cycles for each other (the dependent
    // latency of v_mfma_f32_16x16x4_f32) instead of issuing every 32, so the channel sum is kept in two accumulators
    // (channels 4kq + {0,2} and 4kq + {1,3}) that the epilogue adds.
    constexpr int KA = RT == 1 ? 2 : 1;
    f32x4 acc[NPH][RT][KA];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ka = 0; ka < KA; ++ka) acc[ph][rt][ka] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging plan: global -> registers (a chunk ahead) -> the other LDS buffer ----
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, p.w_bytes, 0x00020000);
    const int sink = 2 * buf_floats + 4 * tid;      // per-thread 16-B sink for slots without data
    constexpr int NST = MAXX + NWU;                 // staging units per thread: window slots, then weight slots
    unsigned so[NST];      // byte offset of the unit's float4 (without the chunk term)
    int sl[NST], sd[NST];  // float offset inside buffer 0 (or the sink) and what to add for buffer 1 (0 for the sink)
#pragma unroll
    for (int j = 0; j < MAXX; ++j) {
        const int idx = tid + 256 * j;
        so[j] = 0x80000000u;          // beyond num_records: the hardware returns 0
        sl[j] = sink;
        sd[j] = 0;
        if (idx < nrows * 4) {
            const int row = idx >> 2, q = idx & 3;
            const int seg = (TB == 1) ? 0 : row / R, r = row - seg * R;
            const int b = b0 + seg, tin = tin0 + r;
            sl[j] = row * SX + 4 * q;
            sd[j] = buf_floats;
            if (b < p.B && tin >= 0 && tin < p.Tin)
                so[j] = ((unsigned)b * (unsigned)p.xbs + (unsigned)(tin * p.Cin + 4 * q)) * 4u;
        }
    }
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
        const int u = tid + 256 * i;
        so[MAXX + i] = 0x80000000u;
        sl[MAXX + i] = sink;
        sd[MAXX + i] = 0;
        if (u < 4 * K5 * BN) {
            const int pl = u >> 5, n = u & 31;
            sl[MAXX + i] = xs_floats + u * 4;
            sd[MAXX + i] = buf_floats;
            if (n0 + n < p.N) so[MAXX + i] = ((unsigned)pl * (unsigned)p.N + (unsigned)(n0 + n)) * 16u;
        }
    }
    const unsigned wchunk = (unsigned)(K5 * p.N) * 16u;     // bytes between consecutive channel quads' plane groups
    f32x4 sr[NST];
    auto bload = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned soff) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };
    auto load_unit = [&](int u, unsigned xs, unsigned ws) { sr[u] = u < MAXX ? bload(xrsrc, so[u], xs) : bload(wrsrc, so[u], ws); };
    auto store_unit = [&](int u, int buf) { *lds4(sl[u] + (buf ? sd[u] : 0)) = sr[u]; };

    // One chunk = 5 slots of 4*RT MFMAs: slot q pairs tap k with a window-row offset ro (and, transposed, a phase):
    //   gather     : (k, ro) = (q, q)
    //   transposed : phase 0 (t = 2u):   k = 0 <- row u+1, k = 2 <- row u, k = 4 <- row u-1
    //                phase 1 (t = 2u+1): k = 1 <- row u+1, k = 3 <- row u            (window row 0 is u-1)
    auto slot_row = [](int q) { return TR2 ? (q < 2 ? 2 : (q < 4 ? 1 : 0)) : q; };
    auto slot_ph = [](int q) { return TR2 ? (q & 1) : 0; };
    f32x4 fa[2][RT], fb[2];
    // piece g of slot q's operands: g < RT -> the window rows of row tile g, g == RT -> the weight columns
    auto frag_piece = [&](int boff, int q, int g, f32x4 (&A)[RT], f32x4& Bv) {
        if (g < RT) A[g] = *lds4(boff + abase[g < RT ? g : 0] + slot_row(q) * SX);
        else Bv = *lds4(boff + bbase + q * WPLANE);
    };

    // Gap-scheduled loop (conv_mfma.hip): a wave issues in order, so whatever is to hide under the
    // matrix pipe sits in the gap right behind an MFMA, one memory instruction per gap (two where the gaps run out),
    // pinned by sched_barrier; left to hipcc the prefetch loads shared registers with the operand reads and every
    // ds_read waited for global memory.  Per chunk: GPS = 4*RT MFMAs per slot;
    //   gaps 0..RT of slot q      the RT+1 operand reads of slot q+1 (other register set)
    //   the remaining gaps of slots 0-3   the six staging units: registers (next chunk) -> other LDS buffer, then the
    //                                     registers reloaded with the chunk after that -- all stores before all loads
    //   slot 4                     gap 0: the chunk's barrier; gaps 1..RT+1: operand reads of the NEXT chunk's slot 0
    // No branches: past the end the last chunk is re-staged into the idle buffer.
    constexpr int GPS = 4 * RT, NSLOT = K5, NFR = RT + 1;
    constexpr int FREE = GPS - NFR;
    constexpr int NOPS = 2 * NST;
    constexpr int OPG = (NOPS + 4 * FREE - 1) / (4 * FREE);
    static_assert(NFR + 1 <= GPS && OPG * 4 * FREE >= NOPS, "gap plan");
    auto chunk = [&](auto parity, int c_next2) {
        constexpr int P = decltype(parity)::value;      // LDS buffer being read; also the operand register set of slot 0
        const int cur = P ? buf_floats : 0, oth = buf_floats - cur;
        const unsigned xs = 4u * (unsigned)c_next2, ws = (unsigned)(c_next2 >> 2) * wchunk;
#pragma unroll
        for (int m = 0; m < NSLOT * GPS; ++m) {
            const int q = m / GPS, g = m % GPS;
            const int s4 = g / RT, rt = g % RT, set = (q + P) & 1;
            acc[slot_ph(q)][rt][s4 % KA] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][rt][s4], fb[set][s4],
                                                                                 acc[slot_ph(q)][rt][s4 % KA], 0, 0, 0);
            if (q + 1 < NSLOT) {
                if (g < NFR) {
                    frag_piece(cur, q + 1, g, fa[set ^ 1], fb[set ^ 1]);
                } else {
                    const int idx = q * FREE + (g - NFR);
#pragma unroll
                    for (int o = idx * OPG; o < (idx + 1) * OPG && o < NOPS; ++o) {
                        if (o < NST) store_unit(o, 1 - P);
                        else load_unit(o - NST, xs, ws);
                    }
                }
            } else {
                if (g == 0) __syncthreads();            // the other buffer is complete, this one is read out
                else if (g <= NFR) frag_piece(oth, 0, g - 1, fa[set ^ 1], fb[set ^ 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int c_last = p.Cin - BKC;
    {
#pragma unroll
        for (int u = 0; u < NST; ++u) load_unit(u, 0u, 0u);
#pragma unroll
        for (int u = 0; u < NST; ++u) store_unit(u, 0);
        const int c1 = min(BKC, c_last);
#pragma unroll
        for (int u = 0; u < NST; ++u) load_unit(u, 4u * (unsigned)c1, (unsigned)(c1 >> 2) * wchunk);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NFR; ++g) frag_piece(0, 0, g, fa[0], fb[0]);
    }
    for (int c0 = 0;;) {
        chunk(std::integral_constant<int, 0>{}, min(c0 + 2 * BKC, c_last));
        c0 += BKC;
        if (c0 >= p.Cin) break;
        chunk(std::integral_constant<int, 1>{}, min(c0 + 2 * BKC, c_last));
        c0 += BKC;
        if (c0 >= p.Cin) break;
    }

    // ---- epilogue: lane holds column n, rows 4*kq + r of each 16x16 tile ----
    const mg_epilogue& E = p.e;
    const int n = n0 + 16 * wn + li;
    if (n >= p.N) return;
    const float bias = E.bias ? E.bias[n] : 0.f;
    const float scale = E.scale ? E.scale[n] : 1.f, shift = E.scale ? E.shift[n] : 0.f;
    const float gscale = E.gscale ? E.gscale[n] : 1.f;
    float st1 = 0.f, cnt = 0.f;
    double bs1 = 0.0, bs2 = 0.0;        // fp64 like the reduction pass they replace (a handful of values per lane)
    const float bnb_mu = (RID && p.bnb_part) ? p.bnb_mean[n] : 0.f, bnb_is = (RID && p.bnb_part) ? p.bnb_invstd[n] : 0.f;
    const bool want_stats = RID && (p.part || p.pool);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            f32x4 a = acc[ph][rt][0];
            if constexpr (KA == 2) a += acc[ph][rt][1];
            unsigned di[4], yi[4];
            bool ok[4], mixrow[4];
            int brow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int im = (wm * RT + rt) * 16 + 4 * kq + r;
                const int seg = im >> p.tt_log2, tl = im & (TT - 1);
                const int b = b0 + seg, t = t0 + tl;
                const int tout = TR2 ? 2 * t + ph : t;
                di[r] = (unsigned)((b * p.Tout + tout) * p.N + n);
                yi[r] = (unsigned)(b * (int)p.ybs + ((RID && p.y_perm) ? n * p.Tout + tout : tout * p.N + n));
                ok[r] = b < p.B && t < p.Tm && tout < p.Tout;
                mixrow[r] = RID && ok[r] && b < p.mix_rows;
                brow[r] = mixrow[r] ? b : 0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = (a[r] + bias) * scale + shift;
            if (E.zout) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok[r]) E.zout[di[r]] = a[r];
            }
            if (E.act == MG_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = mg_act(MG_ACT_RELU, a[r]);
            } else if (E.act == MG_ACT_LRELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = mg_act(MG_ACT_LRELU, a[r]);
            } else if (E.act == MG_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = mg_act(MG_ACT_GELU, a[r]);
            } else if (E.act == MG_ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = mg_act(MG_ACT_TANH, a[r]);
            }
            // elementwise operands: unconditional row-clamped loads, all in flight, applied afterwards (behind `if (ok)`
            // hipcc emitted branch + load + wait per element); rows outside the tensor are never stored
            if (E.gref) {
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = E.gref[ok[r] ? di[r] : 0u];
                if (E.gact == MG_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= mg_act_grad(MG_ACT_RELU, g[r]);
                } else if (E.gact == MG_ACT_LRELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= mg_act_grad(MG_ACT_LRELU, g[r]);
                } else if (E.gact == MG_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= mg_act_grad(MG_ACT_GELU, g[r]);
                } else if (E.gact == MG_ACT_TANH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= mg_act_grad(MG_ACT_TANH, g[r]);
                }
            }
            if (E.emul) {
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = E.emul[ok[r] ? di[r] : 0u];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] *= g[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] *= gscale;
            if (want_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok[r]) {
                        st1 += a[r];
                        cnt += 1.f;
                    }
                acc[ph][rt][0] = a;        // kept for the centred second pass below
            }
            if (E.accumulate) {
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = p.y[ok[r] ? yi[r] : 0u];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] += g[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ok[r]) p.y[yi[r]] = a[r];
            if (RID && p.bnb_part) {      // operands loaded unconditionally (row-clamped), all in flight, like gref
                float av[4], zv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    av[r] = p.bnb_a[ok[r] ? di[r] : 0u];
                    zv[r] = p.bnb_z[ok[r] ? di[r] : 0u];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok[r]) {
                        const double gy = (double)(a[r] * mg_act_grad(p.bnb_act, av[r]));
                        bs1 += gy;
                        bs2 += gy * (((double)zv[r] - (double)bnb_mu) * (double)bnb_is);
                    }
            }
            if (RID && p.mix_out) {
                float rv[4], al[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rv[r] = p.mix_real[mixrow[r] ? yi[r] : 0u];
                    al[r] = p.mix_alpha[brow[r]];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (mixrow[r]) p.mix_out[yi[r]] = fmaf(al[r], rv[r], (1.f - al[r]) * a[r]);
            }
        }
    if (RID && p.bnb_part) {      // the four lanes of a column -> one partial pair per wave and column
        bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
        bs2 += __shfl_xor(bs2, 16, 64); bs2 += __shfl_xor(bs2, 32, 64);
        if (kq == 0) {
            double* dst = p.bnb_part + (long)(2 * blockIdx.x + wm) * 2 * p.N + n;
            dst[0] = bs1;
            dst[p.N] = bs2;
        }
    }
    if (want_stats) {        // the four lanes of a column (row groups kq = 0..3) -> one partial per wave and column
        st1 += __shfl_xor(st1, 16, 64); st1 += __shfl_xor(st1, 32, 64);
        if (kq == 0 && p.pool && 2 * (int)blockIdx.x + wm < p.B) p.pool[(long)(2 * blockIdx.x + wm) * p.N + n] = st1 * p.pool_scale;
        if (p.part) {
            // Sum of squares about the WAVE's own mean (its <= 64 values are still in registers): the combining kernel adds
            // the partials by the parallel-variance rule in fp64, so nothing is ever formed as E[x^2] - mean^2 -- a channel
            // whose |mean| is far above its deviation keeps its variance digits.
            cnt += __shfl_xor(cnt, 16, 64); cnt += __shfl_xor(cnt, 32, 64);
            const float mw = cnt > 0.f ? st1 / cnt : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int im = (wm * RT + rt) * 16 + 4 * kq + r;
                        const int seg = im >> p.tt_log2, tl = im & (TT - 1);
                        const int b = b0 + seg, t = t0 + tl;
                        const int tout = TR2 ? 2 * t + ph : t;
                        const float d = acc[ph][rt][0][r] - mw;
                        if (b < p.B && t < p.Tm && tout < p.Tout) m2 += d * d;
                    }
            m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
            if (kq == 0) {
                float* dst = p.part + (long)(2 * blockIdx.x + wm) * 3 * p.N + n;
                dst[0] = st1;
                dst[p.N] = m2;
                dst[2 * p.N] = cnt;
            }
        }
    }
}

// wq[((c/4)*K + k)*N + n][c%4] = w[n*sn + c*sc + k]
__global__ void wq_relayout_kernel(const float* __restrict__ w, float* __restrict__ wq, int N, int Cc, int K, int sn, int sc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Cc * K) return;
    const int e = (int)(i & 3);
    const long r = i >> 2;
    const int n = (int)(r % N);
    const long r2 = r / N;
    const int k = (int)(r2 % K), q = (int)(r2 / K);
    wq[i] = w[(long)n * sn + (long)(4 * q + e) * sc + k];
}

template <bool TR2, int RT, bool RID>
int launch16(Conv16P p, hipStream_t stream) {
    constexpr int BM = 32 * RT;
    constexpr int SX = PitchOf<TR2>::value;
    constexpr int SA = TR2 ? 1 : 2;
    constexpr int NR = TR2 ? 3 : 5;
    int lg = mg_ilog2_ceil(p.Tm);
    const int lgbm = mg_ilog2_ceil(BM);
    if (lg > lgbm) lg = lgbm;
    p.tt_log2 = lg;
    const int TT = 1 << lg, TB = BM >> lg;
    p.n_ttiles = (int)mg_cdiv(p.Tm, TT);
    p.nt_magic = p.n_ttiles > 1 ? (unsigned)((1ULL << 32) / (unsigned)p.n_ttiles) : 0xFFFFFFFFu;
    const int R = (TT - 1) * SA + NR;
    if (TB * R * 4 > 256 * MAXX) return MG_EUNSUP;        // too many window rows for the staging plan (tiny Tm)
    const size_t lds = 2 * ((size_t)TB * R * SX + WSLAB) * sizeof(float) + 256 * 4 * sizeof(float);
    auto kernel = &conv16_kernel<TR2, RT, RID>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            mg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MG_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.n_ttiles * mg_cdiv(p.B, TB)), (unsigned)mg_cdiv(p.N, BN));
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, p);
    MG_CHECK_LAUNCH("conv16");
    return MG_OK;
}

}  // namespace

extern "C" int mg_wq_relayout(const float* w, float* wq, int N, int Cc, int K, int w_sn, int w_sc, mg_stream_t stream) {
    MG_CHECK_ARG(w && wq && N > 0 && Cc > 0 && K > 0 && Cc % 4 == 0 && w_sn > 0 && w_sc > 0, "mg_wq_relayout: bad args");
    const long total = (long)N * Cc * K;
    hipLaunchKernelGGL(wq_relayout_kernel, dim3((unsigned)mg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wq, N, Cc, K,
                       w_sn, w_sc);
    MG_CHECK_LAUNCH("wq_relayout");
    return MG_OK;
}

// Rows per tile: 64 (RT = 2: half the prologues / epilogues, weights staged half as often) unless 32-row tiles fill the
// chip's workgroup slots more evenly.  A launch of n workgroups on 256 CUs runs ceil(n / 256) deep on the fullest CU, so
// its balance is n / (ceil(n / 256) * 256): 384 workgroups are 1.5 per CU = 0.75, the same layer in 32-row tiles is 768
// = 3 per CU = 1.0.  Measured (tools/conv16_bench.py, MG_CONV16_RT=1|2): the critic's 3B data-gradients conv.2 14.5 -> 12.7
// us and conv.4 23.9 -> 20.9 us at 768 instead of 384 workgroups; every 128-workgroup layer 30-50 % faster at 256; layers at
// 256 / 512 / >= 768 workgroups are faster or equal with 64-row tiles.
static int pick_rt(long m_rows, int N) {
    if (const char* f = getenv("MG_CONV16_RT")) return atoi(f) == 1 ? 1 : 2;
    auto balance = [](long n) { return n >= 1024 ? 1.0 : (double)n / (double)(mg_cdiv(n, 256) * 256); };
    const long n2 = mg_cdiv(m_rows, 64) * mg_cdiv(N, BN), n1 = mg_cdiv(m_rows, 32) * mg_cdiv(N, BN);
    return balance(n1) > balance(n2) + 0.1 ? 1 : 2;
}

// 1 if mg_conv16 supports the shape (otherwise the caller uses mg_conv1d_gather / mg_conv1d_scatter2)
extern "C" int mg_conv16_supported(int B, int Tin, int Cin, int N, int transposed, int Tout) {
    if (B <= 0 || Tin <= 0 || Cin % BKC || N % BN) return 0;
    const int Tm = transposed ? Tin : (Tin + 4 - K5) / 2 + 1;
    if (Tm < 4) return 0;                                      // tiles of a few positions: not worth a plan
    if (transposed && !(Tout == 2 * Tin || Tout == 2 * Tin - 1)) return 0;
    const long xe = (long)B * Tin * Cin * 4, we = (long)N * Cin * K5 * 4, ye = (long)B * (transposed ? Tout : Tm) * N;
    if (!(xe < (1L << 31) && we < (1L << 31) && ye < (1L << 31))) return 0;
    // window rows of one tile must fit the staging plan (MAXX float4 slots per thread)
    const int BM = 32 * pick_rt((long)B * Tm, N);
    int lg = mg_ilog2_ceil(Tm);
    if (lg > mg_ilog2_ceil(BM)) lg = mg_ilog2_ceil(BM);
    const int TT = 1 << lg, TB = BM >> lg, R = (TT - 1) * (transposed ? 1 : 2) + (transposed ? 3 : 5);
    return TB * R * 4 <= 256 * MAXX;
}

// The tiling mg_conv16 picks for a shape: batch rows per tile (a tile never straddles a multiple of it), the number of
// partial-statistics rows a launch with `part` writes (2 per workgroup column block: grid.x * 2), and the positions per
// tile (64 or 32: which instantiation conv16_kernel<transposed, tile_rows / 32> runs -- lets a profiler label launches).
extern "C" int mg_conv16_plan(int B, int Tin, int N, int transposed, int* batch_rows_per_tile, int* part_rows, int* tile_rows) {
    const int Tm = transposed ? Tin : (Tin + 4 - K5) / 2 + 1;
    const int BM = 32 * pick_rt((long)B * Tm, N);
    int lg = mg_ilog2_ceil(Tm);
    if (lg > mg_ilog2_ceil(BM)) lg = mg_ilog2_ceil(BM);
    const int TT = 1 << lg, TB = BM >> lg;
    if (batch_rows_per_tile) *batch_rows_per_tile = TB;
    if (tile_rows) *tile_rows = BM;
    if (part_rows) *part_rows = 2 * (int)(mg_cdiv(Tm, TT) * mg_cdiv(B, TB));
    return MG_OK;
}

static int conv16_launch(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                         long xbs, long ybs, const mg_epilogue* epi, const mg_conv16_extra& ex, mg_stream_t stream);

// 1 if a launch of this shape can also write the temporal mean of its output: gather form, every sample's time axis is
// exactly one wave's rows of a tile (Tout = 32 with 64-row tiles, 16 with 32-row tiles)
extern "C" int mg_conv16_poolable(int B, int Tin, int Cin, int N) {
    if (!mg_conv16_supported(B, Tin, Cin, N, 0, 0)) return 0;
    const int Tm = (Tin + 4 - K5) / 2 + 1;
    return Tm == 16 * pick_rt((long)B * Tm, N);
}

extern "C" int mg_conv16_pool(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, long xbs, long ybs,
                              const mg_epilogue* epi, float* pool, float pool_scale, mg_stream_t stream) {
    MG_CHECK_ARG(pool != nullptr, "mg_conv16_pool: null pool tensor");
    MG_CHECK_ARG(mg_conv16_poolable(B, Tin, Cin, N), "mg_conv16_pool: shape B=%d Tin=%d Cin=%d N=%d is not poolable", B, Tin, Cin, N);
    MG_CHECK_ARG(!(epi && epi->accumulate), "mg_conv16_pool: the mean of an accumulating launch is not defined");
    mg_conv16_extra ex{};
    ex.pool = pool;
    ex.pool_scale = pool_scale;
    return conv16_launch(x, wq, y, B, Tin, Cin, N, 0, 0, xbs, ybs, epi, ex, stream);
}

extern "C" int mg_conv16(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                         long xbs, long ybs, const mg_epilogue* epi, mg_stream_t stream) {
    return conv16_launch(x, wq, y, B, Tin, Cin, N, transposed, Tout, xbs, ybs, epi, mg_conv16_extra{}, stream);
}

extern "C" int mg_conv16_stats(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                               long xbs, long ybs, const mg_epilogue* epi, float* part, mg_stream_t stream) {
    mg_conv16_extra ex{};
    ex.part = part;
    return conv16_launch(x, wq, y, B, Tin, Cin, N, transposed, Tout, xbs, ybs, epi, ex, stream);
}

extern "C" int mg_conv16_ex(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                            long xbs, long ybs, const mg_epilogue* epi, const mg_conv16_extra* extra, mg_stream_t stream) {
    return conv16_launch(x, wq, y, B, Tin, Cin, N, transposed, Tout, xbs, ybs, epi, extra ? *extra : mg_conv16_extra{}, stream);
}

static int conv16_launch(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                         long xbs, long ybs, const mg_epilogue* epi, const mg_conv16_extra& ex, mg_stream_t stream) {
    float* const part = ex.part;
    float* const pool = ex.pool;
    const float pool_scale = ex.pool_scale;
    MG_CHECK_ARG(x && wq && y, "mg_conv16: null tensor");
    MG_CHECK_ARG(!(part && epi && epi->accumulate), "mg_conv16: statistics of an accumulating launch are not defined");
    MG_CHECK_ARG(!pool || (!transposed && mg_conv16_poolable(B, Tin, Cin, N)), "mg_conv16: shape is not poolable");
    MG_CHECK_ARG(!ex.mix_out || (ex.mix_real && ex.mix_alpha && ex.mix_rows > 0 && ex.mix_rows <= B && !ex.y_perm),
                 "mg_conv16: mix needs real, alpha, 0 < rows <= B and the plain output order");
    MG_CHECK_ARG(!ex.y_perm || (ybs == 0 || ybs == (long)N * (transposed ? Tout : (Tin + 4 - K5) / 2 + 1)),
                 "mg_conv16: the permuted output order needs a dense y");
    const int Tm = transposed ? Tin : (Tin + 4 - K5) / 2 + 1;
    if (!transposed) Tout = Tm;
    MG_CHECK_ARG(mg_conv16_supported(B, Tin, Cin, N, transposed, Tout), "mg_conv16: unsupported shape B=%d Tin=%d Cin=%d N=%d", B, Tin, Cin, N);
    Conv16P p{};
    p.x = x; p.wq = wq; p.y = y;
    p.B = B; p.Tin = Tin; p.Cin = Cin; p.Tm = Tm; p.Tout = Tout; p.N = N;
    p.xbs = xbs ? xbs : (long)Tin * Cin;
    p.ybs = ybs ? ybs : (long)Tout * N;
    const long xb = ((long)(B - 1) * p.xbs + (long)Tin * Cin) * 4, yb = (long)(B - 1) * p.ybs + (long)Tout * N;
    MG_CHECK_ARG(xb < (1L << 31) && yb < (1L << 31) && p.ybs < (1L << 31), "mg_conv16: tensor exceeds the 2^31 limit");
    p.x_bytes = (int)xb;
    p.w_bytes = (int)((long)N * Cin * K5 * 4);
    if (epi) {
        p.e = *epi;
        MG_CHECK_ARG(!(p.e.scale && !p.e.shift), "epilogue: scale without shift");
    }
    p.part = part;
    p.pool = pool;
    p.pool_scale = pool_scale;
    p.y_perm = ex.y_perm ? 1 : 0;
    p.mix_real = ex.mix_real; p.mix_alpha = ex.mix_alpha; p.mix_out = ex.mix_out;
    p.mix_rows = ex.mix_out ? ex.mix_rows : 0;
    MG_CHECK_ARG(!ex.bnb_part || (ex.bnb_a && ex.bnb_z && ex.bnb_mean && ex.bnb_invstd && !ex.y_perm && !(epi && epi->accumulate) &&
                                  (ex.bnb_act == MG_ACT_RELU || ex.bnb_act == MG_ACT_LRELU)),
                 "mg_conv16: the BatchNorm-backward sums need a, z, mean, invstd, a ReLU / LeakyReLU layer, the plain output "
                 "order and a non-accumulating launch");
    p.bnb_a = ex.bnb_a; p.bnb_z = ex.bnb_z; p.bnb_mean = ex.bnb_mean; p.bnb_invstd = ex.bnb_invstd; p.bnb_part = (double*)ex.bnb_part;
    p.bnb_act = ex.bnb_act;
    hipStream_t s = (hipStream_t)stream;
    const int rt = pick_rt((long)B * Tm, N);
    int rc;
    const bool rid = p.part || p.pool || p.y_perm || p.mix_out || p.bnb_part;
    if (rid) {
        if (transposed) rc = rt == 2 ? launch16<true, 2, true>(p, s) : launch16<true, 1, true>(p, s);
        else rc = rt == 2 ? launch16<false, 2, true>(p, s) : launch16<false, 1, true>(p, s);
    } else {
        if (transposed) rc = rt == 2 ? launch16<true, 2, false>(p, s) : launch16<true, 1, false>(p, s);
        else rc = rt == 2 ? launch16<false, 2, false>(p, s) : launch16<false, 1, false>(p, s);
    }
    if (rc == MG_EUNSUP) mg_set_error("mg_conv16: Tm=%d needs more window rows than the staging plan holds", Tm);
    return rc;
}
