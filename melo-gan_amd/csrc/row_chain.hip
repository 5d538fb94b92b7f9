// Row chains: the small per-sample layer stacks of the hot path -- the numeric encoder (LayerNorm + 3 Linear,
// src/gan/feature_encoder.py:16-45) forward and data-gradient, the emotion classifier's tail (project + MLP + head +
// cross-entropy and all their data-gradients, src/emotion_discriminator/ed_model.py:61,86-95,147-165) and the critic's
// tail (fc + scoring head and their data-gradients, src/gan/models.py:149-169) -- as ONE launch each.
//
// Why.  Each of these layers is a 64..192-row x <= 256-column Linear: a few MFLOP.  As launches of their own they sit at
// the dependent-kernel floor (~5 us each: 27 Linear launches were 18 % of the training step for 3 % of its FLOPs).  A
// sample's chain has no cross-sample dependency, so one workgroup walks ONE row through the whole stack with the
// activations in LDS; the weights (<= 1.3 MB per chain) come from L2, which every workgroup shares.  What bounds a chain
// is a CU's L2 bandwidth (~64 B/clk): 200 KB of encoder weights ~1.5 us, the classifier tail's 1.3 MB ~9 us, against 4-10
// launches of ~5-8 us each.
//
// A chain is a short op list (kernel argument): LOAD / STORE between global rows and LDS vector slots, LAYERNORM,
// LIN_FWD (y = act(x W^T + b) [* mask], W (N, K) row-major: a wave reads a weight row with one coalesced 16-B-per-lane
// load, 16 rows in flight, and a butterfly reduces the 16 partial dot products in 17 shuffles), LIN_DGRAD (dx = dy W
// [* act'(gref)] [* mask], lanes along the contiguous input axis, the output axis split over lane groups and summed
// through LDS in a fixed order), the cross-entropy head and the critic's scoring head with their gradients.
// Deterministic: no atomics, fixed summation order.
#include "common.h"

namespace {

constexpr int CH_MAXV = MG_CHAIN_MAX_VEC;      // floats per vector slot
constexpr int CH_SLOTS = MG_CHAIN_SLOTS;
// 16 waves per row.  A chain is a sequence of DEPENDENT steps, each one L2 round trip long if -- and only if -- all of a
// step's loads are in flight at once: a 256x256 layer is 256 KB = 16 float4 per thread at 1024 threads (one round trip),
// against 64 per thread in 4-8 dependent rounds at 256 threads (the first version of this kernel: 40-200 us per chain).
constexpr int CH_THREADS = 1024;
constexpr int CH_WAVES = CH_THREADS / 64;

struct ChainArgs {
    mg_chain_op op[MG_CHAIN_MAX_OPS];
    int n_ops;
};

__device__ __forceinline__ float act_fwd(int act, float v) {
    switch (act) {      // uniform per op: one branch
        case MG_ACT_RELU: return v > 0.f ? v : 0.f;
        case MG_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case MG_ACT_GELU: return mg_gelu(v);
        case MG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ---- y[n] = sum_k x[k] W[n*ld + k], vector path: K % 4 == 0, 16 <= K <= CH_MAXV.  A wave takes 16 outputs per round:
// 16 coalesced 16-B-per-lane row loads in flight, then a butterfly folds the 16 partial dot products in 17 shuffles.
// The loads go through a buffer descriptor: the row (wave-uniform) is the scalar offset, the lane's column the one vector
// offset shared by all 16 -- no address registers, so the 16 x 4 data registers fit the 128-VGPR budget of a 1024-thread
// workgroup and hipcc keeps all 16 in flight (with 64-bit addresses per row it serialised them pairwise to save
// registers: 8 dependent round trips, 23 us per 256x256 layer instead of ~2). ----
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lin_fwd_vec(const mg_chain_op& o, const float* __restrict__ xs, float* __restrict__ raw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = o.n0, N = o.n1, nk4 = K >> 2;
    const unsigned ldb = (unsigned)o.ld0 * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.p0), 0,
                                                                         (int)(((long)(N - 1) * o.ld0 + K) * 4), 0x00020000);
    float4 xr[2];
    unsigned vo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = lane + 64 * t;
        xr[t] = c < nk4 ? reinterpret_cast<const float4*>(xs)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        vo[t] = c < nk4 ? (unsigned)c * 16u : 0x80000000u;          // beyond num_records: the hardware returns 0
    }
    auto ld16 = [&](unsigned voff, unsigned soff) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    for (int nb = wave * 16; nb < N; nb += 16 * CH_WAVES) {
        float4 w0[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w0[j] = ld16(vo[0], (unsigned)min(nb + j, N - 1) * ldb);
        __builtin_amdgcn_sched_barrier(0);
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) p[j] = (w0[j].x * xr[0].x + w0[j].y * xr[0].y) + (w0[j].z * xr[0].z + w0[j].w * xr[0].w);
        if (nk4 > 64) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) w0[j] = ld16(vo[1], (unsigned)min(nb + j, N - 1) * ldb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] += (w0[j].x * xr[1].x + w0[j].y * xr[1].y) + (w0[j].z * xr[1].z + w0[j].w * xr[1].w);
        }
        // butterfly: after the step with mask m the lanes with that bit set keep the upper half of the values
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = 32 >> s, h = 8 >> s;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const float keep = upper ? p[j + h] : p[j];
                const float send = upper ? p[j] : p[j + h];
                p[j] = keep + __shfl_xor(send, m, 64);
            }
        }
        float v = p[0];
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        const int n = nb + (lane >> 2);
        if ((lane & 3) == 0 && n < N) raw[n] = v;
    }
}

// scalar path (K < 16 or K % 4): one thread per output
__device__ __forceinline__ void lin_fwd_scalar(const mg_chain_op& o, const float* __restrict__ xs, float* __restrict__ raw) {
    const int K = o.n0, N = o.n1;
    for (int n = threadIdx.x; n < N; n += CH_THREADS) {
        const float* wr = o.p0 + (long)n * o.ld0;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += xs[k] * wr[k];
        raw[n] = acc;
    }
}

// ---- dx[i] = sum_o dy[o] W[o*ld + i]: lanes along i (float4), lane groups split o (16 rows in flight per thread), the
// groups' partial vectors summed through LDS in a fixed order ----
__device__ __forceinline__ void lin_dgrad(const mg_chain_op& o, const float* __restrict__ dys, float* __restrict__ raw,
                                          float* __restrict__ part) {
    const int OUT = o.n0, IN = o.n1;
    const float* __restrict__ W = o.p0;
    const long ld = o.ld0;
    if ((IN & 3) == 0 && IN >= 16) {
        const int L4 = IN >> 2;                    // float4 lanes per weight row (<= 128)
        const int G = CH_THREADS / L4;             // row groups (>= 8)
        const int c = threadIdx.x % L4, g = threadIdx.x / L4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
            for (int r0 = g; r0 < OUT; r0 += 16 * G) {
                float4 w4[16];
                float d[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {       // clamped rows, zero multiplier: no load behind a branch
                    const int r = r0 + u * G;
                    w4[u] = reinterpret_cast<const float4*>(W + (long)min(r, OUT - 1) * ld)[c];
                    d[u] = r < OUT ? dys[min(r, OUT - 1)] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc.x += d[u] * w4[u].x; acc.y += d[u] * w4[u].y; acc.z += d[u] * w4[u].z; acc.w += d[u] * w4[u].w;
                }
            }
            reinterpret_cast<float4*>(part)[g * L4 + c] = acc;       // part: G x IN floats <= CH_THREADS float4
        }
        __syncthreads();
        for (int i = threadIdx.x; i < IN; i += CH_THREADS) {
            float v = 0.f;
            for (int q = 0; q < G; ++q) v += part[q * IN + i];
            raw[i] = v;
        }
    } else if (IN <= 64) {
        // few outputs (the encoder's first layer: 6): the workgroup's threads split the ROWS, partial sums through LDS
        const int G = CH_THREADS / IN;
        const int i = threadIdx.x % IN, g = threadIdx.x / IN;
        if (g < G) {
            float acc = 0.f;
            for (int r = g; r < OUT; r += G) acc += dys[r] * W[(long)r * ld + i];
            part[g * IN + i] = acc;
        }
        __syncthreads();
        if (threadIdx.x < IN) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;       // four chains: 170 dependent adds would be the whole op
            int q = 0;
            for (; q + 3 < G; q += 4) {
                a0 += part[q * IN + threadIdx.x]; a1 += part[(q + 1) * IN + threadIdx.x];
                a2 += part[(q + 2) * IN + threadIdx.x]; a3 += part[(q + 3) * IN + threadIdx.x];
            }
            for (; q < G; ++q) a0 += part[q * IN + threadIdx.x];
            raw[threadIdx.x] = (a0 + a1) + (a2 + a3);
        }
    } else {
        for (int i = threadIdx.x; i < IN; i += CH_THREADS) {
            float acc = 0.f;
            for (int r = 0; r < OUT; ++r) acc += dys[r] * W[(long)r * ld + i];
            raw[i] = acc;
        }
    }
}

__global__ __launch_bounds__(CH_THREADS) void row_chain_kernel(const ChainArgs A) {
    __shared__ __attribute__((aligned(16))) float slot[CH_SLOTS][CH_MAXV];
    __shared__ __attribute__((aligned(16))) float part[CH_THREADS * 4];
    __shared__ float red[CH_WAVES];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    for (int q = 0; q < A.n_ops; ++q) {
        const mg_chain_op& o = A.op[q];
        switch (o.kind) {
            case MG_CH_LOAD: {       // slot[b][n1 + j] (+)= p0[(row % i1) * ld0 + j], j < n0
                const long r = o.i1 > 0 ? row % o.i1 : row;
                for (int j = tid; j < o.n0; j += CH_THREADS) {
                    const float v = o.p0[r * o.ld0 + j];
                    slot[o.b][o.n1 + j] = o.i0 ? slot[o.b][o.n1 + j] + v : v;
                }
                break;
            }
            case MG_CH_COPY: {       // slot[b][n1 + j] = slot[a][i1 + j], j < n0 (a != b)
                for (int j = tid; j < o.n0; j += CH_THREADS) slot[o.b][o.n1 + j] = slot[o.a][o.i1 + j];
                break;
            }
            case MG_CH_MEAN_T: {     // slot[b][c] = (1/i0) sum_t p0[(row*i0 + t)*ld0 + c], c < n0: AdaptiveAvgPool1d(1); q0 rows <- it
                const int T = o.i0, Cc = o.n0;
                const float* base = o.p0 + row * (long)T * o.ld0;
                if ((Cc & 3) == 0 && Cc >= 16 && (o.ld0 & 3) == 0 && ((((uintptr_t)o.p0) & 15) == 0)) {
                    // float4 lanes along the channels, the remaining lanes along time, 16 rows in flight per thread
                    const int L4 = Cc >> 2, RL = CH_THREADS / L4;
                    const int c = tid % L4, rl = tid / L4;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rl < RL) {
                        for (int t0 = rl; t0 < T; t0 += 16 * RL) {
                            float4 v[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u)
                                v[u] = reinterpret_cast<const float4*>(base + (long)min(t0 + u * RL, T - 1) * o.ld0)[c];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 16; ++u)
                                if (t0 + u * RL < T) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                        }
                        reinterpret_cast<float4*>(part)[rl * L4 + c] = acc;
                    }
                    __syncthreads();
                    for (int i = tid; i < Cc; i += CH_THREADS) {
                        float v = 0.f;
                        for (int g = 0; g < RL; ++g) v += part[g * Cc + i];
                        v /= (float)T;
                        slot[o.b][i] = v;
                        if (o.q0) o.q0[row * o.lq0 + i] = v;
                    }
                } else {
                    for (int c = tid; c < Cc; c += CH_THREADS) {
                        float acc = 0.f;
                        for (int t = 0; t < T; ++t) acc += base[(long)t * o.ld0 + c];
                        const float v = acc / (float)T;
                        slot[o.b][c] = v;
                        if (o.q0) o.q0[row * o.lq0 + c] = v;
                    }
                }
                break;
            }
            case MG_CH_STORE: {      // q0[row * lq0 + j] = slot[a][j]
                for (int j = tid; j < o.n0; j += CH_THREADS) o.q0[row * o.lq0 + j] = slot[o.a][j];
                break;
            }
            case MG_CH_LAYERNORM: {  // slot[b] = LN(slot[a]) * p0 + p1; q0: xhat rows, q1: y rows (optional)
                const int D = o.n0;
                if (tid < D) {
                    float mean = 0.f;
                    for (int j = 0; j < D; ++j) mean += slot[o.a][j];
                    mean /= (float)D;
                    float var = 0.f;
                    for (int j = 0; j < D; ++j) {
                        const float d = slot[o.a][j] - mean;
                        var += d * d;
                    }
                    var /= (float)D;
                    const float xh = (slot[o.a][tid] - mean) * (1.f / sqrtf(var + o.f0));
                    const float y = xh * o.p0[tid] + o.p1[tid];
                    if (o.q0) o.q0[row * o.lq0 + tid] = xh;
                    if (o.q1) o.q1[row * o.lq1 + tid] = y;
                    part[tid] = y;
                }
                __syncthreads();
                if (tid < D) slot[o.b][tid] = part[tid];
                break;
            }
            case MG_CH_LIN_FWD: {    // slot[b] = act(slot[a] W^T + p1) * p2[row]; q0: pre-activation rows, q1: output rows
                float* raw = slot[o.b];
                // the epilogue's operands (one output per thread: n1 <= 512 < threads) are requested BEFORE the weight
                // rows: their latency hides under the reduction instead of following it
                const bool mine = tid < o.n1;
                const float bias = (mine && o.p1) ? o.p1[tid] : 0.f;
                const float mask = (mine && o.p2) ? o.p2[row * o.ld2 + tid] : 1.f;
                if ((o.n0 & 3) == 0 && o.n0 >= 16) lin_fwd_vec(o, slot[o.a], raw);
                else lin_fwd_scalar(o, slot[o.a], raw);
                __syncthreads();
                if (mine) {
                    float v = raw[tid] + bias;
                    if (o.q0) o.q0[row * o.lq0 + tid] = v;
                    v = act_fwd(o.act, v) * mask;
                    if (o.q1) o.q1[row * o.lq1 + tid] = v;
                    raw[tid] = v;
                }
                break;
            }
            case MG_CH_LIN_DGRAD: {  // slot[b] = (slot[a] W) * act'(p1[row]) * p2[row]; q1: output rows
                float* raw = slot[o.b];
                const bool mine = tid < o.n1;
                const float gref = (mine && o.p1) ? o.p1[row * o.ld1 + tid] : 0.f;
                const float mask = (mine && o.p2) ? o.p2[row * o.ld2 + tid] : 1.f;
                lin_dgrad(o, slot[o.a], raw, part);
                __syncthreads();
                if (mine) {
                    float v = raw[tid] * mask;
                    if (o.p1) v *= mg_act_grad(o.act, gref);
                    if (o.q1) o.q1[row * o.lq1 + tid] = v;
                    raw[tid] = v;
                }
                break;
            }
            case MG_CH_SOFTMAX_CE: { // slot[a]: logits (n0 classes); t0: int64 targets; q0[row] = loss; slot[b] = f0 * (softmax - onehot)
                if (tid == 0) {
                    const int Cc = o.n0;
                    const float* z = slot[o.a];
                    float mx = z[0];
                    for (int j = 1; j < Cc; ++j) mx = fmaxf(mx, z[j]);
                    float se = 0.f;
                    for (int j = 0; j < Cc; ++j) se += expf(z[j] - mx);
                    const float lse = mx + logf(se);
                    const int64_t y = o.t0[row];
                    // a target outside [0, C) poisons the row instead of reading z[y] (mg_softmax_ce)
                    const bool bad = y < 0 || y >= Cc;
                    o.q0[row] = bad ? __builtin_nanf("") : lse - z[bad ? 0 : y];
                    for (int j = 0; j < Cc; ++j)
                        part[j] = bad ? __builtin_nanf("") : o.f0 * (expf(z[j] - lse) - (j == y ? 1.f : 0.f));
                }
                __syncthreads();
                if (tid < o.n0) slot[o.b][tid] = part[tid];
                break;
            }
            case MG_CH_DHEAD: {      // critic head: q0[row] = f.w[:F] + emb[row % i1].w[F:] + bias; slot[b] = ds[row] w[:F] lrelu'(f);
                                     // q1[row] = ds[row] w[F:]  (the embedding's gradient; one embedding row per sample)
                const int F = o.n0, E = o.n1;
                const float* f = slot[o.a];
                const float d = o.p3[row];
                float acc = 0.f;
                for (int j = tid; j < F; j += CH_THREADS) {
                    const float fv = f[j], wv = o.p0[j];
                    acc += fv * wv;
                    part[j] = d * wv * (fv > 0.f ? 1.f : 0.2f);
                }
                if (o.p2) {
                    const long er = o.i1 > 0 ? row % o.i1 : row;
                    for (int j = tid; j < E; j += CH_THREADS) {
                        const float wv = o.p0[F + j];
                        acc += o.p2[er * o.ld2 + j] * wv;
                        if (o.q1) o.q1[row * o.lq1 + j] = d * wv;
                    }
                }
                // block sum (fixed order)
                for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
                if ((tid & 63) == 0) red[tid >> 6] = acc;
                __syncthreads();
                if (tid == 0) {
                    float t = 0.f;
                    for (int w = 0; w < CH_WAVES; ++w) t += red[w];
                    o.q0[row] = t + o.p1[0];
                }
                for (int j = tid; j < F; j += CH_THREADS) slot[o.b][j] = part[j];
                break;
            }
            default: break;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int mg_row_chain(const mg_chain_op* ops, int n_ops, int rows, mg_stream_t stream) {
    MG_CHECK_ARG(ops && n_ops > 0 && n_ops <= MG_CHAIN_MAX_OPS && rows > 0, "mg_row_chain: 1..%d ops, rows > 0", MG_CHAIN_MAX_OPS);
    ChainArgs A{};
    A.n_ops = n_ops;
    for (int i = 0; i < n_ops; ++i) {
        const mg_chain_op& o = ops[i];
        auto vec_ok = [](int n) { return n > 0 && n <= CH_MAXV; };
        auto slot_ok = [](int s) { return s >= 0 && s < CH_SLOTS; };
        switch (o.kind) {
            case MG_CH_LOAD:
                MG_CHECK_ARG(o.p0 && slot_ok(o.b) && vec_ok(o.n0) && o.n1 >= 0 && o.n1 + o.n0 <= CH_MAXV && o.ld0 >= o.n0 && o.i1 >= 0,
                             "mg_row_chain: op %d (load)", i);
                break;
            case MG_CH_COPY:
                MG_CHECK_ARG(slot_ok(o.a) && slot_ok(o.b) && o.a != o.b && vec_ok(o.n0) && o.n1 >= 0 && o.n1 + o.n0 <= CH_MAXV &&
                             o.i1 >= 0 && o.i1 + o.n0 <= CH_MAXV, "mg_row_chain: op %d (copy)", i);
                break;
            case MG_CH_MEAN_T:
                MG_CHECK_ARG(o.p0 && slot_ok(o.b) && vec_ok(o.n0) && o.ld0 >= o.n0 && o.i0 > 0 && (!o.q0 || o.lq0 >= o.n0),
                             "mg_row_chain: op %d (mean over time)", i);
                break;
            case MG_CH_STORE:
                MG_CHECK_ARG(o.q0 && slot_ok(o.a) && vec_ok(o.n0) && o.lq0 >= o.n0, "mg_row_chain: op %d (store)", i);
                break;
            case MG_CH_LAYERNORM:
                MG_CHECK_ARG(o.p0 && o.p1 && slot_ok(o.a) && slot_ok(o.b) && o.n0 > 0 && o.n0 <= 64 && (!o.q0 || o.lq0 >= o.n0) &&
                             (!o.q1 || o.lq1 >= o.n0), "mg_row_chain: op %d (layernorm, D <= 64)", i);
                break;
            case MG_CH_LIN_FWD:
                MG_CHECK_ARG(o.p0 && slot_ok(o.a) && slot_ok(o.b) && o.a != o.b && vec_ok(o.n0) && vec_ok(o.n1) && o.ld0 >= o.n0 &&
                             (!o.q0 || o.lq0 >= o.n1) && (!o.p2 || o.ld2 >= o.n1) && (!o.q1 || o.lq1 >= o.n1),
                             "mg_row_chain: op %d (linear forward)", i);
                MG_CHECK_ARG((o.n0 & 3) || o.n0 < 16 || ((((uintptr_t)o.p0) & 15) == 0 && (o.ld0 & 3) == 0),
                             "mg_row_chain: op %d: weight rows must be 16-byte aligned", i);
                break;
            case MG_CH_LIN_DGRAD:
                MG_CHECK_ARG(o.p0 && slot_ok(o.a) && slot_ok(o.b) && o.a != o.b && vec_ok(o.n0) && vec_ok(o.n1) && o.ld0 >= o.n1 &&
                             (!o.p1 || o.ld1 >= o.n1) && (!o.p2 || o.ld2 >= o.n1) && (!o.q1 || o.lq1 >= o.n1),
                             "mg_row_chain: op %d (linear data-gradient)", i);
                MG_CHECK_ARG((o.n1 & 3) || o.n1 < 16 || ((((uintptr_t)o.p0) & 15) == 0 && (o.ld0 & 3) == 0),
                             "mg_row_chain: op %d: weight rows must be 16-byte aligned", i);
                break;
            case MG_CH_SOFTMAX_CE:
                MG_CHECK_ARG(o.t0 && o.q0 && slot_ok(o.a) && slot_ok(o.b) && o.n0 > 0 && o.n0 <= 32, "mg_row_chain: op %d (cross-entropy)", i);
                break;
            case MG_CH_DHEAD:
                MG_CHECK_ARG(o.p0 && o.p1 && o.p3 && o.q0 && slot_ok(o.a) && slot_ok(o.b) && vec_ok(o.n0) && o.n1 >= 0 &&
                             (!o.p2 || (o.n1 > 0 && o.ld2 >= o.n1)) && (!o.q1 || (o.p2 && o.lq1 >= o.n1)),
                             "mg_row_chain: op %d (critic head)", i);
                break;
            default:
                mg_set_error("mg_row_chain: op %d has unknown kind %d", i, o.kind);
                return MG_EARG;
        }
        A.op[i] = o;
    }
    hipLaunchKernelGGL(row_chain_kernel, dim3((unsigned)rows), dim3(CH_THREADS), 0, (hipStream_t)stream, A);
    MG_CHECK_LAUNCH("row_chain");
    return MG_OK;
}
