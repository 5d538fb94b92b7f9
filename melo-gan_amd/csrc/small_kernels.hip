// HBM-bound streaming kernels around the window GEMMs: per-channel reductions, BatchNorm
// (train fwd/bwd, eval), pooling, LayerNorm(6), the critic head, WGAN-GP pieces, losses,
// flat Adam.  All fp32, channels-last; rows are (b, t), columns are channels, so a wave's
// 64 lanes read 64 consecutive channels (256 B) of one row -- coalesced by construction.
#include "common.h"

namespace {

constexpr int RED_SPLITS = 256;  // max row slices for two-stage column reductions

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of one float per thread (blockDim.x multiple of 64, <= 1024)
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

// ---------------- column sums: partial[split][{sum,sumsq}][c] ----------------
// MODE 0: v = x          MODE 1 (BN backward): v1 = dy*mask(a), v2 = v1 * xhat(z)
// A block covers 64 channels as 16 float4 lanes x 16 row lanes (rows r0+ry, step 16), so even C=64 tensors
// spread over many blocks (one per row slice) with 16-byte loads; fp64 accumulation is free at HBM-bound rates.
template <int MODE>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int act, long R,
                                                             int C, long rows_per, double* __restrict__ part,
                                                             int want_sq, const float* __restrict__ gamma = nullptr,
                                                             const float* __restrict__ beta = nullptr) {
    __shared__ double sh[2][16][65];
    // blockIdx.z = row group (BatchNorm over several independent batches in one launch): R rows each, contiguous
    x += (long)blockIdx.z * R * C;
    part += (long)blockIdx.z * (RED_SPLITS + 1) * 2 * C;
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + 4 * cq;
    const long r0 = (long)blockIdx.y * rows_per;
    long r1 = r0 + rows_per;
    if (r1 > R) r1 = R;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const bool vec = ((C & 3) == 0) && (c0 + 3 < C);
    if (c0 < C) {
        float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0}, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
        const bool pre = MODE == 1 && act == MG_ACT_GELU;       // GELU' needs the BN output, not the activation
        if (MODE == 1)
            for (int e = 0; e < 4; ++e)
                if (c0 + e < C) {
                    mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
                    if (pre) { ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; }
                }
        for (long r = r0 + ry; r < r1; r += 16) {
            const long i = r * C + c0;
            float xv[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0}, zv[4] = {0, 0, 0, 0};
            if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(x + i);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                if (MODE == 1) {
                    const float4 ta = *reinterpret_cast<const float4*>(a + i);
                    const float4 tz = *reinterpret_cast<const float4*>(z + i);
                    av[0] = ta.x; av[1] = ta.y; av[2] = ta.z; av[3] = ta.w;
                    zv[0] = tz.x; zv[1] = tz.y; zv[2] = tz.z; zv[3] = tz.w;
                }
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < C) {
                        xv[e] = x[i + e];
                        if (MODE == 1) { av[e] = a[i + e]; zv[e] = z[i + e]; }
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (MODE == 0) {
                    const double v = (double)xv[e];
                    s1[e] += v;
                    s2[e] += v * v;
                } else {
                    const double xh = ((double)zv[e] - (double)mu[e]) * (double)is[e];
                    const float ref = pre ? (float)(xh * (double)ga[e] + (double)be[e]) : av[e];
                    const double dy = (double)(xv[e] * mg_act_grad(act, ref));
                    s1[e] += dy;
                    s2[e] += dy * xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[0][ry][4 * cq + e] = s1[e];
        sh[1][ry][4 * cq + e] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6, cx = threadIdx.x & 63;
        const int c = blockIdx.x * 64 + cx;
        if (c < C && (w == 0 || want_sq)) {
            double t = 0.0;
#pragma unroll
            for (int y = 0; y < 16; ++y) t += sh[w][y][cx];
            part[((long)blockIdx.y * 2 + w) * C + c] = t;
        }
    }
}

struct RedPlan { int nsplit; long rows_per; };
RedPlan red_plan(long R) {
    RedPlan p;
    long ns = mg_cdiv(R, 64);      // 64-row slices: even a 2-MB activation spreads over >= 128 workgroups
    if (ns > RED_SPLITS) ns = RED_SPLITS;
    if (ns < 1) ns = 1;
    p.rows_per = mg_cdiv(R, ns);
    p.nsplit = (int)mg_cdiv(R, p.rows_per);
    return p;
}

// Final stage of the two-stage reductions: 16 channels x 16 groups per 256-thread block; group g sums partial
// slices g, g+16, ... and the group sums are combined through LDS in a fixed order (<= 16 dependent loads per
// thread even with RED_SPLITS slices, and 4x the blocks of a 64-channel split).
constexpr int FIN_CH = 16;      // channels per block of the final kernels
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ part, int nsplit, int C, bool want2,
                                                double& s1, double& s2, int& c_out) {
    __shared__ double sh[2][16][FIN_CH];
    const int cx = threadIdx.x & (FIN_CH - 1), g = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cx;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int k = g; k < nsplit; k += 16) {
            a += part[((long)k * 2) * C + c];
            if (want2) b += part[((long)k * 2 + 1) * C + c];
        }
    sh[0][g][cx] = a;
    sh[1][g][cx] = b;
    __syncthreads();
    c_out = c;
    if (g != 0 || c >= C) return false;
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int y = 0; y < 16; ++y) {
        s1 += sh[0][y][cx];
        s2 += sh[1][y][cx];
    }
    return true;
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, int nsplit, int C,
                                                           float* sum, float* sumsq) {
    double s1, s2;
    int c;
    if (!reduce_partials(part, nsplit, C, sumsq != nullptr, s1, s2, c)) return;
    sum[c] = (float)s1;
    if (sumsq) sumsq[c] = (float)s2;
}

__global__ void bn_eval_kernel(const float* __restrict__ z, float* __restrict__ a, long n, int C,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const float v = (z[i] - rm[c]) / sqrtf(rv[c] + eps) * gamma[c] + beta[c];
    a[i] = mg_act(act, v);
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* cb, float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] + ((cb ? cb[c] : 0.f) - rm[c]) * s;
}

// ---------------- BatchNorm from the producing convolution's partial statistics: ONE launch ----------------
// part[g*np + p][3][C] (float): per-(workgroup, wave) column partials written by mg_conv16_stats -- the sum, the sum of
// squares about that wave's OWN mean, and the number of rows.  A block owns 64 channels x one row slice of one group: it
// first combines ITS channels' partials (fixed order, fp64, parallel-variance rule: n = sum n_p, mean = sum S_p / n,
// M2 = sum M2_p + sum S_p^2 / n_p - (sum S_p)^2 / n -- the between-wave term in fp64 from fp32 inputs, never E[x^2] - mean^2
// of the raw data), then applies.  The (row slice 0, group 0) block of every channel block also publishes the statistics
// of every group and moves the running statistics, group after group -- what two consecutive forward passes do.
// (Was two launches: a finishing kernel and the apply pass; a dependent launch costs ~5 us, the redundant combine < 1.)
constexpr int BNA_THREADS = 1024, BNA_PL = BNA_THREADS / 64;      // 16 partial lanes per channel: the combine is ONE round trip
__device__ __forceinline__ void combine_parts64(const float* __restrict__ part, int np, int C, int c0, double (*sh)[64],
                                                double& mean, double& var) {
    const int cx = threadIdx.x & 63, pl = threadIdx.x >> 6;      // BNA_PL partial lanes per channel
    double A = 0.0, Q = 0.0, Bt = 0.0, n = 0.0;
    if (c0 + cx < C) {
        // eight independent load triples in flight per round (a plain loop waits for every round trip in turn)
        for (int q0 = pl; q0 < np; q0 += 8 * BNA_PL) {
            float va[8], vq[8], vn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // always load (clamped row), select afterwards: a load behind a runtime condition makes hipcc branch
                // around it and wait for each one in turn
                const int q = q0 + BNA_PL * j;
                const long o = ((long)(q < np ? q : np - 1) * 3) * C + c0 + cx;
                va[j] = part[o];
                vq[j] = part[o + C];
                vn[j] = part[o + 2 * (long)C];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q0 + BNA_PL * j < np && vn[j] > 0.f) {
                    const double a = (double)va[j], c = (double)vn[j];
                    A += a; Q += (double)vq[j]; Bt += a * a / c; n += c;
                }
        }
    }
    __syncthreads();
    sh[pl][cx] = A; sh[BNA_PL + pl][cx] = Q; sh[2 * BNA_PL + pl][cx] = Bt; sh[3 * BNA_PL + pl][cx] = n;
    __syncthreads();
    A = Q = Bt = n = 0.0;
#pragma unroll
    for (int l = 0; l < BNA_PL; ++l) {       // fixed order
        A += sh[l][cx]; Q += sh[BNA_PL + l][cx]; Bt += sh[2 * BNA_PL + l][cx]; n += sh[3 * BNA_PL + l][cx];
    }
    mean = n > 0.0 ? A / n : 0.0;
    double m2 = Q + (Bt - A * mean);
    if (m2 < 0.0) m2 = 0.0;
    var = n > 0.0 ? m2 / n : 0.0;
}

// the same combine for the fp64 (sum, sum of squares) partials of colsum_partial_kernel<0> (tensors no conv16 launch produced):
// part[g][split][2][C], mean = S1 / R, var = S2 / R - mean^2 in fp64
__device__ __forceinline__ void combine_colsum64(const double* __restrict__ part, int np, int C, int c0, long R, double (*sh)[64],
                                                 double& mean, double& var) {
    const int cx = threadIdx.x & 63, pl = threadIdx.x >> 6;
    double A = 0.0, Q = 0.0;
    if (c0 + cx < C)
        for (int q0 = pl; q0 < np; q0 += 8 * BNA_PL) {
            double va[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + BNA_PL * j;
                const long o = ((long)(q < np ? q : np - 1) * 2) * C + c0 + cx;
                va[j] = part[o];
                vq[j] = part[o + C];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q0 + BNA_PL * j < np) { A += va[j]; Q += vq[j]; }
        }
    __syncthreads();
    sh[pl][cx] = A; sh[BNA_PL + pl][cx] = Q;
    __syncthreads();
    A = Q = 0.0;
#pragma unroll
    for (int l = 0; l < BNA_PL; ++l) { A += sh[l][cx]; Q += sh[BNA_PL + l][cx]; }
    mean = A / (double)R;
    var = Q / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
}

// PT = float: conv16 partials (np rows of 3 planes per group); PT = double: colsum partials ((RED_SPLITS + 1) * 2 * C per group)
template <typename PT>
__global__ __launch_bounds__(BNA_THREADS) void bn_parts_apply_kernel(const PT* __restrict__ part, int np, int groups, long R, int C,
                                                             float momentum, float eps, float* running_mean, float* running_var,
                                                             float* save_mean, float* save_invstd, const float* __restrict__ z,
                                                             float* __restrict__ a, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int act, long rows_per) {
    __shared__ double sh[4 * BNA_PL][64];
    __shared__ float s_mean[64], s_istd[64];
    const int c0 = blockIdx.x * 64, cx = threadIdx.x & 63;
    const int g = blockIdx.z;
    if (blockIdx.y == 0 && g == 0) {           // the publisher: every group in turn
        for (int gg = 0; gg < groups; ++gg) {
            double mean, var;
            if constexpr (sizeof(PT) == 4) combine_parts64(part + (long)gg * np * 3 * C, np, C, c0, sh, mean, var);
            else combine_colsum64(part + (long)gg * (RED_SPLITS + 1) * 2 * C, np, C, c0, R, sh, mean, var);
            if (threadIdx.x < 64 && c0 + cx < C) {
                const float mf = (float)mean, isf = (float)(1.0 / sqrt(var + (double)eps));
                save_mean[(long)gg * C + c0 + cx] = mf;
                save_invstd[(long)gg * C + c0 + cx] = isf;
                if (gg == 0) { s_mean[cx] = mf; s_istd[cx] = isf; }
                if (running_mean) {
                    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
                    running_mean[c0 + cx] = (float)((1.0 - momentum) * running_mean[c0 + cx] + momentum * mean);
                    running_var[c0 + cx] = (float)((1.0 - momentum) * running_var[c0 + cx] + momentum * unb);
                }
            }
        }
    } else {
        double mean, var;
        if constexpr (sizeof(PT) == 4) combine_parts64(part + (long)g * np * 3 * C, np, C, c0, sh, mean, var);
        else combine_colsum64(part + (long)g * (RED_SPLITS + 1) * 2 * C, np, C, c0, R, sh, mean, var);
        if (threadIdx.x < 64 && c0 + cx < C) {
            s_mean[cx] = (float)mean;
            s_istd[cx] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    // apply: 16 float4 lanes x 64 row lanes over this block's 64 channels and row slice
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    constexpr int RYN = BNA_THREADS / 16;
    const int c = c0 + 4 * cq;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * rows_per;
    long r1 = r0 + rows_per;
    if (r1 > R) r1 = R;
    const bool vec = ((C & 3) == 0) && (c + 3 < C);
    double mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool in = c + e < C;
        mu[e] = (double)s_mean[4 * cq + e];
        is[e] = (double)s_istd[4 * cq + e];
        ga[e] = in ? (double)gamma[c + e] : 0.0;
        be[e] = in ? (double)beta[c + e] : 0.0;
    }
    const float* zg = z + (long)g * R * C;
    float* ag = a + (long)g * R * C;
    for (long r = r0 + ry; r < r1; r += RYN) {
        const long i = r * C + c;
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(zg + i);
            const float zv[4] = {t.x, t.y, t.z, t.w};
            float o[4];
            // fp64 per-element math (free at HBM-bound rates; matches ATen's CPU accumulate type)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (float)(((double)zv[e] - mu[e]) * is[e] * ga[e] + be[e]);
            if (act == MG_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = mg_act(MG_ACT_RELU, o[e]);
            } else if (act != MG_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = mg_act(act, o[e]);
            }
            *reinterpret_cast<float4*>(ag + i) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < C) ag[i + e] = mg_act(act, (float)(((double)zg[i + e] - mu[e]) * is[e] * ga[e] + be[e]));
        }
    }
}

// BatchNorm backward, second (last) launch: every block sums ITS 64 channels' fp64 partials of colsum_partial_kernel<1>
// (fixed order), then applies to its row slice; row slice 0 also writes dgamma / dbeta.  (Was a finishing launch + an apply
// launch.)
template <typename PT>
__global__ __launch_bounds__(BNA_THREADS) void bn_bwd_parts_apply_kernel(const PT* __restrict__ part, int nsplit, int C, long R,
                                                                 const float* __restrict__ da, const float* __restrict__ a,
                                                                 const float* __restrict__ z, float* __restrict__ dz,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* dgamma, float* dbeta, int act, long rows_per) {
    __shared__ double sh[2 * BNA_PL][64];
    __shared__ double s1[64], s2[64];
    const int c0 = blockIdx.x * 64, cx = threadIdx.x & 63, pl = threadIdx.x >> 6;
    {
        double t1 = 0.0, t2 = 0.0;
        if (c0 + cx < C)
            for (int q0 = pl; q0 < nsplit; q0 += 8 * BNA_PL) {
                PT va[8], vb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = q0 + BNA_PL * j;
                    const long o = ((long)(q < nsplit ? q : nsplit - 1) * 2) * C + c0 + cx;
                    va[j] = part[o];
                    vb[j] = part[o + C];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (q0 + BNA_PL * j < nsplit) { t1 += (double)va[j]; t2 += (double)vb[j]; }
            }
        sh[pl][cx] = t1;
        sh[BNA_PL + pl][cx] = t2;
        __syncthreads();
        if (threadIdx.x < 64) {
            double u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int l = 0; l < BNA_PL; ++l) { u1 += sh[l][cx]; u2 += sh[BNA_PL + l][cx]; }
            s1[cx] = u1;
            s2[cx] = u2;
            if (blockIdx.y == 0 && c0 + cx < C) {
                dbeta[c0 + cx] = (float)u1;
                dgamma[c0 + cx] = (float)u2;
            }
        }
        __syncthreads();
    }
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    constexpr int RYN = BNA_THREADS / 16;
    const int c = c0 + 4 * cq;
    if (c >= C) return;
    const long r0 = (long)blockIdx.y * rows_per;
    long r1 = r0 + rows_per;
    if (r1 > R) r1 = R;
    const bool vec = ((C & 3) == 0) && (c + 3 < C);
    const bool pre = act == MG_ACT_GELU;
    const double invR = 1.0 / (double)R;
    double mu[4], is[4], ga[4], be[4], k1[4], k2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool in = c + e < C;
        mu[e] = in ? (double)mean[c + e] : 0.0;
        is[e] = in ? (double)invstd[c + e] : 0.0;
        ga[e] = in ? (double)gamma[c + e] : 0.0;
        be[e] = (in && pre) ? (double)beta[c + e] : 0.0;
        k1[e] = s1[4 * cq + e];
        k2[e] = s2[4 * cq + e];
    }
    for (long r = r0 + ry; r < r1; r += RYN) {
        const long i = r * C + c;
        float dv[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0}, zv[4] = {0, 0, 0, 0};
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(da + i);
            const float4 ta = *reinterpret_cast<const float4*>(a + i);
            const float4 tz = *reinterpret_cast<const float4*>(z + i);
            dv[0] = t.x; dv[1] = t.y; dv[2] = t.z; dv[3] = t.w;
            av[0] = ta.x; av[1] = ta.y; av[2] = ta.z; av[3] = ta.w;
            zv[0] = tz.x; zv[1] = tz.y; zv[2] = tz.z; zv[3] = tz.w;
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < C) { dv[e] = da[i + e]; av[e] = a[i + e]; zv[e] = z[i + e]; }
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double xh = ((double)zv[e] - mu[e]) * is[e];
            const float ref = pre ? (float)(xh * ga[e] + be[e]) : av[e];
            const double dy = (double)(dv[e] * mg_act_grad(act, ref));
            o[e] = (float)(ga[e] * is[e] * (dy - k1[e] * invR - xh * k2[e] * invR));
        }
        if (vec) *reinterpret_cast<float4*>(dz + i) = make_float4(o[0], o[1], o[2], o[3]);
        else
            for (int e = 0; e < 4; ++e)
                if (c + e < C) dz[i + e] = o[e];
    }
}

// ---------------- mean over time ----------------
// block = 64 channels of one batch row: 16 float4 lanes x 16 time lanes (scalar fallback for ragged C)
__global__ __launch_bounds__(256) void meanT_fwd_kernel(const float* __restrict__ a, float* __restrict__ h, int T, int C) {
    const int b = blockIdx.y;
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + 4 * cq;
    __shared__ float sh[16][65];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = ((C & 3) == 0) && ((((uintptr_t)a) & 15) == 0);
    if (c0 < C) {
        for (int t = ry; t < T; t += 16) {
            const float* row = a + ((long)b * T + t) * C + c0;
            if (vec) {
                const float4 v = *reinterpret_cast<const float4*>(row);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < C) s[e] += row[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[ry][4 * cq + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < C) {
            float t = 0.f;
#pragma unroll
            for (int y = 0; y < 16; ++y) t += sh[y][threadIdx.x];
            h[(long)b * C + c] = t / (float)T;
        }
    }
}

// VEC: four channels per thread, 16-byte accesses (C % 4 == 0, aligned tensors)
// Optional rider (mean_out != nullptr): ONE extra block at the end of the grid writes mean_out[0] = mean_scale * mean(mean_src)
// -- the generator's adversarial loss -mean(D(fake)) (src/gan/train_gan.py:224) needs no launch of its own.
template <bool VEC>
__global__ void meanT_bwd_kernel(const float* __restrict__ dh, float* __restrict__ dz, long n, int T, int C,
                                 const float* __restrict__ gref, int gact, const float* __restrict__ gscale,
                                 const float* __restrict__ mean_src, float* mean_out, int mean_n, float mean_scale) {
    constexpr int NV = VEC ? 4 : 1;
    if (mean_out && blockIdx.x == gridDim.x - 1) {
        __shared__ float sh[16];
        float r = 0.f;
        for (int b = threadIdx.x; b < mean_n; b += blockDim.x) r += mean_src[b];
        r = block_sum(r, sh);
        if (threadIdx.x == 0) mean_out[0] = mean_scale * r / (float)mean_n;
        return;
    }
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * NV;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long b = i / ((long)T * C);
    const float invT = 1.f / (float)T;
    float v[NV], r[NV];
    if (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(dh + b * C + c);
        v[0] = t.x * invT; v[NV > 1 ? 1 : 0] = t.y * invT; v[NV > 2 ? 2 : 0] = t.z * invT; v[NV > 3 ? 3 : 0] = t.w * invT;
        if (gref) {
            const float4 g = *reinterpret_cast<const float4*>(gref + i);
            r[0] = g.x; r[NV > 1 ? 1 : 0] = g.y; r[NV > 2 ? 2 : 0] = g.z; r[NV > 3 ? 3 : 0] = g.w;
        }
    } else {
        v[0] = dh[b * C + c] * invT;
        if (gref) r[0] = gref[i];
    }
    if (gref) {          // one uniform branch per activation kind (see mg_apply_epilogue_set)
        if (gact == MG_ACT_GELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_GELU, r[q]);
        } else if (gact == MG_ACT_LRELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_LRELU, r[q]);
        } else if (gact == MG_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_RELU, r[q]);
        } else if (gact == MG_ACT_TANH) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] *= mg_act_grad(MG_ACT_TANH, r[q]);
        }
    }
    if (gscale) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] *= gscale[c + q];
    }
    if (VEC) *reinterpret_cast<float4*>(dz + i) = make_float4(v[0], v[NV > 1 ? 1 : 0], v[NV > 2 ? 2 : 0], v[NV > 3 ? 3 : 0]);
    else dz[i] = v[0];
}

// ---------------- LayerNorm (small D) ----------------
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ xhat,
                                     int B, int D, const float* gamma, const float* beta, float eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float mean = 0.f;
    for (int j = 0; j < D; ++j) mean += x[(long)b * D + j];
    mean /= (float)D;
    float var = 0.f;
    for (int j = 0; j < D; ++j) {
        const float d = x[(long)b * D + j] - mean;
        var += d * d;
    }
    var /= (float)D;
    const float is = 1.f / sqrtf(var + eps);
    for (int j = 0; j < D; ++j) {
        const float xh = (x[(long)b * D + j] - mean) * is;
        if (xhat) xhat[(long)b * D + j] = xh;
        y[(long)b * D + j] = xh * gamma[j] + beta[j];
    }
}

// 256 threads: feature j = tid % 64, row group tid / 64 takes rows rg, rg+4, ... (four partial sums per feature, added in
// a fixed order).  One thread per feature walking all B rows was a chain of B dependent loads: 8 us for 64 rows.
__global__ void layernorm_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, float* dgamma,
                                            float* dbeta, int B, int D) {
    __shared__ float sg[4][64], sb[4][64];
    const int j = threadIdx.x & 63, rg = threadIdx.x >> 6;
    float g = 0.f, bsum = 0.f;
    if (j < D)
        for (int b = rg; b < B; b += 4) {
            const float d = dy[(long)b * D + j];
            g += d * xhat[(long)b * D + j];
            bsum += d;
        }
    sg[rg][j] = g;
    sb[rg][j] = bsum;
    __syncthreads();
    if (rg == 0 && j < D) {
        dgamma[j] = (sg[0][j] + sg[1][j]) + (sg[2][j] + sg[3][j]);
        dbeta[j] = (sb[0][j] + sb[1][j]) + (sb[2][j] + sb[3][j]);
    }
}

// ---------------- critic head ----------------
__global__ void dhead_fwd_kernel(const float* __restrict__ f, const float* __restrict__ emb,
                                 const float* __restrict__ w, const float* __restrict__ bias, float* s, int Be,
                                 int F, int E) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int j = lane; j < F; j += 64) acc += f[(long)b * F + j] * w[j];
    if (emb)
        for (int j = lane; j < E; j += 64) acc += emb[(long)(b % Be) * E + j] * w[F + j];
    acc = wave_sum(acc);
    if (lane == 0) s[b] = acc + bias[0];
}

__global__ void dhead_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ f,
                                 const float* __restrict__ w, float* dU, int B, int F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F) return;
    const int b = (int)(i / F), j = (int)(i % F);
    dU[i] = ds[b] * w[j] * (f[i] > 0.f ? 1.f : 0.2f);
}

__global__ void dhead_demb_kernel(const float* __restrict__ ds, const float* __restrict__ w, float* demb, int Be,
                                  int F, int E, int nb_emb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Be * E) return;
    const int be = (int)(i / E), j = (int)(i % E);
    float s = 0.f;
    for (int b = be; b < nb_emb; b += Be) s += ds[b];
    demb[i] = s * w[F + j];
}

// Critic head forward AND its input gradient in one launch: the upstream coefficient ds is a constant of the step
// (-1/B, +1/B, 1: Wasserstein terms and the penalty's grad_outputs), so the backward does not wait for the forward.
// Block b < B: s[b] and dU[b, :]; blocks B.. (if demb): 64 elements of the embedding gradient each.
__global__ void dhead_fwd_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ f,
                                     const float* __restrict__ emb, const float* __restrict__ w,
                                     const float* __restrict__ bias, float* s, float* dU, float* demb, int B, int Be,
                                     int F, int E, int nb_emb) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) {
        const int i = (b - B) * 64 + lane;
        if (i < Be * E) {
            const int be = i / E, j = i - be * E;
            float t = 0.f;
            for (int r = be; r < nb_emb; r += Be) t += ds[r];
            demb[i] = t * w[F + j];
        }
        return;
    }
    const float d = ds[b];
    float acc = 0.f;
    for (int j = lane; j < F; j += 64) {
        const float fv = f[(long)b * F + j], wv = w[j];
        acc += fv * wv;
        dU[(long)b * F + j] = d * wv * (fv > 0.f ? 1.f : 0.2f);
    }
    if (emb)
        for (int j = lane; j < E; j += 64) acc += emb[(long)(b % Be) * E + j] * w[F + j];
    acc = wave_sum(acc);
    if (lane == 0) s[b] = acc + bias[0];
}

struct DLoss { const float* s; const float* norms; float lambda_gp; float* out; float* gp_out; int nb; };
__device__ __forceinline__ void wgan_d_loss_block(const DLoss& L) {
    __shared__ float sh[16];
    float r = 0.f, f = 0.f, q = 0.f;
    for (int b = threadIdx.x; b < L.nb; b += blockDim.x) {
        r += L.s[b];
        f += L.s[L.nb + b];
        const float d = L.norms[b] - 1.f;
        q += d * d;
    }
    r = block_sum(r, sh);
    f = block_sum(f, sh);
    q = block_sum(q, sh);
    if (threadIdx.x == 0) {
        const float mr = r / (float)L.nb, mf = f / (float)L.nb, pen = q / (float)L.nb;
        L.gp_out[0] = pen;
        L.out[0] = mf - mr + L.lambda_gp * pen;
        L.out[1] = mr;
        L.out[2] = mf;
    }
}

// one wave per output element j: lanes stride over the batch rows, shuffle-reduce.  Optional rider (loss.out != nullptr):
// one extra block at the end of the grid computes the critic's loss scalars (mg_wgan_d_loss_gp) -- logging only, nothing
// in the step depends on them, so they need no launch of their own.
__global__ void dhead_wgrad_kernel(const float* __restrict__ ds, const float* __restrict__ f,
                                   const float* __restrict__ emb, const float* __restrict__ gf, float* dw,
                                   float* dbias, int nb, int ng, int Be, int F, int E, const DLoss loss) {
    if (loss.out && blockIdx.x == gridDim.x - 1) {
        wgan_d_loss_block(loss);
        return;
    }
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (j < F) {
        for (int b = lane; b < nb; b += 64) s += ds[b] * f[(long)b * F + j];
        if (gf)
            for (int b = lane; b < ng; b += 64) s += gf[(long)b * F + j];
    } else if (j < F + E) {
        if (emb)
            for (int b = lane; b < nb; b += 64) s += ds[b] * emb[(long)(b % Be) * E + (j - F)];
    } else if (j == F + E) {
        for (int b = lane; b < nb; b += 64) s += ds[b];
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (j < F + E) dw[j] = s;
        else if (j == F + E) dbias[0] = s;
    }
}

// ---------------- WGAN-GP ----------------
__global__ void gp_interp_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                 const float* __restrict__ alpha, float* __restrict__ xhat, long n, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float a = alpha[i / n];
    xhat[i] = fmaf(a, real[i], (1.f - a) * fake[i]);      // the same expression as conv16's mix rider
}

// One workgroup per sample.  The sample's slice stays in registers between the norm and the scaling pass when it
// fits (n <= 1024 threads x 8 float4: the 128x256 roll is exactly that), so g is read once, with 16-byte loads.
__global__ __launch_bounds__(1024) void gp_norm_kernel(const float* __restrict__ g, float* __restrict__ gbar,
                                                       float* norms, float coef, int B, long n) {
    __shared__ float sh[16];
    const int b = blockIdx.x;
    const float* gb = g + (long)b * n;
    const bool vec = ((n & 3) == 0) && (n <= 1024L * 32) && ((((uintptr_t)g) & 15) == 0) &&
                     (gbar == nullptr || (((uintptr_t)gbar) & 15) == 0);
    float4 v[8];
    float s = 0.f;
    if (vec) {
        const long n4 = n >> 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long i = threadIdx.x + 1024L * j;
            v[j] = i < n4 ? reinterpret_cast<const float4*>(gb)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
    } else {
        for (long i = threadIdx.x; i < n; i += blockDim.x) s += gb[i] * gb[i];
    }
    s = block_sum(s, sh);
    const float nrm = sqrtf(s);
    if (threadIdx.x == 0) norms[b] = nrm;
    if (gbar) {
        const float fac = nrm > 0.f ? coef * (2.f / (float)B) * (nrm - 1.f) / nrm : 0.f;
        float* ob = gbar + (long)b * n;
        if (vec) {
            const long n4 = n >> 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long i = threadIdx.x + 1024L * j;
                if (i < n4) reinterpret_cast<float4*>(ob)[i] = make_float4(fac * v[j].x, fac * v[j].y, fac * v[j].z, fac * v[j].w);
            }
        } else {
            for (long i = threadIdx.x; i < n; i += blockDim.x) ob[i] = fac * gb[i];
        }
    }
}

__global__ void gp_final_kernel(const float* norms, float* gp, int B) {
    __shared__ float sh[16];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float d = norms[b] - 1.f;
        s += d * d;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) gp[0] = s / (float)B;
}

// ---------------- losses ----------------
__global__ void wgan_d_loss_kernel(const float* s, const float* gp, const float* norms, float lambda_gp, float* out,
                                   float* gp_out, int nb) {
    __shared__ float sh[16];
    float r = 0.f, f = 0.f, q = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        r += s[b];
        f += s[nb + b];
        if (norms) {
            const float d = norms[b] - 1.f;
            q += d * d;
        }
    }
    r = block_sum(r, sh);
    f = block_sum(f, sh);
    if (norms) q = block_sum(q, sh);
    if (threadIdx.x == 0) {
        const float mr = r / (float)nb, mf = f / (float)nb;
        const float pen = norms ? q / (float)nb : gp[0];
        if (norms) gp_out[0] = pen;
        out[0] = mf - mr + lambda_gp * pen;
        out[1] = mr;
        out[2] = mf;
    }
}

__global__ void neg_mean_kernel(const float* s, float* out, int B, float scale) {
    __shared__ float sh[16];
    float r = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) r += s[b];
    r = block_sum(r, sh);
    if (threadIdx.x == 0) out[0] = scale * r / (float)B;
}

__global__ void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* loss,
                                  float* dlogits, float coef, int B, int C) {
    __shared__ float sh[16];
    float l = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* z = logits + (long)b * C;
        float mx = z[0];
        for (int j = 1; j < C; ++j) mx = fmaxf(mx, z[j]);
        float se = 0.f;
        for (int j = 0; j < C; ++j) se += expf(z[j] - mx);
        const float lse = mx + logf(se);
        const int64_t y = target[b];
        // a target outside [0, C) is an error in the reference (CrossEntropyLoss raises); the host validates labels when
        // a dataset is built (gan/utils.py::check_labels) -- here such a row poisons the loss instead of reading z[y]
        const bool bad = y < 0 || y >= C;
        l += bad ? __builtin_nanf("") : lse - z[bad ? 0 : y];
        if (dlogits)
            for (int j = 0; j < C; ++j)
                dlogits[(long)b * C + j] = bad ? __builtin_nanf("") : coef * (expf(z[j] - lse) - (j == y ? 1.f : 0.f)) / (float)B;
    }
    l = block_sum(l, sh);
    if (threadIdx.x == 0) loss[0] = l / (float)B;
}

// ---------------- elementwise ----------------
__global__ void fill_kernel(float* x, float v, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float a, float b, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}
__global__ void copy_cols_kernel(const float* __restrict__ src, int sld, int soff, float* __restrict__ dst, int dld,
                                 int doff, int rows, int ncols, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * ncols) return;
    const int r = (int)(i / ncols), j = (int)(i % ncols);
    const float v = src[(long)r * sld + soff + j];
    float* d = dst + (long)r * dld + doff + j;
    *d = accumulate ? *d + v : v;
}
struct StageJobs { mg_stage_job j[MG_MAX_STAGE_JOBS]; };
// blockIdx.z = job, blockIdx.y = destination row, blockIdx.x strides over the row
__global__ void stage_rows_kernel(const StageJobs J, int n_rows) {
    const mg_stage_job job = J.j[blockIdx.z];
    const int r = blockIdx.y;
    if (job.rows > 0 && r >= job.rows) return;      // a job may cover fewer rows than the launch
    long sr = r;
    if (job.idx) {
        sr = job.idx[r];
        sr = sr < 0 ? 0 : (sr >= job.src_rows ? job.src_rows - 1 : sr);
    }
    const char* s = static_cast<const char*>(job.src) + sr * job.row_bytes;
    char* d = static_cast<char*>(job.dst) + (long)r * (job.dst_pitch ? job.dst_pitch : job.row_bytes);
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)job.row_bytes) & 15) == 0) {
        const long n = job.row_bytes >> 4;
        for (long i = first; i < n; i += stride) reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
    } else {
        const long n = job.row_bytes >> 2;
        for (long i = first; i < n; i += stride) reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
    }
}
// The same with the source rows picked by a DEVICE-side cursor: position p = ((counter - base) * n_rows + r) mod order_len of
// the epoch's order (order == nullptr: the identity), so a captured training step stages its own batch on every replay --
// no host-side launch between steps.  The counter is advanced by another kernel of the step (the critic's Adam launch
// advances the Philox step counter, which doubles as the batch counter).
__global__ void stage_rows_cursor_kernel(const StageJobs J, int n_rows, const int64_t* __restrict__ order, long order_len,
                                         const unsigned long long* __restrict__ counter, const unsigned long long* __restrict__ base) {
    const mg_stage_job job = J.j[blockIdx.z];
    const int r = blockIdx.y;
    const unsigned long long k = counter[0] - base[0];
    const long pos = (long)((k * (unsigned long long)n_rows + (unsigned long long)r) % (unsigned long long)order_len);
    long sr = order ? order[pos] : pos;
    sr = sr < 0 ? 0 : (sr >= job.src_rows ? job.src_rows - 1 : sr);
    const char* s = static_cast<const char*>(job.src) + sr * job.row_bytes;
    char* d = static_cast<char*>(job.dst) + (long)r * (job.dst_pitch ? job.dst_pitch : job.row_bytes);
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)job.row_bytes) & 15) == 0) {
        const long n = job.row_bytes >> 4;
        for (long i = first; i < n; i += stride) reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
    } else {
        const long n = job.row_bytes >> 2;
        for (long i = first; i < n; i += stride) reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
    }
}
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int L,
                                 const float* __restrict__ gref, int gact) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, l = l0 + tx;
        tile[r][tx] = (c < C && l < L) ? in[((long)b * C + c) * L + l] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, c = c0 + tx;
        if (l < L && c < C) {
            const long o = ((long)b * L + l) * C + c;
            float v = tile[tx][r];
            if (gref) v *= mg_act_grad(gact, gref[o]);
            out[o] = v;
        }
    }
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gref, int gact,
                               const float* __restrict__ emul, float* __restrict__ dx, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = dy[i];
    if (gref) v *= mg_act_grad(gact, gref[i]);
    if (emul) v *= emul[i];
    dx[i] = v;
}

// ---------------- per-step random inputs in one launch ----------------
// Philox4x32-10 counter-based generator.  key = (seed lo, seed hi), counter = (element block, stream id,
// step lo, step hi); the step counter lives in device memory and is advanced by the kernel itself so a
// captured hipGraph draws fresh numbers on every replay.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

struct RngJob { float* dst; long n; int kind; float p0; };   // kind 0: N(0,1); 1: U(0,1); 2: keep-mask*(1/(1-p0))
struct RngJobs { RngJob j[4]; int njobs; };

// tick_state (optional): the Adam state {step, beta1^step, beta2^step} of the optimiser whose update will consume
// these draws is advanced HERE (this kernel does not read it, adam_apply_kernel -- a later launch -- does), and the
// matching adam_apply advances step_ctr (which only this kernel reads): each of the two launches ticks the counter
// the OTHER one reads, so neither needs a launch of its own.
// StageRide (optional): the batch's staging (stage_rows_cursor_kernel's work) as extra blockIdx.y planes of the SAME launch --
// both only read the step counter, and a fused step needs both before anything else: one launch and one dependent-launch
// gap fewer at the head of the step.
struct StageRide { StageJobs J; int n_jobs, n_rows; const int64_t* order; long order_len; const unsigned long long* base; };

__global__ void rng_fill_kernel(const RngJobs jobs, unsigned long long seed, unsigned long long* step_ctr,
                                double* tick_state, double* tick_state2, double beta1, double beta2, const StageRide sr) {
    const unsigned long long step = step_ctr[0];
    const int jid = blockIdx.y;
    if (jid >= jobs.njobs) {
        // a staging plane: the plane's gridDim.x blocks are dealt (row, piece of the row); a thread keeps 8 16-byte loads in
        // flight (one block walking a 128-KB row 4 KB at a time is 32 dependent round trips: 20 us)
        const mg_stage_job job = sr.J.j[jid - jobs.njobs];
        const unsigned long long k = step - sr.base[0];
        const int per_row = max(1, (int)gridDim.x / sr.n_rows);
        const int piece = blockIdx.x / sr.n_rows;
        if (piece >= per_row) return;
        for (int r = blockIdx.x % sr.n_rows; r < sr.n_rows; r += (int)gridDim.x) {
            const long pos = (long)((k * (unsigned long long)sr.n_rows + (unsigned long long)r) % (unsigned long long)sr.order_len);
            long srow = sr.order ? sr.order[pos] : pos;
            srow = srow < 0 ? 0 : (srow >= job.src_rows ? job.src_rows - 1 : srow);
            const char* s = static_cast<const char*>(job.src) + srow * job.row_bytes;
            char* d = static_cast<char*>(job.dst) + (long)r * (job.dst_pitch ? job.dst_pitch : job.row_bytes);
            if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)job.row_bytes) & 15) == 0) {
                const long n = job.row_bytes >> 4;
                const long lo = n * piece / per_row, hi = n * (piece + 1) / per_row;
                for (long i0 = lo + threadIdx.x; i0 < hi; i0 += 8L * blockDim.x) {
                    float4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const long i = i0 + (long)u * blockDim.x;
                        t[u] = reinterpret_cast<const float4*>(s)[i < hi ? i : lo];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const long i = i0 + (long)u * blockDim.x;
                        if (i < hi) reinterpret_cast<float4*>(d)[i] = t[u];
                    }
                }
            } else {
                const long n = job.row_bytes >> 2;
                const long lo = n * piece / per_row, hi = n * (piece + 1) / per_row;
                for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
            }
        }
        return;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        double* const st[2] = {tick_state, tick_state2};      // one draw may serve two updates (the fused critic + generator step)
        for (int k = 0; k < 2; ++k)
            if (st[k]) {
                if (st[k][0] == 0.0) { st[k][1] = 1.0; st[k][2] = 1.0; }
                st[k][0] += 1.0;
                st[k][1] *= beta1;
                st[k][2] *= beta2;
            }
    }
    if (jid < jobs.njobs) {
        const RngJob jb = jobs.j[jid];
        for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; 4 * q < jb.n; q += (long)gridDim.x * blockDim.x) {
            unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32) ^ ((unsigned)jid << 28), (unsigned)step, (unsigned)(step >> 32)};
            philox4(c, (unsigned)seed, (unsigned)(seed >> 32));
            float v[4];
            if (jb.kind == 0) {
                const float r0 = sqrtf(-2.f * logf(u01(c[0]))), r1 = sqrtf(-2.f * logf(u01(c[2])));
                const float t0 = 6.28318530717958647692f * u01(c[1]), t1 = 6.28318530717958647692f * u01(c[3]);
                v[0] = r0 * cosf(t0); v[1] = r0 * sinf(t0); v[2] = r1 * cosf(t1); v[3] = r1 * sinf(t1);
            } else if (jb.kind == 1) {
                for (int e = 0; e < 4; ++e) v[e] = u01(c[e]);
            } else {
                const float sc = 1.f / (1.f - jb.p0);
                for (int e = 0; e < 4; ++e) v[e] = u01(c[e]) >= jb.p0 ? sc : 0.f;
            }
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < jb.n) jb.dst[4 * q + e] = v[e];
        }
    }
}
__global__ void rng_advance_kernel(unsigned long long* step_ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step_ctr[0] += 1;
}

// ---------------- Adam ----------------
__global__ void adam_advance_kernel(double* state, double beta1, double beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (state[0] == 0.0) { state[1] = 1.0; state[2] = 1.0; }
        state[0] += 1.0;
        state[1] *= beta1;
        state[2] *= beta2;
    }
}
// WQ copies kept current by the update (include/melo_gan_hip.h, mg_conv16): for a weight tensor W(n,c,k) =
// w[n*sn + c*sc + k] living at [start, start + N*Cc*K) of the flat parameter buffer, dst[((c/4)*K + k)*N + n][c%4] = W(n,c,k).
struct WqTable { mg_wq_entry e[MG_MAX_WQ_ENTRIES]; int n; long lo, hi; };

__global__ void adam_apply_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                  float wd, const double* __restrict__ state, float grad_scale,
                                  const float* __restrict__ gs_dev, unsigned long long* bump, const WqTable wq) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (bump && i == 0) bump[0] += 1;         // the Philox step counter of the draws this update consumed
    if (i >= n) return;
    const double bc1 = 1.0 - state[1];
    const double bc2 = 1.0 - state[2];
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    float gs = grad_scale;
    if (gs_dev) gs *= gs_dev[0];
    const float gi = g[i] * gs;
    float pi = p[i];
    if (wd != 0.f) pi *= (1.f - lr * wd);  // decoupled (AdamW)
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    if (i >= wq.lo && i < wq.hi) {
        for (int t = 0; t < wq.n; ++t) {
            const mg_wq_entry& e = wq.e[t];
            const long L = i - e.start;
            if (L < 0 || L >= (long)e.N * e.Cc * e.K) continue;
            const int k = (int)(L % e.K);
            const long r = L / e.K;
            int nn, c;
            if (e.w_sc == e.K) { nn = (int)(r / e.Cc); c = (int)(r % e.Cc); }       // w[n][c][k]
            else { c = (int)(r / e.N); nn = (int)(r % e.N); }                        // w[c][n][k]
            e.dst[((long)((c >> 2) * e.K + k) * e.N + nn) * 4 + (c & 3)] = pi;
        }
    }
}

__global__ __launch_bounds__(1024) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* part) {
    __shared__ float sh[16];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        s += g[i] * g[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void norm_clip_final_kernel(const float* part, int nparts, float max_norm, float* out) {
    __shared__ double shd[256];
    double t = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) t += (double)part[i];      // fixed assignment, fixed tree: reproducible
    shd[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double s = shd[0];
        const float nrm = (float)sqrt(s);
        out[0] = nrm;
        const float c = max_norm / (nrm + 1e-6f);
        out[1] = c < 1.f ? c : 1.f;
    }
}

// ---------------- spectral normalisation (torch.nn.utils.spectral_norm, dim 0, one power iteration) ----------------
// One 1024-thread workgroup per layer.  weight_mat = w_orig viewed (rows = out, cols = in * k).  power_iter (training forward):
//   v = normalize(W^T u), u = normalize(W v)  (x / max(|x|_2, eps), in place in the module's buffers),
// then always  sigma = u . (W v),  w_eff = w_orig / sigma  (reference: src/emotion_discriminator/ed_model.py:29-32,79-82 wrap
// Conv1d / Linear in torch.nn.utils.spectral_norm; the algorithm restated is torch's SpectralNorm.compute_weight).
struct SnJobs { mg_sn_job j[MG_MAX_SN_JOBS]; int n; };

__global__ __launch_bounds__(1024) void spectral_norm_fwd_kernel(const SnJobs J, int power_iter, float eps) {
    __shared__ float sh[16];
    const mg_sn_job jb = J.j[blockIdx.x];
    const int R = jb.rows, Cc = jb.cols;
    const float* __restrict__ W = jb.w_orig;
    float* u = jb.u;
    float* v = jb.v;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (power_iter) {
        // v = W^T u: a thread per column (coalesced along the columns), rows in sequence
        for (int j = tid; j < Cc; j += 1024) {
            float acc = 0.f;
            for (int i = 0; i < R; ++i) acc += W[(long)i * Cc + j] * u[i];
            v[j] = acc;
        }
        __threadfence_block();
        __syncthreads();
        float q = 0.f;
        for (int j = tid; j < Cc; j += 1024) q += v[j] * v[j];
        q = block_sum(q, sh);
        const float sv = 1.f / fmaxf(sqrtf(q), eps);
        for (int j = tid; j < Cc; j += 1024) v[j] *= sv;
        __threadfence_block();
        __syncthreads();
        // u = W v: a wave per row
        for (int i = wave; i < R; i += 16) {
            float acc = 0.f;
            for (int j = lane; j < Cc; j += 64) acc += W[(long)i * Cc + j] * v[j];
            acc = wave_sum(acc);
            if (lane == 0) u[i] = acc;
        }
        __threadfence_block();
        __syncthreads();
        float r = 0.f;
        for (int i = tid; i < R; i += 1024) r += u[i] * u[i];
        r = block_sum(r, sh);
        const float su = 1.f / fmaxf(sqrtf(r), eps);
        for (int i = tid; i < R; i += 1024) u[i] *= su;
        __threadfence_block();
        __syncthreads();
    }
    float part = 0.f;
    for (int i = wave; i < R; i += 16) {
        float acc = 0.f;
        for (int j = lane; j < Cc; j += 64) acc += W[(long)i * Cc + j] * v[j];
        acc = wave_sum(acc);
        if (lane == 0) part += u[i] * acc;
    }
    const float sigma = block_sum(part, sh);
    if (tid == 0) jb.sigma[0] = sigma;
    const long n = (long)R * Cc;
    for (long e = tid; e < n; e += 1024) jb.w_eff[e] = W[e] / sigma;
}

// gradient through w_eff = w_orig / sigma, sigma = u . (w_orig v) with u, v constants:
//   d w_orig = (d w_eff - <d w_eff, w_eff> u v^T) / sigma        (in place in dw)
__global__ __launch_bounds__(1024) void spectral_norm_bwd_kernel(const SnJobs J) {
    __shared__ float sh[16];
    const mg_sn_job jb = J.j[blockIdx.x];
    const int Cc = jb.cols;
    const long n = (long)jb.rows * Cc;
    float d = 0.f;
    for (long e = threadIdx.x; e < n; e += 1024) d += jb.dw[e] * jb.w_eff[e];
    d = block_sum(d, sh);
    const float inv = 1.f / jb.sigma[0];
    for (long e = threadIdx.x; e < n; e += 1024) {
        const int i = (int)(e / Cc), j = (int)(e - (long)i * Cc);
        jb.dw[e] = (jb.dw[e] - d * jb.u[i] * jb.v[j]) * inv;
    }
}

// ---------------- VAE ----------------
__global__ void reparam_kernel(const float* mu, const float* lv, const float* eps, float* z, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = mu[i] + eps[i] * expf(0.5f * lv[i]);
}
// dmu = dz + dmu_kld ; dlv = dz*eps*0.5*exp(0.5*lv) + dlv_kld      (backward of z = mu + eps*exp(0.5*lv))
__global__ void reparam_bwd_kernel(const float* dz, const float* lv, const float* eps, const float* dmu_kld,
                                   const float* dlv_kld, float* dmu, float* dlv, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dmu[i] = dz[i] + (dmu_kld ? dmu_kld[i] : 0.f);
    dlv[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]) + (dlv_kld ? dlv_kld[i] : 0.f);
}
// Two launches: every block squares its slice of (recon - x) with 16-byte accesses (and writes drecon), block 0 also does
// the latent terms; per-block partial sums go to `part` and the finish kernel adds them in a fixed order.  (One block
// over the whole reconstruction took 101 us at B=256, T=256: 11 % of the VAE step.)
constexpr int VAE_LOSS_BLOCKS = 256;
__global__ __launch_bounds__(256) void vae_loss_partial_kernel(const float* __restrict__ recon, const float* __restrict__ x,
                                                               long n_x, const float* __restrict__ mu,
                                                               const float* __restrict__ lv, long n_z, float beta,
                                                               double* __restrict__ part, float* drecon, float* dmu,
                                                               float* dlv, int vec) {
    __shared__ float sh[16];
    const float inv = 2.f / (float)n_x;
    float s = 0.f;
    if (vec) {
        const long n4 = n_x >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const float4 r = reinterpret_cast<const float4*>(recon)[i], t = reinterpret_cast<const float4*>(x)[i];
            const float4 d = make_float4(r.x - t.x, r.y - t.y, r.z - t.z, r.w - t.w);
            s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            if (drecon) reinterpret_cast<float4*>(drecon)[i] = make_float4(inv * d.x, inv * d.y, inv * d.z, inv * d.w);
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_x; i += (long)gridDim.x * blockDim.x) {
            const float d = recon[i] - x[i];
            s += d * d;
            if (drecon) drecon[i] = inv * d;
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = (double)s;
    if (blockIdx.x == 0) {
        float k = 0.f;
        for (long i = threadIdx.x; i < n_z; i += blockDim.x) {
            const float e = expf(lv[i]);
            k += 1.f + lv[i] - mu[i] * mu[i] - e;
            if (dmu) dmu[i] = beta * mu[i] / (float)n_z;
            if (dlv) dlv[i] = beta * (-0.5f) * (1.f - e) / (float)n_z;
        }
        k = block_sum(k, sh);
        if (threadIdx.x == 0) part[gridDim.x] = (double)k;
    }
}
__global__ __launch_bounds__(64) void vae_loss_final_kernel(const double* __restrict__ part, int nblocks, long n_x, long n_z,
                                                            float beta, float* out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblocks; ++i) s += part[i];
        const float mse = (float)(s / (double)n_x), kld = (float)(-0.5 * part[nblocks] / (double)n_z);
        out[0] = mse + beta * kld;
        out[1] = mse;
        out[2] = kld;
    }
}

inline unsigned nblk(long n, int bs = 256) { return (unsigned)mg_cdiv(n, bs); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" {

size_t mg_colsum_workspace_bytes(int C) { return ((size_t)RED_SPLITS + 1) * 2 * (size_t)C * sizeof(double); }
size_t mg_bn_workspace_bytes(int C) { return mg_colsum_workspace_bytes(C); }

int mg_colsum(const float* x, long R, int C, float* sum, float* sumsq, void* work, size_t work_bytes,
              mg_stream_t stream) {
    MG_CHECK_ARG(x && sum && R > 0 && C > 0, "mg_colsum: bad args");
    if (!work || work_bytes < mg_colsum_workspace_bytes(C)) { mg_set_error("mg_colsum: workspace too small"); return MG_EWORK; }
    const RedPlan pl = red_plan(R);
    dim3 grid((unsigned)mg_cdiv(C, 64), (unsigned)pl.nsplit);
    hipLaunchKernelGGL(colsum_partial_kernel<0>, grid, dim3(256), 0, ST, x, nullptr, nullptr, nullptr, nullptr, 0, R, C,
                       pl.rows_per, (double*)work, sumsq ? 1 : 0);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(nblk(C, FIN_CH)), dim3(256), 0, ST, (const double*)work, pl.nsplit, C, sum, sumsq);
    MG_CHECK_LAUNCH("colsum");
    return MG_OK;
}

size_t mg_bn_groups_workspace_bytes(int C, int groups) { return (size_t)(groups < 1 ? 1 : groups) * mg_colsum_workspace_bytes(C); }

int mg_bn_train_fwd_groups(const float* z, float* a, long R, int C, int groups, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                           float* save_invstd, int act, void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(z && a && gamma && beta && save_mean && save_invstd && R > 0 && C > 0 && groups >= 1 && groups <= 64,
                 "mg_bn_train_fwd: bad args");
    MG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "mg_bn_train_fwd: running stats must come in pairs");
    if (!work || work_bytes < mg_bn_groups_workspace_bytes(C, groups)) { mg_set_error("mg_bn_train_fwd: workspace too small"); return MG_EWORK; }
    const RedPlan pl = red_plan(R);
    dim3 grid((unsigned)mg_cdiv(C, 64), (unsigned)pl.nsplit, (unsigned)groups);
    hipLaunchKernelGGL(colsum_partial_kernel<0>, grid, dim3(256), 0, ST, z, nullptr, nullptr, nullptr, nullptr, 0, R, C,
                       pl.rows_per, (double*)work, 1);
    // row slices: ~256 blocks of 1024 threads in all, at least 64 rows (one per row lane) each
    const long cb = mg_cdiv(C, 64);
    long slices = 256 / (cb * groups);
    if (slices < 1) slices = 1;
    if (slices > mg_cdiv(R, 64)) slices = mg_cdiv(R, 64);
    const long rows_per = mg_cdiv(R, slices);
    slices = mg_cdiv(R, rows_per);
    hipLaunchKernelGGL(bn_parts_apply_kernel<double>, dim3((unsigned)cb, (unsigned)slices, (unsigned)groups), dim3(BNA_THREADS), 0, ST,
                       (const double*)work, pl.nsplit, groups, R, C, momentum, eps, running_mean, running_var, save_mean,
                       save_invstd, z, a, gamma, beta, act, rows_per);
    MG_CHECK_LAUNCH("bn_train_fwd");
    return MG_OK;
}

int mg_bn_train_fwd(const float* z, float* a, long R, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                    float* save_invstd, int act, void* work, size_t work_bytes, mg_stream_t stream) {
    return mg_bn_train_fwd_groups(z, a, R, C, 1, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                                  save_invstd, act, work, work_bytes, stream);
}

int mg_bn_train_fwd_parts(const float* part, int part_rows_per_group, int groups, const float* z, float* a, long R, int C,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, float* save_mean, float* save_invstd, int act, mg_stream_t stream) {
    MG_CHECK_ARG(part && z && a && gamma && beta && save_mean && save_invstd && R > 0 && C > 0 && groups >= 1 &&
                 part_rows_per_group > 0, "mg_bn_train_fwd_parts: bad args");
    MG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "mg_bn_train_fwd_parts: running stats must come in pairs");
    // row slices: ~512 blocks in all (2 per CU), at least 64 rows each
    // row slices: ~256 blocks of 1024 threads in all, at least 64 rows (one per row lane) each
    const long cb = mg_cdiv(C, 64);
    long slices = 256 / (cb * groups);
    if (slices < 1) slices = 1;
    if (slices > mg_cdiv(R, 64)) slices = mg_cdiv(R, 64);
    const long rows_per = mg_cdiv(R, slices);
    slices = mg_cdiv(R, rows_per);
    hipLaunchKernelGGL(bn_parts_apply_kernel<float>, dim3((unsigned)cb, (unsigned)slices, (unsigned)groups), dim3(BNA_THREADS), 0, ST, part,
                       part_rows_per_group, groups, R, C, momentum, eps, running_mean, running_var, save_mean, save_invstd, z, a,
                       gamma, beta, act, rows_per);
    MG_CHECK_LAUNCH("bn_train_fwd_parts");
    return MG_OK;
}

int mg_bn_train_bwd(const float* da, const float* a, const float* z, float* dz, long R, int C, const float* gamma,
                    const float* beta, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                    int act, void* work, size_t work_bytes, mg_stream_t stream) {
    MG_CHECK_ARG(da && a && z && dz && gamma && save_mean && save_invstd && dgamma && dbeta, "mg_bn_train_bwd: bad args");
    MG_CHECK_ARG(act != MG_ACT_GELU || beta, "mg_bn_train_bwd: GELU needs beta (its derivative is taken at the BN output)");
    if (!work || work_bytes < mg_bn_workspace_bytes(C)) { mg_set_error("mg_bn_train_bwd: workspace too small"); return MG_EWORK; }
    const RedPlan pl = red_plan(R);
    double* part = (double*)work;
    dim3 grid((unsigned)mg_cdiv(C, 64), (unsigned)pl.nsplit);
    hipLaunchKernelGGL(colsum_partial_kernel<1>, grid, dim3(256), 0, ST, da, a, z, save_mean, save_invstd, act, R, C,
                       pl.rows_per, part, 1, gamma, beta);
    const long cb = mg_cdiv(C, 64);
    long slices = 256 / cb;
    if (slices < 1) slices = 1;
    if (slices > mg_cdiv(R, 64)) slices = mg_cdiv(R, 64);
    const long rows_per = mg_cdiv(R, slices);
    slices = mg_cdiv(R, rows_per);
    hipLaunchKernelGGL(bn_bwd_parts_apply_kernel<double>, dim3((unsigned)cb, (unsigned)slices), dim3(BNA_THREADS), 0, ST, (const double*)part,
                       pl.nsplit, C, R, da, a, z, dz, gamma, beta, save_mean, save_invstd, dgamma, dbeta, act, rows_per);
    MG_CHECK_LAUNCH("bn_train_bwd");
    return MG_OK;
}

int mg_bn_train_bwd_parts(const double* part, int part_rows, const float* da, const float* a, const float* z, float* dz, long R,
                          int C, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                          float* dgamma, float* dbeta, int act, mg_stream_t stream) {
    MG_CHECK_ARG(part && part_rows > 0 && da && a && z && dz && gamma && save_mean && save_invstd && dgamma && dbeta && R > 0 && C > 0,
                 "mg_bn_train_bwd_parts: bad args");
    MG_CHECK_ARG(act == MG_ACT_RELU || act == MG_ACT_LRELU, "mg_bn_train_bwd_parts: ReLU / LeakyReLU layers only");
    const long cb = mg_cdiv(C, 64);
    long slices = 256 / cb;
    if (slices < 1) slices = 1;
    if (slices > mg_cdiv(R, 64)) slices = mg_cdiv(R, 64);
    const long rows_per = mg_cdiv(R, slices);
    slices = mg_cdiv(R, rows_per);
    hipLaunchKernelGGL(bn_bwd_parts_apply_kernel<double>, dim3((unsigned)cb, (unsigned)slices), dim3(BNA_THREADS), 0, ST, part, part_rows,
                       C, R, da, a, z, dz, gamma, beta, save_mean, save_invstd, dgamma, dbeta, act, rows_per);
    MG_CHECK_LAUNCH("bn_train_bwd_parts");
    return MG_OK;
}

int mg_bn_eval_fwd(const float* z, float* a, long R, int C, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, float eps, int act, mg_stream_t stream) {
    MG_CHECK_ARG(z && a && gamma && beta && running_mean && running_var, "mg_bn_eval_fwd: bad args");
    hipLaunchKernelGGL(bn_eval_kernel, dim3(nblk(R * C)), dim3(256), 0, ST, z, a, R * C, C, gamma, beta, running_mean,
                       running_var, eps, act);
    MG_CHECK_LAUNCH("bn_eval_fwd");
    return MG_OK;
}

int mg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               const float* conv_bias, float eps, float* scale, float* shift, int C, mg_stream_t stream) {
    MG_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift, "mg_bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(nblk(C)), dim3(256), 0, ST, gamma, beta, running_mean, running_var,
                       conv_bias, eps, scale, shift, C);
    MG_CHECK_LAUNCH("bn_fold");
    return MG_OK;
}

int mg_meanT_fwd(const float* a, float* h, int B, int T, int C, mg_stream_t stream) {
    MG_CHECK_ARG(a && h && B > 0 && T > 0 && C > 0, "mg_meanT_fwd: bad args");
    hipLaunchKernelGGL(meanT_fwd_kernel, dim3((unsigned)mg_cdiv(C, 64), (unsigned)B), dim3(256), 0, ST, a, h, T, C);
    MG_CHECK_LAUNCH("meanT_fwd");
    return MG_OK;
}

int mg_meanT_bwd_mean(const float* dh, float* dz, int B, int T, int C, const float* gref, int gact, const float* gscale,
                      const float* mean_src, float* mean_out, int mean_n, float mean_scale, mg_stream_t stream) {
    MG_CHECK_ARG(dh && dz && B > 0 && T > 0 && C > 0, "mg_meanT_bwd: bad args");
    MG_CHECK_ARG(!mean_out || (mean_src && mean_n > 0), "mg_meanT_bwd_mean: the mean rider needs a source and a length");
    const long n = (long)B * T * C;
    const unsigned extra = mean_out ? 1u : 0u;
    auto al16 = [](const void* q) { return q == nullptr || ((((uintptr_t)q) & 15) == 0); };
    if ((C & 3) == 0 && al16(dh) && al16(dz) && al16(gref))
        hipLaunchKernelGGL(meanT_bwd_kernel<true>, dim3(nblk(n / 4) + extra), dim3(256), 0, ST, dh, dz, n, T, C, gref, gact, gscale,
                           mean_src, mean_out, mean_n, mean_scale);
    else
        hipLaunchKernelGGL(meanT_bwd_kernel<false>, dim3(nblk(n) + extra), dim3(256), 0, ST, dh, dz, n, T, C, gref, gact, gscale,
                           mean_src, mean_out, mean_n, mean_scale);
    MG_CHECK_LAUNCH("meanT_bwd");
    return MG_OK;
}

int mg_meanT_bwd(const float* dh, float* dz, int B, int T, int C, const float* gref, int gact, const float* gscale,
                 mg_stream_t stream) {
    return mg_meanT_bwd_mean(dh, dz, B, T, C, gref, gact, gscale, nullptr, nullptr, 0, 0.f, stream);
}

int mg_layernorm_fwd(const float* x, float* y, float* xhat, int B, int D, const float* gamma, const float* beta,
                     float eps, mg_stream_t stream) {
    MG_CHECK_ARG(x && y && gamma && beta && B > 0 && D > 0 && D <= 64, "mg_layernorm_fwd: bad args (D<=64)");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(nblk(B, 64)), dim3(64), 0, ST, x, y, xhat, B, D, gamma, beta, eps);
    MG_CHECK_LAUNCH("layernorm_fwd");
    return MG_OK;
}

int mg_layernorm_bwd_params(const float* dy, const float* xhat, float* dgamma, float* dbeta, int B, int D,
                            mg_stream_t stream) {
    MG_CHECK_ARG(dy && xhat && dgamma && dbeta && D <= 64, "mg_layernorm_bwd_params: bad args");
    hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(1), dim3(256), 0, ST, dy, xhat, dgamma, dbeta, B, D);
    MG_CHECK_LAUNCH("layernorm_bwd_params");
    return MG_OK;
}

int mg_dhead_fwd(const float* f, const float* emb, const float* w, const float* bias, float* s, int B, int Be,
                 int F, int E, mg_stream_t stream) {
    MG_CHECK_ARG(f && w && bias && s && B > 0 && F > 0, "mg_dhead_fwd: bad args");
    MG_CHECK_ARG(!emb || (Be > 0 && E > 0), "mg_dhead_fwd: bad emb shape");
    hipLaunchKernelGGL(dhead_fwd_kernel, dim3(B), dim3(64), 0, ST, f, emb, w, bias, s, Be > 0 ? Be : 1, F, emb ? E : 0);
    MG_CHECK_LAUNCH("dhead_fwd");
    return MG_OK;
}

int mg_dhead_bwd(const float* ds, const float* f, const float* w, float* dU, float* demb, int B, int Be, int F,
                 int E, int nb_emb, mg_stream_t stream) {
    MG_CHECK_ARG(ds && f && w && dU, "mg_dhead_bwd: bad args");
    hipLaunchKernelGGL(dhead_bwd_kernel, dim3(nblk((long)B * F)), dim3(256), 0, ST, ds, f, w, dU, B, F);
    if (demb)
        hipLaunchKernelGGL(dhead_demb_kernel, dim3(nblk((long)Be * E)), dim3(256), 0, ST, ds, w, demb, Be, F, E, nb_emb);
    MG_CHECK_LAUNCH("dhead_bwd");
    return MG_OK;
}

int mg_dhead_fwd_bwd(const float* ds, const float* f, const float* emb, const float* w, const float* bias, float* s,
                     float* dU, float* demb, int B, int Be, int F, int E, int nb_emb, mg_stream_t stream) {
    MG_CHECK_ARG(ds && f && w && bias && s && dU && B > 0 && F > 0, "mg_dhead_fwd_bwd: bad args");
    MG_CHECK_ARG((!emb && !demb) || (Be > 0 && E > 0), "mg_dhead_fwd_bwd: bad emb shape");
    hipLaunchKernelGGL(dhead_fwd_bwd_kernel, dim3(B + (demb ? (unsigned)mg_cdiv((long)Be * E, 64) : 0)), dim3(64), 0, ST, ds, f, emb, w, bias, s, dU, demb, B,
                       Be > 0 ? Be : 1, F, (emb || demb) ? E : 0, nb_emb);
    MG_CHECK_LAUNCH("dhead_fwd_bwd");
    return MG_OK;
}

int mg_dhead_wgrad_loss(const float* ds, const float* f, const float* emb, const float* gf, float* dw, float* dbias,
                        int nb, int ng, int Be, int F, int E, const float* s, const float* norms, float lambda_gp,
                        float* loss_out, float* gp_out, int nb_loss, mg_stream_t stream) {
    MG_CHECK_ARG(ds && f && dw && dbias, "mg_dhead_wgrad: bad args");
    MG_CHECK_ARG(!loss_out || (s && norms && gp_out && nb_loss > 0), "mg_dhead_wgrad_loss: the loss rider needs s, norms, gp_out");
    const DLoss loss{s, norms, lambda_gp, loss_out, gp_out, nb_loss};
    hipLaunchKernelGGL(dhead_wgrad_kernel, dim3(nblk(F + E + 1, 4) + (loss_out ? 1u : 0u)), dim3(256), 0, ST, ds, f, emb, gf, dw,
                       dbias, nb, ng, Be > 0 ? Be : 1, F, emb ? E : 0, loss);
    MG_CHECK_LAUNCH("dhead_wgrad");
    return MG_OK;
}

int mg_dhead_wgrad(const float* ds, const float* f, const float* emb, const float* gf, float* dw, float* dbias,
                   int nb, int ng, int Be, int F, int E, mg_stream_t stream) {
    return mg_dhead_wgrad_loss(ds, f, emb, gf, dw, dbias, nb, ng, Be, F, E, nullptr, nullptr, 0.f, nullptr, nullptr, 0, stream);
}

int mg_gp_interp(const float* real, const float* fake, const float* alpha, float* xhat, int B, long n,
                 mg_stream_t stream) {
    MG_CHECK_ARG(real && fake && alpha && xhat && B > 0 && n > 0, "mg_gp_interp: bad args");
    hipLaunchKernelGGL(gp_interp_kernel, dim3(nblk(B * n)), dim3(256), 0, ST, real, fake, alpha, xhat, n, B * n);
    MG_CHECK_LAUNCH("gp_interp");
    return MG_OK;
}

int mg_gp_penalty(const float* g, float* gbar, float* norms, float* gp, float coef, int B, long n,
                  mg_stream_t stream) {
    MG_CHECK_ARG(g && norms && B > 0 && n > 0, "mg_gp_penalty: bad args");
    hipLaunchKernelGGL(gp_norm_kernel, dim3(B), dim3(1024), 0, ST, g, gbar, norms, coef, B, n);
    if (gp) hipLaunchKernelGGL(gp_final_kernel, dim3(1), dim3(256), 0, ST, (const float*)norms, gp, B);
    MG_CHECK_LAUNCH("gp_penalty");
    return MG_OK;
}

int mg_wgan_d_loss(const float* s, const float* gp, float lambda_gp, float* out, int nb, mg_stream_t stream) {
    MG_CHECK_ARG(s && gp && out && nb > 0, "mg_wgan_d_loss: bad args");
    hipLaunchKernelGGL(wgan_d_loss_kernel, dim3(1), dim3(256), 0, ST, s, gp, (const float*)nullptr, lambda_gp, out,
                       (float*)nullptr, nb);
    MG_CHECK_LAUNCH("wgan_d_loss");
    return MG_OK;
}

int mg_wgan_d_loss_gp(const float* s, const float* norms, float lambda_gp, float* out, float* gp_out, int nb,
                      mg_stream_t stream) {
    MG_CHECK_ARG(s && norms && out && gp_out && nb > 0, "mg_wgan_d_loss_gp: bad args");
    hipLaunchKernelGGL(wgan_d_loss_kernel, dim3(1), dim3(256), 0, ST, s, (const float*)nullptr, norms, lambda_gp, out,
                       gp_out, nb);
    MG_CHECK_LAUNCH("wgan_d_loss_gp");
    return MG_OK;
}

int mg_softmax_ce(const float* logits, const int64_t* target, float* loss, float* dlogits, float coef, int B, int C,
                  mg_stream_t stream) {
    MG_CHECK_ARG(logits && target && loss && B > 0 && C > 0 && C <= 32, "mg_softmax_ce: bad args");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, ST, logits, target, loss, dlogits, coef, B, C);
    MG_CHECK_LAUNCH("softmax_ce");
    return MG_OK;
}

int mg_neg_mean(const float* s, float* out, int B, mg_stream_t stream) {
    MG_CHECK_ARG(s && out && B > 0, "mg_neg_mean: bad args");
    hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(256), 0, ST, s, out, B, -1.f);
    MG_CHECK_LAUNCH("neg_mean");
    return MG_OK;
}

int mg_mean_scaled(const float* src, float* out, int n, float scale, mg_stream_t stream) {
    MG_CHECK_ARG(src && out && n > 0, "mg_mean_scaled: bad args");
    hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(256), 0, ST, src, out, n, scale);
    MG_CHECK_LAUNCH("mean_scaled");
    return MG_OK;
}

// One lane writes the constant-rate (100 MHz) device clock: a time stamp INSIDE a captured graph, where the profiler's view of
// a forked graph is not to be trusted (tools/step_stamps.py).
__global__ void stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

int mg_stamp(unsigned long long* dst, mg_stream_t stream) {
    MG_CHECK_ARG(dst, "mg_stamp: bad args");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ST, dst);
    MG_CHECK_LAUNCH("stamp");
    return MG_OK;
}

int mg_fill(float* x, float v, long n, mg_stream_t stream) {
    MG_CHECK_ARG(x && n >= 0, "mg_fill: bad args");
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(nblk(n)), dim3(256), 0, ST, x, v, n);
    MG_CHECK_LAUNCH("fill");
    return MG_OK;
}

int mg_axpby(const float* x, float* y, float a, float b, long n, mg_stream_t stream) {
    MG_CHECK_ARG(x && y && n > 0, "mg_axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(nblk(n)), dim3(256), 0, ST, x, y, a, b, n);
    MG_CHECK_LAUNCH("axpby");
    return MG_OK;
}

int mg_copy_cols(const float* src, int sld, int soff, float* dst, int dld, int doff, int rows, int ncols,
                 int accumulate, mg_stream_t stream) {
    MG_CHECK_ARG(src && dst && rows > 0 && ncols > 0 && soff + ncols <= sld && doff + ncols <= dld, "mg_copy_cols: bad args");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk((long)rows * ncols)), dim3(256), 0, ST, src, sld, soff, dst, dld, doff,
                       rows, ncols, accumulate);
    MG_CHECK_LAUNCH("copy_cols");
    return MG_OK;
}

int mg_stage_rows(const mg_stage_job* jobs, int n_jobs, int n_rows, mg_stream_t stream) {
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_STAGE_JOBS && n_rows > 0 && n_rows <= 65535,
                 "mg_stage_rows: need 1..%d jobs and 1..65535 rows", MG_MAX_STAGE_JOBS);
    StageJobs J = {};
    long widest = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const mg_stage_job& j = jobs[i];
        MG_CHECK_ARG(j.src && j.dst && j.row_bytes > 0 && (j.row_bytes & 3) == 0 && j.src_rows > 0 &&
                         ((((uintptr_t)j.src | (uintptr_t)j.dst)) & 3) == 0 && j.rows >= 0 && j.rows <= n_rows &&
                         (j.idx || j.src_rows >= (j.rows > 0 ? j.rows : n_rows)),
                     "mg_stage_rows: job %d: rows must be non-empty multiples of 4 bytes, 4-byte aligned, and the source "
                     "must hold n_rows rows when it is not indexed", i);
        MG_CHECK_ARG(j.dst_pitch == 0 || (j.dst_pitch >= j.row_bytes && (j.dst_pitch & 3) == 0),
                     "mg_stage_rows: job %d: dst_pitch must be 0 or a multiple of 4 that holds a row", i);
        J.j[i] = j;
        widest = j.row_bytes > widest ? j.row_bytes : widest;
    }
    long bx = mg_cdiv(widest >> 4, 256);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(stage_rows_kernel, dim3((unsigned)bx, (unsigned)n_rows, (unsigned)n_jobs), dim3(256), 0, ST, J, n_rows);
    MG_CHECK_LAUNCH("stage_rows");
    return MG_OK;
}

int mg_stage_rows_cursor(const mg_stage_job* jobs, int n_jobs, int n_rows, const int64_t* order, long order_len,
                         const uint64_t* counter, const uint64_t* base, mg_stream_t stream) {
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_STAGE_JOBS && n_rows > 0 && n_rows <= 65535 && order_len > 0 && counter && base,
                 "mg_stage_rows_cursor: need 1..%d jobs, 1..65535 rows, a positive order length, counter and base", MG_MAX_STAGE_JOBS);
    StageJobs J = {};
    long widest = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const mg_stage_job& j = jobs[i];
        MG_CHECK_ARG(j.src && j.dst && j.row_bytes > 0 && (j.row_bytes & 3) == 0 && j.src_rows > 0 && !j.idx && j.rows == 0 &&
                         ((((uintptr_t)j.src | (uintptr_t)j.dst)) & 3) == 0 && (order || j.src_rows >= order_len),
                     "mg_stage_rows_cursor: job %d: 4-byte aligned rows of a multiple of 4 bytes, no per-job index / row count, and "
                     "without an order array the source must hold order_len rows", i);
        MG_CHECK_ARG(j.dst_pitch == 0 || (j.dst_pitch >= j.row_bytes && (j.dst_pitch & 3) == 0),
                     "mg_stage_rows_cursor: job %d: dst_pitch must be 0 or a multiple of 4 that holds a row", i);
        J.j[i] = j;
        widest = j.row_bytes > widest ? j.row_bytes : widest;
    }
    long bx = mg_cdiv(widest >> 4, 256);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(stage_rows_cursor_kernel, dim3((unsigned)bx, (unsigned)n_rows, (unsigned)n_jobs), dim3(256), 0, ST, J, n_rows,
                       order, order_len, (const unsigned long long*)counter, (const unsigned long long*)base);
    MG_CHECK_LAUNCH("stage_rows_cursor");
    return MG_OK;
}

int mg_transpose_bcl_blc(const float* in, float* out, int B, int C, int L, const float* gref, int gact,
                         mg_stream_t stream) {
    MG_CHECK_ARG(in && out && B > 0 && C > 0 && L > 0, "mg_transpose_bcl_blc: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)mg_cdiv(L, 32), (unsigned)mg_cdiv(C, 32), (unsigned)B), dim3(256),
                       0, ST, in, out, C, L, gref, gact);
    MG_CHECK_LAUNCH("transpose");
    return MG_OK;
}

int mg_act_bwd(const float* dy, const float* gref, int gact, const float* emul, float* dx, long n,
               mg_stream_t stream) {
    MG_CHECK_ARG(dy && dx && n > 0, "mg_act_bwd: bad args");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(n)), dim3(256), 0, ST, dy, gref, gact, emul, dx, n);
    MG_CHECK_LAUNCH("act_bwd");
    return MG_OK;
}

static int rng_fill_impl(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                         float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter,
                         double* tick_state, double* tick_state2, float beta1, float beta2, mg_stream_t stream,
                         const StageRide* ride = nullptr) {
    MG_CHECK_ARG(step_counter != nullptr, "mg_rng_fill: null step counter");
    MG_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mg_rng_fill: bad dropout probability");
    RngJobs jobs{};
    int n = 0;
    long mx = 0;
    auto add = [&](float* d, long cnt, int kind) {
        if (d && cnt > 0) { jobs.j[n++] = RngJob{d, cnt, kind, p_drop}; if (cnt > mx) mx = cnt; }
    };
    add(normal, n_normal, 0);
    add(uniform, n_uniform, 1);
    add(mask0, n_mask0, 2);
    add(mask1, n_mask1, 2);
    jobs.njobs = n;
    if (n == 0 && !ride) return MG_OK;
    MG_CHECK_ARG(n > 0, "mg_rng_fill: the staging rider needs at least one tensor to draw");
    unsigned gx = (unsigned)mg_cdiv(mg_cdiv(mx, 4), 256);
    if (gx > 256) gx = 256;
    StageRide sr{};
    if (ride) {
        sr = *ride;
        if (gx < 256) gx = 256;        // row pieces for the staging planes
    }
    hipLaunchKernelGGL(rng_fill_kernel, dim3(gx, n + sr.n_jobs), dim3(256), 0, ST, jobs, (unsigned long long)seed,
                       (unsigned long long*)step_counter, tick_state, tick_state2, (double)beta1, (double)beta2, sr);
    if (!tick_state)
        hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, ST, (unsigned long long*)step_counter);
    MG_CHECK_LAUNCH("rng_fill");
    return MG_OK;
}

int mg_rng_fill(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                           float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter,
                           mg_stream_t stream) {
    return rng_fill_impl(normal, n_normal, uniform, n_uniform, mask0, n_mask0, mask1, n_mask1, p_drop, seed,
                         step_counter, nullptr, nullptr, 0.f, 0.f, stream);
}

int mg_rng_fill_tick(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0,
                                long n_mask0, float* mask1, long n_mask1, float p_drop, uint64_t seed,
                                uint64_t* step_counter, double* adam_state, float beta1, float beta2,
                                mg_stream_t stream) {
    MG_CHECK_ARG(adam_state, "mg_rng_fill_tick: null adam_state");
    return rng_fill_impl(normal, n_normal, uniform, n_uniform, mask0, n_mask0, mask1, n_mask1, p_drop, seed,
                         step_counter, adam_state, nullptr, beta1, beta2, stream);
}

int mg_rng_fill_tick2(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0,
                      long n_mask0, float* mask1, long n_mask1, float p_drop, uint64_t seed,
                      uint64_t* step_counter, double* adam_state, double* adam_state2, float beta1, float beta2,
                      mg_stream_t stream) {
    MG_CHECK_ARG(adam_state && adam_state2 && adam_state != adam_state2, "mg_rng_fill_tick2: two distinct adam states");
    return rng_fill_impl(normal, n_normal, uniform, n_uniform, mask0, n_mask0, mask1, n_mask1, p_drop, seed,
                         step_counter, adam_state, adam_state2, beta1, beta2, stream);
}

int mg_rng_fill_tick2_stage(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0, float* mask1,
                            long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter, double* adam_state,
                            double* adam_state2, float beta1, float beta2, const mg_stage_job* jobs, int n_jobs, int n_rows,
                            const int64_t* order, long order_len, const uint64_t* base, mg_stream_t stream) {
    MG_CHECK_ARG(adam_state && adam_state2 && adam_state != adam_state2, "mg_rng_fill_tick2_stage: two distinct adam states");
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_STAGE_JOBS && n_rows > 0 && order_len > 0 && base && step_counter,
                 "mg_rng_fill_tick2_stage: need 1..%d jobs, rows, a positive order length, counter and base", MG_MAX_STAGE_JOBS);
    StageRide sr{};
    sr.n_jobs = n_jobs; sr.n_rows = n_rows; sr.order = order; sr.order_len = order_len;
    sr.base = (const unsigned long long*)base;
    for (int i = 0; i < n_jobs; ++i) {
        const mg_stage_job& j = jobs[i];
        MG_CHECK_ARG(j.src && j.dst && j.row_bytes > 0 && (j.row_bytes & 3) == 0 && j.src_rows > 0 && !j.idx && j.rows == 0 &&
                         ((((uintptr_t)j.src | (uintptr_t)j.dst)) & 3) == 0 && (order || j.src_rows >= order_len),
                     "mg_rng_fill_tick2_stage: job %d: the rules of mg_stage_rows_cursor", i);
        MG_CHECK_ARG(j.dst_pitch == 0 || (j.dst_pitch >= j.row_bytes && (j.dst_pitch & 3) == 0),
                     "mg_rng_fill_tick2_stage: job %d: dst_pitch must be 0 or a multiple of 4 that holds a row", i);
        sr.J.j[i] = j;
    }
    return rng_fill_impl(normal, n_normal, uniform, n_uniform, mask0, n_mask0, mask1, n_mask1, p_drop, seed, step_counter,
                         adam_state, adam_state2, beta1, beta2, stream, &sr);
}

int mg_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, double* state, float grad_scale, const float* gs_dev,
                 mg_stream_t stream) {
    MG_CHECK_ARG(p && g && m && v && state && n > 0, "mg_adam_flat: bad args");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, ST, state, (double)beta1, (double)beta2);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(nblk(n)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (const double*)state, grad_scale, gs_dev, (unsigned long long*)nullptr, WqTable{});
    MG_CHECK_LAUNCH("adam_flat");
    return MG_OK;
}

int mg_adam_flat_ticked(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, const double* state, float grad_scale, const float* gs_dev,
                        uint64_t* rng_step, mg_stream_t stream) {
    /* rng_step may be NULL: of the two updates one fused draw serves (mg_rng_fill_tick2) only one advances the counter */
    MG_CHECK_ARG(p && g && m && v && state && n > 0, "mg_adam_flat_ticked: bad args");
    hipLaunchKernelGGL(adam_apply_kernel, dim3(nblk(n)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, state, grad_scale, gs_dev, (unsigned long long*)rng_step, WqTable{});
    MG_CHECK_LAUNCH("adam_flat_ticked");
    return MG_OK;
}

int mg_adam_flat_wq(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, double* state, float grad_scale, const float* gs_dev,
                    int state_ticked, uint64_t* rng_step, const mg_wq_entry* table, int n_table, mg_stream_t stream) {
    MG_CHECK_ARG(p && g && m && v && state && n > 0, "mg_adam_flat_wq: bad args");
    MG_CHECK_ARG(n_table >= 0 && n_table <= MG_MAX_WQ_ENTRIES && (n_table == 0 || table), "mg_adam_flat_wq: bad table");
    WqTable t{};
    t.n = n_table;
    t.lo = n; t.hi = 0;
    for (int i = 0; i < n_table; ++i) {
        const mg_wq_entry& e = table[i];
        const long cnt = (long)e.N * e.Cc * e.K;
        MG_CHECK_ARG(e.dst && e.N > 0 && e.Cc > 0 && e.Cc % 4 == 0 && e.K > 0 && e.start >= 0 && e.start + cnt <= n &&
                     ((e.w_sc == e.K && e.w_sn == e.Cc * e.K) || (e.w_sn == e.K && e.w_sc == e.N * e.K)),
                     "mg_adam_flat_wq: table entry %d is not a dense (N,C,K) or (C,N,K) tensor inside the flat buffer", i);
        t.e[i] = e;
        if (e.start < t.lo) t.lo = e.start;
        if (e.start + cnt > t.hi) t.hi = e.start + cnt;
    }
    if (!state_ticked)
        hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, ST, state, (double)beta1, (double)beta2);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(nblk(n)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (const double*)state, grad_scale, gs_dev, (unsigned long long*)rng_step, t);
    MG_CHECK_LAUNCH("adam_flat_wq");
    return MG_OK;
}

static int sn_jobs(const mg_sn_job* jobs, int n_jobs, bool bwd, SnJobs& J, const char* who) {
    MG_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= MG_MAX_SN_JOBS, "%s: 1..%d jobs", who, MG_MAX_SN_JOBS);
    J.n = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        const mg_sn_job& q = jobs[i];
        MG_CHECK_ARG(q.w_eff && q.u && q.v && q.sigma && q.rows > 0 && q.cols > 0 && (bwd ? q.dw != nullptr : q.w_orig != nullptr),
                     "%s: job %d: null tensor or empty shape", who, i);
        J.j[i] = q;
    }
    return MG_OK;
}

int mg_spectral_norm_fwd(const mg_sn_job* jobs, int n_jobs, int power_iterations, float eps, mg_stream_t stream) {
    SnJobs J{};
    if (int rc = sn_jobs(jobs, n_jobs, false, J, "mg_spectral_norm_fwd")) return rc;
    MG_CHECK_ARG(power_iterations == 0 || power_iterations == 1, "mg_spectral_norm_fwd: 0 (eval) or 1 power iteration");
    hipLaunchKernelGGL(spectral_norm_fwd_kernel, dim3((unsigned)n_jobs), dim3(1024), 0, ST, J, power_iterations, eps);
    MG_CHECK_LAUNCH("spectral_norm_fwd");
    return MG_OK;
}

int mg_spectral_norm_bwd(const mg_sn_job* jobs, int n_jobs, mg_stream_t stream) {
    SnJobs J{};
    if (int rc = sn_jobs(jobs, n_jobs, true, J, "mg_spectral_norm_bwd")) return rc;
    hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3((unsigned)n_jobs), dim3(1024), 0, ST, J);
    MG_CHECK_LAUNCH("spectral_norm_bwd");
    return MG_OK;
}

size_t mg_grad_norm_workspace_bytes(long n) { return 1024 * sizeof(float); }

int mg_grad_norm_clip(const float* g, long n, float max_norm, float* out, void* work, size_t work_bytes,
                      mg_stream_t stream) {
    MG_CHECK_ARG(g && out && n > 0, "mg_grad_norm_clip: bad args");
    if (!work || work_bytes < mg_grad_norm_workspace_bytes(n)) { mg_set_error("mg_grad_norm_clip: workspace too small"); return MG_EWORK; }
    int nparts = (int)mg_cdiv(n, 1024 * 8);
    if (nparts > 1024) nparts = 1024;
    if (nparts < 1) nparts = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nparts), dim3(1024), 0, ST, g, n, (float*)work);
    hipLaunchKernelGGL(norm_clip_final_kernel, dim3(1), dim3(256), 0, ST, (const float*)work, nparts, max_norm, out);
    MG_CHECK_LAUNCH("grad_norm_clip");
    return MG_OK;
}

int mg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, mg_stream_t stream) {
    MG_CHECK_ARG(mu && logvar && eps && z && n > 0, "mg_reparam_fwd: bad args");
    hipLaunchKernelGGL(reparam_kernel, dim3(nblk(n)), dim3(256), 0, ST, mu, logvar, eps, z, n);
    MG_CHECK_LAUNCH("reparam");
    return MG_OK;
}

int mg_reparam_bwd(const float* dz, const float* logvar, const float* eps, const float* dmu_kld,
                   const float* dlv_kld, float* dmu, float* dlv, long n, mg_stream_t stream) {
    MG_CHECK_ARG(dz && logvar && eps && dmu && dlv && n > 0, "mg_reparam_bwd: bad args");
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(nblk(n)), dim3(256), 0, ST, dz, logvar, eps, dmu_kld, dlv_kld, dmu, dlv, n);
    MG_CHECK_LAUNCH("reparam_bwd");
    return MG_OK;
}

size_t mg_vae_loss_workspace_bytes(void) { return (VAE_LOSS_BLOCKS + 1) * sizeof(double); }

int mg_vae_loss(const float* recon, const float* x, long n_x, const float* mu, const float* logvar, long n_z,
                float beta, float* out, float* drecon, float* dmu_kld, float* dlv_kld, void* work, size_t work_bytes,
                mg_stream_t stream) {
    MG_CHECK_ARG(recon && x && mu && logvar && out && n_x > 0 && n_z > 0, "mg_vae_loss: bad args");
    if (!work || work_bytes < mg_vae_loss_workspace_bytes()) { mg_set_error("mg_vae_loss: workspace too small"); return MG_EWORK; }
    auto al16 = [](const void* q) { return q == nullptr || ((((uintptr_t)q) & 15) == 0); };
    const int vec = ((n_x & 3) == 0) && al16(recon) && al16(x) && al16(drecon);
    long nb = mg_cdiv(n_x, 4 * 256);
    if (nb > VAE_LOSS_BLOCKS) nb = VAE_LOSS_BLOCKS;
    hipLaunchKernelGGL(vae_loss_partial_kernel, dim3((unsigned)nb), dim3(256), 0, ST, recon, x, n_x, mu, logvar, n_z, beta,
                       (double*)work, drecon, dmu_kld, dlv_kld, vec);
    hipLaunchKernelGGL(vae_loss_final_kernel, dim3(1), dim3(64), 0, ST, (const double*)work, (int)nb, n_x, n_z, beta, out);
    MG_CHECK_LAUNCH("vae_loss");
    return MG_OK;
}

}  // extern "C"
