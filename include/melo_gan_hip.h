/*
 * melo_gan_hip.h -- C-ABI of libmelogan_hip.so (gfx950 / MI355X).
 *
 * The reference (kaushik87599/Melo-GAN) has no FFI: its hot path is a set of
 * torch.nn modules (ATen ops).  Each entry point below replaces the ATen op(s)
 * that the cited reference lines dispatch; the host side (melo-gan_amd/) calls
 * them through ctypes with raw device pointers owned by PyTorch's allocator.
 *
 * Conventions
 *   - All tensors are fp32, CHANNELS-LAST: activations (B, T, C) contiguous in C
 *     (the layout of the reference's `notes` tensors; the reference permutes to
 *     (B, C, T) for torch's Conv1d -- src/gan/models.py:159, ed_model.py:65 --
 *     this library never does).
 *   - Weights keep the reference's state_dict layouts: Conv1d (Cout, Cin, K),
 *     ConvTranspose1d (Cin, Cout, K), Linear (out, in).
 *   - Every function enqueues on `stream` (a hipStream_t) and returns without
 *     synchronising.  No allocation, no retained pointers.
 *   - Return 0 on success; <0 on error: -1 bad argument/shape, -2 unsupported
 *     configuration, -3 workspace too small, -4 HIP runtime error.  The message is
 *     available from mg_last_error() (thread-local).
 */
#ifndef MELO_GAN_HIP_H
#define MELO_GAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mg_stream_t; /* hipStream_t */

/* activation ids (epilogue `act`, and `gact` = which derivative to apply) */
enum { MG_ACT_NONE = 0, MG_ACT_RELU = 1, MG_ACT_LRELU = 2, MG_ACT_GELU = 3, MG_ACT_TANH = 4 };

/* Fused epilogue of the window-GEMM kernels.  For an output element v=acc:
 *   v += bias[n];  v = v*scale[n] + shift[n];  zout[idx] = v;  v = act(v);
 *   v *= act'(gref[idx]) (gact);  v *= emul[idx];  v *= gscale[n];
 *   y = accumulate ? y + v : v
 * Null pointers skip the corresponding stage.  gref semantics per gact:
 *   RELU  : gref is the forward OUTPUT a (mask a>0)
 *   LRELU : gref is the forward output a (a>0 ? 1 : 0.2)
 *   GELU  : gref is the forward PRE-activation z (exact erf form)
 *   TANH  : gref is the forward output a (1 - a*a)
 */
typedef struct mg_epilogue {
    const float* bias;
    const float* scale;
    const float* shift;
    float* zout;
    int act;
    const float* gref;
    int gact;
    const float* emul;
    const float* gscale;
    int accumulate;
} mg_epilogue;

int mg_version(void);
const char* mg_last_error(void);

/* ---- window GEMM: Conv1d / ConvTranspose1d / Linear, forward and data-gradient ----
 *
 * mg_conv1d_gather: y[b,t,n] = EPI( sum_{k,c} x[b, t*stride + k - (K-1)/2, c] * W(n,c,kw) )
 *   kw = flip ? K-1-k : k ;  W(n,c,k) = w[n*w_sn + c*w_sc + k].
 *   Tout = (Tin + 2*((K-1)/2) - K)/stride + 1.   stride in {1,2}, K in {1,3,5}.
 *   Replaces: nn.Conv1d forward (src/gan/models.py:141-145, ed_model.py:28, src/ae/model.py:11-19)
 *             with w_sn=Cin*K, w_sc=K;  nn.Linear forward (K=1,T=1; w_sn=in, w_sc=1);
 *             ConvTranspose1d data-gradient (stride 2; w_sn=Cout*K, w_sc=K);
 *             Conv1d stride-1 data-gradient (flip=1; w_sn=K, w_sc=Cin*K);
 *             Linear data-gradient (K=1; w_sn=1, w_sc=in).
 *   x batch stride xbs, y batch stride ybs (elements; 0 => dense).
 *   work/work_bytes: optional scratch (mg_conv_workspace_bytes(B, Tout, N)); when given and the output
 *   tiling alone would leave most CUs idle, the channel reduction is split over workgroups into partial
 *   slabs that a second kernel sums in fixed order before the epilogue.  NULL => never split.
 */
size_t mg_conv_workspace_bytes(int B, int Tout, int N);
/* Launch hint for the window-GEMM launches that FOLLOW (process-wide, baked into a hipGraph at capture): `bytes` of extra LDS per
 * workgroup, i.e. fewer resident workgroups per CU.  For a branch that runs on a side stream beside the step's critical path
 * (the frozen emotion discriminator): its 1024-workgroup convolutions otherwise fill every CU's registers and the critical
 * path's small dependent kernels wait for their workgroups to retire.  0 restores the default. */
int mg_conv_set_lds_pad(long bytes);
int mg_conv1d_gather(const float* x, const float* w, float* y,
                     int B, int Tin, int Cin, int N, int K, int stride, int flip,
                     int w_sn, int w_sc, long xbs, long ybs,
                     const mg_epilogue* epi, void* work, size_t work_bytes, mg_stream_t stream);

/* mg_conv1d_scatter2: stride-2, K=5, padding 2, output_padding 1 transposed convolution
 *   y[b,t,n] = EPI( sum_{k,c : t+2-k even} x[b,(t+2-k)/2,c] * W(n,c,k) ),  t < Tout,
 *   Tout = 2*Tin (ConvTranspose1d forward) or 2*Tin-1 (data-gradient of a stride-2 Conv1d whose
 *   input length was odd).
 *   Replaces: nn.ConvTranspose1d forward (src/gan/models.py:56-62, src/ae/model.py:66-74)
 *             with w_sn=K, w_sc=Cout*K;  Conv1d stride-2 data-gradient (src/gan/models.py:141-145
 *             backward; w_sn=K... i.e. n=Cin: w_sn=K, w_sc=Cin*K).
 */
int mg_conv1d_scatter2(const float* x, const float* w, float* y,
                       int B, int Tin, int Cin, int N, int Tout,
                       int w_sn, int w_sc, long xbs, long ybs,
                       const mg_epilogue* epi, void* work, size_t work_bytes, mg_stream_t stream);

/* ---- the stride-2, 5-tap window GEMMs on 16x16x4 MFMA tiles (csrc/conv16_mfma.hip) ----
 * The same arithmetic as mg_conv1d_gather(stride 2, K 5) [transposed = 0: nn.Conv1d(k5,s2,p2) forward,
 * src/gan/models.py:141-145, src/ae/model.py:11-19; nn.ConvTranspose1d data-gradient] and mg_conv1d_scatter2
 * [transposed = 1: nn.ConvTranspose1d(k5,s2,p2,op1) forward, src/gan/models.py:56-62; Conv1d stride-2 data-gradient,
 * Tout = 2*Tin or 2*Tin-1] in 64x32 / 32x32 output tiles with the whole channel reduction inside each workgroup: no
 * split-K workspace, no second launch, bitwise run-to-run reproducible.  Weights in the WQ layout
 *     wq[((c/4)*5 + k)*N + n][c%4] = W(n, c, k)        (N output columns, c reduction channel, Cin % 16 == 0, N % 32 == 0)
 * which mg_wq_relayout derives from a reference-layout tensor (W(n,c,k) = w[n*w_sn + c*w_sc + k]) and mg_adam_flat_wq
 * keeps current after every optimiser step.  mg_conv16_supported tells whether a shape is covered (else use the calls above). */
int mg_wq_relayout(const float* w, float* wq, int N, int Cc, int K, int w_sn, int w_sc, mg_stream_t stream);
int mg_conv16_supported(int B, int Tin, int Cin, int N, int transposed, int Tout);
int mg_conv16(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
              long xbs, long ybs, const mg_epilogue* epi, mg_stream_t stream);
/* The same launch that ALSO leaves per-column partial statistics of the values v it stores, so that the train-mode
 * BatchNorm that follows needs no reduction pass of its own: for every (tile, wave row) p and column n
 *   part[p][0][n] = sum_rows v[row, n],   [1][n] = sum_rows (v - mean_p)^2 about that wave's OWN mean,   [2][n] = rows summed
 * (combined by the parallel-variance rule in fp64: nothing is formed as E[x^2] - mean^2).
 * part: 3 * part_rows * N floats; part_rows and the batch rows a tile spans from mg_conv16_plan (a caller that stacks
 * several BatchNorm groups along the batch needs the group size to be a multiple of batch_rows_per_tile).
 * mg_bn_train_fwd_parts finishes the statistics (fixed order, fp64), moves the running ones and applies, all in ONE launch:
 * the semantics of mg_bn_train_fwd_groups (nn.BatchNorm1d training forward, src/gan/models.py:57-61). */
int mg_conv16_plan(int B, int Tin, int N, int transposed, int* batch_rows_per_tile, int* part_rows, int* tile_rows);
int mg_conv16_stats(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                    long xbs, long ybs, const mg_epilogue* epi, float* part, mg_stream_t stream);
/* The same launch also writing the temporal mean of what it stores -- AdaptiveAvgPool1d(1) behind the critic's last
 * convolution (src/gan/models.py:148): pool[b][n] = pool_scale * sum_t y[b][t][n].  Gather form only; mg_conv16_poolable says
 * whether a shape qualifies (every sample's time axis is one wave's rows of a tile: Tout = 32, or 16 with 32-row tiles). */
int mg_conv16_poolable(int B, int Tin, int Cin, int N);
int mg_conv16_pool(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, long xbs, long ybs,
                   const mg_epilogue* epi, float* pool, float pool_scale, mg_stream_t stream);
/* One launch with any of the riders: statistics (part), temporal mean (pool), the permuted output order
 *   y_perm: y[b*ybs + n*Tout + tout] -- the (B, N*Tout) order behind the reference's `view(B, 256, L)` (src/gan/models.py:70),
 *           so the data-gradient of the first deconvolution lands in decoder.pre.2's output order with no transpose launch
 *           (zout / gref / emul keep the dense (b, tout, n) index);
 * and the gradient penalty's interpolate (src/gan/utils.py:76-79): for batch rows b < mix_rows ALSO
 *   mix_out[i] = mix_alpha[b] * mix_real[i] + (1 - mix_alpha[b]) * y[i]   at the element's y index i. */
typedef struct mg_conv16_extra {
    float* part;
    float* pool;
    float pool_scale;
    int y_perm;
    const float* mix_real;
    const float* mix_alpha;
    float* mix_out;
    int mix_rows;
    /* bnb_*: the launch's output y is the gradient that reaches a train-mode BatchNorm + ReLU / LeakyReLU layer from above
     * (src/gan/models.py:57-61 backwards); with that layer's forward tensors a, z (laid out like y) and saved statistics it also
     * leaves bnb_part[p][0][n] = sum_rows g, [1][n] = sum_rows g * x_hat (g = y * act'(a), x_hat = (z - mean) * invstd; p, rows
     * as for `part`: 2 * part_rows * N DOUBLES, accumulated in fp64 like the reduction pass they replace), so the BatchNorm's
     * backward is ONE launch, mg_bn_train_bwd_parts */
    const float* bnb_a;
    const float* bnb_z;
    const float* bnb_mean;
    const float* bnb_invstd;
    double* bnb_part;
    int bnb_act;
} mg_conv16_extra;
int mg_conv16_ex(const float* x, const float* wq, float* y, int B, int Tin, int Cin, int N, int transposed, int Tout,
                 long xbs, long ybs, const mg_epilogue* epi, const mg_conv16_extra* extra, mg_stream_t stream);
/* mg_bn_train_bwd behind a conv16 launch that left the two column sums (bnb_part, part_rows rows of 2 x C doubles): the sums
 * are added in row order and the gradient applied -- one launch. */
int mg_bn_train_bwd_parts(const double* part, int part_rows, const float* da, const float* a, const float* z, float* dz, long R,
                          int C, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                          float* dgamma, float* dbeta, int act, mg_stream_t stream);
int mg_bn_train_fwd_parts(const float* part, int part_rows_per_group, int groups, const float* z, float* a, long R, int C,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, float* save_mean, float* save_invstd, int act, mg_stream_t stream);

/* ---- row chains (csrc/row_chain.hip): a sample's small layer stack as ONE launch -- one workgroup walks one row through
 *      an op list with the activations in LDS vector slots (MG_CHAIN_SLOTS slots of MG_CHAIN_MAX_VEC floats).  Used for the
 *      numeric encoder (LayerNorm + 3 Linear, src/gan/feature_encoder.py:16-45) forward and data-gradient, the emotion
 *      classifier's tail (ed_model.py:61,86-95,147-165: project, MLP, head, cross-entropy and their data-gradients) and the
 *      critic's tail (src/gan/models.py:149-169: fc, scoring head and their data-gradients).  Row r of every global operand
 *      is at ptr + r * ld.  Ops (fields not named are ignored):
 *   LOAD       slot b [n1, n1+n0) (+= if i0) p0 row (r % i1 if i1 > 0)
 *   COPY       slot b [n1, n1+n0) = slot a [i1, i1+n0)
 *   STORE      q0 row <- slot a [0,n0)
 *   MEAN_T     slot b [0,n0) = mean over t < i0 of p0[(r*i0 + t)*ld0 + c]  (AdaptiveAvgPool1d(1) of a (rows, i0, ld0) tensor); q0 rows <- it
 *   LAYERNORM  slot b = LN(slot a [0,n0<=64)) * p0 + p1, eps f0; q0 rows <- xhat, q1 rows <- output (optional)
 *   LIN_FWD    slot b [0,n1) = act(slot a [0,n0) W^T + p1) * p2 row;  W(n,k) = p0[n*ld0 + k];  q0 rows <- pre-activation,
 *              q1 rows <- output (optional)                                              (nn.Linear forward + act + dropout mask)
 *   LIN_DGRAD  slot b [0,n1) = (slot a [0,n0) W) * act'(p1 row) * p2 row;  W(o,i) = p0[o*ld0 + i];  q1 rows <- output
 *   SOFTMAX_CE slot a = logits [0,n0<=32), t0 = int64 targets: q0[r] = -log softmax[target];  slot b = f0 * (softmax - onehot)
 *              (F.cross_entropy forward + backward per row; an out-of-range target poisons the row with NaN)
 *   DHEAD      critic head on slot a = f [0,n0): q0[r] = f . w[0,n0) + emb row (r % i1) . w[n0, n0+n1) + p1[0], w = p0, emb = p2;
 *              slot b = p3[r] * w[0,n0) * lrelu'(f);  q1 rows <- p3[r] * w[n0, n0+n1)       (src/gan/models.py:160-169 + backward)
 */
#define MG_CHAIN_MAX_OPS 16
#define MG_CHAIN_SLOTS 6
#define MG_CHAIN_MAX_VEC 512
enum { MG_CH_LOAD = 1, MG_CH_STORE = 2, MG_CH_LAYERNORM = 3, MG_CH_LIN_FWD = 4, MG_CH_LIN_DGRAD = 5, MG_CH_SOFTMAX_CE = 6,
       MG_CH_DHEAD = 7, MG_CH_MEAN_T = 8, MG_CH_COPY = 9 };
typedef struct mg_chain_op {
    int kind;
    int a, b;
    int n0, n1;
    int act;
    int i0, i1;
    float f0;
    const float* p0; long ld0;
    const float* p1; long ld1;
    const float* p2; long ld2;
    const float* p3; long ld3;
    float* q0; long lq0;
    float* q1; long lq1;
    const int64_t* t0;
} mg_chain_op;
int mg_row_chain(const mg_chain_op* ops, int n_ops, int rows, mg_stream_t stream);
/* out[0] = scale * mean(src[0..n)) (one block): the loss scalar behind a chain's per-row cross-entropy terms */
int mg_mean_scaled(const float* src, float* out, int n, float scale, mg_stream_t stream);
/* dst[0] = the device's constant-rate clock (wall_clock64, 100 MHz) when the node runs: a time stamp inside a captured graph
 * (measurement only: tools/step_stamps.py) */
int mg_stamp(unsigned long long* dst, mg_stream_t stream);

/* ---- skinny GEMM for nn.Linear forward / data-gradient with few rows (M = batch) ----
 *   y[M,N] = EPI( x[M,K] @ W^T ),  W(n,c) = w[n*w_sn + c*w_sc], one of the strides must be 1:
 *   nn.Linear forward (src/gan/models.py:24-26,47-49,151; feature_encoder.py:23,38; ed_model.py:61,78,90):
 *       w (out,in): w_sn = in, w_sc = 1;   data-gradient dx = dy @ w: w_sn = 1, w_sc = in(=N).
 *   Deep K is split over workgroups into partial slabs in `work` (mg_linear_workspace_bytes, may be 0). */
size_t mg_linear_workspace_bytes(int M, int N, int K);
int mg_linear(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, int w_sc,
              const mg_epilogue* epi, void* work, size_t work_bytes, mg_stream_t stream);
/* The same with the OUTPUT COLUMNS PERMUTED: column n' of y is weight row (n' % C) * perm_L + n' / C, C = N / perm_L (bias,
 * scale, gscale follow the weight row; zout / gref / emul the stored order): y comes out as the channels-last (M, perm_L, C)
 * tensor the reference reaches by `view(B, 256, L)` + permute(0, 2, 1) on the way into the first ConvTranspose1d
 * (src/gan/models.py:70-73) -- no transpose launch.  perm_L = 0: mg_linear. */
int mg_linear_perm(const float* x, const float* w, float* y, int M, int K, int N, int w_sn, int w_sc,
                   const mg_epilogue* epi, int perm_L, void* work, size_t work_bytes, mg_stream_t stream);

/* ---- stride-1 three-tap Conv1d (padding 1) by minimal filtering F(2,3) along time (csrc/conv_wino.hip) ----
 * Same result as mg_conv1d_gather(K = 3, stride = 1) up to rounding (the operands are transformed: a few ulp), with 2/3 of
 * its matrix-pipe work: four channel GEMMs per PAIR of outputs instead of six.  For the frozen emotion discriminator's
 * conv1-3 and their input gradients (src/emotion_discriminator/ed_model.py:24-46 inside src/gan/train_gan.py:228-236).
 *   mg_wino3_weights: wt[Cin/4][4][N][4] = the filter transform of g_k = W(n, c, flip ? 2-k : k), W(n,c,k) = w[n*w_sn + c*w_sc + k]
 *                     (forward: w (N,Cin,3), w_sn = 3 Cin, w_sc = 3; data gradient of a Conv1d whose weight is (Cout,Cin,3):
 *                     N = Cin, "Cin" = Cout, w_sn = 3, w_sc = 3 Cin_conv, flip = 1).  Cin % 4 == 0.
 *   mg_conv1d_wino3:  y[b,t,n] = EPI( sum_{k,c} x[b, t+k-1, c] * g_k(n,c) ), x (B,T,Cin) and y (B,T,N) dense, 16-byte aligned;
 *                     T even, Cin % 16 == 0, N % 64 == 0 (mg_conv1d_wino3_supported).  Honours mg_conv_set_lds_pad. */
int mg_conv1d_wino3_supported(int B, int T, int Cin, int N);
int mg_wino3_weights(const float* w, float* wt, int N, int Cin, long w_sn, long w_sc, int flip, mg_stream_t stream);
int mg_conv1d_wino3(const float* x, const float* wt, float* y, int B, int T, int Cin, int N, const mg_epilogue* epi,
                    mg_stream_t stream);
/* mg_wino3_weights for several filters in ONE launch (a network in training transforms all its layers' weights, forward and
 * data-gradient images, at the top of every step: emotion_discriminator/engine.py) */
#define MG_MAX_WINO_WJOBS 8
typedef struct mg_wino3_wjob {
    const float* w;
    float* wt;
    int N, Cin;
    long w_sn, w_sc;
    int flip;
} mg_wino3_wjob;
int mg_wino3_weights_multi(const mg_wino3_wjob* jobs, int n_jobs, mg_stream_t stream);

/* Which instantiation of conv_wgemm_kernel<S,K,TR2,TM,TN> a call launches: TM*10+TN (22: 128x128 tile,
 * 12: 64x128, 11: 64x64).  m_rows = B*Tout (gather) or B*Tin (scatter2).  Lets a profiler label
 * launches by kernel symbol. */
int mg_conv_tile_config(long m_rows, int N, int scatter2);

/* The layers with a <= 8 channel reduction or output side (NOTE_DIM = 4, reference config/gan_config.yaml:43-44: critic
 * conv.0 src/gan/models.py:129, generator deconv.6 src/gan/models.py:67-70, emotion discriminator conv0
 * src/emotion_discriminator/ed_model.py:24, the VAE's first / last layer) do not run on the MFMA tile kernels:
 * mg_conv1d_gather / mg_conv1d_scatter2 route them to VALU kernels (csrc/conv_thin.hip).  This reports the route a call
 * takes: 0 = MFMA tile kernel, 1 = thin_in_kernel, 2 / 3 = thin_out_kernel<TR2, 4 / 8>.  MG_CONV_THIN=0 disables it. */
int mg_conv_thin_route(const float* x, long xbs, int Cin, int N, int K, int stride, int transposed);

/* ---- bf16-storage / fp32-accumulate variant of the frozen emotion-discriminator branch (SECONDARY configuration) ----
 * The branch (src/gan/train_gan.py:228-236: ED(generated) -> cross-entropy -> gradient w.r.t. the generated notes;
 * src/emotion_discriminator/ed_model.py:24-69) has no trained parameters in the GAN step, so it can store activations and
 * its folded weights in bf16 without touching optimiser state.  Products accumulate in fp32 (v_mfma_f32_32x32x16_bf16),
 * epilogue arithmetic is fp32.  Never the default; bench.py reports it as a separate line.
 *
 * mg_epilogue_bf16: v = acc; v = v*scale[n] + shift[n]; zout[di] = bf16(v); v = act(v); v *= act'(gact, gref[di]);
 *                   v *= gscale[n]; y = accumulate ? y + v : v (accumulate: fp32 y only).  zout / gref are bf16 tensors.
 * mg_wb_relayout:   wb[k][n][c] = bf16(w[n*w_sn + c*w_sc + (flip ? K-1-k : k)])  -- the weight image of the kernel below
 *                   (forward: w_sn = Cin*K, w_sc = K; stride-1 data gradient: flip = 1, w_sn = K, w_sc = Cin*K).
 * mg_conv1d_s1_bf16: y[b,t,n] = EPI( sum_{k,c} x[b, t + k - (K-1)/2, c] * wb[k][n][c] ), K in {3,5}; x is bf16 or (x_f32)
 *                   fp32 converted on the way into LDS; y is bf16 or (y_f32) fp32.  Needs T % 128 == 0, Cin % 32 == 0,
 *                   N % 64 == 0 (mg_conv1d_s1_bf16_supported), dense 16-byte aligned tensors.
 * mg_meanT_fwd_bf16 / mg_meanT_bwd_bf16: the temporal mean (ed_model.py:65) over a bf16 activation and its backward
 *                   fused with act'(gref) * gscale, writing a bf16 gradient. */
typedef struct mg_epilogue_bf16 {
    const float* scale;
    const float* shift;
    void* zout;
    int act;
    const void* gref;
    int gact;
    const float* gscale;
    int accumulate;
} mg_epilogue_bf16;
int mg_wb_relayout(const float* w, void* wb, int N, int Cc, int K, int w_sn, int w_sc, int flip, mg_stream_t stream);
int mg_conv1d_s1_bf16_supported(int B, int T, int Cin, int N, int K);
int mg_conv1d_s1_bf16(const void* x, int x_f32, const void* wb, void* y, int y_f32, int B, int T, int Cin, int N, int K,
                      const mg_epilogue_bf16* epi, mg_stream_t stream);
int mg_meanT_fwd_bf16(const void* a, float* h, int B, int T, int C, mg_stream_t stream);
int mg_meanT_bwd_bf16(const float* dh, void* dz, const void* gref, int gact, const float* gscale, int B, int T, int C,
                      mg_stream_t stream);

/* ---- weight gradient ----
 * out[a][b][k] = sum over segments, batches, u of  S[bt,u,a] * L[bt, u*stride + k - (K-1)/2, b]
 *   S: (nb, Ts, A) "small" tensor, L: (nb, Tl, Bc) "large" tensor (zero outside [0,Tl)).
 *   Conv1d   wgrad: S=dy (A=Cout), L=x  (Bc=Cin)  -> (Cout,Cin,K)
 *   ConvT1d  wgrad: S=x  (A=Cin),  L=dy (Bc=Cout) -> (Cin,Cout,K)
 *   Linear   wgrad: Ts=Tl=1,K=1: S=dy (A=out), L=x (Bc=in) -> (out,in)
 * Up to two (S,L,nb) segments are summed (segment 1 may have nb1=0).
 * Fused bias gradient (optional): bias_from = 1 -> bias_out[a] = sum of S over segment 0's rows (Conv1d / Linear
 * bias); bias_from = 2 -> bias_out[b] = sum of L over segment 0's rows (ConvTranspose1d bias); 0/NULL -> none.
 * `work` must hold mg_wgrad_workspace_bytes(...) bytes.  Deterministic (no atomics).
 */
size_t mg_wgrad_workspace_bytes(int A, int Bc, int K, int nb_total, int Ts);
int mg_wgrad(const float* s0, const float* l0, int nb0,
             const float* s1, const float* l1, int nb1,
             float* out, float* bias_out, int bias_from, int Ts, int Tl, int A, int Bc, int K, int stride,
             void* work, size_t work_bytes, mg_stream_t stream);

/* Several independent weight gradients of ONE (K, stride) in one launch (+ one reduce launch if any of them is split):
 * the small layers' gradients are each a few workgroups at the launch floor.  Each job has the meaning of one
 * mg_wgrad call; `work` must hold the SUM over jobs of mg_wgrad_workspace_bytes(A, Bc, K, nb0 + nb1, Ts), each rounded
 * up to 256 bytes.  Outputs of different jobs must not overlap. */
#define MG_MAX_WGRAD_JOBS 8
typedef struct mg_wgrad_job {
    const float* s0; const float* l0; int nb0;
    const float* s1; const float* l1; int nb1;
    float* out; float* bias_out; int bias_from;
    int Ts, Tl, A, Bc;
} mg_wgrad_job;
int mg_wgrad_multi(const mg_wgrad_job* jobs, int n_jobs, int K, int stride, void* work, size_t work_bytes,
                   mg_stream_t stream);

/* ---- per-channel column reductions over rows of a (R, C) matrix ----
 * sum[c] = sum_r x[r,c] (and sumsq if sumsq != NULL).  Used for bias gradients and BN statistics.
 * `work`: mg_colsum_workspace_bytes(C). */
size_t mg_colsum_workspace_bytes(int C);
int mg_colsum(const float* x, long R, int C, float* sum, float* sumsq,
              void* work, size_t work_bytes, mg_stream_t stream);

/* ---- BatchNorm1d, training mode, fused with ReLU (src/gan/models.py:57-61, src/ae/model.py:12-21) ----
 * z: (R, C) pre-BN (R = B*T).  Computes batch mean / biased var, a = relu((z-mean)*invstd*gamma+beta),
 * saves mean/invstd, updates running_mean/var (momentum 0.1, unbiased var) -- also under no_grad.
 * act: any MG_ACT_* (the generator uses RELU, the emotion discriminator's pre-training GELU). */
size_t mg_bn_workspace_bytes(int C);
int mg_bn_train_fwd(const float* z, float* a, long R, int C,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* save_mean, float* save_invstd, int act,
                    void* work, size_t work_bytes, mg_stream_t stream);
/* The same over `groups` independent batches of R rows each, stacked along the rows of z / a (the generator forward of
 * the critic step and of the generator step run as one 2B-row pass: src/gan/train_gan.py:186-189 and :216-219 use the
 * same generator weights): batch statistics per group (save_mean / save_invstd: (groups, C)), running statistics
 * updated group after group -- what consecutive forward calls do.  `work`: mg_bn_groups_workspace_bytes(C, groups). */
size_t mg_bn_groups_workspace_bytes(int C, int groups);
int mg_bn_train_fwd_groups(const float* z, float* a, long R, int C, int groups, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps,
                           float* save_mean, float* save_invstd, int act,
                           void* work, size_t work_bytes, mg_stream_t stream);
/* backward: da (grad wrt a), a (forward output: the ReLU / LeakyReLU mask, tanh'), z.  Produces dz, dgamma, dbeta.
 * beta: only read for act = MG_ACT_GELU, whose derivative is taken at the BN output (recomputed from z); else may be NULL. */
int mg_bn_train_bwd(const float* da, const float* a, const float* z, float* dz, long R, int C,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                    float* dgamma, float* dbeta, int act,
                    void* work, size_t work_bytes, mg_stream_t stream);
/* eval mode: a = act((z-running_mean)/sqrt(running_var+eps)*gamma+beta) */
int mg_bn_eval_fwd(const float* z, float* a, long R, int C, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, float eps, int act,
                   mg_stream_t stream);
/* fold eval-mode BN + conv bias into per-channel scale/shift (ed_model.py:37, eval mode) */
int mg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               const float* conv_bias, float eps, float* scale, float* shift, int C, mg_stream_t stream);

/* ---- mean over time (AdaptiveAvgPool1d(1); src/gan/models.py:148, ed_model.py:60) ---- */
int mg_meanT_fwd(const float* a, float* h, int B, int T, int C, mg_stream_t stream);
/* dz[b,t,c] = dh[b,c]/T * act'(gref[b,t,c]) * gscale[c]   (gref/gscale may be NULL) */
int mg_meanT_bwd(const float* dh, float* dz, int B, int T, int C,
                 const float* gref, int gact, const float* gscale, mg_stream_t stream);

/* ---- LayerNorm over the last dim (src/gan/feature_encoder.py:19), D <= 64 ---- */
int mg_layernorm_fwd(const float* x, float* y, float* xhat, int B, int D,
                     const float* gamma, const float* beta, float eps, mg_stream_t stream);
int mg_layernorm_bwd_params(const float* dy, const float* xhat, float* dgamma, float* dbeta,
                            int B, int D, mg_stream_t stream);

/* ---- critic head (src/gan/models.py:157,165-169): s[b] = <[f[b], emb[b % Be]], w> + bias ---- */
int mg_dhead_fwd(const float* f, const float* emb, const float* w, const float* bias, float* s,
                 int B, int Be, int F, int E, mg_stream_t stream);
/* dU[b,j] = ds[b]*w[j]*lrelu'(f[b,j]); demb[be,j] (+)= sum over b%Be==be of ds[b]*w[F+j] (if demb) */
int mg_dhead_bwd(const float* ds, const float* f, const float* w, float* dU, float* demb,
                 int B, int Be, int F, int E, int nb_emb, mg_stream_t stream);
/* mg_dhead_fwd and mg_dhead_bwd in one launch: ds is a constant of the step, so the backward does not wait for s. */
int mg_dhead_fwd_bwd(const float* ds, const float* f, const float* emb, const float* w, const float* bias, float* s,
                     float* dU, float* demb, int B, int Be, int F, int E, int nb_emb, mg_stream_t stream);
/* dw[j<F] = sum_{b<nb} ds[b] f[b,j] + sum_{b<ng} gf[b,j];  dw[F+j] = sum_{b<nb} ds[b] emb[b%Be,j];
 * dbias = sum_{b<nb} ds[b] */
int mg_dhead_wgrad(const float* ds, const float* f, const float* emb, const float* gf,
                   float* dw, float* dbias, int nb, int ng, int Be, int F, int E, mg_stream_t stream);

/* ---- WGAN-GP pieces (src/gan/utils.py:75-90) ---- */
/* xhat[b,:] = alpha[b]*real[b,:] + (1-alpha[b])*fake[b,:]   (n = T*C elements per sample) */
int mg_gp_interp(const float* real, const float* fake, const float* alpha, float* xhat,
                 int B, long n, mg_stream_t stream);
/* norms[b] = ||g[b,:]||_2 ; gp = mean((norm-1)^2) ; gbar[b,:] = coef*(2/B)*(norm-1)/norm * g[b,:]
 * gp may be NULL: the mean is then left to mg_wgan_d_loss_gp (one launch fewer) */
int mg_gp_penalty(const float* g, float* gbar, float* norms, float* gp, float coef,
                  int B, long n, mg_stream_t stream);

/* ---- losses ---- */
/* loss_d = mean(s[nb:2nb]) - mean(s[0:nb]) + lambda_gp*gp ; out[0]=loss_d out[1]=mean_real out[2]=mean_fake */
int mg_wgan_d_loss(const float* s, const float* gp, float lambda_gp, float* out, int nb, mg_stream_t stream);
/* the same with the penalty computed from the per-sample gradient norms: gp_out[0] = mean((norms-1)^2) */
int mg_wgan_d_loss_gp(const float* s, const float* norms, float lambda_gp, float* out, float* gp_out, int nb,
                      mg_stream_t stream);
/* cross entropy over C<=32 classes: loss = mean_b(-log softmax[b,y_b]); dlogits = coef*(softmax-onehot)/B */
int mg_softmax_ce(const float* logits, const int64_t* target, float* loss, float* dlogits,
                  float coef, int B, int C, mg_stream_t stream);
/* out[0] = -mean(s[0:B]) */
/* mg_dhead_wgrad with the critic's loss scalars riding in the same launch (one extra block): mg_wgan_d_loss_gp's outputs
 * (loss_out = {loss_d, mean real, mean fake}, gp_out = mean((norm-1)^2)); loss_out = NULL: mg_dhead_wgrad. */
int mg_dhead_wgrad_loss(const float* ds, const float* f, const float* emb, const float* gf, float* dw, float* dbias,
                        int nb, int ng, int Be, int F, int E, const float* s, const float* norms, float lambda_gp,
                        float* loss_out, float* gp_out, int nb_loss, mg_stream_t stream);
/* mg_meanT_bwd with a scalar mean riding in the same launch (one extra block): mean_out[0] = mean_scale * mean(mean_src[0..n))
 * -- the generator's adversarial loss -mean(D(fake)), src/gan/train_gan.py:224.  mean_out = NULL: mg_meanT_bwd. */
int mg_meanT_bwd_mean(const float* dh, float* dz, int B, int T, int C, const float* gref, int gact, const float* gscale,
                      const float* mean_src, float* mean_out, int mean_n, float mean_scale, mg_stream_t stream);
int mg_neg_mean(const float* s, float* out, int B, mg_stream_t stream);

/* ---- elementwise helpers ---- */
int mg_fill(float* x, float v, long n, mg_stream_t stream);
int mg_axpby(const float* x, float* y, float a, float b, long n, mg_stream_t stream); /* y = a*x + b*y */
/* dst[r, doff + j] = src[r, soff + j] for j < ncols (row-major 2-D copy; accumulate adds) */
int mg_copy_cols(const float* src, int sld, int soff, float* dst, int dld, int doff,
                 int rows, int ncols, int accumulate, mg_stream_t stream);
/* (B, C, L) <-> (B, L, C): out[b, l, c] = in[b, c, l]  (src/gan/models.py:70 view + :73 permute); with gref (laid
 * out like `out`) the result is multiplied by act'(gref): the activation backward of the layer behind the view. */
int mg_transpose_bcl_blc(const float* in, float* out, int B, int C, int L, const float* gref, int gact,
                         mg_stream_t stream);
/* y = x * act'(gref) * emul   (standalone epilogue pieces for tiny tensors) */
int mg_act_bwd(const float* dy, const float* gref, int gact, const float* emul, float* dx, long n,
               mg_stream_t stream);

/* ---- batch staging (the device side of the DataLoader collate + .to(device) of src/gan/train_gan.py:80,172-178):
 *      for every job, dst row r <- src row (idx ? idx[r] : r) for r < n_rows, ALL jobs in one launch.  Rows are
 *      row_bytes long (a multiple of 4; 16-byte aligned rows take the 16-byte path); dst rows are dst_pitch bytes
 *      apart (0: dense), so a job can also fill a column block of a wider matrix.  idx is a
 *      device array of n_rows int64 indices, clamped into [0, src_rows) on the device (an out-of-range index never
 *      reads outside the source). */
/* ---- spectral normalisation of a weight (torch.nn.utils.spectral_norm, dim 0, n_power_iterations 1, eps 1e-12) ----
 * The reference wraps the emotion discriminator's Conv1d / Linear layers in it when `use_spectral_norm` is set
 * (src/emotion_discriminator/ed_model.py:29-32,79-82; FeatureEncoder(use_sn): src/gan/feature_encoder.py:24-31).
 * weight_mat = w_orig as (rows = out, cols = everything else).  One launch serves several layers.
 *   mg_spectral_norm_fwd: power_iterations = 1 (training forward): v = normalize(W^T u), u = normalize(W v), in place;
 *                         then (also with 0 = eval) sigma = u . (W v) and w_eff = w_orig / sigma.
 *   mg_spectral_norm_bwd: dw (the gradient w.r.t. w_eff, in place) <- (dw - <dw, w_eff> u v^T) / sigma = the gradient w.r.t.
 *                         w_orig (u, v constants, as autograd sees them). */
#define MG_MAX_SN_JOBS 8
typedef struct mg_sn_job {
    const float* w_orig;
    float* w_eff;
    float* u;       /* (rows) */
    float* v;       /* (cols) */
    float* sigma;   /* (1) */
    float* dw;      /* bwd only */
    int rows, cols;
} mg_sn_job;
int mg_spectral_norm_fwd(const mg_sn_job* jobs, int n_jobs, int power_iterations, float eps, mg_stream_t stream);
int mg_spectral_norm_bwd(const mg_sn_job* jobs, int n_jobs, mg_stream_t stream);

#define MG_MAX_STAGE_JOBS 8
typedef struct mg_stage_job {
    const void* src;
    void* dst;
    const int64_t* idx; /* NULL: straight copy of the first n_rows rows */
    long row_bytes;
    long src_rows;
    long dst_pitch;     /* bytes between destination rows; 0 = row_bytes */
    long rows;          /* rows of THIS job (<= n_rows); 0 = n_rows */
} mg_stage_job;
int mg_stage_rows(const mg_stage_job* jobs, int n_jobs, int n_rows, mg_stream_t stream);
/* The same with the source rows picked on the DEVICE: row r of every job comes from position
 *   p = ((counter[0] - base[0]) * n_rows + r) mod order_len   of the epoch's order (order[p]; NULL = the identity),
 * i.e. batch number (counter - base) of an HBM-resident split's shuffled order (the DataLoader's sampler,
 * src/gan/train_gan.py:80) -- so that a captured training step stages its own batch on every replay, with no host launch
 * between steps.  counter is advanced by another launch of the step (the Philox step counter doubles as batch counter);
 * the host writes `order` and `base` once per epoch.  Jobs carry no idx / rows of their own. */
int mg_stage_rows_cursor(const mg_stage_job* jobs, int n_jobs, int n_rows, const int64_t* order, long order_len,
                         const uint64_t* counter, const uint64_t* base, mg_stream_t stream);

/* ---- per-step random inputs (replaces torch.randn / torch.rand / nn.Dropout's bernoulli draws:
 *      src/gan/train_gan.py:188,218; src/gan/utils.py:76; src/gan/feature_encoder.py:34) in ONE launch:
 *      normal[n_normal] ~ N(0,1), uniform[n_uniform] ~ U(0,1), mask{0,1} = keep-mask * 1/(1-p_drop).
 *      Philox4x32-10 keyed by `seed`; *step_counter (device, uint64) is read and then advanced on device,
 *      so a captured hipGraph draws fresh numbers at every replay.  Any pointer may be NULL. */
int mg_rng_fill(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter, mg_stream_t stream);

/* The same draw WITHOUT advancing *step_counter; instead the Adam state of the optimiser whose update will consume
 * the draws is advanced here (it is not read by this launch).  Pair it with mg_adam_flat_ticked, which applies the
 * update without advancing its state and advances *rng_step instead: two launches per sub-step instead of four. */
int mg_rng_fill_tick(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                     float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter,
                     double* adam_state, float beta1, float beta2, mg_stream_t stream);
/* One draw serving TWO updates (critic and generator step fused into one graph): both Adam states are advanced. */
int mg_rng_fill_tick2(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                      float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter,
                      double* adam_state, double* adam_state2, float beta1, float beta2, mg_stream_t stream);
/* mg_rng_fill_tick2 and mg_stage_rows_cursor (counter = step_counter) as ONE launch: the staging rides as extra block planes.
 * Both only read the step counter and a fused step needs both first -- one launch and one dependent-launch gap fewer. */
int mg_rng_fill_tick2_stage(float* normal, long n_normal, float* uniform, long n_uniform, float* mask0, long n_mask0,
                            float* mask1, long n_mask1, float p_drop, uint64_t seed, uint64_t* step_counter,
                            double* adam_state, double* adam_state2, float beta1, float beta2, const mg_stage_job* jobs,
                            int n_jobs, int n_rows, const int64_t* order, long order_len, const uint64_t* base,
                            mg_stream_t stream);

/* ---- fused flat Adam / AdamW (torch.optim.Adam defaults; src/gan/train_gan.py:136-145,
 *      src/ae/train_ae.py:79).  state: double[4] = {step, beta1^step, beta2^step, unused},
 *      advanced on device so the launch is hipGraph-replayable.  grad_scale multiplies g first
 *      (used for 1/world_size and for clip_grad_norm_ via a device scalar if gs_dev != NULL). */
int mg_adam_flat(float* p, const float* g, float* m, float* v, long n,
                 float lr, float beta1, float beta2, float eps, float weight_decay,
                 double* state, float grad_scale, const float* gs_dev, mg_stream_t stream);
/* The update alone, for a state that mg_rng_fill_tick has already advanced; advances *rng_step (see there) unless
 * rng_step is NULL (the second of the two updates behind one mg_rng_fill_tick2 draw). */
int mg_adam_flat_ticked(float* p, const float* g, float* m, float* v, long n,
                        float lr, float beta1, float beta2, float eps, float weight_decay,
                        const double* state, float grad_scale, const float* gs_dev, uint64_t* rng_step,
                        mg_stream_t stream);
/* The update that ALSO keeps WQ-layout copies (mg_conv16) of some weight tensors current: entry = a dense tensor
 * W(n,c,k) = w[n*w_sn + c*w_sc + k] at [start, start + N*Cc*K) of the flat buffer, in (N,Cc,K) order (w_sn = Cc*K,
 * w_sc = K) or (Cc,N,K) order (w_sn = K, w_sc = N*K); dst (N*Cc*K floats) receives the updated values in WQ order.
 * state_ticked: the Adam state was advanced by mg_rng_fill_tick(2) (rng_step, if not NULL, is advanced here); otherwise
 * it is advanced by this call. */
#define MG_MAX_WQ_ENTRIES 8
typedef struct mg_wq_entry {
    long start;
    int N, Cc, K, w_sn, w_sc;
    float* dst;
} mg_wq_entry;
int mg_adam_flat_wq(float* p, const float* g, float* m, float* v, long n,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    double* state, float grad_scale, const float* gs_dev, int state_ticked, uint64_t* rng_step,
                    const mg_wq_entry* table, int n_table, mg_stream_t stream);
/* out[0] = sqrt(sum g^2) ; out[1] = min(1, max_norm/(norm+1e-6))  (clip_grad_norm_, train_ae.py:121) */
int mg_grad_norm_clip(const float* g, long n, float max_norm, float* out, void* work, size_t work_bytes,
                      mg_stream_t stream);
size_t mg_grad_norm_workspace_bytes(long n);

/* ---- VAE extras (src/ae/model.py:127-133, src/ae/train_ae.py:35-51) ---- */
/* z = mu + eps*exp(0.5*logvar) */
int mg_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, mg_stream_t stream);
/* backward of the reparameterisation: dmu = dz + dmu_kld ; dlv = dz*eps*0.5*exp(0.5*logvar) + dlv_kld */
int mg_reparam_bwd(const float* dz, const float* logvar, const float* eps, const float* dmu_kld,
                   const float* dlv_kld, float* dmu, float* dlv, long n, mg_stream_t stream);
/* mse = mean((recon-x)^2); kld = -0.5*mean(1+lv-mu^2-exp(lv)); out = {total, mse, kld};
 * drecon = 2(recon-x)/n_x ; dmu, dlv include beta-weighted KLD grads PLUS the z-path grads are added by caller */
size_t mg_vae_loss_workspace_bytes(void);
int mg_vae_loss(const float* recon, const float* x, long n_x, const float* mu, const float* logvar, long n_z,
                float beta, float* out, float* drecon, float* dmu_kld, float* dlv_kld,
                void* work, size_t work_bytes, mg_stream_t stream);

/* ---- hipGraph capture of a launch sequence issued through this library (or anything else on the stream) ---- */
int mg_graph_begin(mg_stream_t stream);
int mg_graph_end(mg_stream_t stream, void** graph_exec_out);
/* mg_graph_end with the capture instantiated n (1..8) times (launches may then alternate between the executables). */
int mg_graph_end_n(mg_stream_t stream, void** graph_execs_out, int n);
/* kernel nodes of the graph the calling thread captured last (-1: unknown): the launches one replay stands for */
int mg_graph_last_kernel_nodes(void);
int mg_graph_launch(void* graph_exec, mg_stream_t stream);
int mg_graph_destroy(void* graph_exec);

/* ---- timing helper: HIP events on an arbitrary stream (bench.py roofline leg) ---- */
int mg_event_create(void** ev);
int mg_event_record(void* ev, mg_stream_t stream);
int mg_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int mg_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* MELO_GAN_HIP_H */
