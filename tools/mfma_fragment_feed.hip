// fp32 MFMA chain fed by double-buffered ds_read_b128 fragments (the conv loop's pattern, nothing else)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, const float* in, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int off = threadIdx.x * 4;
    f32x4 fa[2], fb[2];
    fa[0] = *(f32x4*)&lds[off & 8191]; fb[0] = *(f32x4*)&lds[(off + 2048) & 8191];
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        const float* base = lds + ((off + i * 16) & 2047);
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl & 1][s4], fb[sl & 1][s4], acc, 0, 0, 0);
                if (s4 == (MODE == 1 ? 2 : 0)) {
                    fa[(sl + 1) & 1] = *(const f32x4*)(base + (sl + 1) * 80);
                    fb[(sl + 1) & 1] = *(const f32x4*)(base + 4096 + (sl + 1) * 256);
                }
                if (MODE == 2 && s4 == 1) { fa[(sl + 1) & 1] = *(const f32x4*)(base + (sl + 1) * 80); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(int iters) {
    float *out, *in; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 16); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, cyc, iters); (void)hipDeviceSynchronize(); }
    long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const char* names[] = {"prefetch in gap 0 (4 MFMAs ahead)", "prefetch in gap 2 (2 MFMAs ahead)"};
    printf("%-50s %.1f cyc/MFMA\n", names[MODE], h[0] / ((double)iters * 24));
}
int main() { run<0>(100); run<1>(100); return 0; }
