// fp32 MFMA dependent chain with filler instructions in the gaps
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, const float* in, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float a = threadIdx.x * 0.001f, b = 1.0f;
    float f0 = a, f1 = b, f2 = a + 1, f3 = b + 2;
    f32x4 v4 = {a, b, a, b};
    int off = threadIdx.x * 4; int sreg = iters;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if ((MODE == 5 || MODE == 6 || MODE == 7) && (u & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (MODE == 1 || MODE == 6) { asm volatile("v_mov_b32 %0, %1" : "=v"(f0) : "v"(f1)); }
            if (MODE == 2 || MODE == 7) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %1, %2\n v_mov_b32 %3, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (MODE == 3 || MODE == 5) { v4 = *(f32x4*)&lds[(off + u * 64) & 8191]; asm volatile("" :: "v"(v4)); }
            if (MODE == 4) { *(f32x4*)&lds[(off + u * 1024) & 8191] = v4; }
            if (MODE == 12) { v4 = *(f32x4*)&lds[(off + u * 64) & 8191]; f32x4 w4 = *(f32x4*)&lds[(off + 2048 + u * 64) & 8191]; asm volatile("" :: "v"(v4), "v"(w4)); }
            if (MODE == 13) { typedef volatile __attribute__((address_space(3))) float lv; lv* d = (lv*)&lds[(off + u * 1024) & 8191]; d[0] = f0; d[1] = f1; d[2] = f2; d[3] = f3; }
            if (MODE == 14) { *(f32x4*)&lds[(off + u * 1024) & 8191] = v4; v4 = *(const f32x4*)&in[(off + u * 64) & 8191]; }
            if (MODE == 15 && (u & 3) == 0) { __syncthreads(); }
            if (MODE == 8) { lds[(off + u * 1024) & 8191] = f0; lds[((off + u * 1024) & 8191) + 1] = f1; }
            if (MODE == 9) { v4 = *(const f32x4*)&in[(off + u * 64) & 8191]; }
            if (MODE == 10) { asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3" : "+s"(sreg)); }
            if (MODE == 11 && u == 0) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %1, %2\n v_mov_b32 %3, %0\n v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %1, %2\n v_mov_b32 %3, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    float s = f0 + f1 + f2 + f3 + v4[0] + v4[1] + v4[2] + v4[3] + sreg;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE>
void run(int iters) {
    float *out, *in; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 16); (void)hipMalloc(&in, 8192 * 4);
    (void)hipMemset(in, 0, 8192 * 4);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, cyc, iters); (void)hipDeviceSynchronize(); }
    long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const char* names[] = {"bare chain", "1 v_mov per gap", "4 v_mov per gap", "1 ds_read_b128 per gap", "1 ds_write_b128 per gap", "ds_read_b128 per gap, 2 alternating accumulators", "1 v_mov per gap, 2 alternating accumulators", "4 v_mov per gap, 2 alternating accumulators", "1 ds_write2_b32 per gap", "1 global_load_dwordx4 per gap", "2 SALU per gap", "8 v_mov in one gap of 8", "2 ds_read_b128 per gap", "4 ds_write_b32 per gap", "ds_write_b128 + global_load per gap", "barrier every 4th gap"};
    printf("%-50s %.1f cyc/MFMA\n", names[MODE], h[0] / ((double)iters * 8));
}
int main() { run<0>(100); run<1>(100); run<2>(100); run<3>(100); run<4>(100); run<5>(100); run<6>(100); run<7>(100); run<8>(100); run<9>(100); run<10>(100); run<11>(100); run<12>(100); run<13>(100); run<14>(100); run<15>(100); return 0; }
