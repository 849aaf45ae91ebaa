// micro-benchmark: dependent-issue latency and shader clock with 1 / 8 / 256 workgroups active
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void dep_chain(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 1e-9f, b = 1.000001f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = fmaf(a, b, 1e-7f);  // dependent chain
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void dep_dpp(float *out, long long *cyc, int iters) {
    float v = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n" : "+v"(v));
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void lds_chain(float *out, long long *cyc, int iters) {
    __shared__ int s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) p = s[p];
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void barrier_chain(float *out, long long *cyc, int iters) {
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = 0;
}
int main() {
    float *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long long h[4];
    for (int blocks : {1, 8, 256}) for (int threads : {64, 256, 1024}) {
        const int iters = 20000;
        for (int which = 0; which < 4; ++which) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) dep_chain<<<blocks, threads>>>(out, cyc, iters);
                if (which == 1) dep_dpp<<<blocks, threads>>>(out, cyc, iters);
                if (which == 2) lds_chain<<<blocks, threads>>>(out, cyc, iters);
                if (which == 3) barrier_chain<<<blocks, threads>>>(out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
            const double per = which == 0 ? iters * 16.0 : (which == 1 ? iters * 4.0 : iters);
            const char *nm[] = {"dep fma", "dep dpp-step(+nop)", "dep lds read", "barrier"};
            printf("blocks=%3d threads=%4d %-20s: %8.1f ns/op  clock64 ticks/op=%7.1f  => tick rate %.2f GHz\n", blocks, threads, nm[which],
                   ms * 1e6 / per, (double)h[0] / per, (double)h[0] / (ms * 1e6));
        }
    }
    return 0;
}
