// micro-benchmark: do VALU instructions overlap a queued MFMA?  One wave per SIMD (256 threads, 1 workgroup) or two (2 workgroups on one CU
// are not guaranteed -- so 512 threads in one workgroup): a loop of 8 independent MFMAs, each followed by M independent v_fma_f32 on other
// registers.  cycles per MFMA = 32 + 4 M (no overlap) or max(32, 4 M) (overlap)?  fp32 16x16x4 and bf16 16x16x32.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int M, int KIND>
__global__ void k(float *out, long long *cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-6f + i;
    const float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999f, p = 1.0001f, q = 1e-6f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(1.0f + i); hb[i] = (__bf16)(0.5f); }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int m = 0; m < M; ++m) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[m % 8]) : "v"(p), "v"(q));
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M, int KIND>
void run(const char *name, float *out, long long *cyc, int threads) {
    const int iters = 20000;
    k<M, KIND><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    k<M, KIND><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // clock64 ticks at 100 MHz on this part: convert through a calibration run with M = 0 (32 / 16 cycles per MFMA is exact)
    printf("%s waves/CU/4=%d VALU per MFMA=%2d : ticks per MFMA of waves 0, 1, 4, 5, last = %7.2f %7.2f %7.2f %7.2f %7.2f\n", name, threads / 256, M, (double)h[0] / (iters * 8.0),
           (double)h[1] / (iters * 8.0), (double)h[4 % (threads / 64)] / (iters * 8.0), (double)h[5 % (threads / 64)] / (iters * 8.0), (double)h[threads / 64 - 1] / (iters * 8.0));
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    for (int threads : {64, 256, 512, 1024}) {
        run<0, 0>("f32 16x16x4 ", out, cyc, threads); run<2, 0>("f32 16x16x4 ", out, cyc, threads); run<4, 0>("f32 16x16x4 ", out, cyc, threads);
        run<6, 0>("f32 16x16x4 ", out, cyc, threads); run<8, 0>("f32 16x16x4 ", out, cyc, threads); run<12, 0>("f32 16x16x4 ", out, cyc, threads);
        run<16, 0>("f32 16x16x4 ", out, cyc, threads);
        run<0, 1>("bf16 16x16x32", out, cyc, threads); run<2, 1>("bf16 16x16x32", out, cyc, threads); run<4, 1>("bf16 16x16x32", out, cyc, threads);
        run<8, 1>("bf16 16x16x32", out, cyc, threads);
    }
    return 0;
}
