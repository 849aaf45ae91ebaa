// micro-benchmark: issue rate of plain and packed fp32 VALU instructions (8 independent chains per wave; 1, 2, 4 waves per SIMD)
//   hipcc --offload-arch=gfx950 -O3 -w scripts/micro/valu_rate.hip -o scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float *out, long long *cyc, int iters) {
    float x[8]; f32x2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-6f + i; y[i] = (f32x2){x[i], x[i] + 1.f}; }
    const float p = 1.0001f, q = 1e-6f; const f32x2 p2 = {p, p}, q2 = {q, q};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(p), "v"(q));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[j]) : "v"(p2), "v"(q2));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[j]) : "v"(q2));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[j]) : "v"(p2));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(q));
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND> void run(const char *name, float *out, long long *cyc) {
    for (int threads : {256, 512, 1024}) {
        const int iters = 20000;
        k<KIND><<<1, threads>>>(out, cyc, iters); hipDeviceSynchronize();
        k<KIND><<<1, threads>>>(out, cyc, iters); hipDeviceSynchronize();
        long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-14s waves per SIMD %d: %6.2f cycles per instruction per wave (first wave), %6.2f (last wave) -> %5.2f cycles per instruction per SIMD\n", name, threads / 256,
               (double)h[0] / (iters * 8.0), (double)h[threads / 64 - 1] / (iters * 8.0), (double)h[threads / 64 - 1] / (iters * 8.0) / (threads / 256));
    }
}
int main() {
    float *out; long long *cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    run<0>("v_fma_f32", out, cyc); run<4>("v_add_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<2>("v_pk_add_f32", out, cyc); run<3>("v_pk_mul_f32", out, cyc);
    return 0;
}
