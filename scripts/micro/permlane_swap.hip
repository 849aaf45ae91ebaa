// gfx950 v_permlane16_swap / v_permlane32_swap against __shfl_xor(x, 16 / 32) (ds_bpermute): semantics check.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/permlane_swap.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor16(float x) {
    unsigned a = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    // r[0] = vdst after, r[1] = src0 after
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32(float x) {
    unsigned a = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__global__ void k(float *o, const float *in) {
    float x = in[threadIdx.x];
    o[threadIdx.x] = xor16(x);
    o[64 + threadIdx.x] = xor32(x);
    o[128 + threadIdx.x] = __shfl_xor(x, 16);
    o[192 + threadIdx.x] = __shfl_xor(x, 32);
    unsigned a = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    o[256 + threadIdx.x] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    o[320 + threadIdx.x] = fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
int main() {
    float *d_in, *d_o; float h[64], o[384];
    for (int i = 0; i < 64; ++i) h[i] = (i * 37) % 64;
    hipMalloc(&d_in, 256); hipMalloc(&d_o, 1536);
    hipMemcpy(d_in, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_o, d_in);
    hipMemcpy(o, d_o, 1536, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { if (o[i] != o[128 + i]) bad++; if (o[64 + i] != o[192 + i]) bad++; }
    int badm = 0;
    for (int i = 0; i < 64; ++i) { if (o[256 + i] != fmaxf(h[i], h[i ^ 16])) badm++; if (o[320 + i] != fmaxf(h[i], h[i ^ 32])) badm++; }
    printf("badmax=%d\n", badm);
    printf("bad=%d  x16: %g %g %g %g | x32: %g %g\n", bad, o[0], o[16], o[17], o[40], o[64], o[64 + 33]);
    return 0;
}
