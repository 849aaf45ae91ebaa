// diagnostic: effective shader clock while other work runs.  One wave spins a dependent VALU chain and stamps the shader-cycle counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) before and after:  f_shader = d(memtime) / d(realtime) * 100 MHz.
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_probe_kernel(long long *out, int iters) {
    float a = threadIdx.x * 1e-9f;
    long long r0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = __builtin_fmaf(a, 1.000001f, 1e-7f);
    }
    long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (long long)a; }
}
extern "C" int clock_probe(long long *out, int iters, hipStream_t s) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, out, iters);
    return (int)hipGetLastError();
}
