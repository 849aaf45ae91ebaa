// How fast can the waves of a persistent launch claim units from ONE device counter?  (round 6, dynamic unit scheduling)
// Each of `nwg` x 4 waves performs `m` dependent agent-scope fetch-adds on the same address (latency chain per wave); reports latency of a
// lone chain and the aggregate rate with the chip full.   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) int gint;
__global__ void __launch_bounds__(256) claims(int *ctr, int m, int stride, long long *cyc) {
    const int lane = threadIdx.x & 63;
    int v = 0;
    int *p = ctr + (size_t)(blockIdx.x % stride) * 64;   // stride = 1: everybody on one address; > 1: spread over that many lines
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < m; ++i) {
        if (lane == 0) v += __hip_atomic_fetch_add((gint *)p, 1 + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (v == 0x7fffffff) ctr[1] = v;
}
int main() {
    int *ctr; long long *cyc;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int stride : {1, 8, 64}) for (int nwg : {1, 64, 512, 2048}) for (int m : {64}) {
        hipMemset(ctr, 0, 1 << 20);
        hipLaunchKernelGGL(claims, dim3(nwg), dim3(256), 0, 0, ctr, m, stride, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(claims, dim3(nwg), dim3(256), 0, 0, ctr, m, stride, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)nwg * 4 * m;
        printf("lines %2d  workgroups %4d x 4 waves x %d claims: %8.1f us, %7.1f ns per claim aggregate (%6.1f claims/us), one chain: %lld ticks per claim\n", stride, nwg, m, ms * 1e3,
               ms * 1e6 / n, n / (ms * 1e3), c / m);
    }
    return 0;
}
