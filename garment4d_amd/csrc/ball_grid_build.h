// The cell-grid build of ball_grid.hip as a device function: ball_grid.hip wraps it in its own kernel, fps_bucket.hip runs it as a second
// ROLE of the level-1 sampling launch (the grid of a cloud depends on the cloud only, like its FPS).
#pragma once
#include "g4d_common.h"

namespace g4d {

struct GridHdr {  // 32 bytes at the start of each cloud's workspace
    float lox, loy, loz, inv_c;
    int gx, gy, gz, ncells;
};

constexpr int kGridHdrBytes = 64;
constexpr int kBuildThreads = 1024;
constexpr int kCap = 512;        // hits kept per (query, scale); more = dense ball = scan fallback
constexpr int kCapRegs = kCap / 64;  // list elements per lane when the list is held in registers
constexpr float kCellSlack = 1.01f;

__host__ __device__ inline int grid_cmax(int n) {
    int c = 512;
    while (c < n && c < 32768) c <<= 1;
    return c;
}
__host__ __device__ inline size_t grid_cloud_bytes(int n) {
    const size_t cells = ((size_t)grid_cmax(n) + 1) * 4;
    return kGridHdrBytes + ((cells + 63) & ~(size_t)63) + (size_t)n * 16;
}

__device__ __forceinline__ int cell_of(float v, float lo, float inv_c, int g) {
    // clamp makes NaN -> 0, -inf -> 0, +inf -> g-1 (such points can never be hits); finite points of the box are untouched
    const float u = fminf(fmaxf((v - lo) * inv_c, 0.f), (float)(g - 1));
    return (int)u;
}

__device__ __forceinline__ float wave_min_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ void ball_grid_build_body(int n, int cmax, int budget, float cell_req, const float *__restrict__ xyz_all,
                                                     unsigned char *__restrict__ ws_all, size_t ws_stride, int cloud) {
    extern __shared__ __attribute__((aligned(16))) int hist[];  // [cmax + 1], then 16 x 8 floats of reduction scratch
    float *red = reinterpret_cast<float *>(hist + cmax + 1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int T = kBuildThreads, W = T / 64;
    const float *xyz = xyz_all + (size_t)cloud * n * 3;
    unsigned char *ws = ws_all + (size_t)cloud * ws_stride;
    GridHdr *hdr = reinterpret_cast<GridHdr *>(ws);
    int *cellstart = reinterpret_cast<int *>(ws + kGridHdrBytes);
    float4 *sorted = reinterpret_cast<float4 *>(ws + kGridHdrBytes + ((((size_t)cmax + 1) * 4 + 63) & ~(size_t)63));
    const float INF = __builtin_inff();

    // 1. bounding box of the finite points
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    for (int k = t; k < n; k += T) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        if (fabsf(x) < INF) { lx = fminf(lx, x); hx = fmaxf(hx, x); }  // false for NaN and +-inf
        if (fabsf(y) < INF) { ly = fminf(ly, y); hy = fmaxf(hy, y); }
        if (fabsf(z) < INF) { lz = fminf(lz, z); hz = fmaxf(hz, z); }
    }
    lx = wave_min_all(lx); ly = wave_min_all(ly); lz = wave_min_all(lz);
    hx = wave_max_all(hx); hy = wave_max_all(hy); hz = wave_max_all(hz);
    if (lane == 0) { red[wave * 8 + 0] = lx; red[wave * 8 + 1] = ly; red[wave * 8 + 2] = lz;
                     red[wave * 8 + 3] = hx; red[wave * 8 + 4] = hy; red[wave * 8 + 5] = hz; }
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        lx = fminf(lx, red[w * 8 + 0]); ly = fminf(ly, red[w * 8 + 1]); lz = fminf(lz, red[w * 8 + 2]);
        hx = fmaxf(hx, red[w * 8 + 3]); hy = fmaxf(hy, red[w * 8 + 4]); hz = fmaxf(hz, red[w * 8 + 5]);
    }
    if (!(lx <= hx)) { lx = 0.f; hx = 0.f; }  // no finite coordinate on this axis
    if (!(ly <= hy)) { ly = 0.f; hy = 0.f; }
    if (!(lz <= hz)) { lz = 0.f; hz = 0.f; }
    // 2. grid: cell edge c >= kCellSlack * r_max, grown by 2^(1/3) until <= 1024 cells per axis and <= cmax cells in all
    // cell_req <= 0: the finest grid that fits `budget` (<= cmax) cells (three_nn: no radius is given) -- start just below the edge that gives
    // cmax cells over the box volume and let the loop grow it
    float c = fmaxf(cell_req, 1e-30f);
    if (!(cell_req > 0.f)) {
        const float ex = fmaxf(hx - lx, 0.f), ey = fmaxf(hy - ly, 0.f), ez = fmaxf(hz - lz, 0.f);
        const float big = fmaxf(ex, fmaxf(ey, ez));
        const float vol = fmaxf(ex, big * 1e-3f) * fmaxf(ey, big * 1e-3f) * fmaxf(ez, big * 1e-3f);
        c = vol > 0.f && vol < INF ? 0.5f * cbrtf(vol / (float)budget) : 1e-30f;
        c = fmaxf(c, big * (1.0f / 1023.0f));
        if (!(c > 0.f && c < INF)) c = 1e-30f;
    }
    int gx, gy, gz;
    float inv_c;
    for (int it = 0; it < 400; ++it) {
        inv_c = 1.0f / c;
        const float fx = floorf((hx - lx) * inv_c), fy = floorf((hy - ly) * inv_c), fz = floorf((hz - lz) * inv_c);
        if (fx < 1024.f && fy < 1024.f && fz < 1024.f) {  // also false for inf / NaN quotients
            gx = (int)fx + 1; gy = (int)fy + 1; gz = (int)fz + 1;
            if ((long long)gx * gy * gz <= (long long)budget) break;
        }
        c *= 1.2599211f;
        gx = gy = gz = 1;
        inv_c = 0.f;  // reached only if the loop runs out: one cell holding everything (still exact)
    }
    const int ncells = gx * gy * gz;
    // 3. histogram
    __syncthreads();
    for (int i = t; i <= ncells; i += T) hist[i] = 0;
    __syncthreads();
    for (int k = t; k < n; k += T) {
        const int cell = (cell_of(xyz[k * 3 + 2], lz, inv_c, gz) * gy + cell_of(xyz[k * 3 + 1], ly, inv_c, gy)) * gx +
                         cell_of(xyz[k * 3 + 0], lx, inv_c, gx);
        atomicAdd(&hist[cell], 1);
    }
    __syncthreads();
    // 4. exclusive scan of hist[0 .. ncells) in place; hist[ncells] = n
    const int chunk = (ncells + T - 1) / T;
    const int c0 = min(t * chunk, ncells), c1 = min(c0 + chunk, ncells);
    int sum = 0;
    for (int i = c0; i < c1; ++i) sum += hist[i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    int *wsum = reinterpret_cast<int *>(red);
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base + incl - sum;
    for (int i = c0; i < c1; ++i) {
        const int v = hist[i];
        hist[i] = run;
        run += v;
    }
    if (t == 0) hist[ncells] = n;
    __syncthreads();
    for (int i = t; i <= ncells; i += T) cellstart[i] = hist[i];
    if (t == 0) {
        GridHdr h;
        h.lox = lx; h.loy = ly; h.loz = lz; h.inv_c = inv_c;
        h.gx = gx; h.gy = gy; h.gz = gz; h.ncells = ncells;
        *hdr = h;
    }
    __syncthreads();
    // 5. scatter into cell order (order inside a cell is irrelevant: the query sorts hits by index)
    for (int k = t; k < n; k += T) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        const int cell = (cell_of(z, lz, inv_c, gz) * gy + cell_of(y, ly, inv_c, gy)) * gx + cell_of(x, lx, inv_c, gx);
        const int dst = atomicAdd(&hist[cell], 1);
        sorted[dst] = make_float4(x, y, z, __int_as_float(k));
    }
}


}  // namespace g4d
