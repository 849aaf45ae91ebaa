// Cell-bucketed ball query for gfx950 -- the "LDS-bucketed neighbour search" form of ball_query_kernel_fast
// (/root/reference/modules/pointnet2/pointnet2/src/ball_query_gpu.cu:9-67) for large clouds whose balls are small
// compared with the cloud (BASELINE configs 2 and 5: a radius-0.1 ball holds 0.4 % of a unit cloud, so > 99 % of the
// B*M*N pair tests of the scan are avoidable).  Output is bit-identical to g4d_ball_query_msg_f32 for ANY input: the same
// distance expression (dist2<FM>, g4d_common.h) decides membership, the cells only choose which points get tested.
//
//   build (one workgroup per cloud): bounding box of the finite points -> uniform grid, cell edge c = 1.01 * r_max (grown
//     until the grid fits `cmax` cells and 1024 cells per axis) -> counting sort: histogram with LDS atomics, exclusive
//     scan in LDS, scatter of (x, y, z, index) records into cell order.  Cell ids run x-fastest, so the 3 x-neighbours of
//     a cell are one contiguous range of records.
//   query (one wave per query): a point within r_max of the query lies in one of the 27 cells around the query's cell
//     (|u_p - u_q| < r_max / c + rounding < 1 in cell units; the margin 1 - 1/1.01 = 1e-2 is 40x the worst rounding of the
//     two cell coordinates, 2 * 1024 * 2^-23 = 2.4e-4, and a query OUTSIDE the box is handled by clipping the un-clamped
//     cell range) = 9 contiguous record ranges, walked as ONE virtual sequence (lane -> range by 8 compares against the
//     wave-uniform prefix sums of the range lengths): all 64 lanes busy whatever the ranges' lengths, four record loads in
//     flight per lane.  Every record is tested against every radius of the layer; the hits' ORIGINAL indices are appended
//     to a per-(wave, scale) LDS list (ballot + mbcnt).  "First nsample hits in
//     ascending index order" is then a selection + rank computation on that list: with more hits than slots the nsample-th
//     smallest index is found by bisection on the value (list in registers: compare + ballot + popcount per step, no LDS),
//     the survivors are ranked (rank = number of smaller indices; LDS broadcast reads) and written straight to out[rank];
//     padding = the rank-0 index, rows without a hit are zeros.
//   dense ball (more than kCap = 512 hits: far more candidates than slots): the first nsample hits by index sit
//     near the start of the cloud, so that (query, scale) falls back to the index-ordered scan with early exit -- the
//     kernel never does worse than the scan it replaces.
//
// Algorithmic bytes: build 12*N read + 16*N written per cloud; query 12*M + 4*M*sum(nsample) + the visited records.
#include <cstdlib>

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

__global__ void __launch_bounds__(kBuildThreads) ball_grid_build_kernel(int n, int cmax, float cell_req, const float *__restrict__ xyz_all,
                                                                       unsigned char *__restrict__ ws_all, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) int hist[];  // [cmax + 1], then 16 x 8 floats of reduction scratch
    float *red = reinterpret_cast<float *>(hist + cmax + 1);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int T = kBuildThreads, W = T / 64;
    const float *xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    unsigned char *ws = ws_all + (size_t)blockIdx.x * ws_stride;
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
    float c = fmaxf(cell_req, 1e-30f);
    int gx, gy, gz;
    float inv_c;
    for (int it = 0; it < 400; ++it) {
        inv_c = 1.0f / c;
        const float fx = floorf((hx - lx) * inv_c), fy = floorf((hy - ly) * inv_c), fz = floorf((hz - lz) * inv_c);
        if (fx < 1024.f && fy < 1024.f && fz < 1024.f) {  // also false for inf / NaN quotients
            gx = (int)fx + 1; gy = (int)fy + 1; gz = (int)fz + 1;
            if ((long long)gx * gy * gz <= (long long)cmax) break;
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

struct BgArgs {
    float radius2[4];
    int nsample[4];
    int *idx[4];
};

// un-clamped cell coordinate of a query along one axis, limited to [-2, g + 1] (NaN -> -2: an empty range)
__device__ __forceinline__ int query_cell(float v, float lo, float inv_c, int g) {
    const float u = floorf((v - lo) * inv_c);
    return (int)fminf(fmaxf(u, -2.f), (float)(g + 1));
}

template <int NS, int FM>
__global__ void __launch_bounds__(256) ball_grid_query_kernel(int n, int m, int qpw, const BgArgs a, const float *__restrict__ new_xyz_all,
                                                             const float *__restrict__ xyz_all, const unsigned char *__restrict__ ws_all,
                                                             size_t ws_stride, int cmax) {
    __shared__ int lst[4][NS][kCap];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.y;
    const unsigned char *ws = ws_all + (size_t)b * ws_stride;
    const GridHdr *hdr = reinterpret_cast<const GridHdr *>(ws);
    const int *cellstart = reinterpret_cast<const int *>(ws + kGridHdrBytes);
    const float4 *sorted = reinterpret_cast<const float4 *>(ws + kGridHdrBytes + ((((size_t)cmax + 1) * 4 + 63) & ~(size_t)63));
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float lox = hdr->lox, loy = hdr->loy, loz = hdr->loz, inv_c = hdr->inv_c;
    const int gx = hdr->gx, gy = hdr->gy, gz = hdr->gz;
    const int ry = lane % 3 - 1, rz = (lane / 3) % 3 - 1;  // lanes 0..8 look up the 9 record ranges of a query

    for (int qi = 0; qi < qpw; ++qi) {
        const int q = (blockIdx.x * 4 + wave) * qpw + qi;  // wave-uniform
        if (q >= m) break;
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const int cx = query_cell(qx, lox, inv_c, gx), cy = query_cell(qy, loy, inv_c, gy), cz = query_cell(qz, loz, inv_c, gz);
        // lane r < 9: cells (x0..x1, cy + ry, cz + rz) = records [s, e)
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gx - 1);
        const int yy = cy + ry, zz = cz + rz;
        int s = 0, e = 0;
        if (lane < 9 && x0 <= x1 && yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
            const int row = (zz * gy + yy) * gx;
            s = cellstart[row + x0];
            e = cellstart[row + x1 + 1];
        }
        // the 9 ranges as ONE virtual sequence of `total` records: entry j lives at record j + off[r] for cum[r] <= j < cum[r+1]
        int cum[10], off[9];
        cum[0] = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int rs = __builtin_amdgcn_readlane(s, r), re = __builtin_amdgcn_readlane(e, r);
            off[r] = rs - cum[r];
            cum[r + 1] = cum[r] + (re - rs);
        }
        const int total = cum[9];
        int h[NS];
#pragma unroll
        for (int sc = 0; sc < NS; ++sc) h[sc] = 0;

        // 4 x 64 entries per step: the four record loads of a lane are issued together (the walk is a chain of L2 round trips
        // otherwise), then each is tested against every radius; hits are appended in arrival order, the selection below sorts
        for (int j0 = 0; j0 < total; j0 += 4 * 64) {
            bool all_dense = true;  // every scale already holds more than kCap hits: the lists are useless, the scan fallback decides
#pragma unroll
            for (int sc = 0; sc < NS; ++sc) all_dense &= h[sc] > kCap;
            if (all_dense) break;
            float4 p[4];
            bool valid[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 64 + lane;
                valid[u] = j < total;
                int o = off[0];
#pragma unroll
                for (int r = 1; r < 9; ++r) o = (j >= cum[r]) ? off[r] : o;
                p[u] = sorted[valid[u] ? j + o : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u * 64 >= total) break;  // wave-uniform
                const float d2 = dist2<FM>(qx - p[u].x, qy - p[u].y, qz - p[u].z);  // ball_query_gpu.cu:30 under the contraction contract
#pragma unroll
                for (int sc = 0; sc < NS; ++sc) {
                    const bool hit = valid[u] && d2 < a.radius2[sc];
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    if (mask != 0ull) {
                        const int slot = h[sc] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit && slot < kCap) lst[wave][sc][slot] = __float_as_int(p[u].w);
                        h[sc] += __builtin_popcountll(mask);
                    }
                }
            }
        }

#pragma unroll
        for (int sc = 0; sc < NS; ++sc) {
            const int ns = a.nsample[sc];
            int *out = a.idx[sc] + ((size_t)b * m + q) * ns;
            const int hs = h[sc];
            if (hs == 0) {
                for (int l = lane; l < ns; l += 64) out[l] = 0;  // no hit: the reference leaves the caller's zeros
            } else if (hs <= kCap) {
                // The ns smallest indices among the hs hits, in ascending order.  LDS accesses of one wave execute in order, so
                // the list written above is visible here without a barrier.
                int *L = lst[wave][sc];
                int hacc = hs;
                if (hs > ns) {
                    // more hits than slots: find t = the smallest value with count(v < t) == ns by bisection on the VALUE
                    // (indices are distinct, so the count steps by one) -- lanes hold the list in registers, a step is
                    // compare + ballot + popcount, no LDS -- then keep exactly the ns elements below t
                    int v[kCapRegs];
#pragma unroll
                    for (int e = 0; e < kCapRegs; ++e) v[e] = (e * 64 + lane < hs) ? L[e * 64 + lane] : 0x7fffffff;
                    const int ne = (hs + 63) >> 6;
                    int lo = 0, hi = n;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        int c = 0;
#pragma unroll
                        for (int e = 0; e < kCapRegs; ++e)
                            if (e < ne) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v[e] < mid));
                        if (c >= ns) hi = mid; else lo = mid + 1;
                    }
                    int w = 0;
#pragma unroll
                    for (int e = 0; e < kCapRegs; ++e)
                        if (e < ne) {
                            const bool keep = v[e] < lo;
                            const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
                            const int slot = w + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            if (keep) L[slot] = v[e];   // slot <= position of the element: never overwrites an unread one (all were read)
                            w += __builtin_popcountll(mask);
                        }
                    hacc = w;  // == ns
                }
                // rank = number of smaller indices (LDS broadcast reads), written straight to out[rank]
                int first = 0;
                for (int i0 = 0; i0 < hacc; i0 += 64) {
                    const int i = i0 + lane;
                    const int v = i < hacc ? L[i] : 0x7fffffff;
                    int rank = 0;
#pragma unroll 4
                    for (int j = 0; j < hacc; ++j) rank += (L[j] < v) ? 1 : 0;
                    if (i < hacc && rank < ns) out[rank] = v;
                    const unsigned long long z = __builtin_amdgcn_ballot_w64(i < hacc && rank == 0);
                    if (z != 0ull) first = __builtin_amdgcn_readlane(v, __builtin_ctzll(z));
                }
                for (int l = hacc + lane; l < ns; l += 64) out[l] = first;  // ball_query_gpu.cu:32-36
            } else {
                // dense ball: index-ordered scan with early exit (what ball_query.hip does), straight from the cloud
                const float r2 = a.radius2[sc];
                int cnt = 0, first = 0;
                for (int base = 0; base < n && cnt < ns; base += 64) {
                    const int k = base + lane;
                    const int kc = min(k, n - 1);
                    const float d2 = dist2<FM>(qx - xyz[kc * 3 + 0], qy - xyz[kc * 3 + 1], qz - xyz[kc * 3 + 2]);
                    const bool hit = k < n && d2 < r2;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    if (mask != 0ull) {
                        if (cnt == 0) first = base + __builtin_ctzll(mask);
                        const int slot = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (hit && slot < ns) out[slot] = k;
                        cnt += __builtin_popcountll(mask);
                    }
                }
                for (int l = cnt + lane; l < ns; l += 64) out[l] = first;
            }
        }
    }
}

int grid_build(int b, int n, float rmax, const float *xyz, void *ws, hipStream_t st) {  // also used by ball_query.hip (query sorting)
    const int cmax = grid_cmax(n);
    const size_t lds = ((size_t)cmax + 1) * 4 + 16 * 8 * 4;
    static unsigned long long attr = 0;  // one bit per device
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(ball_grid_build_kernel), 160 * 1024 - 1024, attr, "g4d_ball_grid_build_f32"))
        return rc;
    hipLaunchKernelGGL(ball_grid_build_kernel, dim3(b), dim3(kBuildThreads), lds, st, n, cmax, rmax * kCellSlack, xyz,
                       reinterpret_cast<unsigned char *>(ws), grid_cloud_bytes(n));
    return check_launch("g4d_ball_grid_build_f32");
}

size_t grid_bytes_per_cloud(int n) { return grid_cloud_bytes(n); }
size_t grid_records_offset(int n) { return kGridHdrBytes + ((((size_t)grid_cmax(n) + 1) * 4 + 63) & ~(size_t)63); }

template <int NS, int FM>
static void grid_query_launch(dim3 grid, hipStream_t st, int n, int m, int qpw, const BgArgs &a, const float *new_xyz, const float *xyz,
                              const void *ws) {
    hipLaunchKernelGGL((ball_grid_query_kernel<NS, FM>), grid, dim3(256), 0, st, n, m, qpw, a, new_xyz, xyz,
                       reinterpret_cast<const unsigned char *>(ws), grid_cloud_bytes(n), grid_cmax(n));
}

static int grid_query(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz, const float *xyz,
                      int *const *idx, const void *ws, hipStream_t st) {
    BgArgs a = {};
    for (int s = 0; s < nscales; ++s) {
        a.radius2[s] = radii[s] * radii[s];  // ball_query_gpu.cu:23, rounded once in fp32
        a.nsample[s] = nsamples[s];
        a.idx[s] = idx[s];
    }
    const long long queries = (long long)b * m;
    int qpw = 1;
    while (qpw < 8 && queries / (qpw * 2) >= 32768) qpw <<= 1;  // keep >= 32k waves (4 per SIMD lane of the chip) before batching
    dim3 grid((unsigned)((m + 4 * qpw - 1) / (4 * qpw)), (unsigned)b);
    G4D_WITH_FM(distance_contraction(), switch (nscales) {
        case 1: grid_query_launch<1, FM>(grid, st, n, m, qpw, a, new_xyz, xyz, ws); break;
        case 2: grid_query_launch<2, FM>(grid, st, n, m, qpw, a, new_xyz, xyz, ws); break;
        case 3: grid_query_launch<3, FM>(grid, st, n, m, qpw, a, new_xyz, xyz, ws); break;
        default: grid_query_launch<4, FM>(grid, st, n, m, qpw, a, new_xyz, xyz, ws); break;
    })
    return check_launch("g4d_ball_grid_query_f32");
}

}  // namespace g4d

extern "C" size_t g4d_ball_grid_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (size_t)b * g4d::grid_cloud_bytes(n);
}

extern "C" int g4d_ball_grid_build_f32(int b, int n, float rmax, const float *xyz, void *grid, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && b <= 65535, "g4d_ball_grid_build_f32: bad sizes (b=%d n=%d)", b, n);
    if (b == 0 || n == 0) return G4D_OK;
    G4D_REQUIRE(xyz && grid, "g4d_ball_grid_build_f32: null pointer");
    G4D_REQUIRE(rmax > 0.f && rmax < __builtin_inff(), "g4d_ball_grid_build_f32: rmax must be a positive finite radius");
    return grid_build(b, n, rmax, xyz, grid, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_ball_grid_query_f32(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz,
                                       const float *xyz, int *const *idx, const void *grid, float grid_rmax, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nscales >= 1 && nscales <= 4 && b <= 65535, "g4d_ball_grid_query_f32: bad sizes");
    G4D_REQUIRE(radii && nsamples && idx, "g4d_ball_grid_query_f32: null pointer");
    if (b == 0 || m == 0) return G4D_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int s = 0; s < nscales; ++s) {
        G4D_REQUIRE(nsamples[s] > 0 && idx[s], "g4d_ball_grid_query_f32: bad scale %d", s);
        G4D_REQUIRE(radii[s] <= grid_rmax, "g4d_ball_grid_query_f32: radius %g exceeds the radius the grid was built for (%g)", radii[s], grid_rmax);
        if (n == 0) {
            hipError_t e = hipMemsetAsync(idx[s], 0, sizeof(int) * (size_t)b * m * nsamples[s], st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (n == 0) return G4D_OK;
    G4D_REQUIRE(new_xyz && xyz && grid, "g4d_ball_grid_query_f32: null pointer");
    return grid_query(b, n, m, nscales, radii, nsamples, new_xyz, xyz, idx, grid, st);
}

extern "C" int g4d_ball_query_grid_f32(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz,
                                       const float *xyz, int *const *idx, void *grid, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(nscales >= 1 && nscales <= 4 && radii, "g4d_ball_query_grid_f32: bad scales");
    float rmax = radii[0];
    for (int s = 1; s < nscales; ++s) rmax = radii[s] > rmax ? radii[s] : rmax;
    if (b > 0 && n > 0 && m > 0) {
        const int rc = g4d_ball_grid_build_f32(b, n, rmax, xyz, grid, stream);
        if (rc != G4D_OK) return rc;
    }
    return g4d_ball_grid_query_f32(b, n, m, nscales, radii, nsamples, new_xyz, xyz, idx, grid, rmax, stream);
}
