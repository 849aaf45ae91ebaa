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
#include "ball_grid_build.h"

namespace g4d {

__global__ void __launch_bounds__(kBuildThreads) ball_grid_build_kernel(int n, int cmax, int budget, float cell_req, const float *__restrict__ xyz_all,
                                                                       unsigned char *__restrict__ ws_all, size_t ws_stride) {
    ball_grid_build_body(n, cmax, budget, cell_req, xyz_all, ws_all, ws_stride, blockIdx.x);
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

// ---- three nearest neighbours over the cell grid ------------------------------------------------------------------------------
// three_nn_kernel_fast (/root/reference/modules/pointnet2/pointnet2/src/interpolate_gpu.cu:9-52) scans all m known points per
// unknown point; the 3 nearest of a point on a surface sampled by m points sit within a few sample spacings.  The known points
// are counting-sorted into the finest grid that fits the cell budget (ball_grid_build_kernel, cell_req <= 0); a lane owns one
// unknown point and walks the 3 x 3 x 3 cells around its own cell (9 contiguous record ranges), keeping the 3 smallest
// (d2, ORIGINAL index) pairs in lexicographic order -- the visiting order is not the index order, and the lexicographic top-3 is
// exactly what the reference's ascending scan with strict `<` produces.  Every point outside the cube of radius R cells is
// farther than R cell edges along some axis, so the result is final once the third distance is below (0.99 R edge)^2 (1 % covers
// the rounding of the cell coordinates, 2.4e-4 cells, and of d2); otherwise the lane restarts on the 5 x 5 x 5 cube, then on the
// whole record list -- also the route for a query that is not finite or lies more than a cell outside the box.  Same distance
// expression as the scan (dist2<FM>): bit-identical output for any input.
__device__ __forceinline__ void nn3_insert_lex(float d, int k, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3) {
    const bool lt1 = d < b1 || (d == b1 && k < i1), lt2 = d < b2 || (d == b2 && k < i2), lt3 = d < b3 || (d == b3 && k < i3);
    const float nb3 = lt2 ? b2 : (lt3 ? d : b3);
    const int ni3 = lt2 ? i2 : (lt3 ? k : i3);
    const float nb2 = lt1 ? b1 : (lt2 ? d : b2);
    const int ni2 = lt1 ? i1 : (lt2 ? k : i2);
    b1 = lt1 ? d : b1; i1 = lt1 ? k : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}

// STAGE: the cloud's records and cell table are copied into LDS first (known sets up to 2048 points: 40 KB) -- the walk is a chain
// of dependent reads (cell bounds -> records), ~100 cycles each from LDS against ~1 us from L2.
template <int FM, bool STAGE>
__global__ void __launch_bounds__(256) three_nn_grid_kernel(int n, int m, const float *__restrict__ unknown_all,
                                                           const unsigned char *__restrict__ ws_all, size_t ws_stride, int cmax,
                                                           float *__restrict__ dist2_all, int *__restrict__ idx_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nn_smem[];
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const unsigned char *ws = ws_all + (size_t)b * ws_stride;
    const GridHdr h = *reinterpret_cast<const GridHdr *>(ws);
    const int *cellstart = reinterpret_cast<const int *>(ws + kGridHdrBytes);
    const float4 *rec = reinterpret_cast<const float4 *>(ws + kGridHdrBytes + ((((size_t)cmax + 1) * 4 + 63) & ~(size_t)63));
    if constexpr (STAGE) {
        float4 *srec = reinterpret_cast<float4 *>(nn_smem);
        int *scell = reinterpret_cast<int *>(nn_smem + (size_t)m * 16);
        for (int k = threadIdx.x; k < m; k += 256) srec[k] = rec[k];
        for (int k = threadIdx.x; k <= h.ncells; k += 256) scell[k] = cellstart[k];
        __syncthreads();
        rec = srec;
        cellstart = scell;
    }
    const float *u = unknown_all + ((size_t)b * n + min(p, n - 1)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float INF = __builtin_inff();
    float b1 = INF, b2 = INF, b3 = INF;   // interpolate_gpu.cu:24-25: double 1e40 against a float d == a float compare against +inf
    int i1 = 0, i2 = 0, i3 = 0;
    auto visit = [&](int k0, int k1) {  // two records in flight per step
        int k = k0;
        for (; k + 1 < k1; k += 2) {
            const float4 r0 = rec[k], r1 = rec[k + 1];
            const float d0 = dist2<FM>(ux - r0.x, uy - r0.y, uz - r0.z);   // interpolate_gpu.cu:33 under the contraction contract
            const float d1 = dist2<FM>(ux - r1.x, uy - r1.y, uz - r1.z);
            nn3_insert_lex(d0, __float_as_int(r0.w), b1, b2, b3, i1, i2, i3);
            nn3_insert_lex(d1, __float_as_int(r1.w), b1, b2, b3, i1, i2, i3);
        }
        if (k < k1) {
            const float4 r0 = rec[k];
            nn3_insert_lex(dist2<FM>(ux - r0.x, uy - r0.y, uz - r0.z), __float_as_int(r0.w), b1, b2, b3, i1, i2, i3);
        }
    };
    const float fx = (ux - h.lox) * h.inv_c, fy = (uy - h.loy) * h.inv_c, fz = (uz - h.loz) * h.inv_c;
    // the ring argument needs the query's real cell: finite and at most one cell outside the box (NaN fails every compare)
    bool pending = true;
    const bool local = fx > -1.f && fx < (float)h.gx + 1.f && fy > -1.f && fy < (float)h.gy + 1.f && fz > -1.f && fz < (float)h.gz + 1.f;
    if (local) {
        const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
        const float edge = h.inv_c > 0.f ? 1.0f / h.inv_c : INF;   // inv_c == 0: one cell holds everything
        {   // R = 1: the bounds of the 9 rows first (independent reads), then the records
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.gx - 1);
            int rs[9], re[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int y = cy + (q % 3) - 1, z = cz + (q / 3) - 1;
                const bool ok = x0 <= x1 && y >= 0 && y < h.gy && z >= 0 && z < h.gz;
                const int row = ok ? (z * h.gy + y) * h.gx : 0;
                rs[q] = ok ? cellstart[row + x0] : 0;
                re[q] = ok ? cellstart[row + x1 + 1] : 0;
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) visit(rs[q], re[q]);
            const float reach = edge * 0.99f;
            pending = !(b3 < reach * reach);
        }
        if (pending) {  // R = 2 (rare: sparse neighbourhood): the whole 5 x 5 x 5 cube from scratch
            b1 = b2 = b3 = INF; i1 = i2 = i3 = 0;
            const int x0 = max(cx - 2, 0), x1 = min(cx + 2, h.gx - 1);
            if (x0 <= x1) {
                for (int dz = -2; dz <= 2; ++dz) {
                    const int z = cz + dz;
                    if (z < 0 || z >= h.gz) continue;
                    for (int dy = -2; dy <= 2; ++dy) {
                        const int y = cy + dy;
                        if (y < 0 || y >= h.gy) continue;
                        const int row = (z * h.gy + y) * h.gx;
                        visit(cellstart[row + x0], cellstart[row + x1 + 1]);
                    }
                }
            }
            const float reach = 2.f * edge * 0.99f;
            pending = !(b3 < reach * reach);
        }
    }
    if (pending) {  // exact for anything: every record, lexicographic inserts
        b1 = b2 = b3 = INF; i1 = i2 = i3 = 0;
        visit(0, m);
    }
    if (p < n) {
        float *d2 = dist2_all + ((size_t)b * n + p) * 3;
        int *ix = idx_all + ((size_t)b * n + p) * 3;
        d2[0] = b1; d2[1] = b2; d2[2] = b3;
        ix[0] = i1; ix[1] = i2; ix[2] = i3;
    }
}

void grid_sorted_layout(int n, size_t *offset_bytes, size_t *stride_bytes) {
    *offset_bytes = kGridHdrBytes + ((((size_t)grid_cmax(n) + 1) * 4 + 63) & ~(size_t)63);
    *stride_bytes = grid_cloud_bytes(n);
}

int grid_build(int b, int n, float rmax, const float *xyz, void *ws, hipStream_t st, int budget) {  // also used by ball_query.hip (query sorting)
    const int cmax = grid_cmax(n);
    if (budget <= 0 || budget > cmax) budget = cmax;
    const size_t lds = ((size_t)cmax + 1) * 4 + 16 * 8 * 4;
    static unsigned long long attr = 0;  // one bit per device
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(ball_grid_build_kernel), 160 * 1024 - 1024, attr, "g4d_ball_grid_build_f32"))
        return rc;
    hipLaunchKernelGGL(ball_grid_build_kernel, dim3(b), dim3(kBuildThreads), lds, st, n, cmax, budget, rmax > 0.f ? rmax * kCellSlack : 0.f, xyz,
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
    static const int qpw_env = getenv("G4D_BG_QPW") ? atoi(getenv("G4D_BG_QPW")) : 0;   // tuning hook
    if (qpw_env > 0) qpw = qpw_env;
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

extern "C" int g4d_three_nn_grid_f32(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, void *grid,
                                     g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && b <= 65535, "g4d_three_nn_grid_f32: bad sizes (b=%d n=%d m=%d)", b, n, m);
    if ((long long)b * n == 0) return G4D_OK;
    G4D_REQUIRE(unknown && dist2 && idx, "g4d_three_nn_grid_f32: null pointer");
    if (m == 0) return g4d_three_nn_f32(b, n, m, unknown, known, dist2, idx, stream);
    G4D_REQUIRE(known && grid, "g4d_three_nn_grid_f32: null pointer (grid scratch = g4d_ball_grid_bytes(b, m))");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // Cell budget = one cell per known point of the box volume (the finest grid the workspace holds).  Measured (MI355X, device time,
    // build + search, scripts/time_three_nn.py): coarser cells only lose -- a lane's walk costs ~10x the scan's per candidate (its
    // LDS reads are scattered, the scan's are wave broadcasts with a wave-uniform skip of the inserts), so the win has to come from
    // visiting few candidates: 30 x 4096 <- 6890 (config 4's interpenetration search) 334 -> 205 us, 8 x 8192 <- 1024 43 -> 40-79 us
    // (surface / volume cloud), 8 x 1024 <- 256 8 -> 55 us.  Callers therefore take this route from m = 4096 on.
    static const int per_cell = [] { const char *e = getenv("G4D_NN_PER_CELL"); return e && atoi(e) > 0 ? atoi(e) : 1; }();
    if (const int rc = grid_build(b, m, 0.f, known, grid, st, m / per_cell > 8 ? m / per_cell : 8)) return rc;
    dim3 g((unsigned)((n + 255) / 256), (unsigned)b);
    const unsigned char *wsb = reinterpret_cast<const unsigned char *>(grid);
    if (m <= 2048) {
        const size_t lds = (size_t)m * 16 + ((size_t)grid_cmax(m) + 1) * 4;   // <= 40 KB
        G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_grid_kernel<FM, true>), g, dim3(256), lds, st, n, m, unknown, wsb,
                                                              grid_cloud_bytes(m), grid_cmax(m), dist2, idx))
    } else {
        G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_grid_kernel<FM, false>), g, dim3(256), 0, st, n, m, unknown, wsb,
                                                              grid_cloud_bytes(m), grid_cmax(m), dist2, idx))
    }
    return check_launch("g4d_three_nn_grid_f32");
}
