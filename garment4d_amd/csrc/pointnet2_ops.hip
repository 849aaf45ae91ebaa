// gather / group / three_nn / three_interpolate (+ the three scatter-add backward kernels) for gfx950.
// Replaces /root/reference/modules/pointnet2/pointnet2/src/{sampling_gpu.cu:8-83, group_points_gpu.cu:8-86,
// interpolate_gpu.cu:9-161}.  All HBM/L2-bound copies or scans: coalesced along the point axis, the index
// row is read ONCE per thread and reused for a chunk of channels (the reference re-reads it per channel),
// 64-bit offsets (the reference's int32 offsets wrap at 2^31 elements).
#include <cstdlib>

#include "g4d_common.h"
#include "three_nn_body.h"

namespace g4d {

constexpr int kCT = 8;  // channels per thread in the copy kernels

// out[b,c,e] = points[b,c,idx[b,e]]  (e over npoints*nsample; gather is nsample == 1)
__global__ void __launch_bounds__(256) group_kernel(int c, int n, long long e_total, const float *__restrict__ points,
                                                   const int *__restrict__ idx, float *__restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= e_total) return;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * kCT;
    const int k = idx[(size_t)b * e_total + e];
    const float *src = points + ((size_t)b * c + c0) * n + k;
    float *dst = out + ((size_t)b * c + c0) * e_total + e;
    const int cn = min(kCT, c - c0);
    float v[kCT];
#pragma unroll
    for (int i = 0; i < kCT; ++i)
        if (i < cn) v[i] = src[(size_t)i * n];
#pragma unroll
    for (int i = 0; i < kCT; ++i)
        if (i < cn) dst[(size_t)i * e_total] = v[i];
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e]
__global__ void __launch_bounds__(256) group_grad_kernel(int c, int n, long long e_total, const float *__restrict__ grad_out,
                                                        const int *__restrict__ idx, float *__restrict__ grad_points) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= e_total) return;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * kCT;
    const int k = idx[(size_t)b * e_total + e];
    float *dst = grad_points + ((size_t)b * c + c0) * n + k;
    const float *src = grad_out + ((size_t)b * c + c0) * e_total + e;
    const int cn = min(kCT, c - c0);
#pragma unroll
    for (int i = 0; i < kCT; ++i)
        if (i < cn) atomicAdd(dst + (size_t)i * n, src[(size_t)i * e_total]);
}

template <int FM>
__global__ void __launch_bounds__(256) three_nn_kernel(int n, int m, const float *__restrict__ unknown_all,
                                                      const float *__restrict__ known_all, float *__restrict__ dist2_all,
                                                      int *__restrict__ idx_all) {
    three_nn_body<FM>(n, m, unknown_all, known_all, dist2_all, idx_all, blockIdx.x, blockIdx.y);
}

template <int FM>
__global__ void __launch_bounds__(256) three_nn_multi_kernel(const NNMulti q) {
    three_nn_multi_role<FM>(q, blockIdx.x, blockIdx.y);
}

// Variant for large unknown sets (n >= 4096: the last feature-propagation level): a wave owns 64 unknown points and scans ALL known
// points in ascending index order (no slices, no merge -- the reference's own scan order, so plain strict `<` inserts), and every one of
// the 4 candidates of a step has its OWN wave-uniform test.  Whether the ordered insert runs is decided per wave: with a running third
// distance over K scanned points a candidate improves a lane with probability 3 / K, i.e. some lane of the wave with ~192 / K.  The
// sliced kernel restarts K at every slice (256 points: an insert is needed at nearly every step, and all 4 candidates went through
// it together: ~60 of the ~85 instructions per step); scanning the whole set lets K grow to m and the per-candidate tests skip
// half of the inserts.
// `qrec` (optional): the unknown cloud as (x, y, z, original index) records in CELL ORDER -- the `sorted` array of its ball-grid
// workspace (ball_grid.hip), one array of n records per cloud at stride qstride bytes.  The 64 queries of a wave are then neighbours:
// a known point is close to all of them or to none, so the wave-uniform tests around the inserts -- which pass when ANY lane improves
// -- fail for most candidates instead of passing for most of them (64 unrelated queries x 4 candidates: ~80 % at K = 512).
// SPLIT = 2 | 4: the workgroup owns 256 / SPLIT queries instead of 256 and its waves scan one contiguous 1/SPLIT of every staged chunk each;
// the SPLIT top-3 lists are merged through LDS under (distance, index) -- the scan's own result, because a strict `<` scan in
// index order keeps, among equal distances, the lowest indices.  4096 waves instead of 1024 on the last FP level (one per SIMD before).
template <int FM, int SPLIT>
__global__ void __launch_bounds__(256) three_nn_wide_kernel(int n, int m, const float *__restrict__ unknown_all,
                                                           const float *__restrict__ known_all, float *__restrict__ dist2_all,
                                                           int *__restrict__ idx_all, const unsigned char *__restrict__ qrec, size_t qstride,
                                                           int sorted_out) {
    __shared__ __attribute__((aligned(16))) float skx[kNNChunk], sky[kNNChunk], skz[kNNChunk];
    constexpr int QPB = 256 / SPLIT;   // queries per workgroup
    __shared__ float sd[SPLIT > 1 ? SPLIT - 1 : 1][SPLIT > 1 ? QPB : 1][3];   // the slices' top-3 lists (none un-split)
    __shared__ int si[SPLIT > 1 ? SPLIT - 1 : 1][SPLIT > 1 ? QPB : 1][3];
    const int b = blockIdx.y;
    const int slice = (int)threadIdx.x / QPB, ql = (int)threadIdx.x % QPB;   // wave-uniform slice (QPB is a multiple of 64)
    int p = (int)blockIdx.x * QPB + ql;
    const float *known = known_all + (size_t)b * m * 3;
    float ux, uy, uz;
    int orig = p;
    if (qrec) {
        const float4 r = reinterpret_cast<const float4 *>(qrec + (size_t)b * qstride)[min(p, n - 1)];
        ux = r.x; uy = r.y; uz = r.z; orig = sorted_out ? p : __float_as_int(r.w);   // sorted_out: results stay in cell order
    } else {
        const float *u = unknown_all + ((size_t)b * n + min(p, n - 1)) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += kNNChunk) {
        const int cm = min(kNNChunk, m - base);
        __syncthreads();
        for (int j = threadIdx.x; j < kNNChunk; j += 256) {  // beyond the cloud: +inf coordinates, d = +inf, never inserted
            const bool ok = j < cm;
            const float *kp = known + (size_t)(base + (ok ? j : 0)) * 3;
            const float inf = __builtin_inff();
            skx[j] = ok ? kp[0] : inf; sky[j] = ok ? kp[1] : inf; skz[j] = ok ? kp[2] : inf;
        }
        __syncthreads();
        const int jn = (cm + 3) & ~3;   // block-uniform trip count
        const int qlen = SPLIT > 1 ? (((jn / SPLIT) + 3) & ~3) : jn;   // this wave's share of the chunk (wave-uniform bounds)
        const int jlo = slice * qlen, jhi = min(jlo + qlen, jn);
        for (int j = jlo; j < jhi; j += 4) {
            const float4 kx = *reinterpret_cast<const float4 *>(&skx[j]), ky = *reinterpret_cast<const float4 *>(&sky[j]),
                         kz = *reinterpret_cast<const float4 *>(&skz[j]);
            const float d0 = dist2<FM>(ux - kx.x, uy - ky.x, uz - kz.x);   // interpolate_gpu.cu:33 under the contraction contract
            const float d1 = dist2<FM>(ux - kx.y, uy - ky.y, uz - kz.y);
            const float d2 = dist2<FM>(ux - kx.z, uy - ky.z, uz - kz.z);
            const float d3 = dist2<FM>(ux - kx.w, uy - ky.w, uz - kz.w);
            if (__builtin_amdgcn_ballot_w64(fminf(fminf(d0, d1), fminf(d2, d3)) < b3) == 0ull) continue;
            if (__builtin_amdgcn_ballot_w64(d0 < b3) != 0ull) nn_insert(d0, base + j, b1, b2, b3, i1, i2, i3);
            if (__builtin_amdgcn_ballot_w64(d1 < b3) != 0ull) nn_insert(d1, base + j + 1, b1, b2, b3, i1, i2, i3);
            if (__builtin_amdgcn_ballot_w64(d2 < b3) != 0ull) nn_insert(d2, base + j + 2, b1, b2, b3, i1, i2, i3);
            if (__builtin_amdgcn_ballot_w64(d3 < b3) != 0ull) nn_insert(d3, base + j + 3, b1, b2, b3, i1, i2, i3);
        }
    }
    if constexpr (SPLIT > 1) {
        if (slice > 0) {
            sd[slice - 1][ql][0] = b1; sd[slice - 1][ql][1] = b2; sd[slice - 1][ql][2] = b3;
            si[slice - 1][ql][0] = i1; si[slice - 1][ql][1] = i2; si[slice - 1][ql][2] = i3;
        }
        __syncthreads();
        if (slice > 0) return;
#pragma unroll
        for (int w = 0; w < SPLIT - 1; ++w)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = sd[w][ql][c];
                if (d < __builtin_inff()) nn_insert_lex(d, si[w][ql][c], b1, b2, b3, i1, i2, i3);
            }
    }
    if (p < n) {
        float *d2o = dist2_all + ((size_t)b * n + orig) * 3;
        int *ix = idx_all + ((size_t)b * n + orig) * 3;
        d2o[0] = b1; d2o[1] = b2; d2o[2] = b3;
        ix[0] = i1; ix[1] = i2; ix[2] = i3;
    }
}

// Measured on cfg2's last FP level (8 x 8192 queries <- 1024 known, cell-ordered): 49.3 / 29.2 / 24.6 us alone for SPLIT = 1 / 2 / 4 and a
// single batch's latency 1.073 / 1.053 / 1.041 ms -- but the 20-batch mix drops from 34.7k to 34.5k / 34.1k frames/s: the split scans cost
// more wave time in total (every slice restarts its running third distance), and wave time is what the mix is short of.  So: split only
// when the launch would leave SIMDs empty (fewer than 1024 waves un-split); G4D_NN_SPLIT = 1 | 2 | 4 overrides.
static int nn_wide_split(long long queries) {
    static const int v = [] { const char *e = getenv("G4D_NN_SPLIT"); return e ? atoi(e) : 0; }();
    if (v == 1 || v == 2 || v == 4) return v;
    return queries <= 65536 ? 4 : (queries <= 131072 ? 2 : 1);   // (round 4: throughput comes from coalesced calls of >= 1 M queries, un-split; a lone B = 8 step -- 65536 queries, one wave per SIMD un-split -- is a latency case)
}
#define G4D_NN_WIDE_LAUNCH(n_, b_, st_, ...)                                                                                                  \
    if (nn_wide_split((long long)(n_) * (b_)) == 4) {                                                                                                               \
        dim3 gridw(((n_) + 63) / 64, (b_));                                                                                                   \
        G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_wide_kernel<FM, 4>), gridw, dim3(256), 0, st_, __VA_ARGS__))         \
    } else if (nn_wide_split((long long)(n_) * (b_)) == 2) {                                                                                                        \
        dim3 gridw(((n_) + 127) / 128, (b_));                                                                                                 \
        G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_wide_kernel<FM, 2>), gridw, dim3(256), 0, st_, __VA_ARGS__))         \
    } else {                                                                                                                                  \
        dim3 gridw(((n_) + 255) / 256, (b_));                                                                                                 \
        G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_wide_kernel<FM, 1>), gridw, dim3(256), 0, st_, __VA_ARGS__))         \
    }

// out[b,c,p] = w0*pts[b,c,i0] + w1*pts[b,c,i1] + w2*pts[b,c,i2]   (left-to-right, unfused)
__global__ void __launch_bounds__(256) three_interp_kernel(int c, int m, int n, const float *__restrict__ points,
                                                          const int *__restrict__ idx, const float *__restrict__ weight,
                                                          float *__restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * kCT;
    const int *ix = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *src = points + ((size_t)b * c + c0) * m;
    float *dst = out + ((size_t)b * c + c0) * n + p;
    const int cn = min(kCT, c - c0);
#pragma unroll
    for (int i = 0; i < kCT; ++i)
        if (i < cn) {
            const float *row = src + (size_t)i * m;
            dst[(size_t)i * n] = w0 * row[i0] + w1 * row[i1] + w2 * row[i2];
        }
}

__global__ void __launch_bounds__(256) three_interp_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                                               const int *__restrict__ idx, const float *__restrict__ weight,
                                                               float *__restrict__ grad_points) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * kCT;
    const int *ix = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *src = grad_out + ((size_t)b * c + c0) * n + p;
    float *dst = grad_points + ((size_t)b * c + c0) * m;
    const int cn = min(kCT, c - c0);
#pragma unroll
    for (int i = 0; i < kCT; ++i)
        if (i < cn) {
            const float g = src[(size_t)i * n];
            float *row = dst + (size_t)i * m;
            atomicAdd(row + i0, g * w0);
            atomicAdd(row + i1, g * w1);
            atomicAdd(row + i2, g * w2);
        }
}

// point-major row gather: out[b,j,:] = in[b,idx[b,j],:]  (rows of `c` floats; used for new_xyz = xyz[fps idx])
__global__ void __launch_bounds__(256) gather_rows_kernel(int n, int m, int c, const float *__restrict__ in,
                                                         const int *__restrict__ idx, float *__restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // over m*c
    if (e >= (long long)m * c) return;
    const int b = blockIdx.y;
    const int j = (int)(e / c), ch = (int)(e - (long long)j * c);
    out[((size_t)b * m + j) * c + ch] = in[((size_t)b * n + idx[(size_t)b * m + j]) * c + ch];
}

// point-major three_interpolate + skip concat (PointnetFPModule.forward, pointnet2_modules.py:139-149):
// out[b,p,:] = [ sum_i w_i known[b,idx_i,:] (C2) | skip[b,p,:] (C1) ],  w_i = normalised 1/(sqrt(dist2_i)+1e-8).
// One thread per (row, 4 channels): the three source rows are contiguous C2-float vectors.
__global__ void __launch_bounds__(256) interp_concat_kernel(long long rows, int n, int m, int C2, int C1,
                                                           const float *__restrict__ known, const float *__restrict__ skip,
                                                           const float *__restrict__ dist2, const int *__restrict__ nn_idx,
                                                           float *__restrict__ out) {
    const int K = C2 + C1;
    const int per_row = (K + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * per_row) return;
    const long long row = gid / per_row;
    const int k0 = (int)(gid - row * per_row) * 4;
    const int b = (int)(row / n);
    const int *ix = nn_idx + row * 3;
    const float *d2 = dist2 + row * 3;
    const float r0 = 1.0f / (__fsqrt_rn(d2[0]) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(d2[1]) + 1e-8f), r2 = 1.0f / (__fsqrt_rn(d2[2]) + 1e-8f);
    const float norm = (r0 + r1) + r2;
    const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
    const float *a0 = known + ((size_t)b * m + ix[0]) * C2, *a1 = known + ((size_t)b * m + ix[1]) * C2,
                *a2 = known + ((size_t)b * m + ix[2]) * C2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k >= K) break;
        out[row * K + k] = k < C2 ? (w0 * a0[k] + w1 * a1[k] + w2 * a2[k]) : skip[row * C1 + (k - C2)];
    }
}

// Batched SpMM of GraphConvolution.forward (modules/pygcn/layers.py:44-47): out[f,v,:] = sum_u Ahat[v,u] * S[f,u,:] (+ bias).
// The reference folds the batch into columns (transpose, reshape, torch.spmm, reshape, transpose); here one thread owns
// 4 consecutive channels of one (frame, vertex) row: the ~5 neighbour rows are read as contiguous float4 segments
// (lanes = consecutive channels -> coalesced) and nothing is transposed.
__global__ void __launch_bounds__(256) spmm_rows_kernel(long long rows, int vg, int c, const float *__restrict__ S,
                                                       const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                       const float *__restrict__ vals, const float *__restrict__ bias,
                                                       int relu, float *__restrict__ out) {
    const int per_row = (c + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * per_row) return;
    const long long row = gid / per_row;
    const int c0 = (int)(gid - row * per_row) * 4;
    const long long f = row / vg;
    const int v = (int)(row - f * vg);
    const int beg = rowptr[v], end = rowptr[v + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = (c0 + 3 < c) && ((c & 3) == 0);
    for (int e = beg; e < end; ++e) {
        const float a = vals[e];
        const float *src = S + ((size_t)f * vg + colidx[e]) * c + c0;
        if (full) {
            const float4 x = *reinterpret_cast<const float4 *>(src);
            acc[0] = fmaf(a, x.x, acc[0]); acc[1] = fmaf(a, x.y, acc[1]); acc[2] = fmaf(a, x.z, acc[2]); acc[3] = fmaf(a, x.w, acc[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < c) acc[j] = fmaf(a, src[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < c) {
            const float y = acc[j] + (bias ? bias[c0 + j] : 0.f);
            out[(size_t)row * c + c0 + j] = relu ? fmaxf(y, 0.f) : y;
        }
}

static inline int launch_group(int b, int c, int n, long long e_total, const float *points, const int *idx, float *out,
                               hipStream_t s, const char *what) {
    if (b == 0 || c == 0 || e_total == 0) return G4D_OK;
    dim3 grid((unsigned)((e_total + 255) / 256), (c + kCT - 1) / kCT, b);
    hipLaunchKernelGGL(group_kernel, grid, dim3(256), 0, s, c, n, e_total, points, idx, out);
    return check_launch(what);
}

static inline int launch_group_grad(int b, int c, int n, long long e_total, const float *grad_out, const int *idx,
                                    float *grad_points, hipStream_t s, const char *what) {
    if (b == 0 || c == 0 || e_total == 0) return G4D_OK;
    dim3 grid((unsigned)((e_total + 255) / 256), (c + kCT - 1) / kCT, b);
    hipLaunchKernelGGL(group_grad_kernel, grid, dim3(256), 0, s, c, n, e_total, grad_out, idx, grad_points);
    return check_launch(what);
}

}  // namespace g4d

using namespace g4d;
#define G4D_STREAM(s) reinterpret_cast<hipStream_t>(s)
#define G4D_DIMS_OK(name, ...)                                         \
    do {                                                               \
        const long long dims_[] = {__VA_ARGS__};                       \
        for (long long d_ : dims_) G4D_REQUIRE(d_ >= 0, name ": negative size"); \
    } while (0)

extern "C" int g4d_gather_f32(int b, int c, int n, int m, const float *points, const int *idx, float *out,
                              g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_gather_f32", b, c, n, m);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_gather_f32: b or c too large for the grid");
    if ((long long)b * c * m == 0) return G4D_OK;
    G4D_REQUIRE(points && idx && out, "g4d_gather_f32: null pointer");
    return launch_group(b, c, n, m, points, idx, out, G4D_STREAM(stream), "g4d_gather_f32");
}

extern "C" int g4d_gather_grad_f32(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points,
                                   g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_gather_grad_f32", b, c, n, m);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_gather_grad_f32: b or c too large for the grid");
    if ((long long)b * c * m == 0) return G4D_OK;
    G4D_REQUIRE(grad_out && idx && grad_points, "g4d_gather_grad_f32: null pointer");
    return launch_group_grad(b, c, n, m, grad_out, idx, grad_points, G4D_STREAM(stream), "g4d_gather_grad_f32");
}

extern "C" int g4d_group_f32(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                             g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_group_f32", b, c, n, npoints, nsample);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_group_f32: b or c too large for the grid");
    if ((long long)b * c * npoints * nsample == 0) return G4D_OK;
    G4D_REQUIRE(points && idx && out, "g4d_group_f32: null pointer");
    return launch_group(b, c, n, (long long)npoints * nsample, points, idx, out, G4D_STREAM(stream), "g4d_group_f32");
}

extern "C" int g4d_group_grad_f32(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                  float *grad_points, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_group_grad_f32", b, c, n, npoints, nsample);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_group_grad_f32: b or c too large for the grid");
    if ((long long)b * c * npoints * nsample == 0) return G4D_OK;
    G4D_REQUIRE(grad_out && idx && grad_points, "g4d_group_grad_f32: null pointer");
    return launch_group_grad(b, c, n, (long long)npoints * nsample, grad_out, idx, grad_points, G4D_STREAM(stream),
                             "g4d_group_grad_f32");
}

extern "C" int g4d_three_nn_f32(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_three_nn_f32", b, n, m);
    G4D_REQUIRE(b <= 65535, "g4d_three_nn_f32: b > 65535 not supported");
    if ((long long)b * n == 0) return G4D_OK;
    G4D_REQUIRE(unknown && dist2 && idx && (known || m == 0), "g4d_three_nn_f32: null pointer");
    static const int wide_min_n = [] { const char *e = getenv("G4D_NN_WIDE_MIN_N"); return e ? atoi(e) : 4096; }();
    if (n >= wide_min_n && m >= 256) {   // 64 queries per wave over the whole known set: see three_nn_wide_kernel
        G4D_NN_WIDE_LAUNCH(n, b, G4D_STREAM(stream), n, m, unknown, known, dist2, idx, (const unsigned char *)nullptr, (size_t)0, 0)
        return check_launch("g4d_three_nn_f32");
    }
    dim3 grid((n + 63) / 64, b);
    G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL(three_nn_kernel<FM>, grid, dim3(256), 0, G4D_STREAM(stream), n, m, unknown, known, dist2, idx))
    return check_launch("g4d_three_nn_f32");
}

// Up to four three_nn problems of the same batch size in one launch (three_nn_multi_kernel); each result identical to g4d_three_nn_f32.
extern "C" int g4d_three_nn_multi_f32(int b, int count, const int *n, const int *m, const float *const *unknown, const float *const *known,
                                      float *const *dist2, int *const *idx, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && b <= 65535 && count >= 1 && count <= 4 && n && m && unknown && known && dist2 && idx, "g4d_three_nn_multi_f32: bad arguments (1..4 problems)");
    if (b == 0) return G4D_OK;
    NNMulti q = {};
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        G4D_REQUIRE(n[i] > 0 && m[i] > 0 && unknown[i] && known[i] && dist2[i] && idx[i], "g4d_three_nn_multi_f32: problem %d: empty or null", i);
        q.n[i] = n[i]; q.m[i] = m[i]; q.unknown[i] = unknown[i]; q.known[i] = known[i]; q.dist2[i] = dist2[i]; q.idx[i] = idx[i];
        blocks += (n[i] + 63) / 64;
        q.blk_end[i] = blocks;
    }
    q.count = count;
    dim3 grid((unsigned)blocks, (unsigned)b);
    G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL(three_nn_multi_kernel<FM>, grid, dim3(256), 0, G4D_STREAM(stream), q))
    return check_launch("g4d_three_nn_multi_f32");
}

// three_nn with the ball-grid workspace of the UNKNOWN cloud at hand (g4d_ball_grid_build_f32 on `unknown`, any radius): the same scan,
// the queries taken in the workspace's cell order (see three_nn_wide_kernel).  Output identical to g4d_three_nn_f32.
extern "C" int g4d_three_nn_cells_f32(int b, int n, int m, const float *unknown, const void *unknown_grid, const float *known, float *dist2,
                                      int *idx, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_three_nn_cells_f32", b, n, m);
    G4D_REQUIRE(b <= 65535, "g4d_three_nn_cells_f32: b > 65535 not supported");
    if ((long long)b * n == 0) return G4D_OK;
    if (!unknown_grid || m == 0) return g4d_three_nn_f32(b, n, m, unknown, known, dist2, idx, stream);
    G4D_REQUIRE(unknown && dist2 && idx && known, "g4d_three_nn_cells_f32: null pointer");
    size_t off = 0, stride = 0;
    grid_sorted_layout(n, &off, &stride);
    G4D_NN_WIDE_LAUNCH(n, b, G4D_STREAM(stream), n, m, unknown, known, dist2, idx, reinterpret_cast<const unsigned char *>(unknown_grid) + off, stride, 0)
    return check_launch("g4d_three_nn_cells_f32");
}

// ... with the results LEFT in cell order: dist2 / idx row p of cloud b belong to the b-th cloud's p-th grid record (whose 4th dword is
// the point's original index).  For consumers that walk the points in that order (g4d_mlp_chain_table_cells_f32).
extern "C" int g4d_three_nn_cells_sorted_f32(int b, int n, int m, const void *unknown_grid, const float *known, float *dist2, int *idx,
                                             g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_three_nn_cells_sorted_f32", b, n, m);
    G4D_REQUIRE(b <= 65535, "g4d_three_nn_cells_sorted_f32: b > 65535 not supported");
    if ((long long)b * n == 0) return G4D_OK;
    G4D_REQUIRE(unknown_grid && dist2 && idx && known && m > 0, "g4d_three_nn_cells_sorted_f32: null pointer / empty known set");
    size_t off = 0, stride = 0;
    grid_sorted_layout(n, &off, &stride);
    G4D_NN_WIDE_LAUNCH(n, b, G4D_STREAM(stream), n, m, (const float *)nullptr, known, dist2, idx, reinterpret_cast<const unsigned char *>(unknown_grid) + off, stride, 1)
    return check_launch("g4d_three_nn_cells_sorted_f32");
}

extern "C" int g4d_three_interp_f32(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                    float *out, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_three_interp_f32", b, c, m, n);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_three_interp_f32: b or c too large for the grid");
    if ((long long)b * c * n == 0) return G4D_OK;
    G4D_REQUIRE(points && idx && weight && out, "g4d_three_interp_f32: null pointer");
    dim3 grid((n + 255) / 256, (c + kCT - 1) / kCT, b);
    hipLaunchKernelGGL(three_interp_kernel, grid, dim3(256), 0, G4D_STREAM(stream), c, m, n, points, idx, weight, out);
    return check_launch("g4d_three_interp_f32");
}

extern "C" int g4d_three_interp_grad_f32(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                         const float *weight, float *grad_points, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_three_interp_grad_f32", b, c, n, m);
    G4D_REQUIRE(b <= 65535 && (c + kCT - 1) / kCT <= 65535, "g4d_three_interp_grad_f32: b or c too large for the grid");
    if ((long long)b * c * n == 0) return G4D_OK;
    G4D_REQUIRE(grad_out && idx && weight && grad_points, "g4d_three_interp_grad_f32: null pointer");
    dim3 grid((n + 255) / 256, (c + kCT - 1) / kCT, b);
    hipLaunchKernelGGL(three_interp_grad_kernel, grid, dim3(256), 0, G4D_STREAM(stream), c, n, m, grad_out, idx, weight,
                       grad_points);
    return check_launch("g4d_three_interp_grad_f32");
}

extern "C" int g4d_gather_rows_f32(int b, int n, int m, int c, const float *in, const int *idx, float *out,
                                   g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_gather_rows_f32", b, n, m, c);
    G4D_REQUIRE(b <= 65535, "g4d_gather_rows_f32: b > 65535 not supported");
    if ((long long)b * m * c == 0) return G4D_OK;
    G4D_REQUIRE(in && idx && out, "g4d_gather_rows_f32: null pointer");
    dim3 grid((unsigned)(((long long)m * c + 255) / 256), b);
    hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, G4D_STREAM(stream), n, m, c, in, idx, out);
    return check_launch("g4d_gather_rows_f32");
}

extern "C" int g4d_interp_concat_f32(int b, int n, int m, int c2, int c1, const float *known_feats, const float *skip,
                                     const float *dist2, const int *nn_idx, float *out, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_interp_concat_f32", b, n, m, c2, c1);
    const long long rows = (long long)b * n;
    if (rows == 0 || c2 + c1 == 0) return G4D_OK;
    G4D_REQUIRE(known_feats && dist2 && nn_idx && out && (c1 == 0 || skip), "g4d_interp_concat_f32: null pointer");
    const long long work = rows * ((c2 + c1 + 3) / 4);
    G4D_REQUIRE((work + 255) / 256 < (1ll << 31), "g4d_interp_concat_f32: too large");
    hipLaunchKernelGGL(interp_concat_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, G4D_STREAM(stream), rows, n, m, c2, c1,
                       known_feats, skip, dist2, nn_idx, out);
    return check_launch("g4d_interp_concat_f32");
}

extern "C" int g4d_spmm_rows_f32(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx,
                                 const float *vals, const float *bias, int relu, float *out, g4d_stream_t stream) {
    G4D_DIMS_OK("g4d_spmm_rows_f32", frames, vg, c);
    const long long rows = (long long)frames * vg;
    if (rows == 0 || c == 0) return G4D_OK;
    G4D_REQUIRE(S && rowptr && colidx && vals && out, "g4d_spmm_rows_f32: null pointer");
    const long long work = rows * ((c + 3) / 4);
    G4D_REQUIRE((work + 255) / 256 < (1ll << 31), "g4d_spmm_rows_f32: too large");
    hipLaunchKernelGGL(spmm_rows_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, G4D_STREAM(stream), rows, vg, c, S, rowptr,
                       colidx, vals, bias, relu, out);
    return check_launch("g4d_spmm_rows_f32");
}
