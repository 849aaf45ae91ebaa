// three_nn building blocks shared by pointnet2_ops.hip (g4d_three_nn_f32 / _multi_f32) and ball_query.hip (g4d_search_multi_f32: the small
// ball queries and three-NN searches of a step in one launch).
#pragma once
#include "g4d_common.h"

namespace g4d {

// three nearest known points per unknown point.  The reference runs one thread per unknown point scanning all m
// known points; that leaves one wave per SIMD and a long serial loop.  Here a workgroup = 64 unknown points x 4
// slices of the known set: the known points are staged through LDS in 16-byte slots (wave-broadcast ds_read_b128),
// each wave keeps the 3 smallest (d, index) of its slice (ascending index, strict `<` = the reference's cascade),
// the top-3 insertion only runs when some lane needs it, and wave 0 merges the four partial lists.  The result is
// the 3 smallest under (d, index) lexicographic order -- exactly what the sequential scan produces.
constexpr int kNNChunk = 1024;
__device__ __forceinline__ void nn_insert(float d, int k, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3) {
    const bool lt1 = d < b1, lt2 = d < b2, lt3 = d < b3;  // cascade of interpolate_gpu.cu:31-42, branch-free
    const float nb3 = lt2 ? b2 : (lt3 ? d : b3);
    const int ni3 = lt2 ? i2 : (lt3 ? k : i3);
    const float nb2 = lt1 ? b1 : (lt2 ? d : b2);
    const int ni2 = lt1 ? i1 : (lt2 ? k : i2);
    b1 = lt1 ? d : b1; i1 = lt1 ? k : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}
// merge-time insertion: ties between slices resolve to the lower index
__device__ __forceinline__ void nn_insert_lex(float d, int k, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3) {
    const bool lt1 = d < b1 || (d == b1 && k < i1), lt2 = d < b2 || (d == b2 && k < i2), lt3 = d < b3 || (d == b3 && k < i3);
    const float nb3 = lt2 ? b2 : (lt3 ? d : b3);
    const int ni3 = lt2 ? i2 : (lt3 ? k : i3);
    const float nb2 = lt1 ? b1 : (lt2 ? d : b2);
    const int ni2 = lt1 ? i1 : (lt2 ? k : i2);
    b1 = lt1 ? d : b1; i1 = lt1 ? k : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}

template <int FM>
__device__ __forceinline__ void three_nn_body(int n, int m, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                                              float *__restrict__ dist2_all, int *__restrict__ idx_all, int bx, int b) {
    __shared__ __attribute__((aligned(16))) float skx[kNNChunk], sky[kNNChunk], skz[kNNChunk];  // SoA: 4 points per ds_read_b128
    __shared__ float sd[3][64][3];
    __shared__ int si[3][64][3];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int p = bx * 64 + lane;
    const float *known = known_all + (size_t)b * m * 3;
    const int pc = min(p, n - 1);
    const float *u = unknown_all + ((size_t)b * n + pc) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    // interpolate_gpu.cu:24-25: double best = 1e40 compared against a float d == float compare against +inf
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
        // each of the 4 waves scans a quarter of the chunk (multiple of 4 points), 4 points per step: three 16-byte LDS
        // broadcasts, four distances, ONE wave-uniform test; the ordered inserts run only when some lane needs one
        const int per = (((cm + 3) >> 2) + 3) & ~3;
        const int j0 = ks * per, j1 = min((cm + 3) & ~3, j0 + per);
        for (int j = j0; j < j1; j += 4) {
            const float4 kx = *reinterpret_cast<const float4 *>(&skx[j]), ky = *reinterpret_cast<const float4 *>(&sky[j]),
                         kz = *reinterpret_cast<const float4 *>(&skz[j]);
            const float d0 = dist2<FM>(ux - kx.x, uy - ky.x, uz - kz.x);   // interpolate_gpu.cu:33 under the contraction contract
            const float d1 = dist2<FM>(ux - kx.y, uy - ky.y, uz - kz.y);
            const float d2 = dist2<FM>(ux - kx.z, uy - ky.z, uz - kz.z);
            const float d3 = dist2<FM>(ux - kx.w, uy - ky.w, uz - kz.w);
            if (__builtin_amdgcn_ballot_w64(fminf(fminf(d0, d1), fminf(d2, d3)) < b3) != 0ull) {  // wave-uniform skip
                nn_insert(d0, base + j, b1, b2, b3, i1, i2, i3);
                nn_insert(d1, base + j + 1, b1, b2, b3, i1, i2, i3);
                nn_insert(d2, base + j + 2, b1, b2, b3, i1, i2, i3);
                nn_insert(d3, base + j + 3, b1, b2, b3, i1, i2, i3);
            }
        }
    }
    if (ks > 0) {
        sd[ks - 1][lane][0] = b1; sd[ks - 1][lane][1] = b2; sd[ks - 1][lane][2] = b3;
        si[ks - 1][lane][0] = i1; si[ks - 1][lane][1] = i2; si[ks - 1][lane][2] = i3;
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = sd[w][lane][c];
                if (d < __builtin_inff()) nn_insert_lex(d, si[w][lane][c], b1, b2, b3, i1, i2, i3);
            }
        if (p < n) {
            float *d2 = dist2_all + ((size_t)b * n + p) * 3;
            int *ix = idx_all + ((size_t)b * n + p) * 3;
            d2[0] = b1; d2[1] = b2; d2[2] = b3;
            ix[0] = i1; ix[1] = i2; ix[2] = i3;
        }
    }
}

// Several small three_nn problems of the same batch in ONE launch (the inner feature-propagation levels: 256 <- 64 and 1024 <- 256 points
// are a few microseconds of work each, and every launch costs the 16-batch mix 3-5 us): blockIdx.x runs over the problems' 64-query tiles.
struct NNMulti {
    int count;
    int n[4], m[4], blk_end[4];
    const float *unknown[4], *known[4];
    float *dist2[4];
    int *idx[4];
};
// one workgroup of the multi-problem launch: bx = its index among the problems' 64-query tiles, by = the cloud
template <int FM>
__device__ __forceinline__ void three_nn_multi_role(const NNMulti &q, int bx, int by) {
    int k = 0, first = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (k == i && k + 1 < q.count && bx >= q.blk_end[i]) { first = q.blk_end[i]; k = i + 1; }
    // the problem index is block-uniform; the selects below keep the pointers in scalar registers
    const int n = k == 0 ? q.n[0] : (k == 1 ? q.n[1] : (k == 2 ? q.n[2] : q.n[3]));
    const int m = k == 0 ? q.m[0] : (k == 1 ? q.m[1] : (k == 2 ? q.m[2] : q.m[3]));
    const float *u = k == 0 ? q.unknown[0] : (k == 1 ? q.unknown[1] : (k == 2 ? q.unknown[2] : q.unknown[3]));
    const float *kn = k == 0 ? q.known[0] : (k == 1 ? q.known[1] : (k == 2 ? q.known[2] : q.known[3]));
    float *d2 = k == 0 ? q.dist2[0] : (k == 1 ? q.dist2[1] : (k == 2 ? q.dist2[2] : q.dist2[3]));
    int *ix = k == 0 ? q.idx[0] : (k == 1 ? q.idx[1] : (k == 2 ? q.idx[2] : q.idx[3]));
    three_nn_body<FM>(n, m, u, kn, d2, ix, bx - first, by);
}
}  // namespace g4d
