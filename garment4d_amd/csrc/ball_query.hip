// Ball query for gfx950 -- replaces ball_query_kernel_fast / ball_query_kernel_launcher_fast
// (/root/reference/modules/pointnet2/pointnet2/src/ball_query_gpu.cu:9-67), single- and multi-scale.
//
// Semantics kept: for query q, hits = ascending k with d2 = dist2<FM>(qx-x, qy-y, qz-z) < r*r
// (fp32 under the process-wide contraction contract of g4d.h -- nvcc's fused shape by default; r*r rounded once in fp32); out = first `nsample` hits, remaining slots
// = first hit, all 0 when there is no hit.
//
// Layout for the hardware (the reference runs one THREAD per query, each scanning N points serially with
// broadcast loads and a divergent early exit, one launch per radius):
//   * one WAVE owns QW queries; its 64 lanes test 64 consecutive points per step, each point reused for the QW
//     queries (centres in SGPRs) and for all NS radii of a multi-scale layer (one distance, NS compares);
//   * a workgroup (4 waves = 4*QW queries) streams the cloud through LDS in 1024-point SoA stages: stage s+1 is
//     loaded into registers while stage s is consumed, so the L2 stream is shared by the 4 waves and its
//     latency sits behind 16 steps of VALU work (the first version re-loaded every 64-point step per wave from L2
//     with nothing in flight: 65 us per radius at B=8, N=8192, M=1024);
//   * ascending-index order falls out of ballot + mbcnt prefix (slot = cnt + #hits in lower lanes); the common case
//     -- no lane within the LARGEST radius -- costs one compare and one branch for all scales (a radius-0.1 ball
//     holds ~0.4 % of a unit cloud), the per-scale bookkeeping runs only behind it;
//   * a wave stops testing once its queries are full; the workgroup leaves when all four are.
// Algorithmic bytes: 12*B*(N+M) + 4*B*M*sum(nsample); B*M*N distance evaluations worst case.
#include <cmath>
#include <cstdlib>

#include "g4d_common.h"
#include "three_nn_body.h"

namespace g4d {

struct BqArgs {
    float radius2_max;
    float radius2[4];
    int nsample[4];
    int *idx[4];
};

typedef g4d_f32x2 f32x2;
constexpr int kStage = 1024;  // points per LDS stage (SoA: 3 x 4 KB), double buffered; a multiple of 128 (two points per lane and step)

// boxes: per (frame, G-point block) axis-aligned bounds [lo.xyz, hi.xyz]; G = 64 (one wave per block: the lanes kernel) or
// 16 (one 16-lane row per sub-block: ball_query_sub_kernel)
template <int G>
__global__ void __launch_bounds__(256) ball_boxes_kernel(int n, int nblk, long long total, const float *__restrict__ xyz_all,
                                                        float *__restrict__ boxes) {
    constexpr int PER_WAVE = 64 / G;
    const int lane = threadIdx.x & 63;
    const long long w = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PER_WAVE + lane / G;
    const bool live = w < total;
    const long long wc = live ? w : total - 1;
    const long long b = wc / nblk;
    const int blk = (int)(wc - b * nblk);
    const int k = blk * G + (lane % G);
    const float inf = __builtin_inff();
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = k < n ? xyz_all[((size_t)b * n + k) * 3 + d] : 0.f;
        lo[d] = k < n ? v : inf;
        hi[d] = k < n ? v : -inf;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    if (live && (lane % G) == 0) {
        float *o = boxes + (size_t)w * 6;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
    }
}

template <int QW, int NS, int FM>
__device__ __forceinline__ void ball_query_body(int n, int m, const BqArgs &a, const float *__restrict__ new_xyz_all,
                                                const float *__restrict__ xyz_all, int bx, int b) {
    __shared__ float sp[2][3][kStage];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int q0 = (bx * 4 + wave) * QW;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *new_xyz = new_xyz_all + (size_t)b * m * 3;

    float qx[QW], qy[QW], qz[QW];
    int cnt[QW][NS], first[QW][NS];
    int open = 0;  // (query, scale) pairs still collecting
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = min(q0 + i, m - 1);
        qx[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 0])));
        qy[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 1])));
        qz[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 2])));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cnt[i][s] = (q0 + i < m) ? 0 : a.nsample[s];  // out-of-range query slots start "full"
            first[i][s] = 0;
            open += (q0 + i < m) ? 1 : 0;
        }
    }

    // stage loader: thread t owns points t, t+256, t+512, t+768 of a stage (coalesced 12-byte records)
    float rx[4], ry[4], rz[4];
    auto load_stage = [&](int base) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = base + t + j * 256;
            const bool ok = k < n;
            const int kc = ok ? k : n - 1;
            const float inf = __builtin_inff();
            rx[j] = ok ? xyz[kc * 3 + 0] : inf; ry[j] = ok ? xyz[kc * 3 + 1] : inf; rz[j] = ok ? xyz[kc * 3 + 2] : inf;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sp[buf][0][t + j * 256] = rx[j]; sp[buf][1][t + j * 256] = ry[j]; sp[buf][2][t + j * 256] = rz[j];
        }
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (int base = 0; base < n; base += kStage) {
        const bool more = base + kStage < n;
        if (more) load_stage(base + kStage);  // in flight while this stage is consumed
        const int cn = min(kStage, n - base);
        for (int c = 0; c < cn && open > 0; c += 128) {
            // two points per lane and step (c + lane, c + 64 + lane): the distance arithmetic runs on the packed fp32 pipe
            // (v_pk_add / v_pk_mul: same IEEE results per component, half the instructions); lanes past the end of the cloud
            // hold +inf coordinates (stage loader) -> never a hit, no `valid` predicate
            const f32x2 x = {sp[buf][0][c + lane], sp[buf][0][c + 64 + lane]};
            const f32x2 y = {sp[buf][1][c + lane], sp[buf][1][c + 64 + lane]};
            const f32x2 z = {sp[buf][2][c + lane], sp[buf][2][c + 64 + lane]};
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const f32x2 dx = f32x2{qx[i], qx[i]} - x, dy = f32x2{qy[i], qy[i]} - y, dz = f32x2{qz[i], qz[i]} - z;
                const f32x2 d2v = dist2<FM>(dx, dy, dz);
                // common case: nobody within the LARGEST radius -> two compares, one branch for all NS scales
                const unsigned long long any0 = __builtin_amdgcn_ballot_w64(d2v[0] < a.radius2_max);
                const unsigned long long any1 = __builtin_amdgcn_ballot_w64(d2v[1] < a.radius2_max);
                if ((any0 | any1) == 0ull) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // ascending index order: first the low half, then the high half
                    if ((h ? any1 : any0) == 0ull) continue;
                    const float d2 = h ? d2v[1] : d2v[0];
                    const int k = base + c + h * 64 + lane;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        if (cnt[i][s] < a.nsample[s]) {  // wave-uniform
                            const bool hit = d2 < a.radius2[s];
                            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                            if (mask != 0ull) {
                                if (cnt[i][s] == 0) first[i][s] = base + c + h * 64 + __builtin_ctzll(mask);
                                const int slot = cnt[i][s] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                if (hit && slot < a.nsample[s]) a.idx[s][((size_t)b * m + q0 + i) * a.nsample[s] + slot] = k;
                                cnt[i][s] += __builtin_popcountll(mask);
                                if (cnt[i][s] >= a.nsample[s]) --open;
                            }
                        }
                    }
                }
            }
        }
        if (!more) break;
        if (__syncthreads_or(open > 0) == 0) break;  // everybody done: skip the rest of the cloud (also the WAR barrier)
        store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // pad with the first hit (ball_query_gpu.cu:32-36); rows without a hit are zeros
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (q0 + i < m && cnt[i][s] < a.nsample[s])
                for (int l = cnt[i][s] + lane; l < a.nsample[s]; l += 64) a.idx[s][((size_t)b * m + q0 + i) * a.nsample[s] + l] = first[i][s];
}

template <int QW, int NS, int FM>
__global__ void __launch_bounds__(256) ball_query_kernel(int n, int m, const BqArgs a, const float *__restrict__ new_xyz_all,
                                                        const float *__restrict__ xyz_all) {
    ball_query_body<QW, NS, FM>(n, m, a, new_xyz_all, xyz_all, blockIdx.x, blockIdx.y);
}

// Two small ball queries of the same batch (the inner set-abstraction levels: 256 and 64 centroids per cloud) in ONE launch: a launch
// costs the 16-batch mix 3-5 us, these searches are microseconds of work.  blockIdx.x runs over both problems' 4-query workgroups.
struct BqMulti {
    int n[2], m[2], blk0;
    BqArgs a[2];
    const float *new_xyz[2], *xyz[2];
};
template <int NS, int FM>
__global__ void __launch_bounds__(256) ball_query_multi_kernel(const BqMulti q) {
    if ((int)blockIdx.x < q.blk0) ball_query_body<1, NS, FM>(q.n[0], q.m[0], q.a[0], q.new_xyz[0], q.xyz[0], blockIdx.x, blockIdx.y);
    else ball_query_body<1, NS, FM>(q.n[1], q.m[1], q.a[1], q.new_xyz[1], q.xyz[1], (int)blockIdx.x - q.blk0, blockIdx.y);
}

// The small searches of a step that depend on sampled coordinates only -- two ball queries and up to four three_nn problems (the inner
// SA / FP levels of the encoder) -- in ONE launch: workgroups [0, bq_blocks) are ball-query tiles, the rest three_nn tiles.  Each of
// these launches is a few microseconds of work and costs the many-streams regime about its own duration (launches of different
// streams barely overlap when they are this short: profiles/r03_launch_cost_isolated_vs_16streams.txt).
// QW queries per wave: 1 for a lone B = 8 step (more, smaller workgroups: latency), 4 for a coalesced call (a workgroup's LDS staging of
// the cloud and its fixed costs are shared by 16 queries instead of 4; the per-query arithmetic is the same: identical outputs).
template <int NS, int FM, int QW>
__global__ void __launch_bounds__(256) search_multi_kernel(const BqMulti q, const NNMulti nq, int bq_blocks) {
    if ((int)blockIdx.x < bq_blocks) {
        if ((int)blockIdx.x < q.blk0) ball_query_body<QW, NS, FM>(q.n[0], q.m[0], q.a[0], q.new_xyz[0], q.xyz[0], blockIdx.x, blockIdx.y);
        else ball_query_body<QW, NS, FM>(q.n[1], q.m[1], q.a[1], q.new_xyz[1], q.xyz[1], (int)blockIdx.x - q.blk0, blockIdx.y);
    } else {
        three_nn_multi_role<FM>(nq, (int)blockIdx.x - bq_blocks, blockIdx.y);
    }
}

// Sub-block-culled ball query for index-coherent clouds (g4d_ball_query_boxes_f32: the body / garment queries of
// modules/mesh_encoder.py:452-464, where consecutive vertex indices are neighbours in space, so most of the cloud lies wholly
// outside a query's largest still-open ball).
//   * The cloud is cut into SUB-BLOCKS of 16 consecutive points whose bounds a pre-pass wrote (ball_boxes_kernel<16>).  Lane l
//     tests sub-block chunk*64 + l against the query with the SAME fp32 expression (same contraction shape FM) as the point test:
//     rounding is monotone, so box d2 <= d2 of every point inside the box and a sub-block holding a hit is never skipped.
//   * The surviving sub-blocks are visited four per step in ascending order -- lanes 0-15 the lowest, 16-31 the next, ... -- so lane
//     order is still index order and the ballot + mbcnt prefix still yields "the first nsample hits by index".
//   * When a scale fills, the pruning radius drops to the largest radius still collecting and the candidate mask is re-filtered.
//   * No LDS staging, no barrier: a query visits a few percent of the cloud, so streaming every 1024-point stage through LDS (and
//     the two barriers per stage that tie a workgroup's waves together) cost more than the visited points (800-880 us staged,
//     580-690 us direct).  The survivors are read straight from global memory (192 contiguous bytes each: L1 / L2 hits -- a
//     frame's cloud is 83 KB).
// Same hit order, same rounding, same output as ball_query_kernel for ANY input (coherent or not: only the speed depends on it).
// History: round 1 culled 64-point blocks inside the LDS-staged scan (1.3 ms on the same case, no better than the plain scan).
template <int QW, int NS, int FM>
__global__ void __launch_bounds__(256) ball_query_sub_kernel(int n, int m, const BqArgs a, const float *__restrict__ new_xyz_all,
                                                            const float *__restrict__ xyz_all, const float *__restrict__ boxes_all) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * QW;
    if (q0 >= m) return;  // no barrier in this kernel
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *new_xyz = new_xyz_all + (size_t)b * m * 3;
    float qx[QW], qy[QW], qz[QW];
    int cnt[QW][NS], first[QW][NS];
    int open = 0;
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = min(q0 + i, m - 1);
        qx[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 0])));
        qy[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 1])));
        qz[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(new_xyz[q * 3 + 2])));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cnt[i][s] = (q0 + i < m) ? 0 : a.nsample[s];
            first[i][s] = 0;
            open += (q0 + i < m) ? 1 : 0;
        }
    }
    const int nsub = (n + 15) >> 4;
    const float *boxes = boxes_all + (size_t)b * nsub * 6;
    const int grp = lane >> 4, sub = lane & 15;
    // Cost model (measured, 240 x 4096 queries on 6890 points): ~4 steps and ~3.5 box chunks per query; the walk is bound by
    // instruction issue (each wave64 VALU op takes 4 cycles), not by load latency -- keeping the chains of 2 or 4 queries of a wave
    // in flight together (all point loads of a step issued before any is looked at) changed nothing (QW = 1: 597-690 us, QW = 2:
    // 690-755, QW = 4: 704-920), so a wave owns ONE query and only the next chunk's boxes are prefetched.
    float nb[6];
    {
        const float *bx = boxes + (size_t)min(lane, nsub - 1) * 6;
#pragma unroll
        for (int d = 0; d < 6; ++d) nb[d] = bx[d];
    }
    for (int sb0 = 0; sb0 < nsub && open > 0; sb0 += 64) {
        const int nss = min(64, nsub - sb0);
        const float lox = nb[0], loy = nb[1], loz = nb[2], hix = nb[3], hiy = nb[4], hiz = nb[5];
        if (sb0 + 64 < nsub) {
            const float *bx = boxes + (size_t)min(sb0 + 64 + lane, nsub - 1) * 6;
#pragma unroll
            for (int d = 0; d < 6; ++d) nb[d] = bx[d];
        }
        unsigned long long cand[QW];
        float bd2[QW], r2open[QW];
        unsigned long long anyc = 0ull;
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            r2open[i] = -1.f;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (cnt[i][s] < a.nsample[s]) r2open[i] = fmaxf(r2open[i], a.radius2[s]);
            const float ex = fmaxf(fmaxf(lox - qx[i], qx[i] - hix), 0.f), ey = fmaxf(fmaxf(loy - qy[i], qy[i] - hiy), 0.f),
                        ez = fmaxf(fmaxf(loz - qz[i], qz[i] - hiz), 0.f);
            bd2[i] = dist2<FM>(ex, ey, ez);
            cand[i] = __builtin_amdgcn_ballot_w64(lane < nss && bd2[i] < r2open[i]);  // r2open = -1: nothing passes
            anyc |= cand[i];
        }
        while (anyc) {
            float x[QW], y[QW], z[QW];
            int k[QW];
            bool live[QW], had[QW];
#pragma unroll
            for (int i = 0; i < QW; ++i) {  // the four lowest surviving sub-blocks of every query (64 = none left); loads issued for all
                unsigned long long c = cand[i];
                had[i] = c != 0ull;
                // ffs - 1 = s_ff1_i32_b64 (-1 for an empty mask); c & (c - 1) clears the lowest set bit and leaves 0 at 0 -- no guards:
                // four scalar instructions per pick (the guarded ctz / conditional clears compiled to ~11 each, through VALU compares)
                const int s0 = __builtin_ffsll((long long)c) - 1;
                c &= c - 1;
                const int s1 = __builtin_ffsll((long long)c) - 1;
                c &= c - 1;
                const int s2 = __builtin_ffsll((long long)c) - 1;
                c &= c - 1;
                const int s3 = __builtin_ffsll((long long)c) - 1;
                c &= c - 1;
                cand[i] = c;
                const int mine = grp == 0 ? s0 : (grp == 1 ? s1 : (grp == 2 ? s2 : s3));
                const int kk = (sb0 + mine) * 16 + sub;
                live[i] = mine >= 0 && kk < n;
                k[i] = live[i] ? kk : 0;
                x[i] = xyz[k[i] * 3 + 0]; y[i] = xyz[k[i] * 3 + 1]; z[i] = xyz[k[i] * 3 + 2];
            }
            anyc = 0ull;
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                if (had[i]) {  // wave-uniform
                    const float dx = qx[i] - x[i], dy = qy[i] - y[i], dz = qz[i] - z[i];
                    const float d2 = live[i] ? dist2<FM>(dx, dy, dz) : __builtin_inff();
                    if (__builtin_amdgcn_ballot_w64(d2 < r2open[i]) != 0ull) {
                        bool any_open = false, closed = false;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            if (cnt[i][s] < a.nsample[s]) {  // wave-uniform
                                const bool hit = d2 < a.radius2[s];
                                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                                if (mask != 0ull) {
                                    if (cnt[i][s] == 0) first[i][s] = __builtin_amdgcn_readlane(k[i], __builtin_ctzll(mask));
                                    const int slot = cnt[i][s] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                    if (hit && slot < a.nsample[s]) a.idx[s][((size_t)b * m + q0 + i) * a.nsample[s] + slot] = k[i];
                                    cnt[i][s] += __builtin_popcountll(mask);
                                    if (cnt[i][s] >= a.nsample[s]) { --open; closed = true; }
                                }
                                any_open |= cnt[i][s] < a.nsample[s];
                            }
                        }
                        if (!any_open) cand[i] = 0ull;
                        else if (closed) {  // a scale filled: shrink the pruning radius, drop the sub-blocks it no longer reaches
                            r2open[i] = -1.f;
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                if (cnt[i][s] < a.nsample[s]) r2open[i] = fmaxf(r2open[i], a.radius2[s]);
                            cand[i] &= __builtin_amdgcn_ballot_w64(bd2[i] < r2open[i]);
                        }
                    }
                }
                anyc |= cand[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QW; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (q0 + i < m && cnt[i][s] < a.nsample[s])
                for (int l = cnt[i][s] + lane; l < a.nsample[s]; l += 64) a.idx[s][((size_t)b * m + q0 + i) * a.nsample[s] + l] = first[i][s];
}

// "Lanes = queries" ball query for SPATIALLY COHERENT query sets against an index-coherent cloud (the body / garment queries of
// modules/mesh_encoder.py:452-464: 64 consecutive garment vertices form a compact patch, 64 consecutive body vertices a compact
// block).  The dense regime -- balls holding 100-2000 points of which 8-32 are wanted -- is decided by early exit in index
// order, not by spatial cells, and the wave-per-query scan pays its ballot / prefix bookkeeping once per (query, block).  Here a
// wave owns 64 queries, one per lane, and walks the cloud's 64-point blocks in index order:
//   * a block is skipped for the whole wave when its box is at least r_open away from the bounding box of the wave's still
//     collecting queries (the same dist2<FM> on the box-to-box gaps: monotone rounding, never skips a hit; r_open = the largest
//     radius some lane still collects for);
//   * the points of a visited block are wave-uniform (scalar loads); a lane tests its own query, appends hits to its own row:
//     ascending index order is the visiting order, no ballot, no prefix sum; one ballot per point skips the per-scale work when
//     no lane is within the largest open radius;
//   * the walk ends when every lane has every scale full.
// Output identical to ball_query_kernel for ANY input; it is FASTER only when the 64 queries of a wave are close together.
// SORTED: the queries are taken from the cell-ordered records of a grid built over them (ball_grid.hip) -- 64 consecutive records
// are spatially compact whatever the queries' own order -- and each lane writes the row of its record's ORIGINAL index.
template <int NS, int FM, bool SORTED>
__global__ void __launch_bounds__(256) ball_query_lanes_kernel(int n, int m, const BqArgs a, const float *__restrict__ new_xyz_all,
                                                              const float *__restrict__ xyz_all, const float *__restrict__ boxes_all,
                                                              const unsigned char *__restrict__ qgrid, size_t qgrid_stride, size_t qrec_off) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    const bool qok = slot < m;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    float qx, qy, qz;
    int q;
    if constexpr (SORTED) {
        const float4 rec = reinterpret_cast<const float4 *>(qgrid + (size_t)b * qgrid_stride + qrec_off)[qok ? slot : m - 1];
        qx = rec.x; qy = rec.y; qz = rec.z; q = __float_as_int(rec.w);
    } else {
        q = qok ? slot : m - 1;
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        qx = qp[0]; qy = qp[1]; qz = qp[2];
    }
    const int nblk = (n + 63) >> 6;
    const float *boxes = boxes_all + (size_t)b * nblk * 6;
    int cnt[NS], first[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { cnt[s] = qok ? 0 : a.nsample[s]; first[s] = 0; }
    const float INF = __builtin_inff();
    for (int blk = 0; blk < nblk; ++blk) {
        // wave state: which scales still collect, and the box of the lanes that do (NaN query coordinates drop out of min / max)
        float r2open = -1.f;
        bool lane_open = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool o = cnt[s] < a.nsample[s];
            lane_open |= o;
            if (__builtin_amdgcn_ballot_w64(o) != 0ull) r2open = fmaxf(r2open, a.radius2[s]);
        }
        if (r2open < 0.f) break;  // wave-uniform: every lane full
        float lx = lane_open ? qx : INF, ly = lane_open ? qy : INF, lz = lane_open ? qz : INF;
        float hx = lane_open ? qx : -INF, hy = lane_open ? qy : -INF, hz = lane_open ? qz : -INF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lx = fminf(lx, __shfl_xor(lx, o)); ly = fminf(ly, __shfl_xor(ly, o)); lz = fminf(lz, __shfl_xor(lz, o));
            hx = fmaxf(hx, __shfl_xor(hx, o)); hy = fmaxf(hy, __shfl_xor(hy, o)); hz = fmaxf(hz, __shfl_xor(hz, o));
        }
        // skip ahead over blocks whose box cannot hold a hit of any collecting lane; the wave box is recomputed only here
        int nb = blk;
        for (; nb < nblk; ++nb) {
            const float *bx = boxes + (size_t)nb * 6;
            const float gx = fmaxf(fmaxf(bx[0] - hx, lx - bx[3]), 0.f), gy = fmaxf(fmaxf(bx[1] - hy, ly - bx[4]), 0.f),
                        gz = fmaxf(fmaxf(bx[2] - hz, lz - bx[5]), 0.f);
            if (dist2<FM>(gx, gy, gz) < r2open) break;  // wave-uniform (all operands are)
        }
        blk = nb;
        if (blk >= nblk) break;
        const int k0 = blk << 6, kn = min(64, n - k0);
        for (int p = 0; p < kn; ++p) {
            const int k = k0 + p;
            const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];  // wave-uniform address: scalar loads
            const float d2 = dist2<FM>(qx - x, qy - y, qz - z);                        // ball_query_gpu.cu:30 under the contraction contract
            if (__builtin_amdgcn_ballot_w64(d2 < r2open) == 0ull) continue;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool take = d2 < a.radius2[s] && cnt[s] < a.nsample[s];
                if (take) {
                    if (cnt[s] == 0) first[s] = k;
                    a.idx[s][((size_t)b * m + q) * a.nsample[s] + cnt[s]] = k;
                    ++cnt[s];
                }
            }
        }
    }
    // pad with the first hit (ball_query_gpu.cu:32-36); rows without a hit are zeros (first = 0)
    if (qok) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            for (int l = cnt[s]; l < a.nsample[s]; ++l) a.idx[s][((size_t)b * m + q) * a.nsample[s] + l] = first[s];
    }
}

template <int NS, int FM>
static void launch_bq_sub_fm(dim3 grid, hipStream_t st, int n, int m, const BqArgs &a, const float *new_xyz, const float *xyz, const float *boxes) {
    hipLaunchKernelGGL((ball_query_sub_kernel<1, NS, FM>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz, boxes);
}
template <int NS>
static void launch_bq_sub(dim3 grid, hipStream_t st, int n, int m, const BqArgs &a, const float *new_xyz, const float *xyz, const float *boxes) {
    G4D_WITH_FM(distance_contraction(), (launch_bq_sub_fm<NS, FM>(grid, st, n, m, a, new_xyz, xyz, boxes)))
}

template <int NS, int FM>
static void launch_bq_fm(int qw, dim3 grid, hipStream_t st, int n, int m, const BqArgs &a, const float *new_xyz, const float *xyz) {
    if (qw == 4) hipLaunchKernelGGL((ball_query_kernel<4, NS, FM>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz);
    else if (qw == 2) hipLaunchKernelGGL((ball_query_kernel<2, NS, FM>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz);
    else hipLaunchKernelGGL((ball_query_kernel<1, NS, FM>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz);
}

template <int NS>
static void launch_bq(int qw, dim3 grid, hipStream_t st, int n, int m, const BqArgs &a, const float *new_xyz, const float *xyz) {
    G4D_WITH_FM(distance_contraction(), (launch_bq_fm<NS, FM>(qw, grid, st, n, m, a, new_xyz, xyz)))
}

}  // namespace g4d

static int ball_query_msg_impl(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz,
                               const float *xyz, int *const *idx, float *boxes, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nscales >= 1 && nscales <= 4 && b <= 65535, "g4d_ball_query_msg_f32: bad sizes");
    G4D_REQUIRE(radii && nsamples && idx, "g4d_ball_query_msg_f32: null pointer");
    if (b == 0 || m == 0) return G4D_OK;
    G4D_REQUIRE(new_xyz && xyz, "g4d_ball_query_msg_f32: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BqArgs a = {};
    for (int s = 0; s < nscales; ++s) {
        G4D_REQUIRE(nsamples[s] > 0 && idx[s], "g4d_ball_query_msg_f32: bad scale %d", s);
        a.radius2[s] = radii[s] * radii[s];  // ball_query_gpu.cu:23, rounded once in fp32
        a.radius2_max = s == 0 ? a.radius2[0] : (a.radius2[s] > a.radius2_max ? a.radius2[s] : a.radius2_max);
        a.nsample[s] = nsamples[s];
        a.idx[s] = idx[s];
        if (n == 0) {
            hipError_t e = hipMemsetAsync(idx[s], 0, sizeof(int) * (size_t)b * m * nsamples[s], st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (n == 0) return G4D_OK;
    // queries per wave: keep >= ~4096 waves in the launch when the problem allows
    const long long queries = (long long)b * m;
    static const long long min_waves = getenv("G4D_BQ_MIN_WAVES") ? atoll(getenv("G4D_BQ_MIN_WAVES")) : 4096;  // measured: SA1 (8192 queries) 55 us at 4 queries per wave, 44 at 2, 58 at 1
    int qw = 4;
    while (qw > 1 && queries / qw < min_waves) qw >>= 1;
    dim3 grid((m + 4 * qw - 1) / (4 * qw), b);
    if (boxes) {
        const int nblk = (n + 15) / 16;   // 16-point sub-blocks
        const long long total = (long long)b * nblk;
        hipLaunchKernelGGL(ball_boxes_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, n, nblk, total, xyz, boxes);
        grid = dim3((m + 3) / 4, b);  // one query per wave: see the cost model in ball_query_sub_kernel
        switch (nscales) {
            case 1: launch_bq_sub<1>(grid, st, n, m, a, new_xyz, xyz, boxes); break;
            case 2: launch_bq_sub<2>(grid, st, n, m, a, new_xyz, xyz, boxes); break;
            case 3: launch_bq_sub<3>(grid, st, n, m, a, new_xyz, xyz, boxes); break;
            default: launch_bq_sub<4>(grid, st, n, m, a, new_xyz, xyz, boxes); break;
        }
        return check_launch("g4d_ball_query_boxes_f32");
    }
    switch (nscales) {
        case 1: launch_bq<1>(qw, grid, st, n, m, a, new_xyz, xyz); break;
        case 2: launch_bq<2>(qw, grid, st, n, m, a, new_xyz, xyz); break;
        case 3: launch_bq<3>(qw, grid, st, n, m, a, new_xyz, xyz); break;
        default: launch_bq<4>(qw, grid, st, n, m, a, new_xyz, xyz); break;
    }
    return check_launch("g4d_ball_query_msg_f32");
}

extern "C" int g4d_ball_query_msg_f32(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                      const float *new_xyz, const float *xyz, int *const *idx, g4d_stream_t stream) {
    return ball_query_msg_impl(b, n, m, nscales, radii, nsamples, new_xyz, xyz, idx, nullptr, stream);
}

extern "C" int g4d_ball_query_boxes_f32(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                        const float *new_xyz, const float *xyz, int *const *idx, float *boxes, g4d_stream_t stream) {
    G4D_REQUIRE(boxes != nullptr, "g4d_ball_query_boxes_f32: boxes scratch is NULL");
    return ball_query_msg_impl(b, n, m, nscales, radii, nsamples, new_xyz, xyz, idx, boxes, stream);
}

extern "C" size_t g4d_ball_query_lanes_qsort_bytes(int b, int m) { return (b <= 0 || m <= 0) ? 0 : (size_t)b * g4d::grid_bytes_per_cloud(m); }

extern "C" int g4d_ball_query_lanes_f32(int b, int n, int m, int nscales, const float *radii, const int *nsamples, const float *new_xyz,
                                        const float *xyz, int *const *idx, float *boxes, void *qsort, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nscales >= 1 && nscales <= 4 && b <= 65535, "g4d_ball_query_lanes_f32: bad sizes");
    G4D_REQUIRE(radii && nsamples && idx, "g4d_ball_query_lanes_f32: null pointer");
    if (b == 0 || m == 0) return G4D_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BqArgs a = {};
    for (int s = 0; s < nscales; ++s) {
        G4D_REQUIRE(nsamples[s] > 0 && idx[s], "g4d_ball_query_lanes_f32: bad scale %d", s);
        a.radius2[s] = radii[s] * radii[s];
        a.radius2_max = s == 0 ? a.radius2[0] : (a.radius2[s] > a.radius2_max ? a.radius2[s] : a.radius2_max);
        a.nsample[s] = nsamples[s];
        a.idx[s] = idx[s];
        if (n == 0) {
            hipError_t e = hipMemsetAsync(idx[s], 0, sizeof(int) * (size_t)b * m * nsamples[s], st);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (n == 0) return G4D_OK;
    G4D_REQUIRE(new_xyz && xyz && boxes, "g4d_ball_query_lanes_f32: null pointer (boxes scratch = b * ceil(n/64) * 6 floats)");
    const int nblk = (n + 63) / 64;
    const long long total = (long long)b * nblk;
    hipLaunchKernelGGL(ball_boxes_kernel<64>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, n, nblk, total, xyz, boxes);
    dim3 grid((unsigned)((m + 255) / 256), (unsigned)b);
    const unsigned char *qg = reinterpret_cast<const unsigned char *>(qsort);
    const size_t qstride = grid_bytes_per_cloud(m), qoff = grid_records_offset(m);
    if (qsort) {
        // sort the queries into cells of half the largest radius: 64 consecutive records = a compact wave
        const int rc = grid_build(b, m, 0.5f * sqrtf(a.radius2_max) / 1.01f, new_xyz, qsort, st);
        if (rc != G4D_OK) return rc;
    }
#define G4D_LANES(NSV)                                                                                                            \
    if (qsort) hipLaunchKernelGGL((ball_query_lanes_kernel<NSV, FM, true>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz, boxes, qg, qstride, qoff); \
    else hipLaunchKernelGGL((ball_query_lanes_kernel<NSV, FM, false>), grid, dim3(256), 0, st, n, m, a, new_xyz, xyz, boxes, qg, qstride, qoff);
    G4D_WITH_FM(distance_contraction(), switch (nscales) {
        case 1: G4D_LANES(1) break;
        case 2: G4D_LANES(2) break;
        case 3: G4D_LANES(3) break;
        default: G4D_LANES(4) break;
    })
#undef G4D_LANES
    return check_launch("g4d_ball_query_lanes_f32");
}

extern "C" int g4d_ball_query_f32(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                                  int *idx, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "g4d_ball_query_f32: negative size");
    if (b == 0 || m == 0 || nsample == 0) return G4D_OK;
    G4D_REQUIRE(new_xyz && xyz && idx, "g4d_ball_query_f32: null pointer");
    int *ip[1] = {idx};
    return g4d_ball_query_msg_f32(b, n, m, 1, &radius, &nsample, new_xyz, xyz, ip, stream);
}

static int bq_multi_fill(g4d::BqMulti &q, int b, int nscales, int n0, int m0, const float *radii0, const int *nsamples0, const float *new_xyz0,
                         const float *xyz0, int *const *idx0, int n1, int m1, const float *radii1, const int *nsamples1, const float *new_xyz1,
                         const float *xyz1, int *const *idx1, const char *who) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && b <= 65535 && nscales >= 1 && nscales <= 4 && n0 > 0 && m0 > 0 && n1 > 0 && m1 > 0, "%s: bad sizes", who);
    G4D_REQUIRE(radii0 && nsamples0 && new_xyz0 && xyz0 && idx0 && radii1 && nsamples1 && new_xyz1 && xyz1 && idx1, "%s: null pointer", who);
    const int ns[2] = {n0, n1}, ms[2] = {m0, m1};
    const float *rr[2] = {radii0, radii1};
    const int *nsm[2] = {nsamples0, nsamples1};
    int *const *ix[2] = {idx0, idx1};
    for (int k = 0; k < 2; ++k) {
        q.n[k] = ns[k]; q.m[k] = ms[k];
        for (int s = 0; s < nscales; ++s) {
            G4D_REQUIRE(nsm[k][s] > 0 && ix[k][s], "%s: bad scale %d of problem %d", who, s, k);
            q.a[k].radius2[s] = rr[k][s] * rr[k][s];
            q.a[k].radius2_max = s == 0 ? q.a[k].radius2[0] : (q.a[k].radius2[s] > q.a[k].radius2_max ? q.a[k].radius2[s] : q.a[k].radius2_max);
            q.a[k].nsample[s] = nsm[k][s];
            q.a[k].idx[s] = ix[k][s];
        }
    }
    q.new_xyz[0] = new_xyz0; q.xyz[0] = xyz0; q.new_xyz[1] = new_xyz1; q.xyz[1] = xyz1;
    q.blk0 = (m0 + 3) / 4;
    return G4D_OK;
}

// Two ball queries (same batch size, same number of scales) in one launch; each output identical to g4d_ball_query_msg_f32's.
extern "C" int g4d_ball_query_msg2_f32(int b, int nscales, int n0, int m0, const float *radii0, const int *nsamples0, const float *new_xyz0,
                                       const float *xyz0, int *const *idx0, int n1, int m1, const float *radii1, const int *nsamples1,
                                       const float *new_xyz1, const float *xyz1, int *const *idx1, g4d_stream_t stream) {
    using namespace g4d;
    BqMulti q = {};
    if (const int rc = bq_multi_fill(q, b, nscales, n0, m0, radii0, nsamples0, new_xyz0, xyz0, idx0, n1, m1, radii1, nsamples1, new_xyz1, xyz1, idx1,
                                     "g4d_ball_query_msg2_f32")) return rc;
    if (b == 0) return G4D_OK;
    dim3 grid((unsigned)(q.blk0 + (m1 + 3) / 4), (unsigned)b);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define G4D_BQ2(NSV) G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((ball_query_multi_kernel<NSV, FM>), grid, dim3(256), 0, st, q))
    switch (nscales) {
        case 1: G4D_BQ2(1) break;
        case 2: G4D_BQ2(2) break;
        case 3: G4D_BQ2(3) break;
        default: G4D_BQ2(4) break;
    }
#undef G4D_BQ2
    return check_launch("g4d_ball_query_msg2_f32");
}

// g4d_ball_query_msg2_f32 and g4d_three_nn_multi_f32 (count <= 4 problems) of the same batch in ONE launch (search_multi_kernel); every
// output identical to the separate calls'.  The problems must not depend on each other's outputs.
extern "C" int g4d_search_multi_f32(int b, int nscales, int n0, int m0, const float *radii0, const int *nsamples0, const float *new_xyz0,
                                    const float *xyz0, int *const *idx0, int n1, int m1, const float *radii1, const int *nsamples1,
                                    const float *new_xyz1, const float *xyz1, int *const *idx1, int nn_count, const int *nn_n, const int *nn_m,
                                    const float *const *nn_unknown, const float *const *nn_known, float *const *nn_dist2, int *const *nn_idx,
                                    g4d_stream_t stream) {
    using namespace g4d;
    BqMulti q = {};
    if (const int rc = bq_multi_fill(q, b, nscales, n0, m0, radii0, nsamples0, new_xyz0, xyz0, idx0, n1, m1, radii1, nsamples1, new_xyz1, xyz1, idx1,
                                     "g4d_search_multi_f32")) return rc;
    G4D_REQUIRE(nn_count >= 1 && nn_count <= 4 && nn_n && nn_m && nn_unknown && nn_known && nn_dist2 && nn_idx, "g4d_search_multi_f32: bad three_nn arguments (1..4 problems)");
    if (b == 0) return G4D_OK;
    NNMulti nq = {};
    int nblocks = 0;
    for (int i = 0; i < nn_count; ++i) {
        G4D_REQUIRE(nn_n[i] > 0 && nn_m[i] > 0 && nn_unknown[i] && nn_known[i] && nn_dist2[i] && nn_idx[i], "g4d_search_multi_f32: three_nn problem %d: empty or null", i);
        nq.n[i] = nn_n[i]; nq.m[i] = nn_m[i]; nq.unknown[i] = nn_unknown[i]; nq.known[i] = nn_known[i]; nq.dist2[i] = nn_dist2[i]; nq.idx[i] = nn_idx[i];
        nblocks += (nn_n[i] + 63) / 64;
        nq.blk_end[i] = nblocks;
    }
    nq.count = nn_count;
    static const int qw_env = getenv("G4D_SEARCH_MULTI_QW") ? atoi(getenv("G4D_SEARCH_MULTI_QW")) : 0;   // tuning hook: 1 | 4
    const int qw = qw_env == 1 || qw_env == 4 ? qw_env : ((long long)b * (m0 + m1) >= 32768 ? 4 : 1);
    q.blk0 = (m0 + 4 * qw - 1) / (4 * qw);
    const int bq_blocks = q.blk0 + (m1 + 4 * qw - 1) / (4 * qw);
    dim3 grid((unsigned)(bq_blocks + nblocks), (unsigned)b);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define G4D_SM(NSV)                                                                                                                                          \
    if (qw == 4) { G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((search_multi_kernel<NSV, FM, 4>), grid, dim3(256), 0, st, q, nq, bq_blocks)) }   \
    else { G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((search_multi_kernel<NSV, FM, 1>), grid, dim3(256), 0, st, q, nq, bq_blocks)) }
    switch (nscales) {
        case 1: G4D_SM(1) break;
        case 2: G4D_SM(2) break;
        case 3: G4D_SM(3) break;
        default: G4D_SM(4) break;
    }
#undef G4D_SM
    return check_launch("g4d_search_multi_f32");
}
