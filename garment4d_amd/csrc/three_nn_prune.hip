// three_nn with exact spatial pruning for the last feature-propagation level (/root/reference/modules/pointnet2/pointnet2/src/
// interpolate_gpu.cu:9-74: for every unknown point the three smallest squared distances over ALL m known points, earlier index first
// among equal distances).  Round 5.
//
// three_nn_wide_kernel (pointnet2_ops.hip) evaluates n x m distances per cloud -- 8192 x 1024 on cfg2's last level, 2.0 G evaluations per
// 240-cloud call, 320 us at 0.56 of the nominal VALU rate.  The known set of that level is an FPS subset: well spread, ~3 of its points
// within 0.09 of any query.  So:
//   * a PRE-PASS (one workgroup per cloud, ~10 us per launch) Morton-sorts the m <= 1024 known points in LDS and writes them as
//     (x, y, z, original index) records plus the bounding box of every 16 consecutive records (a "block") to a workspace;
//   * the search stages a cloud's records and boxes in LDS (23 KB); a wave owns 64 queries taken in the CELL ORDER of the unknown cloud's
//     ball grid (spatial neighbours) or in index order.  Step 1: every lane evaluates its HOME block (the block whose Morton range holds the
//     query's code: per-lane LDS addresses) -- a first, tight third distance; without it one lane that has not met its neighbourhood yet keeps
//     every block alive for the whole wave.  Step 2: lane l tests block l against the bounding box of the wave's queries with the SAME fp32
//     expression as the point distance (dist2<FM> on the per-axis gaps: rounding a difference is monotone and dist2<FM> is monotone in every
//     |argument|, so no point of the block is closer to any query of the wave than that bound -- exactly, under every contraction mode);
//     the blocks whose bound does not exceed the wave's largest third distance are walked in ascending order, each tested once more per lane
//     against its own query right before the visit (a lane's home block is never inserted twice);
//   * a visit is the scan's inner loop (4 candidates per step out of LDS) with the (distance, ORIGINAL index) lexicographic insert of the
//     split scan's merge: the result is the three smallest under that order whatever the visiting order -- exactly what the reference's
//     strict `<` scan in index order leaves.  A block is skipped only when its bound is STRICTLY larger than every lane's third distance, so
//     an equal-distance candidate with a lower index is never lost.
// Measured (240 clouds, 8192 <- 1024 FPS-selected points, scripts/time_three_nn.py): uniform volume clouds 390 -> 303-313 us, body-like surface
// clouds 370 -> 212-218 us (pre-pass 18 us of it); bench 55.0k -> 55.6k frames/s fp32, 103.8k -> 106.1k bf16.  What limits it: the 64 queries of a
// wave are a ROW of 8 grid cells (0.8 x 0.1 x 0.1 of the unit cube), so the union of their neighbourhoods is 16-20 of the 64 blocks; compact
// (Morton-ordered) waves would visit 6-8.  Below ~32 clouds per launch the scan is faster (8 clouds: 26 vs 59-77 us): fused.py switches there.
// Output identical to g4d_three_nn_f32 / _cells_f32 / _cells_sorted_f32 for ANY input (duplicates, zero padding, non-finite coordinates: a NaN
// never passes a `<` / `<=`, an infinite coordinate makes a block's bound 0 or inf, both safe).
#include <cstdlib>

#include "g4d_common.h"
#include "three_nn_body.h"
#include "ball_grid_build.h"

namespace g4d {

namespace {
constexpr int kPrM = 1024;          // known points per cloud, padded
constexpr int kPrBlk = 16;          // records per block
constexpr int kPrNB = kPrM / kPrBlk;   // 64 blocks: one per lane

__device__ __forceinline__ unsigned pr_part1by2(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ float pr_wave_min(float v) {   // NaN-safe (fminf drops a NaN), all lanes end with the result
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float pr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
}  // namespace

// workspace per cloud: kPrM float4 records (sorted; padding: +inf coordinates, index INT_MAX) + kPrNB boxes of 8 floats (lo.xyz, 0, hi.xyz, 0)
//                      + kPrNB Morton codes (the first record's code per block; 0xffffffff: empty) + the sort grid (lo.xyz, scale)
constexpr size_t kPrBoxOff = (size_t)kPrM * 16, kPrCodeOff = kPrBoxOff + (size_t)kPrNB * 32, kPrGridOff = kPrCodeOff + (size_t)kPrNB * 4;
constexpr size_t kPrCloudBytes = kPrGridOff + 16;
constexpr int kPrepT = 512;          // threads of the pre-pass: one compare-exchange pair each

__global__ void __launch_bounds__(kPrepT) nn_prune_prep_kernel(int m, const float *__restrict__ known_all, unsigned char *__restrict__ ws) {
    __shared__ unsigned long long keys[kPrM];
    __shared__ float sx[kPrM], sy[kPrM], sz[kPrM];
    __shared__ float red[6][kPrepT / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float *known = known_all + (size_t)b * m * 3;
    const float INF = __builtin_inff();
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    for (int k = t; k < kPrM; k += kPrepT) {
        const bool ok = k < m;
        const float x = ok ? known[k * 3 + 0] : INF, y = ok ? known[k * 3 + 1] : INF, z = ok ? known[k * 3 + 2] : INF;
        sx[k] = x; sy[k] = y; sz[k] = z;
        if (ok && x - x == 0.f && y - y == 0.f && z - z == 0.f) {   // finite points span the sort's grid (any finite scale is good enough: only coherence matters)
            lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
            hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
        }
    }
    lx = pr_wave_min(lx); ly = pr_wave_min(ly); lz = pr_wave_min(lz);
    hx = pr_wave_max(hx); hy = pr_wave_max(hy); hz = pr_wave_max(hz);
    if (lane == 0) { red[0][wave] = lx; red[1][wave] = ly; red[2][wave] = lz; red[3][wave] = hx; red[4][wave] = hy; red[5][wave] = hz; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kPrepT / 64; ++w) {
        lx = fminf(lx, red[0][w]); ly = fminf(ly, red[1][w]); lz = fminf(lz, red[2][w]);
        hx = fmaxf(hx, red[3][w]); hy = fmaxf(hy, red[4][w]); hz = fmaxf(hz, red[5][w]);
    }
    const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
    const float scale = ext < INF ? 1023.0f / ext : 0.f;
    for (int k = t; k < kPrM; k += kPrepT) {
        unsigned long long key = ~0ull;   // padding sorts to the end
        if (k < m) {
            const float x = sx[k], y = sy[k], z = sz[k];
            // NaN / inf coordinates: fminf / fmaxf clamp them into the grid (a NaN becomes 0): they only need SOME place in the order
            const unsigned cx = (unsigned)fminf(fmaxf((x - lx) * scale, 0.f), 1023.f);
            const unsigned cy = (unsigned)fminf(fmaxf((y - ly) * scale, 0.f), 1023.f);
            const unsigned cz = (unsigned)fminf(fmaxf((z - lz) * scale, 0.f), 1023.f);
            const unsigned code = pr_part1by2(cx) | (pr_part1by2(cy) << 1) | (pr_part1by2(cz) << 2);
            key = ((unsigned long long)code << 32) | (unsigned)k;
        }
        keys[k] = key;
    }
    __syncthreads();
    for (int k = 2; k <= kPrM; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < kPrM / 2; i += kPrepT) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int c = a | j;
                const unsigned long long ka = keys[a], kc = keys[c];
                const bool asc = (a & k) == 0;
                if ((ka > kc) == asc) { keys[a] = kc; keys[c] = ka; }
            }
            __syncthreads();
        }
    }
    float4 *rec = reinterpret_cast<float4 *>(ws + (size_t)b * kPrCloudBytes);
    float *box = reinterpret_cast<float *>(ws + (size_t)b * kPrCloudBytes + kPrBoxOff);
    unsigned *bcode = reinterpret_cast<unsigned *>(ws + (size_t)b * kPrCloudBytes + kPrCodeOff);
    if (t == 0) { float *g = reinterpret_cast<float *>(ws + (size_t)b * kPrCloudBytes + kPrGridOff); g[0] = lx; g[1] = ly; g[2] = lz; g[3] = scale; }
    for (int p = t; p < kPrM; p += kPrepT) {
        const unsigned long long key = keys[p];
        const bool ok = key != ~0ull;
        const int k = ok ? (int)(unsigned)key : 0;
        rec[p] = make_float4(ok ? sx[k] : INF, ok ? sy[k] : INF, ok ? sz[k] : INF, __int_as_float(ok ? k : 0x7fffffff));
    }
    if (t < kPrNB) {   // the box of block t: NaN coordinates are dropped by fminf / fmaxf (a NaN point is never anybody's neighbour); padding: an empty box
        float bl[3] = {INF, INF, INF}, bh[3] = {-INF, -INF, -INF};
        for (int q = 0; q < kPrBlk; ++q) {
            const unsigned long long key = keys[t * kPrBlk + q];
            if (key == ~0ull) continue;
            const int k = (int)(unsigned)key;
            bl[0] = fminf(bl[0], sx[k]); bl[1] = fminf(bl[1], sy[k]); bl[2] = fminf(bl[2], sz[k]);
            bh[0] = fmaxf(bh[0], sx[k]); bh[1] = fmaxf(bh[1], sy[k]); bh[2] = fmaxf(bh[2], sz[k]);
        }
        bcode[t] = (unsigned)(keys[t * kPrBlk] >> 32);   // (0xffffffff for an all-padding block)
        float *o = box + t * 8;
        o[0] = bl[0]; o[1] = bl[1]; o[2] = bl[2]; o[3] = 0.f; o[4] = bh[0]; o[5] = bh[1]; o[6] = bh[2]; o[7] = 0.f;
    }
}

// gap between [lo, hi] and [qlo, qhi] along one axis, >= 0; empty boxes (lo = +inf, hi = -inf) give +inf
__device__ __forceinline__ float pr_gap(float lo, float hi, float qlo, float qhi) { return fmaxf(fmaxf(lo - qhi, qlo - hi), 0.f); }

constexpr int kPrT = 1024;   // queries (threads) per workgroup: one 23 KB staging of the cloud's sorted known set serves 16 waves (256: 0.43 of the wave cycles parked behind it)
template <int FM>
__global__ void __launch_bounds__(kPrT) three_nn_prune_kernel(int n, int m, const float *__restrict__ unknown_all, const unsigned char *__restrict__ ws,
                                                            float *__restrict__ dist2_all, int *__restrict__ idx_all,
                                                            const unsigned char *__restrict__ qrec, size_t qstride, int sorted_out) {
    // 20 floats per block of 16 records: the per-lane reads of step 1 (lanes in different home blocks) would otherwise all fall into the two
    // banks a 64-byte block stride leaves; 16-byte alignment of a block's four-record groups is kept
    constexpr int LB = kPrBlk + 4;
    __shared__ __attribute__((aligned(16))) float skx[kPrNB * LB], sky[kPrNB * LB], skz[kPrNB * LB];
    __shared__ __attribute__((aligned(16))) int ski[kPrNB * LB];
    __shared__ __attribute__((aligned(16))) float sbox[kPrNB * 8];
    __shared__ unsigned scode[kPrNB];
    __shared__ float sgrid[4];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63;
    // Every load of the prologue is issued before any of them is waited for (round 5: as separate guarded loops / ifs each load was followed by
    // its own s_waitcnt vmcnt(0) -- five L2 round trips in a row at the start of every workgroup).  Indices past an array's end are clamped
    // and the value dropped.
    static_assert(kPrM == kPrT && kPrNB * 8 <= kPrT, "one record per thread");
    const unsigned char *cw = ws + (size_t)b * kPrCloudBytes;
    const float4 r = reinterpret_cast<const float4 *>(cw)[t];
    const float boxv = reinterpret_cast<const float *>(cw + kPrBoxOff)[min(t, kPrNB * 8 - 1)];
    const unsigned codev = reinterpret_cast<const unsigned *>(cw + kPrCodeOff)[min(t, kPrNB - 1)];
    const float gridv = reinterpret_cast<const float *>(cw + kPrGridOff)[min(t, 3)];
    const int p = (int)blockIdx.x * kPrT + t;
    float ux, uy, uz;
    int orig = p;
    if (qrec) {
        const float4 q = reinterpret_cast<const float4 *>(qrec + (size_t)b * qstride)[min(p, n - 1)];
        ux = q.x; uy = q.y; uz = q.z; orig = sorted_out ? p : __float_as_int(q.w);
    } else {
        const float *u = unknown_all + ((size_t)b * n + min(p, n - 1)) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    {
        const int jj = (t >> 4) * LB + (t & 15);
        skx[jj] = r.x; sky[jj] = r.y; skz[jj] = r.z; ski[jj] = __float_as_int(r.w);
        if (t < kPrNB * 8) sbox[t] = boxv;
        if (t < kPrNB) scode[t] = codev;
        if (t < 4) sgrid[t] = gridv;
    }
    __syncthreads();
    const float INF = __builtin_inff();
    float b1 = INF, b2 = INF, b3 = INF;
    int i1 = 0, i2 = 0, i3 = 0;
    const int nblk = (m + kPrBlk - 1) / kPrBlk;
    // ---- 1. every lane's HOME block (the block whose Morton range holds the query's code) gives it a first, tight third distance: without it
    //         one lane that has not met its neighbourhood yet keeps every block of the walk alive for the whole wave
    int hb;
    {
        const float gx = sgrid[0], gy = sgrid[1], gz = sgrid[2], gs = sgrid[3];
        const unsigned cx = (unsigned)fminf(fmaxf((ux - gx) * gs, 0.f), 1023.f), cy = (unsigned)fminf(fmaxf((uy - gy) * gs, 0.f), 1023.f),
                       cz = (unsigned)fminf(fmaxf((uz - gz) * gs, 0.f), 1023.f);
        const unsigned cq = pr_part1by2(cx) | (pr_part1by2(cy) << 1) | (pr_part1by2(cz) << 2);
        int pos = 0;
#pragma unroll
        for (int step = kPrNB / 2; step > 0; step >>= 1)
            if (pos + step < nblk && scode[pos + step] <= cq) pos += step;
        hb = pos;
#pragma unroll 4
        for (int s4 = 0; s4 < kPrBlk; ++s4) {
            const int j = hb * LB + s4;   // per-lane LDS addresses
            const float d = dist2<FM>(ux - skx[j], uy - sky[j], uz - skz[j]);
            if (d < INF) nn_insert_lex(d, ski[j], b1, b2, b3, i1, i2, i3);
        }
    }
    // ---- 2. the walk: ascending over the blocks that can still matter to some lane
    const float qlx = pr_wave_min(ux), qly = pr_wave_min(uy), qlz = pr_wave_min(uz);   // the wave's query box (NaN coordinates dropped)
    const float qhx = pr_wave_max(ux), qhy = pr_wave_max(uy), qhz = pr_wave_max(uz);
    const float *mybox = sbox + min(lane, kPrNB - 1) * 8;
    // lane l: lower bound of d(query, point) over every query of the wave and every point of block l
    const float wbound = dist2<FM>(pr_gap(mybox[0], mybox[4], qlx, qhx), pr_gap(mybox[1], mybox[5], qly, qhy), pr_gap(mybox[2], mybox[6], qlz, qhz));
    const bool nanq = !(ux == ux && uy == uy && uz == uz);   // a NaN query inserts nothing (inf / 0 like in the scan): it asks for no block
    const float maxb3 = pr_wave_max(nanq ? -INF : b3);
    unsigned long long rest = __builtin_amdgcn_ballot_w64(lane < nblk && !(wbound > maxb3));   // (NaN bounds -- infinite coordinates on both sides -- are kept)
    for (; rest != 0ull; rest &= rest - 1) {
        const int blk = __builtin_ctzll(rest);
        const float *bx = sbox + blk * 8;   // wave-uniform address: broadcast reads
        const float dq = dist2<FM>(pr_gap(bx[0], bx[4], ux, ux), pr_gap(bx[1], bx[5], uy, uy), pr_gap(bx[2], bx[6], uz, uz));
        const bool want = !nanq && blk != hb && !(dq > b3);   // strictly farther than the lane's third distance: nothing to gain, not even a tie
        if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
#pragma unroll
        for (int s4 = 0; s4 < kPrBlk; s4 += 4) {
            const int j = blk * LB + s4;
            const float4 kx = *reinterpret_cast<const float4 *>(&skx[j]), ky = *reinterpret_cast<const float4 *>(&sky[j]),
                         kz = *reinterpret_cast<const float4 *>(&skz[j]);
            const int4 ki = *reinterpret_cast<const int4 *>(&ski[j]);
            const float d0 = dist2<FM>(ux - kx.x, uy - ky.x, uz - kz.x);   // interpolate_gpu.cu:33 under the contraction contract
            const float d1 = dist2<FM>(ux - kx.y, uy - ky.y, uz - kz.y);
            const float d2 = dist2<FM>(ux - kx.z, uy - ky.z, uz - kz.z);
            const float d3 = dist2<FM>(ux - kx.w, uy - ky.w, uz - kz.w);
            // a lane whose HOME block this is has these points already (`want` is false for it): a second insert would duplicate them.
            // <=: an equal distance with a lower original index replaces; +inf distances (padding, infinite coordinates) never insert, as in
            // the reference's strict `<` against an initial +inf
            const bool ok = blk != hb;
            if (__builtin_amdgcn_ballot_w64(ok && fminf(fminf(d0, d1), fminf(d2, d3)) <= b3) == 0ull) continue;
            if (__builtin_amdgcn_ballot_w64(ok && d0 <= b3 && d0 < INF) != 0ull) { if (ok && d0 < INF) nn_insert_lex(d0, ki.x, b1, b2, b3, i1, i2, i3); }
            if (__builtin_amdgcn_ballot_w64(ok && d1 <= b3 && d1 < INF) != 0ull) { if (ok && d1 < INF) nn_insert_lex(d1, ki.y, b1, b2, b3, i1, i2, i3); }
            if (__builtin_amdgcn_ballot_w64(ok && d2 <= b3 && d2 < INF) != 0ull) { if (ok && d2 < INF) nn_insert_lex(d2, ki.z, b1, b2, b3, i1, i2, i3); }
            if (__builtin_amdgcn_ballot_w64(ok && d3 <= b3 && d3 < INF) != 0ull) { if (ok && d3 < INF) nn_insert_lex(d3, ki.w, b1, b2, b3, i1, i2, i3); }
        }
    }
    if (p < n) {
        float *d2o = dist2_all + ((size_t)b * n + orig) * 3;
        int *ix = idx_all + ((size_t)b * n + orig) * 3;
        d2o[0] = b1; d2o[1] = b2; d2o[2] = b3;
        ix[0] = i1; ix[1] = i2; ix[2] = i3;
    }
}

}  // namespace g4d

using namespace g4d;

// three_nn of (b, n, 3) unknown points over (b, m, 3) known points with exact block pruning.  Supported: 16 <= m <= 1024, b <= 65535.
// unknown_grid (optional): the ball-grid workspace of the unknown cloud -- queries are then taken in its cell order; sorted_out != 0 leaves the
// results in that order (as g4d_three_nn_cells_sorted_f32).  Without a grid `unknown` is read in index order.  ws: >= g4d_three_nn_pruned_ws_bytes(b)
// bytes of device memory, 16-byte aligned, owned by the caller.
extern "C" int g4d_three_nn_pruned_supported(int n, int m) { return n > 0 && m >= 16 && m <= kPrM; }
extern "C" long long g4d_three_nn_pruned_ws_bytes(int b) { return (long long)(b > 0 ? b : 0) * (long long)kPrCloudBytes; }
extern "C" int g4d_three_nn_pruned_f32(int b, int n, int m, const float *unknown, const void *unknown_grid, const float *known, float *dist2, int *idx,
                                       int sorted_out, void *ws, long long ws_bytes, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && b <= 65535 && g4d_three_nn_pruned_supported(n, m), "g4d_three_nn_pruned_f32: needs 16 <= m <= 1024, n > 0, b <= 65535 (use g4d_three_nn_f32)");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(known && dist2 && idx && ws && (unknown || unknown_grid) && (!sorted_out || unknown_grid), "g4d_three_nn_pruned_f32: null pointer");
    G4D_REQUIRE(ws_bytes >= g4d_three_nn_pruned_ws_bytes(b) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "g4d_three_nn_pruned_f32: workspace too small or not 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nn_prune_prep_kernel, dim3(b), dim3(kPrepT), 0, st, m, known, reinterpret_cast<unsigned char *>(ws));
    if (const int rc = check_launch("g4d_three_nn_pruned_f32(prep)")) return rc;
    size_t off = 0, stride = 0;
    const unsigned char *qrec = nullptr;
    if (unknown_grid) {
        grid_sorted_layout(n, &off, &stride);
        qrec = reinterpret_cast<const unsigned char *>(unknown_grid) + off;
    }
    dim3 grid((n + kPrT - 1) / kPrT, b);
    G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((three_nn_prune_kernel<FM>), grid, dim3(kPrT), 0, st, n, m, unknown, reinterpret_cast<const unsigned char *>(ws), dist2, idx,
                                                           qrec, stride, sorted_out))
    return check_launch("g4d_three_nn_pruned_f32");
}
