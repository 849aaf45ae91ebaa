// Set-abstraction stack over xyz-only neighbourhoods: QueryAndGroup(use_xyz=True, features=None) -> SharedMLP [3, C1, C2, C3] -> pool
// (/root/reference/modules/pointnet2/pointnet2/pointnet2_modules.py:40-53 with pointnet2_utils.py:242-265; the first level of
// Pointnet2MSGSEG, pointnet2encoder.py:41-53: [3,16,16,32] over 16 samples and [3,32,32,64] over 32 samples of 1024 centres).
//
// The generic register-chain kernel (mlp_chain.hip) treats this like any stack: 32 rows per wave, every wave streams the weights from
// L2 and pays a prologue / epilogue per 32 rows -- 4.7k cycles per wave around 0.8k cycles of MFMAs for the 16-wide stack.  These two
// stacks are tiny (3.2k weights) and their rows are many (131k + 262k per 8 clouds), so here, like the positional encoders
// (pos_encode.hip):
//   * waves are persistent and autonomous (no LDS, no barrier); ALL weights, scales and shifts live in registers for the whole launch;
//   * layer 1 (K = 3) is ONE K = 4-padded MFMA per (row tile, 16 channels) since round 6 (rounds 1-5: three VALU FMAs per value -- and VALU
//     work is ADDED to the matrix pipe's time on this chip, profiles/r04_mfma_valu_overlap.txt): A = W1 (lane (channel fi, k = fq) holds
//     W1[16 ks + fi][fq], 0 for k = 3), B = the lane's OWN component fq of its row's offset x_j - q (one 4-byte load instead of a 12-byte
//     row, one subtraction instead of three).  v_mfma_f32_16x16x4_f32 is the k-ascending FMA chain fma(w2, dz, fma(w1, dy, w0 * dx)) the
//     VALU form computed (pos_encode.hip, round 5: bit-identical), and the result arrives in the operand layout layer 2 wants: lane
//     (fi = l & 15, fq = l >> 4) holds channels {16 ks + 4 fq + e} of row fi of a 16-row tile;
//   * layer 2 is evaluated TRANSPOSED (A = W2, B = h1): the accumulators come out as lane (row fi, channels 16 ct + 4 fq + r), i.e.
//     already in the operand layout of the next layer -- affine + ReLU in place, no shuffle;
//   * layer 3 in the normal orientation (A = h2, B = W3) gives lane (channel fi, rows 4 fq + r): the max / mean over the S rows of a
//     neighbourhood is register maxima + two cross-lane steps, as in pos_encode.hip.
// A pass = 32 rows (2 tiles): 96 MFMAs for [32, 32, 64].  Roofline: fp32 MFMA for the two contracted layers.
#include <cstdlib>

#include "g4d_common.h"

namespace g4d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SaXyzArgs {
    int n, p, S, logS, pool;          // pool: 1 max, 2 avg
    long long rows;                   // b * p * S
    const float *xyz, *new_xyz;
    const int *idx;
    const float *W1, *sc1, *sh1;      // W1 (C1, ldw1) row-major: columns 0..2 used
    int ldw1;
    const float *W2f, *sc2, *sh2;     // fragment order [tile][k-step of 16][lane = 16 fq + fi][4], kst2 / kst3 k-steps per tile (Kpad / 16)
    const float *W3f, *sc3, *sh3;
    int kst2, kst3;
    float *out;
    int ldo, col0;
};

struct __attribute__((packed, aligned(4))) F3s { float x, y, z; };
__device__ __forceinline__ F3s ld3(const float *base, unsigned elem) {
    return *reinterpret_cast<const F3s *>(reinterpret_cast<const char *>(base) + (elem << 2));
}

// max over the 16 rows of FOUR row-major tiles at once.  v_c (c = 0..3) holds, in lane (fi, fq), a value already reduced over that lane's
// rows 4 fq + r; what is left is the max over the four 16-lane rows of the wave, per tile.  v_permlane16_swap(a, b) leaves
// {a.row0, b.row0, a.row2, b.row2} / {a.row1, b.row1, a.row3, b.row3}, so ONE swap + ONE max halves two tiles at once and parks them in
// alternating rows; v_permlane32_swap does the same for the two halves of the wave.  Result: lane 16 c + fi = max over all 16 rows of
// tile c at column fi -- 64 different outputs in one register (one 256-byte store) after 3 swaps + 3 max, where reducing the tiles one by
// one (lane_xor16 / lane_xor32: copies, swap, select, max per step) took ~40 instructions and four quarter-wave stores.
__device__ __forceinline__ float pool4_rows_max(float v0, float v1, float v2, float v3) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
    const float m01 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));   // rows: tile 0 (fq 0|1), tile 1 (fq 0|1), tile 0 (fq 2|3), tile 1 (fq 2|3)
    const float m23 = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(m01), __float_as_uint(m23), false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));              // rows: tile 0, tile 1, tile 2, tile 3
}

// One stack over the passes `first, first + stride, ...` of a wave (a pass = 32 consecutive grouped rows).  LOGS / MAXP are compile-time:
// with the pooling mode and window resolved at run time the epilogue was ~45 % of the loop's VALU instructions (both max and sum
// computed and selected, a branch maze per channel tile; 5.0 VALU instructions per MFMA, SQ counters at 240 clouds per launch), and at
// ~5 VALU per 32-cycle MFMA the two waves of a SIMD cannot keep the matrix pipe busy (0.41).
template <int C1, int C2, int C3, int LOGS, bool MAXP>
__device__ __forceinline__ void sa_xyz_body(const SaXyzArgs &a, int first, int stride) {
    constexpr int T1 = C1 / 16, T2 = C2 / 16, T3 = C3 / 16;
    constexpr int S = 1 << LOGS;
    static_assert(S == 16 || S == 32, "16 or 32 samples");
    static_assert(T3 == 2 || T3 == 4, "32 or 64 output channels");
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, fq = lane >> 4;
    // ---- weights into registers, once per wave
    float w1f[T1], s1[T1][4], h1s[T1][4];   // w1f: A fragment of layer 1, lane (channel fi, k = fq): W1[16 ks + fi][fq], 0 for the padding column k = 3
#pragma unroll
    for (int ks = 0; ks < T1; ++ks) {
        const float w = a.W1[(ks * 16 + fi) * a.ldw1 + min(fq, 2)];
        w1f[ks] = fq < 3 ? w : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = ks * 16 + fq * 4 + e;
            s1[ks][e] = a.sc1[c]; h1s[ks][e] = a.sh1[c];
        }
    }
    const unsigned cfq = (unsigned)min(fq, 2);   // the coordinate this lane feeds to layer 1 (lanes with fq = 3 feed the zero padding)
    f32x4 w2[T2][T1], w3[T3][T2];
#pragma unroll
    for (int ct = 0; ct < T2; ++ct)
#pragma unroll
        for (int ks = 0; ks < T1; ++ks) w2[ct][ks] = *reinterpret_cast<const f32x4 *>(a.W2f + ((size_t)(ct * a.kst2 + ks) * 64 + lane) * 4);
#pragma unroll
    for (int ct = 0; ct < T3; ++ct)
#pragma unroll
        for (int ks = 0; ks < T2; ++ks) w3[ct][ks] = *reinterpret_cast<const f32x4 *>(a.W3f + ((size_t)(ct * a.kst3 + ks) * 64 + lane) * 4);
    f32x4 s2[T2], h2s[T2];   // channels 16 ct + 4 fq + r
#pragma unroll
    for (int ct = 0; ct < T2; ++ct) {
        s2[ct] = *reinterpret_cast<const f32x4 *>(a.sc2 + ct * 16 + fq * 4);
        h2s[ct] = *reinterpret_cast<const f32x4 *>(a.sh2 + ct * 16 + fq * 4);
    }
    float s3[T3], h3s[T3];   // channel 16 ct + fi
#pragma unroll
    for (int ct = 0; ct < T3; ++ct) { s3[ct] = a.sc3[ct * 16 + fi]; h3s[ct] = a.sh3[ct * 16 + fi]; }

    const int rows = (int)a.rows;            // < 2^31 (launcher)
    const int npass = (rows + 31) >> 5;      // 32 rows per pass; S = 16 divides it, S = 32 is it
    // Two-level software pipeline over the wave's passes (the gather is two dependent loads: index -> coordinates): while pass k is on
    // the VALU / MFMA, the coordinates of pass k + 1 are in flight and so are the indices of pass k + 2.
    struct Rows { float px[2], pq[2]; };   // component cfq of the neighbour / of the centre
    int ivn[2];
    auto load_idx = [&](int pass, int (&v)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) v[mt] = a.idx[min(min(pass, npass - 1) * 32 + mt * 16 + fi, rows - 1)];   // unconditional (see load_rows)
    };
    const int nq = (int)(a.rows >> LOGS);
    auto load_rows = [&](int pass, const int (&v)[2], Rows &rw) {
        // No `if (pass < npass)` around these loads: past the end the last pass is read again and never used.  A conditional block
        // makes the number of loads in flight unknown at the join, and the wait for the CURRENT pass's rows (older than these)
        // becomes s_waitcnt vmcnt(0) -- i.e. the prefetch just issued is waited for on the spot.
        pass = min(pass, npass - 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int qi = __builtin_amdgcn_readfirstlane(min((pass * 32 + mt * 16) >> LOGS, nq - 1));   // S >= 16: a tile belongs to one query
            const int f = qi / a.p;
            rw.px[mt] = a.xyz[(unsigned)(f * a.n + v[mt]) * 3u + cfq];
            rw.pq[mt] = a.new_xyz[(unsigned)qi * 3u + cfq];
        }
    };
    // One pass: rows `use` (requested a pass ago) through the three layers while the next pass's rows arrive in `fill` and the indices of the
    // one after in ivn.  Round 5: the loop below runs passes in PAIRS with the two row buffers swapping roles, instead of `cur = nxt` at the end
    // of a pass -- the copy (and the register shuffles the allocator hung on it) needed the prefetched rows right behind the loads that
    // fetched them, so every pass began by waiting for its own prefetch (s_waitcnt vmcnt(3) / vmcnt(2) five instructions after the loads).
    // `alive` (uniform): false for the second pass of the last pair when the wave has an odd number of passes -- computed on clamped rows,
    // nothing stored.
    auto do_pass = [&](int pass, const Rows &cur, Rows &fill, bool alive) {
        load_rows(pass + stride, ivn, fill);      // level 2 of the next pass
        load_idx(pass + 2 * stride, ivn);         // level 1 of the one after
        // (without this fence the machine scheduler sinks both prefetches to the BOTTOM of the pass: ~15 MFMAs of cover)
        __builtin_amdgcn_sched_barrier(0);
        const int row0 = pass * 32;
        // ---- layer 1 on the matrix pipe (K = 3 padded to one k-step), in operand layout
        f32x4 h1[2][T1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float d = cur.px[mt] - cur.pq[mt];          // pointnet2_utils.py:254, this lane's component
            const float comp = fq < 3 ? d : 0.f;
#pragma unroll
            for (int ks = 0; ks < T1; ++ks) {
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[ks], comp, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) h1[mt][ks][e] = fmaxf(__builtin_fmaf(acc[e], s1[ks][e], h1s[ks][e]), 0.f);
            }
        }
        // ---- layer 2, transposed: D[out channel 4 fq + r][row fi]
        f32x4 h2[2][T2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ct = 0; ct < T2; ++ct) h2[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < T1; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int ct = 0; ct < T2; ++ct)
                        h2[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[ct][ks][e], h1[mt][ks][e], h2[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ct = 0; ct < T2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[mt][ct][r] = fmaxf(__builtin_fmaf(h2[mt][ct][r], s2[ct][r], h2s[ct][r]), 0.f);
        // ---- layer 3, normal orientation: D[row 4 fq + r][channel fi]
        f32x4 acc[2][T3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) acc[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < T2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int ct = 0; ct < T3; ++ct)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[mt][ks][e], w3[ct][ks][e], acc[mt][ct], 0, 0, 0);
        // ---- affine + ReLU + pool over the S rows of each neighbourhood
        if constexpr (MAXP) {
            // max_r relu(y_r) = relu(max_r y_r) exactly, so the ReLU is applied once per output; v[mt][ct] = this lane's four rows of tile mt
            float v[2][T3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ct = 0; ct < T3; ++ct) {
                    const float y0 = __builtin_fmaf(acc[mt][ct][0], s3[ct], h3s[ct]), y1 = __builtin_fmaf(acc[mt][ct][1], s3[ct], h3s[ct]),
                                y2 = __builtin_fmaf(acc[mt][ct][2], s3[ct], h3s[ct]), y3 = __builtin_fmaf(acc[mt][ct][3], s3[ct], h3s[ct]);
                    v[mt][ct] = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
                }
            if constexpr (S == 32) {   // the two tiles of the pass are one neighbourhood: out row = pass
                float x[T3];
#pragma unroll
                for (int ct = 0; ct < T3; ++ct) x[ct] = fmaxf(v[0][ct], v[1][ct]);
                float *o = a.out + (size_t)pass * a.ldo + a.col0;
                if constexpr (T3 == 4) {
                    const float m = fmaxf(pool4_rows_max(x[0], x[1], x[2], x[3]), 0.f);
                    if (alive) o[lane] = m;                                                             // lane = channel: one 256-byte store
                } else {
                    const float m = fmaxf(pool4_rows_max(x[0], x[1], x[0], x[1]), 0.f);                 // rows: tile 0, tile 1, tile 0, tile 1
                    if (lane < 32 && alive) o[lane] = m;
                }
            } else {                   // S == 16: tile mt is neighbourhood 2 pass + mt
                if constexpr (T3 == 2) {
                    const float m = fmaxf(pool4_rows_max(v[0][0], v[0][1], v[1][0], v[1][1]), 0.f);     // lanes 0..31: tile 0's 32 channels, 32..63: tile 1's
                    const int g = 2 * pass + (lane >> 5);
                    if (g < nq && alive) a.out[(size_t)g * a.ldo + a.col0 + (lane & 31)] = m;
                } else {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const float m = fmaxf(pool4_rows_max(v[mt][0], v[mt][1], v[mt][2], v[mt][3]), 0.f);
                        if (2 * pass + mt < nq && alive) a.out[(size_t)(2 * pass + mt) * a.ldo + a.col0 + lane] = m;
                    }
                }
            }
        } else {   // mean pooling: the ReLU does not commute with the sum
            constexpr float inv = 1.f / (float)S;
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) {
                float v[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(__builtin_fmaf(acc[mt][ct][r], s3[ct], h3s[ct]), 0.f);
                    float x = (y[0] + y[1]) + (y[2] + y[3]);
                    x = x + lane_xor16(x);
                    x = x + lane_xor32(x);
                    v[mt] = x;      // the 16 rows of tile mt, in every lane
                }
                const int ch = ct * 16 + fi;
                if constexpr (S == 16) {
                    if (lane < 16) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const int first_row = row0 + mt * 16;
                            if (first_row < rows && alive) a.out[(size_t)(first_row >> 4) * a.ldo + a.col0 + ch] = v[mt] * inv;
                        }
                    }
                } else {
                    if (lane < 16 && row0 < rows && alive) a.out[(size_t)(row0 >> 5) * a.ldo + a.col0 + ch] = (v[0] + v[1]) * inv;
                }
            }
        }
    };
    int pass = first;
    Rows ra, rb;
    load_idx(pass, ivn);
    load_rows(pass, ivn, ra);
    load_idx(pass + stride, ivn);
    for (; pass < npass; pass += 2 * stride) {
        do_pass(pass, ra, rb, true);
        do_pass(pass + stride, rb, ra, pass + stride < npass);
    }
}

template <int C1, int C2, int C3>
__device__ __forceinline__ void sa_xyz_dispatch(const SaXyzArgs &a, int first, int stride) {
    if (a.pool == 1) {
        if (a.S == 16) sa_xyz_body<C1, C2, C3, 4, true>(a, first, stride);
        else sa_xyz_body<C1, C2, C3, 5, true>(a, first, stride);
    } else {
        if (a.S == 16) sa_xyz_body<C1, C2, C3, 4, false>(a, first, stride);
        else sa_xyz_body<C1, C2, C3, 5, false>(a, first, stride);
    }
}

template <int C1, int C2, int C3>
__global__ void __launch_bounds__(256, 2) sa_xyz_kernel(const SaXyzArgs a) {
    sa_xyz_dispatch<C1, C2, C3>(a, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), gridDim.x * 4);
}

// Both xyz-only scales of an MSG level (16-16-32 at 16 samples and 32-32-64 at 32, max pool) in ONE launch.  Every wave takes its share
// of the wide stack's passes, then its share of the narrow one's (weights re-loaded in between): one persistent pool.  (Two pools sized
// by MFMA work -- round 2 -- left 32 % of the wave slots empty at the end, the narrow stack's passes being costlier per MFMA: 1.36
// resident waves per SIMD of 2, SQ counters at 240 clouds per launch.)
template <bool MAXP>
__global__ void __launch_bounds__(256, 2) sa_xyz_pair_kernel(const SaXyzArgs a0, const SaXyzArgs a1) {
    const int first = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), stride = gridDim.x * 4;   // (uniform: pass numbers, output rows and the loop test on the scalar unit)
    sa_xyz_body<32, 32, 64, 5, MAXP>(a1, first, stride);
    sa_xyz_body<16, 16, 32, 4, MAXP>(a0, first, stride);
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_sa_xyz_mlp3_supported(int c1, int c2, int c3, int nsample) {
    return ((c1 == 16 && c2 == 16 && c3 == 32) || (c1 == 32 && c2 == 32 && c3 == 64)) && (nsample == 16 || nsample == 32);
}

static int sa_xyz_fill(SaXyzArgs &a, int b, int n, int p, int nsample, const float *xyz, const float *new_xyz, const int *idx, int c1, int c2, int c3,
                       const float *W1, int ldw1, const float *scale1, const float *shift1, const float *W2_frag, int kpad2, const float *scale2,
                       const float *shift2, const float *W3_frag, int kpad3, const float *scale3, const float *shift3, int pool, float *out, int ldo,
                       int col0) {
    G4D_REQUIRE(b >= 0 && n > 0 && p >= 0, "g4d_sa_xyz_mlp3_f32: bad sizes");
    G4D_REQUIRE(g4d_sa_xyz_mlp3_supported(c1, c2, c3, nsample), "g4d_sa_xyz_mlp3_f32: widths %d-%d-%d over %d samples are not instantiated", c1, c2,
                c3, nsample);
    G4D_REQUIRE(pool == 1 || pool == 2, "g4d_sa_xyz_mlp3_f32: pool must be 1 (max) or 2 (avg)");
    const long long rows = (long long)b * p * nsample;
    G4D_REQUIRE(xyz && new_xyz && idx && W1 && scale1 && shift1 && W2_frag && scale2 && shift2 && W3_frag && scale3 && shift3 && out && ldw1 >= 3,
                "g4d_sa_xyz_mlp3_f32: null pointer");
    G4D_REQUIRE(ldo >= col0 + c3 && col0 >= 0, "g4d_sa_xyz_mlp3_f32: output window out of range");
    G4D_REQUIRE(rows < (1ll << 31) - 64 && (long long)b * n * 12 < (1ll << 32), "g4d_sa_xyz_mlp3_f32: needs rows < 2^31 and b*n*12 B < 4 GB");
    a.n = n; a.p = p; a.S = nsample; a.logS = nsample == 16 ? 4 : 5; a.pool = pool; a.rows = rows;
    a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.W1 = W1; a.sc1 = scale1; a.sh1 = shift1; a.ldw1 = ldw1;
    G4D_REQUIRE(kpad2 % 16 == 0 && kpad2 >= c1 && kpad3 % 16 == 0 && kpad3 >= c2, "g4d_sa_xyz_mlp3_f32: Kpad of layers 2 / 3 must be multiples of 16 covering c1 / c2");
    a.W2f = W2_frag; a.sc2 = scale2; a.sh2 = shift2; a.W3f = W3_frag; a.sc3 = scale3; a.sh3 = shift3; a.kst2 = kpad2 / 16; a.kst3 = kpad3 / 16;
    a.out = out; a.ldo = ldo; a.col0 = col0;
    return G4D_OK;
}

// Both scales of an xyz-only MSG level in one launch: scale 0 must be the 16-16-32 stack, scale 1 the 32-32-64 one (sa_xyz_pair_kernel);
// same centroids / cloud, each scale with its own neighbour indices, weights and output window.  Results identical to two
// g4d_sa_xyz_mlp3_f32 calls.
extern "C" int g4d_sa_xyz_mlp3_pair_f32(int b, int n, int p, const float *xyz, const float *new_xyz, int pool, float *out, int ldo,
                                        int nsample0, const int *idx0, const float *W1_0, int ldw1_0, const float *scale1_0, const float *shift1_0,
                                        const float *W2_frag0, int kpad2_0, const float *scale2_0, const float *shift2_0, const float *W3_frag0,
                                        int kpad3_0, const float *scale3_0, const float *shift3_0, int col0_0,
                                        int nsample1, const int *idx1, const float *W1_1, int ldw1_1, const float *scale1_1, const float *shift1_1,
                                        const float *W2_frag1, int kpad2_1, const float *scale2_1, const float *shift2_1, const float *W3_frag1,
                                        int kpad3_1, const float *scale3_1, const float *shift3_1, int col0_1, g4d_stream_t stream) {
    SaXyzArgs a0, a1;
    if (int rc = sa_xyz_fill(a0, b, n, p, nsample0, xyz, new_xyz, idx0, 16, 16, 32, W1_0, ldw1_0, scale1_0, shift1_0, W2_frag0, kpad2_0, scale2_0, shift2_0,
                             W3_frag0, kpad3_0, scale3_0, shift3_0, pool, out, ldo, col0_0)) return rc;
    if (int rc = sa_xyz_fill(a1, b, n, p, nsample1, xyz, new_xyz, idx1, 32, 32, 64, W1_1, ldw1_1, scale1_1, shift1_1, W2_frag1, kpad2_1, scale2_1, shift2_1,
                             W3_frag1, kpad3_1, scale3_1, shift3_1, pool, out, ldo, col0_1)) return rc;
    if (a0.rows == 0) return G4D_OK;
    G4D_REQUIRE(nsample0 == 16 && nsample1 == 32, "g4d_sa_xyz_mlp3_pair_f32: scale 0 takes 16 samples, scale 1 takes 32");
    // one persistent pool, three workgroups per CU (round 6: with layer 1 on the matrix pipe the kernel needs 155 registers instead of 190, three
    // waves per SIMD fit: 534 -> 513 us at 240 clouds), never more waves than the wide stack has passes
    static const int bpc = [] { const char *e = getenv("G4D_SA_XYZ_BLOCKS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 3; }();
    const long long want = ((a1.rows + 31) / 32 + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * bpc ? want : 256 * bpc);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (pool == 1) hipLaunchKernelGGL(sa_xyz_pair_kernel<true>, dim3(grid), dim3(256), 0, st, a0, a1);
    else hipLaunchKernelGGL(sa_xyz_pair_kernel<false>, dim3(grid), dim3(256), 0, st, a0, a1);
    return check_launch("g4d_sa_xyz_mlp3_pair_f32");
}

extern "C" int g4d_sa_xyz_mlp3_f32(int b, int n, int p, int nsample, const float *xyz, const float *new_xyz, const int *idx, int c1, int c2,
                                   int c3, const float *W1, int ldw1, const float *scale1, const float *shift1, const float *W2_frag, int kpad2,
                                   const float *scale2, const float *shift2, const float *W3_frag, int kpad3, const float *scale3, const float *shift3,
                                   int pool, float *out, int ldo, int col0, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && n > 0 && p >= 0, "g4d_sa_xyz_mlp3_f32: bad sizes");
    G4D_REQUIRE(g4d_sa_xyz_mlp3_supported(c1, c2, c3, nsample), "g4d_sa_xyz_mlp3_f32: widths %d-%d-%d over %d samples are not instantiated", c1, c2,
                c3, nsample);
    G4D_REQUIRE(pool == 1 || pool == 2, "g4d_sa_xyz_mlp3_f32: pool must be 1 (max) or 2 (avg)");
    const long long rows = (long long)b * p * nsample;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(xyz && new_xyz && idx && W1 && scale1 && shift1 && W2_frag && scale2 && shift2 && W3_frag && scale3 && shift3 && out && ldw1 >= 3,
                "g4d_sa_xyz_mlp3_f32: null pointer");
    G4D_REQUIRE(ldo >= col0 + c3 && col0 >= 0, "g4d_sa_xyz_mlp3_f32: output window out of range");
    G4D_REQUIRE(rows < (1ll << 31) - 64 && (long long)b * n * 12 < (1ll << 32), "g4d_sa_xyz_mlp3_f32: needs rows < 2^31 and b*n*12 B < 4 GB");
    SaXyzArgs a;
    a.n = n; a.p = p; a.S = nsample; a.logS = nsample == 16 ? 4 : 5; a.pool = pool; a.rows = rows;
    a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.W1 = W1; a.sc1 = scale1; a.sh1 = shift1; a.ldw1 = ldw1;
    G4D_REQUIRE(kpad2 % 16 == 0 && kpad2 >= c1 && kpad3 % 16 == 0 && kpad3 >= c2, "g4d_sa_xyz_mlp3_f32: Kpad of layers 2 / 3 must be multiples of 16 covering c1 / c2");
    a.W2f = W2_frag; a.sc2 = scale2; a.sh2 = shift2; a.W3f = W3_frag; a.sc3 = scale3; a.sh3 = shift3; a.kst2 = kpad2 / 16; a.kst3 = kpad3 / 16;
    a.out = out; a.ldo = ldo; a.col0 = col0;
    const long long npass = (rows + 31) / 32, want = (npass + 3) / 4;
    static const int bpc = [] { const char *e = getenv("G4D_SA_XYZ_BLOCKS_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 3; }();
    const unsigned grid = (unsigned)(want < 256 * bpc ? want : 256 * bpc);   // persistent waves (3 per SIMD): the weights are loaded once per wave
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (c1 == 16) hipLaunchKernelGGL((sa_xyz_kernel<16, 16, 32>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sa_xyz_kernel<32, 32, 64>), dim3(grid), dim3(256), 0, st, a);
    return check_launch("g4d_sa_xyz_mlp3_f32");
}
