// Bucketed furthest point sampling for LARGE clouds, 8192 < N <= 32768 (SURVEY.md 8a-a1's stress shape 32 x 32768 -> 8192): same
// algorithm and bit-identical output as fps_bucket.hip / the reference (sampling_gpu.cu:93-253), laid out for a cloud that does not
// fit a compute unit's registers AND LDS:
//   * one workgroup of 16 waves per cloud; the cloud is Morton-sorted once (bitonic sort of 32-bit keys = coarse Morton code | index,
//     128 KB of LDS) and cut into 64-point buckets, P = 16 or 32 consecutive buckets per wave;
//   * a lane keeps, for its P points, the running min-distance (P registers) and the tie RANK, which also encodes the original index
//     (15 bits at these sizes -- bit-reversed k mod 1024, then k / 1024 -- two per register: P / 2 registers); the COORDINATES stay in global memory (a 32768-point cloud is 384 KB: it lives in L2) and are fetched only for
//     the buckets a round actually sweeps -- with the exact box pruning of fps_bucket.hip that is ~3 % of them;
//   * the lane arg-max is two-level: slots are grouped by 8, a group's (value, rank) candidate is cached and recomputed only when
//     one of its buckets was swept, the lane candidate is the best of the P / 8 group candidates;
//   * workgroup exchange as in fps_bucket.hip (one LDS atomic max per wave on a rotating slot, one barrier); the winner's coordinates
//     are one broadcast load from global memory.
// Round 6: multi-pick rounds (fps_big_kernel<., ., MULTI = true>, the default; G4D_FPS_BIG_MULTI=0 restores one pick per round): 32 x 32768 -> 8192 in
// 6.6 ms instead of 12.6 (0.80 us per pick).
// What bounds a single-pick round: the dependent chain box test -> gather of the swept buckets (one L2 round trip per active group of 8 slots) ->
// group / lane / wave arg-max -> barrier -> winner decode -> winner's coordinates (another L2 round trip).  Measured: see DESIGN.md.
// The generic kernel this replaces for these sizes (fps.hip: 1024 threads striding over global min-distances, the reference's own
// shape) runs ~3 us per round at N = 8192 and ~10 us at N = 32768.
#include "g4d_common.h"

namespace g4d {

constexpr int kBigHdr = 2048;

__device__ __forceinline__ unsigned fpsg_rank(int k, int bs, int log2bs) {
    const unsigned c = (unsigned)k & (unsigned)(bs - 1);
    const unsigned q = (unsigned)k >> log2bs;
    const unsigned br = log2bs ? (__builtin_bitreverse32(c) >> (32 - log2bs)) : 0u;
    return (br << 16) | q;
}
__device__ __forceinline__ int fpsg_index(unsigned rank, int log2bs) {   // inverse of fpsg_rank
    const unsigned c = log2bs ? (__builtin_bitreverse32(rank >> 16) >> (32 - log2bs)) : 0u;
    return (int)(((rank & 0xffffu) << log2bs) | c);
}
__device__ __forceinline__ float fpsg_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fpsg_max(float a, float b) {  // bare v_max_f32 (inputs are never NaN)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fpsg_wave_min(float v) {
    int out;
    asm volatile(
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\tv_readlane_b32 %1, %0, 63\n\ts_nop 3"
        : "+v"(v), "=s"(out));
    return __int_as_float(out);
}
__device__ __forceinline__ unsigned fpsg_part1by2(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// 15-bit tie rank of point k when bs = 1024 (every n > 1024): (bit-reversed (k mod 1024)) << 5 | k / 1024 -- the order of fpsg_rank, k < 32768
__device__ __forceinline__ unsigned fpsg_rank16(int k) { return ((__builtin_bitreverse32((unsigned)k & 1023u) >> 22) << 5) | ((unsigned)k >> 10); }
__device__ __forceinline__ int fpsg_index16(unsigned r16) { return (int)(((r16 & 31u) << 10) | (__builtin_bitreverse32(r16 >> 5) >> 22)); }
struct __attribute__((packed, aligned(4))) F3g { float x, y, z; };
__device__ __forceinline__ F3g fpsg_ld3(const float *base, int k) { return *reinterpret_cast<const F3g *>(base + (size_t)k * 3); }

// W = 16 waves, P slots per lane (16 | 32): N <= 1024 P.  LDS: header + 4 * 1024 P sort keys (the pick list reuses that region).
// MULTI (round 6): multi-pick rounds as in fps_bucket.hip -- every wave publishes its two best keys and its third-best VALUE, one pass of pair tests
// decides how long a prefix of the merged order goes out this round (up to 16 samples), see the loop below.
template <int P, int FM, bool MULTI>
__global__ void __launch_bounds__(1024) fps_big_kernel(int n, int m, int bs, int log2bs, int idxbits, const float *__restrict__ xyz_all,
                                                       float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ nx_all) {
    constexpr int W = 16, T = 1024, NPAD = T * P, G = P / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_raw);   // [3] rotating arg-max slots
    float *red = reinterpret_cast<float *>(smem_raw + 256);                          // [6][16] bbox partials
    unsigned *keys = reinterpret_cast<unsigned *>(smem_raw + kBigHdr);               // [NPAD] during the sort
    int *spick = reinterpret_cast<int *>(smem_raw + kBigHdr);                        // [m] afterwards
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cloud = blockIdx.x;
    const float *xyz = xyz_all + (size_t)cloud * n * 3;
    float *temp = temp_all ? temp_all + (size_t)cloud * n : nullptr;
    int *idx = idx_all + (size_t)cloud * m;
    float *nx = nx_all ? nx_all + (size_t)cloud * m * 3 : nullptr;
    const float INF = __builtin_inff();

    // ---- A. bounding box, Morton keys -------------------------------------------------------------------------------------------
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    for (int k = t; k < n; k += T) {
        const F3g p = fpsg_ld3(xyz, k);
        lx = fminf(lx, p.x); ly = fminf(ly, p.y); lz = fminf(lz, p.z);
        hx = fmaxf(hx, p.x); hy = fmaxf(hy, p.y); hz = fmaxf(hz, p.z);
    }
    lx = fpsg_wave_min(lx); ly = fpsg_wave_min(ly); lz = fpsg_wave_min(lz);
    hx = wave_max_f32(hx); hy = wave_max_f32(hy); hz = wave_max_f32(hz);
    if (lane == 0) { red[0 * 16 + wave] = lx; red[1 * 16 + wave] = ly; red[2 * 16 + wave] = lz;
                     red[3 * 16 + wave] = hx; red[4 * 16 + wave] = hy; red[5 * 16 + wave] = hz; }
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        lx = fminf(lx, red[0 * 16 + w]); ly = fminf(ly, red[1 * 16 + w]); lz = fminf(lz, red[2 * 16 + w]);
        hx = fmaxf(hx, red[3 * 16 + w]); hy = fmaxf(hy, red[4 * 16 + w]); hz = fmaxf(hz, red[5 * 16 + w]);
    }
    // spatially coherent is all the sort has to be: a 30-bit Morton code cut to the 32 - idxbits bits the key has room for
    const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
    const float scale = 1023.0f / ext;
    const int mshift = 30 - (32 - idxbits);
    for (int q = t; q < NPAD; q += T) {
        unsigned key = 0xffffffffu;  // padding sorts to the end (a real key never reaches it: its index field is < 2^idxbits - 1)
        if (q < n) {
            const F3g p = fpsg_ld3(xyz, q);
            const unsigned cx = (unsigned)fminf(fmaxf((p.x - lx) * scale, 0.f), 1023.f);
            const unsigned cy = (unsigned)fminf(fmaxf((p.y - ly) * scale, 0.f), 1023.f);
            const unsigned cz = (unsigned)fminf(fmaxf((p.z - lz) * scale, 0.f), 1023.f);
            const unsigned code = fpsg_part1by2(cx) | (fpsg_part1by2(cy) << 1) | (fpsg_part1by2(cz) << 2);
            key = ((code >> mshift) << idxbits) | (unsigned)q;
        }
        keys[q] = key;
    }
    __syncthreads();
    // ---- B. bitonic sort of NPAD 32-bit keys in LDS -------------------------------------------------------------------------------
    for (int k = 2; k <= NPAD; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < NPAD / 2; i += T) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b2 = a | j;
                const unsigned ka = keys[a], kb = keys[b2];
                const bool asc = (a & k) == 0;
                if ((ka > kb) == asc) { keys[a] = kb; keys[b2] = ka; }
            }
            __syncthreads();
        }
    }
    // ---- C. wave w owns the P consecutive buckets [w P, (w + 1) P) of the sorted order; slot i of lane l = point 64 (w P + i) + l ---
    float md[P];
    unsigned rkp[P / 2];     // 16-bit tie ranks of the slots' points, two per register (slot 2 i in the low half); 0xffff = padding
    const unsigned imask = (1u << idxbits) - 1u;
#pragma unroll
    for (int i = 0; i < P; i += 2) {
        const unsigned k0 = keys[(wave * P + i) * 64 + lane], k1 = keys[(wave * P + i + 1) * 64 + lane];
        const unsigned r0 = k0 == 0xffffffffu ? 0xffffu : fpsg_rank16((int)(k0 & imask)), r1 = k1 == 0xffffffffu ? 0xffffu : fpsg_rank16((int)(k1 & imask));
        rkp[i / 2] = r0 | (r1 << 16);
    }
#define G4D_RK(i) (((i) & 1) ? (rkp[(i) / 2] >> 16) : (rkp[(i) / 2] & 0xffffu))
    __syncthreads();  // the sort region becomes the pick list
    // bucket boxes: lane i holds the box of bucket i of the wave (P <= 32 lanes used)
    float blx = INF, bly = INF, blz = INF, bhx = -INF, bhy = -INF, bhz = -INF;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool ok = G4D_RK(i) != 0xffffu;
        const int k = ok ? fpsg_index16(G4D_RK(i)) : 0;
        const F3g p = fpsg_ld3(xyz, k);
        md[i] = ok ? (temp ? temp[k] : 1e10f) : -2.f;   // -2: below every real min-distance, never a candidate
        const float a0 = fpsg_wave_min(ok ? p.x : INF), a1 = fpsg_wave_min(ok ? p.y : INF), a2 = fpsg_wave_min(ok ? p.z : INF);
        const float a3 = wave_max_f32(ok ? p.x : -INF), a4 = wave_max_f32(ok ? p.y : -INF), a5 = wave_max_f32(ok ? p.z : -INF);
        if (lane == i) { blx = a0; bly = a1; blz = a2; bhx = a3; bhy = a4; bhz = a5; }
        asm volatile("" ::: "memory");   // one slot at a time: left alone the scheduler hoists all P coordinate loads (3 P registers) to the top
    }
    if (t == 0) { spick[0] = 0; slots[0] = 0ull; slots[1] = 0ull; slots[2] = 0ull; }
    __syncthreads();

    if constexpr (MULTI) {
        // ---- multi-pick rounds (the protocol of fps_bucket.hip's multi-pick loop, laid out for coordinates that live in global memory) -----------
        //   1. the round's samples (<= 16, in LDS) are tested against the wave's P boxes one sample at a time (lane l: bucket l), the swept buckets'
        //      rows are gathered ONCE per round (one L2 round trip per active half-group of 4 slots) and every sample that reaches them is applied;
        //   2. a swept wave recomputes its two best keys and an upper bound of its third-best value and publishes {key, v3}; barrier A;
        //   3. pair tests: lane (c, jj) holds (own candidate c, candidate jj), both points' coordinates fetched from the cloud by the index the key's
        //      rank encodes (one L2 round trip for all 32 candidates at once); rank = number of larger keys; a candidate stops the walk if a larger
        //      key's point would lower it, if a hidden key could outrank it (both keys of some wave above it, value not above that wave's third), or
        //      at value 0; accepted candidates write their coordinates and index at slot `rank`; barrier B;
        //   4. everybody reads the round's length.
        constexpr int KE = 16;
        unsigned long long *rec = reinterpret_cast<unsigned long long *>(smem_raw);   // [2 W] records of 32 bytes: key at +0, third-best value at +20
        float *recf = reinterpret_cast<float *>(smem_raw);
        unsigned *nstop = reinterpret_cast<unsigned *>(smem_raw + 1024);               // [2] the round's length (LDS atomic min), alternating
        float *res = reinterpret_cast<float *>(smem_raw + 1024 + 256);                 // [KE] the round's samples: x, y, z, value
        static_assert(2 * W * 32 <= 1024 && 1024 + 256 + KE * 16 <= kBigHdr, "fps_big multi-pick: the exchange areas must fit the LDS header");
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        if (lane < 2) { rec[(wave * 2 + lane) * 4] = 0ull; recf[(wave * 2 + lane) * 8 + 5] = -2.f; }
        if (t == 0) {
            const F3g p0 = fpsg_ld3(xyz, 0);
            nstop[0] = 0xffu; nstop[1] = 0xffu;
            res[0] = p0.x; res[1] = p0.y; res[2] = p0.z; res[3] = INF;
        }
        __syncthreads();
        int ns = 1, j = 1, rpar = 0;
        float gval = INF;
        // box tests of `cnt` samples + sweeps of the buckets they reach (pruning bound `bound`); returns the union of the swept buckets
        auto sweep = [&](int cnt, float bound) -> unsigned {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(&res[min(lane, KE - 1) * 4]);   // lane i: sample i
            unsigned am = 0u, U = 0u;
            for (int i = 0; i < cnt; ++i) {
                const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.x), i)), ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.y), i)),
                            az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.z), i));
                const float gx = fmaxf(fmaxf(blx - ax, ax - bhx), 0.f), gy = fmaxf(fmaxf(bly - ay, ay - bhy), 0.f), gz = fmaxf(fmaxf(blz - az, az - bhz), 0.f);
                const unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(lane < P && dist2<FM>(gx, gy, gz) < bound);
                am = lane == i ? mask : am;   // lane i keeps sample i's buckets
                U |= mask;
            }
            if (U == 0u) return 0u;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (((U >> (8 * g)) & 0xffu) == 0u) continue;   // wave-uniform
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (((U >> (8 * g + 4 * h)) & 0xfu) == 0u) continue;   // wave-uniform
                    F3g p[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        unsigned r = G4D_RK(8 * g + 4 * h + s4);
                        asm volatile("" : "+v"(r));   // (see the single-pick loop: keeps the row addresses from being hoisted and spilled)
                        p[s4] = fpsg_ld3(xyz, r == 0xffffu ? 0 : fpsg_index16(r));
                    }
                    for (int i = 0; i < cnt; ++i) {
                        const unsigned mi = ((unsigned)__builtin_amdgcn_readlane((int)am, i) >> (8 * g + 4 * h)) & 0xfu;
                        if (mi == 0u) continue;   // wave-uniform
                        const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.x), i)), ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.y), i)),
                                    az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv.z), i));
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            if ((mi >> s4) & 1u) {
                                const float dx = p[s4].x - ax, dy = p[s4].y - ay, dz = p[s4].z - az;
                                md[8 * g + 4 * h + s4] = fpsg_min(dist2<FM>(dx, dy, dz), md[8 * g + 4 * h + s4]);
                            }
                    }
                }
            }
            return U;
        };
        while (j < m) {
            if (sweep(ns, gval) != 0u) {   // wave-uniform
                // the lane's best and second-best slot (value, then smallest rank)
                float b1 = md[0];
#pragma unroll
                for (int i = 1; i < P; ++i) b1 = fpsg_max(b1, md[i]);
                unsigned r1 = 0xffffu;
#pragma unroll
                for (int i = 0; i < P; ++i) r1 = min(r1, md[i] == b1 ? G4D_RK(i) : 0xffffu);
                float b2 = -2.f;
#pragma unroll
                for (int i = 0; i < P; ++i) b2 = fpsg_max(b2, G4D_RK(i) == r1 ? -2.f : md[i]);   // ranks of real points are unique: drops exactly the best slot
                unsigned r2 = 0xffffu;
#pragma unroll
                for (int i = 0; i < P; ++i) r2 = min(r2, (md[i] == b2 && G4D_RK(i) != r1) ? G4D_RK(i) : 0xffffu);
                // the wave's two best keys and an upper bound of its third-best value
                const float c1v = wave_max_f32(b1);
                unsigned long long hit = __builtin_amdgcn_ballot_w64(b1 == c1v);
                unsigned c1r, c2r;
                int h1, h2;
                if (__builtin_popcountll(hit) == 1) { h1 = __builtin_ctzll(hit); c1r = (unsigned)__builtin_amdgcn_readlane((int)r1, h1); }
                else { c1r = wave_min_u32(b1 == c1v ? r1 : 0xffffu); h1 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(b1 == c1v && r1 == c1r) | (1ull << 63)); }
                const float v2 = lane == h1 ? b2 : b1;
                const unsigned q2 = lane == h1 ? r2 : r1;
                const float c2v = wave_max_f32(v2);
                hit = __builtin_amdgcn_ballot_w64(v2 == c2v);
                if (__builtin_popcountll(hit) == 1) { h2 = __builtin_ctzll(hit); c2r = (unsigned)__builtin_amdgcn_readlane((int)q2, h2); }
                else { c2r = wave_min_u32(v2 == c2v ? q2 : 0xffffu); h2 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(v2 == c2v && q2 == c2r) | (1ull << 63)); }
                const float v3w = wave_max_f32((lane == h1 || lane == h2) ? b2 : b1);   // (a lane owning both published points offers its second value: larger, safe)
                if (lane < 2) {
                    const float cvv = lane == 0 ? c1v : c2v;
                    const unsigned c16 = lane == 0 ? c1r : c2r;
                    const unsigned r32 = ((c16 >> 5) << 16) | (c16 & 31u);   // the 32-bit rank of fpsg_rank
                    rec[(wave * 2 + lane) * 4] = (cvv < 0.f || c16 == 0xffffu) ? 0ull : (((unsigned long long)__float_as_uint(cvv) << 32) | (unsigned)(~r32));
                    recf[(wave * 2 + lane) * 8 + 5] = v3w;
                }
            }
            __syncthreads();
            {
                const int c = lane >> 5, jj = lane & 31;
                const unsigned long long kj = rec[jj * 4];
                const int me = wave * 2 + c;
                const unsigned long long km = rec[me * 4];
                const float v3j = recf[(jj >> 1) * 16 + 5];
                int ij = fpsg_index(~(unsigned)kj, log2bs), im = fpsg_index(~(unsigned)km, log2bs);
                ij = (kj != 0ull && (unsigned)ij < (unsigned)n) ? ij : 0;   // (an empty slot reads point 0: never used)
                im = (km != 0ull && (unsigned)im < (unsigned)n) ? im : 0;
                const F3g pj = fpsg_ld3(xyz, ij), pm = fpsg_ld3(xyz, im);
                const float mv = __uint_as_float((unsigned)(km >> 32));
                const bool gt = kj > km;
                const bool aff = gt && !(dist2<FM>(pm.x - pj.x, pm.y - pj.y, pm.z - pj.z) >= mv);
                const unsigned long long gtm = __builtin_amdgcn_ballot_w64(gt), afm = __builtin_amdgcn_ballot_w64(aff);
                const unsigned gth = c ? (unsigned)(gtm >> 32) : (unsigned)gtm, afh = c ? (unsigned)(afm >> 32) : (unsigned)afm;
                const int rank = __builtin_popcount(gth);
                const unsigned both = gth & (gth >> 1) & 0x55555555u;      // bit 2 w: both keys of wave w are above this candidate
                const bool hid = ((both >> (jj & ~1)) & 1u) != 0u && !(mv > v3j);
                const unsigned long long hdm = __builtin_amdgcn_ballot_w64(hid);
                const unsigned hdh = c ? (unsigned)(hdm >> 32) : (unsigned)hdm;
                const bool zero = !(mv > 0.f);
                const bool bad = km == 0ull || afh != 0u || hdh != 0u || (zero && rank > 0);
                if (jj == 0) {
                    atomicMin(&nstop[rpar], bad ? (unsigned)rank : (zero ? (unsigned)rank + 1u : 0xffu));
                    if (rank < KE) {   // ranks are unique among real keys; empty slots all carry point 0
                        *reinterpret_cast<f32x4 *>(&res[rank * 4]) = (f32x4){pm.x, pm.y, pm.z, mv};
                        if (j + rank < m) spick[j + rank] = im;
                    }
                }
                if (t == 0) nstop[rpar ^ 1] = 0xffu;
            }
            __syncthreads();
            {
                int nn = __builtin_amdgcn_readfirstlane((int)nstop[rpar]);
                nn = max(1, min(min(nn, KE), m - j));
                gval = res[(nn - 1) * 4 + 3];
                ns = nn;
            }
            rpar ^= 1;
            j += ns;
        }
        // the reference's scratch holds the min-distances to every sample but the LAST one (sampling_gpu.cu:129-141): the final round's others
        if (temp && ns > 1) sweep(ns - 1, INF);
    } else {
    float x1, y1, z1;
    { const F3g p0 = fpsg_ld3(xyz, 0); x1 = p0.x; y1 = p0.y; z1 = p0.z; }
    float gval = INF;          // the last winner's value: no min-distance exceeds it
    float gb[G];               // cached (value, rank) candidate of each group of 8 slots
    unsigned gr[G];            // (16-bit ranks)
#pragma unroll
    for (int g = 0; g < G; ++g) { gb[g] = -2.f; gr[g] = 0xffffu; }
    float cval = -2.f;         // the wave's cached candidate
    unsigned crank = 0xffffffffu;
    for (int j = 1; j < m; ++j) {
        // 1. which buckets can change?  (same exact test as fps_bucket.hip: the gap to the box under the same dist2<FM>)
        const float gx = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f);
        const float gy = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f);
        const float gz = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
        const float dbox = dist2<FM>(gx, gy, gz);
        const unsigned active = (unsigned)__builtin_amdgcn_ballot_w64(lane < P && dbox < gval);
        if (active != 0u) {   // wave-uniform
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned ag = (active >> (8 * g)) & 0xffu;
                if (ag == 0u) continue;   // wave-uniform
                // 2. the swept slots are updated; a half-group's 4 rows are fetched together (one L2 round trip per active half; rows of
                //    un-swept slots are loaded and dropped -- four loads in flight cost less than one dependent round trip each)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned ah = (ag >> (4 * h)) & 0xfu;
                    if (ah == 0u) continue;   // wave-uniform
                    F3g p[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        unsigned r = G4D_RK(8 * g + 4 * h + s);
                        asm volatile("" : "+v"(r));   // the ranks never change: left visible, the row addresses (2 registers per slot) are hoisted out of the round loop and spilled
                        p[s] = fpsg_ld3(xyz, r == 0xffffu ? 0 : fpsg_index16(r));
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if ((ah >> s) & 1u) {
                            const float dx = p[s].x - x1, dy = p[s].y - y1, dz = p[s].z - z1;
                            md[8 * g + 4 * h + s] = fpsg_min(dist2<FM>(dx, dy, dz), md[8 * g + 4 * h + s]);
                        }
                    }
                }
                // 3a. the group's candidate: largest value, smallest rank among the slots holding it
                float tv[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) tv[s] = md[8 * g + s];
#pragma unroll
                for (int w = 8; w > 1; w >>= 1)
#pragma unroll
                    for (int s = 0; s < w / 2; ++s) tv[s] = fpsg_max(tv[s], tv[s + w / 2]);
                unsigned tr[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) tr[s] = (md[8 * g + s] == tv[0]) ? G4D_RK(8 * g + s) : 0xffffu;
#pragma unroll
                for (int w = 8; w > 1; w >>= 1)
#pragma unroll
                    for (int s = 0; s < w / 2; ++s) tr[s] = min(tr[s], tr[s + w / 2]);
                gb[g] = tv[0];
                gr[g] = tr[0];
            }
            // 3b. lane candidate = best group candidate; then the wave's
            float b = gb[0];
#pragma unroll
            for (int g = 1; g < G; ++g) b = fpsg_max(b, gb[g]);
            unsigned r = 0xffffu;
#pragma unroll
            for (int g = 0; g < G; ++g) r = min(r, gb[g] == b ? gr[g] : 0xffffu);
            cval = wave_max_f32(b);
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(b == cval);
            unsigned c16;
            if (__builtin_popcountll(hit) == 1) c16 = (unsigned)__builtin_amdgcn_readlane((int)r, __builtin_ctzll(hit));
            else c16 = wave_min_u32(b == cval ? r : 0xffffu);
            crank = c16 == 0xffffu ? 0xffffffffu : (((c16 >> 5) << 16) | (c16 & 31u));   // the exchange key carries the 32-bit rank of fpsg_rank
        }
        // 4. workgroup arg-max: one LDS atomic max per wave on a rotating slot, one barrier, one read
        if (lane == 0)
            atomicMax(&slots[j % 3], ((unsigned long long)__float_as_uint(fmaxf(cval, 0.f)) << 32) | (unsigned)(~crank));
        __syncthreads();
        const unsigned long long best = slots[j % 3];
        if (t == 0) slots[(j + 2) % 3] = 0ull;
        const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(best >> 32));
        const unsigned rank = ~(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)best);
        gval = __uint_as_float(bhi);
        int old = fpsg_index(rank, log2bs);
        old = (unsigned)old < (unsigned)n ? old : 0;     // (no candidate anywhere -- all-NaN clouds: the reference picks index 0 too)
        const F3g pw = fpsg_ld3(xyz, old);               // the winner's coordinates: one broadcast load (L2)
        x1 = pw.x; y1 = pw.y; z1 = pw.z;
        if (t == 0) spick[j] = old;
    }
    }   // single-pick rounds
    __syncthreads();
    for (int j = t; j < m; j += T) {
        const int k = spick[j];
        idx[j] = k;
        if (nx) { const F3g p = fpsg_ld3(xyz, k); nx[j * 3 + 0] = p.x; nx[j * 3 + 1] = p.y; nx[j * 3 + 2] = p.z; }
    }
    if (temp) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (G4D_RK(i) != 0xffffu) temp[fpsg_index16(G4D_RK(i))] = md[i];
    }
#undef G4D_RK
}

template <int P>
static int launch_big(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    const size_t lds = kBigHdr + (size_t)4 * 1024 * P;   // m <= n <= 1024 P: the pick list fits the sort region
    int idxbits = 1;
    while ((1 << idxbits) < n + 1) ++idxbits;            // 2^idxbits > n: the padding key 0xffffffff is never a real key
    static unsigned long long attr[6] = {0, 0, 0, 0, 0, 0};
    const int mode = distance_contraction();
    const void *k = nullptr;
    static const int multi = getenv("G4D_FPS_BIG_MULTI") ? atoi(getenv("G4D_FPS_BIG_MULTI")) : 1;   // 0: one pick per round (the round-3 loop)
    G4D_WITH_FM(mode, k = multi ? reinterpret_cast<const void *>(fps_big_kernel<P, FM, true>) : reinterpret_cast<const void *>(fps_big_kernel<P, FM, false>))
    if (const int rc = ensure_dynamic_lds(k, 160 * 1024 - 1024, attr[(mode == 0 ? 0 : (mode == 1 ? 1 : 2)) + (multi ? 3 : 0)], "g4d_fps_f32(large)")) return rc;
    if (multi) { G4D_WITH_FM(mode, hipLaunchKernelGGL((fps_big_kernel<P, FM, true>), dim3(b), dim3(1024), lds, s, n, m, bs, log2bs, idxbits, xyz, temp, idx, nx)) }
    else { G4D_WITH_FM(mode, hipLaunchKernelGGL((fps_big_kernel<P, FM, false>), dim3(b), dim3(1024), lds, s, n, m, bs, log2bs, idxbits, xyz, temp, idx, nx)) }
    return check_launch("g4d_fps_f32(large)");
}

// Called by fps_impl (fps.hip) for 8192 < n <= 32768 when the register-resident kernel does not take the shape.  -1: not covered.
int fps_big_dispatch(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    static const int on = getenv("G4D_FPS_BIG") ? atoi(getenv("G4D_FPS_BIG")) : 1;
    if (!on || n <= 8192 || n > 32768 || m > n || bs != 1024 || log2bs != 10) return -1;   // (bs = 1024 for every n >= 1024: the 16-bit rank packing relies on it)
    if (n <= 16384) return launch_big<16>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
    return launch_big<32>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
}

}  // namespace g4d
