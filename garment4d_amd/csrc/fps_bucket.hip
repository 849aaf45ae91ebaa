// Bucketed furthest point sampling: identical output to fps.hip / the reference, ~23 % fewer cycles per round at
// N = 8192 (1.085 -> 0.84 us per round, B = 8) and the first FPS here whose work per round does not grow with N.
//
// In round j only points closer to the new sample than their current min-distance change, and every min-distance
// is <= g_j, the value of the sample just selected (the global max).  The cloud is sorted along a Morton curve
// (bitonic sort in LDS, once per launch) and cut into buckets of 64 consecutive points -- one VGPR "slot" of one
// wave; a wave owns P CONSECUTIVE buckets of the Morton order, so the ~14 buckets a round touches sit in 2-3 waves and
// the other waves only republish their cached candidate (dealing them round-robin makes ~14 waves pay the fixed
// arg-max cost every round: 0.81 instead of 0.75 us per round).  Each wave keeps the boxes of its buckets lane-distributed.  A round is then
//
//   1. lane i tests box i against the new sample: d_box = dist2<FM>(gx, gy, gz) with gx = the gap between the sample
//      and the box along x, ... evaluated with the SAME fp32 operations (same contraction shape FM) as the point distance.
//      gx = max(lo - x1, x1 - hi, 0) <= |px - x1| holds for the ROUNDED differences (rounding a difference is monotone),
//      and dist2<FM> is monotone non-decreasing in each |argument| whether its products are rounded separately or fused
//      (fma(a, a, c) = round(a*a + c) is monotone in |a| and in c), so d(p) >= d_box holds for the ROUNDED values of every
//      point p in the box, exactly, without any epsilon, under every contraction mode: if d_box >= g_j the sweep could not lower a single min-distance and the bucket
//      is skipped -- pruning is bit-exact (measured: 14 of 128 buckets swept per round at N=8192, M=1024);
//   2. a wave with no active bucket (about half of them in a typical round) republishes its cached candidate; the
//      others sweep their active buckets and redo ONE arg-max: balanced max / rank-select trees over the lane's 8
//      points, then a DPP wave arg-max;
//   3. the workgroup exchange is one LDS atomic max per wave on a rotating 64-bit slot, one barrier, one read; the
//      winner's coordinates come from the LDS SoA copy of the cloud.
//
// Tie-break = the reference's (smallest bit-reversed (k mod bs), then smallest k) through the same rank key, computed
// from ORIGINAL indices, so the permutation is invisible in the output.  What bounds the round now is plain
// dependent issue: a lone wave retires one dependent instruction per ~6 cycles, a DPP step costs 19, an LDS round
// trip 64, a 16-wave barrier 60 (scripts/micro/clock.hip); the slowest wave's ~190-instruction path is the round.
#include "g4d_common.h"
#include "ball_grid_build.h"

namespace g4d {

constexpr int kFpsHdr = 2048;   // LDS header in front of the sort keys / SoA cloud: exchange slots, candidate records, round results

// --- pieces shared with fps.hip (kept local: both files are self-contained translation units)
__device__ __forceinline__ unsigned fpsb_rank(int k, int bs, int log2bs) {
    const unsigned c = (unsigned)k & (unsigned)(bs - 1);
    const unsigned q = (unsigned)k >> log2bs;
    const unsigned br = log2bs ? (__builtin_bitreverse32(c) >> (32 - log2bs)) : 0u;
    return (br << 16) | q;
}

__device__ __forceinline__ float fpsb_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int W>
__device__ __forceinline__ unsigned long long fpsb_row_max_u64(unsigned long long key) {
#define G4D_STEP(CTRL)                                                                                       \
    {                                                                                                        \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)key, CTRL, 0xf, 0xf, true);        \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(key >> 32), CTRL, 0xf, 0xf, true); \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                    \
        key = o > key ? o : key;                                                                             \
    }
    G4D_STEP(0x111)
    G4D_STEP(0x112)
    if constexpr (W >= 8) G4D_STEP(0x114)
    if constexpr (W >= 16) G4D_STEP(0x118)
#undef G4D_STEP
    return key;
}

__device__ __forceinline__ float fmax_raw(float a, float b) {  // bare v_max_f32: no canonicalising pre-pass (inputs are never NaN)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float wave_min_f32(float v) {
    int out;
    asm volatile(
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %1, %0, 63\n\t"
        "s_nop 3"
        : "+v"(v), "=s"(out));
    return __int_as_float(out);
}

__device__ __forceinline__ unsigned part1by2(unsigned v) {  // spread the low 10 bits: b9..b0 -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// W waves, P buckets (slots) per wave; handles N <= 64*W*P points.  LDS: max(8*Npad sort keys, 12*N SoA) + 512.
// KMAX > 1: the multi-pick round loop (below); KMAX = 1: one sample per round.
// SOA: the LDS copy of the cloud (96 KB at n = 8192) that the winner's coordinates are read from.  SOA = false (multi-pick rounds only,
// round 5): the coordinates come from the owning lane's REGISTERS -- the tie rank carries the slot number in its low bits, the wave's
// best / second-best slot is a wave-uniform register index (s_set_gpr_idx_on + v_mov: the point arrays live in ONE 4 P-wide vector so
// that the compiler indexes it instead of expanding a select chain), three v_readlane -- and the workgroup needs 70 KB of LDS (sort keys
// + pick list) instead of 104: a 72 KB shared-MLP workgroup of another stream fits beside it on the same CU.
template <int W, int P, int FM, int KMAX = 1, bool SOA = true>
__device__ __forceinline__ void fps_bucket_body(int n, int m, int bs, int log2bs, int deal_kcap, int pick_off, const float *__restrict__ xyz_all,
                                                float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ nx_all, int cloud) {
    constexpr int T = 64 * W, NPAD = T * P;
    static_assert(SOA || KMAX > 1, "the register form exists for the multi-pick loop only");
    constexpr int SB = SOA ? 0 : (P == 4 ? 2 : P == 8 ? 3 : P == 16 ? 4 : 5);   // slot bits below the tie rank (register form)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_raw);         // [2][16] candidate keys
    float *red = reinterpret_cast<float *>(smem_raw + 256);                                // [6][16] bbox partials
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem_raw + kFpsHdr);   // [NPAD] during the sort
    float *sx = reinterpret_cast<float *>(smem_raw + kFpsHdr);                                // SoA cloud afterwards
    int *spick = reinterpret_cast<int *>(smem_raw + pick_off);                             // [m] the samples, written out once at the end
    float *sy = sx + n;
    float *sz = sy + n;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kcap = deal_kcap >> 8, deal = deal_kcap & 0xff;   // (two small ints in one kernel argument)
    const float *xyz = xyz_all + (size_t)cloud * n * 3;
    float *temp = temp_all ? temp_all + (size_t)cloud * n : nullptr;
    int *idx = idx_all + (size_t)cloud * m;
    float *nx = nx_all ? nx_all + (size_t)cloud * m * 3 : nullptr;   // optional: the selected points themselves (gather fused)
    const float INF = __builtin_inff();

    // ---- A. Morton keys of the cloud -------------------------------------------------------------------------
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    for (int k = t; k < n; k += T) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
        hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
    }
    lx = wave_min_f32(lx); ly = wave_min_f32(ly); lz = wave_min_f32(lz);
    hx = wave_max_f32(hx); hy = wave_max_f32(hy); hz = wave_max_f32(hz);
    if (lane == 0) { red[0 * 16 + wave] = lx; red[1 * 16 + wave] = ly; red[2 * 16 + wave] = lz;
                     red[3 * 16 + wave] = hx; red[4 * 16 + wave] = hy; red[5 * 16 + wave] = hz; }
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        lx = fminf(lx, red[0 * 16 + w]); ly = fminf(ly, red[1 * 16 + w]); lz = fminf(lz, red[2 * 16 + w]);
        hx = fmaxf(hx, red[3 * 16 + w]); hy = fmaxf(hy, red[4 * 16 + w]); hz = fmaxf(hz, red[5 * 16 + w]);
    }
    // the sort only has to be spatially coherent, not exact: any finite scale works (NaN/inf coordinates fall into cell 0)
    const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
    const float scale = 1023.0f / ext;
    for (int q = t; q < NPAD; q += T) {
        unsigned long long key = ~0ull;  // padding sorts to the end
        if (q < n) {
            const float x = xyz[q * 3 + 0], y = xyz[q * 3 + 1], z = xyz[q * 3 + 2];
            const unsigned cx = (unsigned)fminf(fmaxf((x - lx) * scale, 0.f), 1023.f);
            const unsigned cy = (unsigned)fminf(fmaxf((y - ly) * scale, 0.f), 1023.f);
            const unsigned cz = (unsigned)fminf(fmaxf((z - lz) * scale, 0.f), 1023.f);
            const unsigned code = part1by2(cx) | (part1by2(cy) << 1) | (part1by2(cz) << 2);
            key = ((unsigned long long)code << 32) | (unsigned)q;
        }
        keys[q] = key;
    }
    __syncthreads();
    // ---- B. bitonic sort of NPAD 64-bit keys in LDS ----------------------------------------------------------------
    for (int k = 2; k <= NPAD; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < NPAD / 2; i += T) {
                // i-th compare-exchange pair of this stage: lower index a with bit j clear
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b2 = a | j;
                const unsigned long long ka = keys[a], kb = keys[b2];
                const bool asc = (a & k) == 0;
                if ((ka > kb) == asc) { keys[a] = kb; keys[b2] = ka; }
            }
            __syncthreads();
        }
    }
    // ---- C. deal buckets of 64 sorted points to the waves, load the points into registers ----------------------------
    int pk[P];                     // original index of the slot's point (or -1)
    typedef float fvec __attribute__((ext_vector_type(4 * P)));
    fvec pv;                       // [0, P) x, [P, 2P) y, [2P, 3P) z, [3P, 4P) running min-distance of the lane's P points
#define px(i) pv[(i)]
#define py(i) pv[P + (i)]
#define pz(i) pv[2 * P + (i)]
#define md(i) pv[3 * P + (i)]
#pragma unroll
    for (int i = 0; i < P; ++i) {
        // buckets are dealt to the waves in runs of `deal` consecutive (Morton-adjacent) buckets: slot i of wave w holds bucket
        // ((i / deal) * W + w) * deal + i % deal.  deal = 1 spreads the ~14 buckets a round touches over ~14 waves (every one of
        // them pays the fixed arg-max cost), deal = P keeps them in 2-3 waves while the others republish their cached candidate
        const int q = (((i / deal) * W + wave) * deal + (i % deal)) * 64 + lane;
        const unsigned long long key = keys[q];
        pk[i] = (key == ~0ull) ? -1 : (int)(unsigned)key;
    }
    __syncthreads();  // everybody has read its keys: the region becomes the SoA cloud
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool ok = pk[i] >= 0;
        const int k = ok ? pk[i] : 0;
        px(i) = ok ? xyz[k * 3 + 0] : 0.f;
        py(i) = ok ? xyz[k * 3 + 1] : 0.f;
        pz(i) = ok ? xyz[k * 3 + 2] : 0.f;
        md(i) = ok ? (temp ? temp[k] : 1e10f) : -2.f;  // -2: below every real min-distance, never a candidate
        if constexpr (SOA) { if (ok) { sx[k] = px(i); sy[k] = py(i); sz[k] = pz(i); } }
    }
    const float x0 = xyz[0], y0 = xyz[1], z0 = xyz[2];   // the first sample is point 0 (sampling_gpu.cu:118)
    // bucket boxes, lane l holds the box of bucket l % P (64 / P copies: the multi-pick loop tests 64 / P samples at once); per-slot tie ranks
    float blx = INF, bly = INF, blz = INF, bhx = -INF, bhy = -INF, bhz = -INF;
    unsigned rk[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool ok = pk[i] >= 0;
        rk[i] = ok ? ((fpsb_rank(pk[i], bs, log2bs) << SB) | (SOA ? 0u : (unsigned)i)) : 0xffffffffu;
        const float a0 = wave_min_f32(ok ? px(i) : INF), a1 = wave_min_f32(ok ? py(i) : INF), a2 = wave_min_f32(ok ? pz(i) : INF);
        const float a3 = wave_max_f32(ok ? px(i) : -INF), a4 = wave_max_f32(ok ? py(i) : -INF), a5 = wave_max_f32(ok ? pz(i) : -INF);
        if ((lane & (P - 1)) == i) { blx = a0; bly = a1; blz = a2; bhx = a3; bhy = a4; bhz = a5; }   // lane l holds the box of bucket l % P
    }
    if (t == 0) { spick[0] = 0; slots[0] = 0ull; slots[1] = 0ull; slots[2] = 0ull; }
    if (KMAX > 1 && t == 0) {   // multi-pick rounds: the first round's only sample (point 0) in the samples' exchange area
        float *res0 = reinterpret_cast<float *>(smem_raw + 1024 + 256);
        res0[0] = x0; res0[1] = y0; res0[2] = z0; res0[3] = INF;
    }
    // register form: the boxes wait in LDS (the sort keys are dead: everybody read its keys before the barrier above) and are re-read at the
    // head of every round -- six registers that are NOT live while the top-two trees run (the kernel is held to 72 VGPRs -- 64 until round 6 -- so that a
    // 256-register shared-MLP wave of another stream shares the SIMD with four FPS waves); a bucket's box = 32 bytes, 8 distinct per wave
    float *sbox = reinterpret_cast<float *>(smem_raw + kFpsHdr) + (wave * P + (lane & (P - 1))) * 8;
    if constexpr (!SOA) {
        typedef float f32x4b __attribute__((ext_vector_type(4)));
        typedef float f32x2b __attribute__((ext_vector_type(2)));
        if (lane < P) {
            *reinterpret_cast<f32x4b *>(sbox) = (f32x4b){blx, bly, blz, bhx};
            *reinterpret_cast<f32x2b *>(sbox + 4) = (f32x2b){bhy, bhz};
        }
        if (lane < 2) {   // = rec[..]: "no candidate" until the wave's first sweep publishes one (third-best value: none)
            slots[(wave * 2 + lane) * 4] = 0ull;
            reinterpret_cast<float *>(smem_raw)[(wave * 2 + lane) * 8 + 5] = -2.f;
        }
    }
    __syncthreads();

#ifdef G4D_FPS_DEBUG
    unsigned dbg_active = 0;
    long long dbg_t[5] = {0, 0, 0, 0, 0}, dbg_c0 = 0;
#define G4D_STAMP(i) { const long long now_ = clock64(); dbg_t[i] += now_ - dbg_c0; dbg_c0 = now_; }
#else
#define G4D_STAMP(i)
#endif
    if constexpr (KMAX > 1) {
        // ---- multi-pick rounds (round 3) ---------------------------------------------------------------------------------------
        // FPS is a chain of m - 1 dependent arg-maxes, but consecutive winners are usually INDEPENDENT: let c1 > c2 > ... be the points
        // in the (value, tie-rank) order of the current min-distances.  Sampling c1 changes md[k] only where d(k, c1) < md[k]; if
        // d(c2, c1) >= md[c2] then c2 keeps its key, every other key stays or drops and c1's own drops to 0, so the NEXT arg-max is c2 --
        // exactly, ties included (keys are a total order) -- without looking at the updated distances.  By induction a whole prefix
        // c1 .. cn goes out in one round as long as each c_i has md > 0 and is unaffected (d(c_i, c_j) >= md[c_i], the very comparison
        // the sweep's min would make, same dist2<FM>) by every c_j before it.  On a uniform 8192-point cloud 1023 picks take ~200
        // rounds instead of 1023 (numpy emulation; the far-apart maxima of neighbouring "holes" of the sample set are independent).
        //   * every wave keeps its TOP TWO keys (recomputed only when one of its buckets was swept) and publishes both: 2 W keys,
        //     one 16-byte LDS store per wave, one barrier;
        //   * every wave then walks the merged order of the 2 W keys redundantly (same data, same code -> same result, no second
        //     barrier): lane l holds key l with its point's coordinates (LDS SoA cloud); arg-max by DPP, accept, mark the candidates
        //     the accepted point would change as dirty; the walk stops at a dirty or zero-valued key, after KE picks, or at a key
        //     that one of a wave's UNPUBLISHED keys could outrank (both of that wave's keys are above it and its value does not exceed the wave's
        //     third-best value, published along; until round 6: after any wave's second key -- half of all stops on a uniform cloud);
        //   * next round: box tests of all accepted samples at once (lane l: bucket l % P against sample l / P), sweeps of the active
        //     (sample, bucket) pairs (min is order-independent), pruning bound = the value of the LAST accepted key (every remaining
        //     min-distance is <= it).
        constexpr int SPP = 64 / P;                           // samples one box-test pass covers (lane l: bucket l % P against sample p0 + l / P)
        constexpr int KE0 = KMAX;                             // samples per round (compile-time bound; round 6: up to two passes of SPP)
        constexpr int NK = 2 * W;                            // keys per round (<= 32)
        const int KE = min(min(KE0, NK), max(1, kcap));      // run-time cap (tuning hook G4D_FPS_KCAP); never more than the keys a round has -- with
                                                             // W = 4 all 8 candidates can be clean, and nobody would have written samples 8 .. 15
        static_assert(NK * 32 <= 1024 && 1024 + 256 + KE0 * 16 <= kFpsHdr, "fps multi-pick: the exchange areas must fit the LDS header");
        static_assert(NK <= 64 && KE0 >= 1, "fps multi-pick: key / sample counts must fit a wave");
        unsigned long long *rec = reinterpret_cast<unsigned long long *>(smem_raw);    // [2 W] candidate records of 32 bytes: key, x, y, z (the bbox partials are dead)
        unsigned *nstop = reinterpret_cast<unsigned *>(smem_raw + 1024);                // [2] the round's length: LDS atomic min over the candidates' stop positions (two slots, alternating)
        float *res = reinterpret_cast<float *>(smem_raw + 1024 + 256);                  // [KE0] the round's samples in rank order: x, y, z, value
        const int ls = lane / P;                             // which sample of a pass this lane tests its box against
        typedef float f32x4 __attribute__((ext_vector_type(4)));   // (the round's samples live in LDS, `res`; round 1: the one sample is point 0, written before the barrier above)
        int ns = 1, j = 1, rpar = 0;
        if (t == 0) { nstop[0] = 0xffu; nstop[1] = 0xffu; }   // (ordered before the first use by the barriers of round 1)
        float gval = INF;
        // this wave's best (lane 0) and second-best (lane 1) candidate: key (0 = none) and the point's coordinates; kept until a bucket of the wave is swept
        // (register form: the record in LDS IS the cache -- an unswept wave simply leaves its two records alone)
        unsigned long long rkey = 0ull;
        float rx = x0, ry = y0, rz = z0;
        float rv3 = -2.f;   // an upper bound of the wave's THIRD-best value (round 6): what its unpublished keys cannot exceed
#ifdef G4D_FPS_DEBUG
        long long dbg_rounds = 0, dbg_stop[4] = {0, 0, 0, 0}, dbg_ph[4] = {0, 0, 0, 0}, dbg_actw = 0, dbg_a[4] = {0, 0, 0, 0}, dbg_pairs = 0, dbg_c1 = 0;   // stop: dirty | zero | cap | second key
#endif
        while (j < m) {
#ifdef G4D_FPS_DEBUG
            dbg_c0 = clock64(); ++dbg_rounds;
#endif
            // 1. which (sample, bucket) pairs can change anything?
            if constexpr (!SOA) {
                typedef float f32x4b __attribute__((ext_vector_type(4)));
                typedef float f32x2b __attribute__((ext_vector_type(2)));
                const f32x4b b0 = *reinterpret_cast<const volatile f32x4b *>(sbox);
                const f32x2b b1_ = *reinterpret_cast<const volatile f32x2b *>(sbox + 4);
                blx = b0.x; bly = b0.y; blz = b0.z; bhx = b0.w; bhy = b1_.x; bhz = b1_.y;
            }
            bool swept = false;
            for (int p0 = 0; p0 < ns; p0 += SPP) {   // wave-uniform: one pass per SPP samples of the round
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(&res[min(p0 + ls, KE0 - 1) * 4]);   // (lanes of samples >= ns read stale slots: masked below)
                const float xs = sv.x, ys = sv.y, zs = sv.z;
                const float gx = fmaxf(fmaxf(blx - xs, xs - bhx), 0.f);
                const float gy = fmaxf(fmaxf(bly - ys, ys - bhy), 0.f);
                const float gz = fmaxf(fmaxf(blz - zs, zs - bhz), 0.f);
                const float dbox = dist2<FM>(gx, gy, gz);
                const unsigned long long active = __builtin_amdgcn_ballot_w64(p0 + ls < ns && dbox < gval);
                if (active == 0ull) continue;   // wave-uniform
                swept = true;
                // 2. sweeps
                const int cnt = min(SPP, ns - p0);
                for (int i = 0; i < cnt; ++i) {
                    const unsigned mi = (unsigned)(active >> (i * P)) & (P >= 32 ? 0xffffffffu : ((1u << (P & 31)) - 1u));
                    if (mi == 0u) continue;
                    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), i * P));
                    const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys), i * P));
                    const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs), i * P));
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        if ((mi >> q) & 1u) {
                            const float dx = px(q) - ax, dy = py(q) - ay, dz = pz(q) - az;
                            md(q) = fpsb_min(dist2<FM>(dx, dy, dz), md(q));
                        }
                    }
                }
#ifdef G4D_FPS_DEBUG
                dbg_pairs += __builtin_popcountll(active);
#endif
            }
            if (swept) {   // wave-uniform
#ifdef G4D_FPS_DEBUG
                { const long long now_ = clock64(); dbg_a[0] += now_ - dbg_c0; dbg_c1 = now_; }
#endif
                // 3. the lane's best and second-best slot (value, then smallest rank), then the wave's
                float tv[P];
#pragma unroll
                for (int i = 0; i < P; ++i) tv[i] = md(i);
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tv[i] = fmax_raw(tv[i], tv[i + w / 2]);
                const float b1 = tv[0];
                unsigned tr[P];
#pragma unroll
                for (int i = 0; i < P; ++i) tr[i] = (md(i) == b1) ? rk[i] : 0xffffffffu;
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tr[i] = min(tr[i], tr[i + w / 2]);
                const unsigned r1 = tr[0];
#pragma unroll
                for (int i = 0; i < P; ++i) tv[i] = (rk[i] == r1) ? -2.f : md(i);   // ranks of real points are unique: this drops exactly the best slot
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tv[i] = fmax_raw(tv[i], tv[i + w / 2]);
                const float b2 = tv[0];
#pragma unroll
                for (int i = 0; i < P; ++i) tr[i] = (md(i) == b2 && rk[i] != r1) ? rk[i] : 0xffffffffu;
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tr[i] = min(tr[i], tr[i + w / 2]);
                const unsigned r2 = tr[0];
#ifdef G4D_FPS_DEBUG
                { const long long now_ = clock64(); dbg_a[1] += now_ - dbg_c1; dbg_c1 = now_; }
#endif
                const float c1v = wave_max_f32(b1);
                unsigned long long hit = __builtin_amdgcn_ballot_w64(b1 == c1v);
                unsigned c1r;
                int h1;
                if (__builtin_popcountll(hit) == 1) {
                    h1 = __builtin_ctzll(hit);
                    c1r = (unsigned)__builtin_amdgcn_readlane((int)r1, h1);
                } else {
                    c1r = wave_min_u32(b1 == c1v ? r1 : 0xffffffffu);
                    h1 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(b1 == c1v && r1 == c1r) | (1ull << 63));
                }
#ifdef G4D_FPS_DEBUG
                { const long long now_ = clock64(); dbg_a[2] += now_ - dbg_c1; dbg_c1 = now_; }
#endif
                const float v2 = lane == h1 ? b2 : b1;
                const unsigned q2 = lane == h1 ? r2 : r1;
                const float c2v = wave_max_f32(v2);
                hit = __builtin_amdgcn_ballot_w64(v2 == c2v);
                unsigned c2r;
                int h2;
                if (__builtin_popcountll(hit) == 1) {
                    h2 = __builtin_ctzll(hit);
                    c2r = (unsigned)__builtin_amdgcn_readlane((int)q2, h2);
                } else {
                    c2r = wave_min_u32(v2 == c2v ? q2 : 0xffffffffu);
                    h2 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(v2 == c2v && q2 == c2r) | (1ull << 63));
                }
                // the wave's third-best VALUE, or rather an upper bound of it: the best value of every lane once the two published points are gone
                // (a lane that owns both published points offers its second value instead of its unknown third: larger, hence safe)
                const float v3w = wave_max_f32((lane == h1 || lane == h2) ? b2 : b1);
                if constexpr (SOA) rv3 = v3w;   // (register form: the record in LDS is the cache, no register lives across the round)
                {
                    const float cvv = lane == 0 ? c1v : c2v;
                    const unsigned crr = lane == 0 ? c1r : c2r;
                    rkey = (cvv < 0.f || crr == 0xffffffffu) ? 0ull : (((unsigned long long)__float_as_uint(cvv) << 32) | (unsigned)(~crr));
                    if constexpr (SOA) {
                        const unsigned ccls = log2bs ? (__builtin_bitreverse32(crr >> 16) >> (32 - log2bs)) : 0u;
                        const int ci = (rkey && lane < 2) ? (int)(((crr & 0xffffu) << log2bs) | ccls) : 0;
                        rx = sx[ci]; ry = sy[ci]; rz = sz[ci];
                    } else {
                        // the two candidates' coordinates out of their owners' registers: slot = the rank's low bits (wave-uniform index)
                        const int s1 = (int)(c1r & (unsigned)(P - 1)), s2 = (int)(c2r & (unsigned)(P - 1));
                        const float ax1 = pv[s1], ay1 = pv[P + s1], az1 = pv[2 * P + s1];
                        const float ax2 = pv[s2], ay2 = pv[P + s2], az2 = pv[2 * P + s2];
                        const float bx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax1), h1)), bx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax2), h2));
                        const float by1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay1), h1)), by2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay2), h2));
                        const float bz1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az1), h1)), bz2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az2), h2));
                        rx = lane == 0 ? bx1 : bx2; ry = lane == 0 ? by1 : by2; rz = lane == 0 ? bz1 : bz2;
                        if (lane < 2) {   // publish here: the record doubles as the wave's cache
                            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<u64x2 *>(&rec[(wave * 2 + lane) * 4]) = (u64x2){rkey, ((unsigned long long)__float_as_uint(ry) << 32) | __float_as_uint(rx)};
                            typedef float f32x2b __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<f32x2b *>(reinterpret_cast<float *>(rec) + (wave * 2 + lane) * 8 + 4) = (f32x2b){rz, v3w};
                        }
                    }
                }
#ifdef G4D_FPS_DEBUG
                { const long long now_ = clock64(); dbg_a[3] += now_ - dbg_c1; dbg_c1 = now_; }
#endif
            }
#ifdef G4D_FPS_DEBUG
            { const long long now_ = clock64(); dbg_ph[0] += now_ - dbg_c0; dbg_c0 = now_; dbg_actw += swept; }
#endif
            // 4. publish the two candidates as records {key, x, y, z}; barrier A
            if (SOA && lane < 2) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2b __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u64x2 *>(&rec[(wave * 2 + lane) * 4]) = (u64x2){rkey, ((unsigned long long)__float_as_uint(ry) << 32) | __float_as_uint(rx)};
                *reinterpret_cast<f32x2b *>(reinterpret_cast<float *>(rec) + (wave * 2 + lane) * 8 + 4) = (f32x2b){rz, rv3};
            }
            __syncthreads();
#ifdef G4D_FPS_DEBUG
            { const long long now_ = clock64(); dbg_ph[1] += now_ - dbg_c0; dbg_c0 = now_; }
#endif
            // 5. every wave ranks ITS two candidates against all 2 W and tests them for independence, in one pass: lane (c = l / 32, jj = l % 32)
            //    holds the pair (own candidate c, candidate jj).  rank = number of larger keys; bad = some larger key's point would lower this
            //    candidate's min-distance (d < value: the comparison the sweep's min makes).  The sequential walk (take keys in descending
            //    order; stop at a candidate that an earlier pick affects, whose value is 0, or that a hidden key could outrank) only ever tests a
            //    candidate against ALL larger keys, so the flags of all candidates can be computed independently, here by 16 waves at once.
            {
                const int c = lane >> 5, jj = lane & 31;
                const float *recf = reinterpret_cast<const float *>(rec);
                const unsigned long long kj = jj < NK ? rec[jj * 4] : 0ull;
                const float jx = recf[jj * 8 + 2], jy = recf[jj * 8 + 3], jz = recf[jj * 8 + 4];
                const int me = wave * 2 + c;
                const unsigned long long km = rec[me * 4];
                const float mx = recf[me * 8 + 2], my = recf[me * 8 + 3], mz = recf[me * 8 + 4];
                const float mv = __uint_as_float((unsigned)(km >> 32));
                const bool gt = kj > km;   // (an empty slot has key 0: never larger; an empty OWN slot ranks behind every real key)
                const bool aff = gt && !(dist2<FM>(mx - jx, my - jy, mz - jz) >= mv);
                const unsigned long long gtm = __builtin_amdgcn_ballot_w64(gt), afm = __builtin_amdgcn_ballot_w64(aff);
                const unsigned gth = c ? (unsigned)(gtm >> 32) : (unsigned)gtm, afh = c ? (unsigned)(afm >> 32) : (unsigned)afm;
                const int rank = __builtin_popcount(gth);
                // Unpublished keys (round 6; until then the walk simply stopped after a wave's SECOND key -- half of all stops on a uniform cloud):
                // a wave publishes two keys, everything else it owns is at most its third-best value v3 (sent along with its records).  If BOTH
                // keys of wave w rank above this candidate, one of w's hidden keys could rank above it too -- unless the candidate's value is
                // strictly larger than v3(w).  Values only: a tie with v3 counts as "could" (safe).  A NaN fails `>`: stops, as everywhere.
                const float v3j = recf[(jj >> 1) * 16 + 5];
                const unsigned both = gth & (gth >> 1) & 0x55555555u;      // bit 2 w: keys 2 w and 2 w + 1 are both above this candidate
                const bool hid = ((both >> (jj & ~1)) & 1u) != 0u && !(mv > v3j);
                const unsigned long long hdm = __builtin_amdgcn_ballot_w64(hid);
                const unsigned hdh = c ? (unsigned)(hdm >> 32) : (unsigned)hdm;
                // the walk stops AT this candidate (bad: affected by a larger key, possibly outranked by a hidden key, empty, or -- unless it is the
                // round's first -- value 0) or right AFTER it (value 0)
                const bool zero = !(mv > 0.f);
                const bool bad = km == 0ull || afh != 0u || hdh != 0u || (zero && rank > 0);
                const bool after = zero;
                if (jj == 0) {
                    // where the walk stops because of this candidate: AT it (bad) or right AFTER it; the round's length is the smallest
                    atomicMin(&nstop[rpar], bad ? (unsigned)rank : (after ? (unsigned)rank + 1u : 0xffu));
                    if (rank < KE) {   // ranks are unique among real keys: slot `rank` of the round has one writer (empty slots all carry point 0)
                        const unsigned crk = (~(unsigned)km) >> SB;
                        const unsigned ccls = log2bs ? (__builtin_bitreverse32(crk >> 16) >> (32 - log2bs)) : 0u;
                        *reinterpret_cast<f32x4 *>(&res[rank * 4]) = (f32x4){mx, my, mz, mv};
                        if (j + rank < m) spick[j + rank] = km ? (int)(((crk & 0xffffu) << log2bs) | ccls) : 0;
                    }
                }
                if (t == 0) nstop[rpar ^ 1] = 0xffu;   // the other slot: last read before barrier A of this round, next used after barrier A of the next
            }
            __syncthreads();
            // 6. the round's length = the first stop; its samples in the lane layout of the next box test; pruning bound = the last sample's value
            {
                int n = __builtin_amdgcn_readfirstlane((int)nstop[rpar]);
                n = max(1, min(min(n, KE), m - j));
                gval = res[(n - 1) * 4 + 3];   // (wave-uniform address: a broadcast read; the samples themselves are read by the next round's passes)
                ns = n;
            }
            rpar ^= 1;
            j += ns;
#ifdef G4D_FPS_DEBUG
            { const long long now_ = clock64(); dbg_ph[2] += now_ - dbg_c0; dbg_c0 = now_; }
#endif
        }
        if (temp && ns > 1) {
            // the reference's scratch ends up holding the min-distances to every sample but the LAST one (sampling_gpu.cu:129-141 updates
            // with idx[j - 1] before it picks idx[j]): apply the final round's samples except its last
            if constexpr (!SOA) {
                blx = sbox[0]; bly = sbox[1]; blz = sbox[2]; bhx = sbox[3]; bhy = sbox[4]; bhz = sbox[5];
            }
            for (int p0 = 0; p0 < ns - 1; p0 += SPP) {
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(&res[min(p0 + ls, KE0 - 1) * 4]);
                const float xs = sv.x, ys = sv.y, zs = sv.z;
                const float gx = fmaxf(fmaxf(blx - xs, xs - bhx), 0.f);
                const float gy = fmaxf(fmaxf(bly - ys, ys - bhy), 0.f);
                const float gz = fmaxf(fmaxf(blz - zs, zs - bhz), 0.f);
                const unsigned long long active = __builtin_amdgcn_ballot_w64(p0 + ls < ns - 1 && dist2<FM>(gx, gy, gz) < INF);
                const int cnt = min(SPP, ns - 1 - p0);
                for (int i = 0; i < cnt; ++i) {
                    const unsigned mi = (unsigned)(active >> (i * P)) & (P >= 32 ? 0xffffffffu : ((1u << (P & 31)) - 1u));
                    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs), i * P));
                    const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ys), i * P));
                    const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zs), i * P));
#pragma unroll
                    for (int q = 0; q < P; ++q)
                        if ((mi >> q) & 1u) md(q) = fpsb_min(dist2<FM>(px(q) - ax, py(q) - ay, pz(q) - az), md(q));
                }
            }
        }
#ifdef G4D_FPS_DEBUG
        __syncthreads();
        if (lane == 0 && temp && (wave == 0 || wave == 5) && cloud == 0) {
            float *o = temp + (wave ? 16 : 0);
            o[0] = (float)dbg_rounds; o[1] = (float)dbg_stop[0]; o[2] = (float)dbg_stop[1]; o[3] = (float)dbg_stop[2]; o[4] = (float)dbg_stop[3];
            o[5] = (float)dbg_ph[0]; o[6] = (float)dbg_ph[1]; o[7] = (float)dbg_ph[2]; o[8] = (float)dbg_actw;
            o[9] = (float)dbg_a[0]; o[10] = (float)dbg_a[1]; o[11] = (float)dbg_a[2]; o[12] = (float)dbg_a[3]; o[13] = (float)dbg_pairs;
        }
        return;
#endif
    } else {
        float x1 = sx[0], y1 = sy[0], z1 = sz[0];
        float gval = INF;          // global max of the min-distances (the last winner's value): nothing can exceed it
        float cval = -2.f;         // this wave's cached candidate (value, rank); refreshed only when one of its buckets was swept
        unsigned crank = 0xffffffffu;
        for (int j = 1; j < m; ++j) {
#ifdef G4D_FPS_DEBUG
            dbg_c0 = clock64();
#endif
            // 1. which buckets can change?  gap between the sample and the box, per axis, same op order as the point distance;
            //    a bucket whose box is at least sqrt(gval) away cannot lower any min-distance (every one is <= gval)
            const float gx = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f);
            const float gy = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f);
            const float gz = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
            const float dbox = dist2<FM>(gx, gy, gz);
            const unsigned active = (unsigned)__builtin_amdgcn_ballot_w64(lane < P && dbox < gval);
#ifdef G4D_FPS_DEBUG
            dbg_active += __builtin_popcount(active);
#endif
            G4D_STAMP(0)
            if (active != 0u) {  // wave-uniform; about half of the waves skip the whole block in a typical round
                // 2. sweep the active buckets
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    if ((active >> i) & 1u) {
                        const float dx = px(i) - x1, dy = py(i) - y1, dz = pz(i) - z1;
                        md(i) = fpsb_min(dist2<FM>(dx, dy, dz), md(i));
                    }
                }
                // 3. lane candidate: max value over its P points, smallest rank among the points holding it; then the wave's
                //    (balanced trees: the round is a dependent-issue chain at ~6 cycles per instruction, depth is what counts)
                float tv[P];
#pragma unroll
                for (int i = 0; i < P; ++i) tv[i] = md(i);
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tv[i] = fmax_raw(tv[i], tv[i + w / 2]);
                const float b = tv[0];
                unsigned tr[P];
#pragma unroll
                for (int i = 0; i < P; ++i) tr[i] = (md(i) == b) ? rk[i] : 0xffffffffu;
#pragma unroll
                for (int w = P; w > 1; w >>= 1)
#pragma unroll
                    for (int i = 0; i < w / 2; ++i) tr[i] = min(tr[i], tr[i + w / 2]);
                const unsigned r = tr[0];
                cval = wave_max_f32(b);
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(b == cval);
                if (__builtin_popcountll(hit) == 1) {
                    crank = (unsigned)__builtin_amdgcn_readlane((int)r, __builtin_ctzll(hit));
                } else {
                    crank = wave_min_u32(b == cval ? r : 0xffffffffu);
                }
            }
            G4D_STAMP(1)
            // 4. workgroup arg-max: ONE LDS atomic max per wave on a rotating slot, one barrier, one read
            if (lane == 0)
                atomicMax(&slots[j % 3], ((unsigned long long)__float_as_uint(fmaxf(cval, 0.f)) << 32) | (unsigned)(~crank));
            __syncthreads();
            G4D_STAMP(2)
            const unsigned long long best = slots[j % 3];
            if (t == 0) slots[(j + 2) % 3] = 0ull;  // next use is after the NEXT barrier; last read was before this one
            const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(best >> 32));
            const unsigned rank = ~(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)best);
            gval = __uint_as_float(bhi);
            const unsigned c = log2bs ? (__builtin_bitreverse32(rank >> 16) >> (32 - log2bs)) : 0u;
            const int old = (int)(((rank & 0xffffu) << log2bs) | c);
            x1 = sx[old]; y1 = sy[old]; z1 = sz[old];
            // The pick goes to LDS, not to global memory: __syncthreads() drains the wave's outstanding global stores (vmcnt(0)), so a
            // store issued here would put its write round trip in front of wave 0's NEXT barrier arrival -- on the critical path of a
            // round in which wave 0 has nothing else to do.  The list (and the gathered coordinates) is written once, after the loop.
            if (t == 0) spick[j] = old;
#ifdef G4D_FPS_DEBUG
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            G4D_STAMP(3)
        }
    }
    __syncthreads();
    for (int j = t; j < m; j += T) {
        const int k = spick[j];
        idx[j] = k;
        if (nx) {
            if constexpr (SOA) { nx[j * 3 + 0] = sx[k]; nx[j * 3 + 1] = sy[k]; nx[j * 3 + 2] = sz[k]; }
            else { nx[j * 3 + 0] = xyz[k * 3 + 0]; nx[j * 3 + 1] = xyz[k * 3 + 1]; nx[j * 3 + 2] = xyz[k * 3 + 2]; }   // (L2 hits: the cloud was read twice above)
        }
    }
    if (temp) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (rk[i] != 0xffffffffu) {   // the slot's original index, decoded from its tie rank (pk[] does not stay live through the round loop)
                const unsigned r = rk[i] >> SB;
                const unsigned c = log2bs ? (__builtin_bitreverse32(r >> 16) >> (32 - log2bs)) : 0u;
                temp[((r & 0xffffu) << log2bs) | c] = md(i);
            }
    }
#undef px
#undef py
#undef pz
#undef md
#ifdef G4D_FPS_DEBUG
    __syncthreads();
    if (lane == 0 && temp) atomicAdd(&temp[0], (float)dbg_active);  // debug only: total active (wave, bucket) sweeps
    if (lane == 0 && temp && (wave == 0 || wave == 5) && cloud == 0)
        for (int q = 0; q < 4; ++q) temp[1 + (wave ? 4 : 0) + q] = (float)dbg_t[q];
#endif
}

template <int W, int P, int FM, int KMAX>
__global__ void __launch_bounds__(64 * W) fps_bucket_kernel(int n, int m, int bs, int log2bs, int deal, int pick_off, const float *__restrict__ xyz_all,
                                                           float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ nx_all) {
    fps_bucket_body<W, P, FM, KMAX, true>(n, m, bs, log2bs, deal, pick_off, xyz_all, temp_all, idx_all, nx_all, blockIdx.x);
}

// the register form (multi-pick, no LDS cloud copy), held to 72 VGPRs (round 6; 64 before: with up to 16 samples per round and the third-best-value
// rule the 64-register build spilled six dwords and ran 3 % slower; measured with both builds on one box: 624-633 vs 604-607 us per 240-cloud launch,
// 68.4-68.6k vs 68.5k frames/s fp32, 110.0-110.6k vs 111.0-111.3k bf16): four of its waves leave 224 registers of a SIMD's 512 to another call's wave
// (the widest fp32 shared-MLP kernel takes 212)
template <int W, int P, int FM, int KMAX>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(7, 8)))
fps_bucket_reg_kernel(int n, int m, int bs, int log2bs, int deal, int pick_off, const float *__restrict__ xyz_all, float *__restrict__ temp_all,
                      int *__restrict__ idx_all, float *__restrict__ nx_all) {
    fps_bucket_body<W, P, FM, KMAX, false>(n, m, bs, log2bs, deal, pick_off, xyz_all, temp_all, idx_all, nx_all, blockIdx.x);
}

// samples per round of the bucketed kernels: up to kFpsPicks (multi-pick, default) or 1 (G4D_FPS_MULTI=1: one arg-max per round, the round-2 loop).
// Round 6: 8 -> 16 (two box-test passes per round when more than 64 / P samples went out) together with the third-best-value rule of the merge:
// on a uniform 8192 -> 1024 cloud 236 -> 166 rounds (numpy emulation: stops after a wave's second key 123 -> hidden-key stops 53, cap 23 -> 7).
constexpr int kFpsPicks = 16;
static int fps_multi() {
    static const int k = getenv("G4D_FPS_MULTI") ? atoi(getenv("G4D_FPS_MULTI")) : 8;
    return k > 1 ? 8 : 1;
}
// Which multi-pick form a launch of b clouds takes.  The LDS-copy form (rounds 3-4: 74 VGPRs, 104 KB) has the shorter round (0.61 us per pick
// against 0.67 at b = 8): it serves the small launches, where the sampling chain is the latency of the step.  The register form (72 VGPRs,
// 70 KB) serves the coalesced calls (b >= 32 clouds): there every CU hosts a sampling workgroup and what counts is which launches of the OTHER
// calls in flight fit beside it -- measured at 240 clouds per call, four calls in flight, both arms on one box: 51.9k -> 53.6k frames/s
// fp32, 94.4k -> 96.8k bf16 (profiles/r05_overlap_pairs_240clouds.txt: next to fp_init 0.98 -> 0.80 of the sum, next to the tiled GEMMs
// 0.97 -> 0.86-0.91).  G4D_FPS_SOA=1 / 0 forces one form.
static bool fps_soa(int b) {
    static const int k = getenv("G4D_FPS_SOA") ? atoi(getenv("G4D_FPS_SOA")) : -1;
    return k < 0 ? b < 32 : k != 0;
}
static int fps_kcap() {   // tuning hook: at most this many samples per round
    static const int k = getenv("G4D_FPS_KCAP") ? atoi(getenv("G4D_FPS_KCAP")) : kFpsPicks;
    return k < 1 ? 1 : (k > 64 ? 64 : k);
}

// The sampling launch with a second ROLE: workgroups [0, b) run the FPS of cloud blockIdx.x, workgroups [b, 2 b) build the ball-query
// cell grid of cloud blockIdx.x - b (ball_grid_build.h).  Both depend on the cloud only; the grid build (17 us alone, one workgroup per
// cloud) disappears behind the 750 us of the sampling and the step has one launch fewer.
template <int FM, int KMAX>
__global__ void __launch_bounds__(1024) fps_bucket_grid_kernel(int b, int n, int m, int bs, int log2bs, int deal, int pick_off,
                                                              const float *__restrict__ xyz_all, int *__restrict__ idx_all, float *__restrict__ nx_all,
                                                              int cmax, float cell_req, unsigned char *__restrict__ ws_all, size_t ws_stride) {
    if ((int)blockIdx.x < b) fps_bucket_body<16, 8, FM, KMAX, true>(n, m, bs, log2bs, deal, pick_off, xyz_all, nullptr, idx_all, nx_all, blockIdx.x);
    else ball_grid_build_body(n, cmax, cmax, cell_req, xyz_all, ws_all, ws_stride, (int)blockIdx.x - b);
}

template <int FM, int KMAX>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(7, 8)))
fps_bucket_grid_reg_kernel(int b, int n, int m, int bs, int log2bs, int deal, int pick_off, const float *__restrict__ xyz_all, int *__restrict__ idx_all,
                           float *__restrict__ nx_all, int cmax, float cell_req, unsigned char *__restrict__ ws_all, size_t ws_stride) {
    if ((int)blockIdx.x < b) fps_bucket_body<16, 8, FM, KMAX, false>(n, m, bs, log2bs, deal, pick_off, xyz_all, nullptr, idx_all, nx_all, blockIdx.x);
    else ball_grid_build_body(n, cmax, cmax, cell_req, xyz_all, ws_all, ws_stride, (int)blockIdx.x - b);
}

template <int W, int P, int FM>
static int launch_bucket_fm(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    const size_t npad = (size_t)64 * W * P;
    const bool multi = fps_multi() > 1;
    const bool soa = !multi || fps_soa(b) || W != 16;   // (the register form is held to 72 VGPRs: 16 waves x 4 or 8 points per lane only)
    const size_t body = (!soa || npad * 8 > (size_t)n * 12) ? npad * 8 : (size_t)n * 12;
    const size_t pick_off = (kFpsHdr + body + 15) & ~(size_t)15;
    const size_t lds = pick_off + (size_t)m * 4;
    if (lds > 160 * 1024 - 1024) return -1;   // the pick list lives in LDS: m beyond ~15.8k at n = 8192 goes to the next route (fps.hip)
    auto kern = multi ? fps_bucket_kernel<W, P, FM, kFpsPicks> : fps_bucket_kernel<W, P, FM, 1>;
    if constexpr (W == 16) { if (!soa) kern = fps_bucket_reg_kernel<W, P, FM, kFpsPicks>; }
    static unsigned long long attr_done[3] = {0, 0, 0};  // one bit per device
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024 - 1024, attr_done[multi ? (soa ? 1 : 2) : 0], "g4d_fps_f32(bucketed)")) return rc;
    // measured at N = 8192, M = 1024, B = 8 (scripts/time_fps.py): deal 1 / 2 / 4 / 8 -> 0.812 / 0.783 / 0.757 / 0.748 us per round
    static const int deal_env = getenv("G4D_FPS_DEAL") ? atoi(getenv("G4D_FPS_DEAL")) : 0;  // tuning hook: 1 | 2 | 4 | ... | P; 0 = P
    const int deal = (deal_env >= 1 && deal_env <= P && P % deal_env == 0) ? deal_env : P;
    hipLaunchKernelGGL(kern, dim3(b), dim3(64 * W), lds, s, n, m, bs, log2bs, deal | (fps_kcap() << 8), (int)pick_off, xyz, temp, idx, nx);
    return check_launch("g4d_fps_f32(bucketed)");
}

template <int W, int P>
static int launch_bucket(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    G4D_WITH_FM(distance_contraction(), return (launch_bucket_fm<W, P, FM>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s)))
    return G4D_OK;
}

// FPS (+ gather) of b clouds and the cell grid of the same clouds in one launch; -1 when the shape is not the bucketed kernel's default one
int fps_bucket_grid_launch(int b, int n, int m, int bs, int log2bs, const float *xyz, int *idx, float *nx, float rmax, void *grid_ws, hipStream_t s) {
    if (!(n > 4096 && n <= 8192)) return -1;
    constexpr int W = 16, P = 8;
    const size_t npad = (size_t)64 * W * P;
    const bool multi = fps_multi() > 1;
    const bool soa = !multi || fps_soa(b);
    const size_t body = (!soa || npad * 8 > (size_t)n * 12) ? npad * 8 : (size_t)n * 12;
    const size_t pick_off = (kFpsHdr + body + 15) & ~(size_t)15;
    const int cmax = grid_cmax(n);
    const size_t lds_fps = pick_off + (size_t)m * 4, lds_grid = ((size_t)cmax + 1) * 4 + 16 * 8 * 4;
    const size_t lds = lds_fps > lds_grid ? lds_fps : lds_grid;
    if (lds > 160 * 1024 - 1024) return -1;   // (pick list too long for LDS: the caller falls back to two launches)
    static unsigned long long attr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int mode = distance_contraction();
    const int slot = (mode == 0 ? 0 : (mode == 1 ? 1 : 2)) + (multi ? (soa ? 3 : 6) : 0);
    const void *k = nullptr;
    G4D_WITH_FM(mode, k = multi ? (soa ? reinterpret_cast<const void *>(fps_bucket_grid_kernel<FM, kFpsPicks>) : reinterpret_cast<const void *>(fps_bucket_grid_reg_kernel<FM, kFpsPicks>))
                                : reinterpret_cast<const void *>(fps_bucket_grid_kernel<FM, 1>))
    if (const int rc = ensure_dynamic_lds(k, 160 * 1024 - 1024, attr[slot], "g4d_fps_gather_grid_f32")) return rc;
#define G4D_GRID_LAUNCH(KM, KERN)                                                                                                                                    \
    G4D_WITH_FM(mode, hipLaunchKernelGGL((KERN<FM, KM>), dim3(2 * b), dim3(1024), lds, s, b, n, m, bs, log2bs, P | (fps_kcap() << 8), (int)pick_off, xyz, idx, nx, \
                                         cmax, rmax * kCellSlack, reinterpret_cast<unsigned char *>(grid_ws), grid_cloud_bytes(n)))
    if (multi && soa) { G4D_GRID_LAUNCH(kFpsPicks, fps_bucket_grid_kernel) } else if (multi) { G4D_GRID_LAUNCH(kFpsPicks, fps_bucket_grid_reg_kernel) } else { G4D_GRID_LAUNCH(1, fps_bucket_grid_kernel) }
#undef G4D_GRID_LAUNCH
    return check_launch("g4d_fps_gather_grid_f32");
}

// Called by g4d_fps_f32 (fps.hip) for 2048 < n <= 8192.  Returns -1 when the shape is not covered.
int fps_bucket_dispatch(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    static const int cfg = getenv("G4D_FPS_BUCKET_W") ? atoi(getenv("G4D_FPS_BUCKET_W")) : 16;  // tuning hook
    if (n > 4096 && n <= 8192) {
        if (cfg == 8) return launch_bucket<8, 16>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
        if (cfg == 4) return launch_bucket<4, 32>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
        return launch_bucket<16, 8>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
    }
    if (n > 2048 && n <= 4096) return launch_bucket<16, 4>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
    return -1;
}

}  // namespace g4d
