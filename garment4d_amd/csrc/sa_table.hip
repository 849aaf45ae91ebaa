// Set-abstraction scale behind a pre-contracted first layer, PERSISTENT and software-pipelined (round 4): the large-launch form of
// g4d_mlp_chain_group_table_f32 (QueryAndGroup + SharedMLP + max pool, pointnet2_modules.py:40-53 / pointnet2_utils.py:232-265; the
// feature part of the first layer is the per-source-point table of fused.sa_level_table).
//
// Why: in-kernel cycle stamps in the MIDDLE of a 240-cloud launch of the register-chain kernel (scripts/dbg_chain_steady.py, SA level 2
// scale 1: 384 MFMAs = 12.3k cycles of matrix pipe per 32-row wave) show a wave living 34k cycles: 8.6k until its row contexts exist
// (kernel arguments -> neighbour index -> coordinates / table row: three dependent round trips), 2.6k until the rows have arrived, ~2k at
// each layer seam and 4.8k in the epilogue, most of it the per-layer scale / shift vectors being fetched from L2 at the moment they are
// needed.  With 2-3 such waves per SIMD the matrix pipe is 53 % busy.  A loop over row blocks alone does not help (measured: slower) --
// what has to go is the exposed latency.  Here
//   * a workgroup is resident for the whole launch; every per-layer constant (xyz weights, the loader's affine, scale / shift of both
//     layers) and, for the 32- / 64-wide stacks, both weight matrices sit in LDS, loaded once;
//   * a wave walks its row blocks with a two-level prefetch (as sa_xyz.hip): while block k is on the VALU / matrix pipe, the table rows
//     and coordinates of block k + 1 are in flight and so are the indices of block k + 2;
//   * a pooling group never spans waves: a wave takes whole neighbourhoods (for 64 samples: four 16-row blocks with a running maximum),
//     so there is no barrier after the start-up copy; the pooled tiles leave through the four-tile swap reduction of sa_xyz.hip,
//     64 channels per store.
// Arithmetic and summation order are those of mlp_chain.hip's table loader and chained layers: results are bit-identical.
#include <cstdlib>
#include <type_traits>

#include "mlp_common.h"

namespace g4d {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct SaTabArgs {
    int rows, N, P;                      // rows = B * P * S grouped rows; N source points per cloud, P centroids per cloud
    const float *xyz, *new_xyz;
    const int *idx;
    const float *tab;                    // (B * N, tab_ld): feature part of the first layer per source point
    int tab_ld;
    const float *wx, *ps, *pf;           // [3][C] xyz columns of the first layer (transposed); [C] its affine
    const float *W2, *sc2, *sh2;         // fragment order [tile][k-step][64][4]
    const float *W3, *sc3, *sh3;         // (T k-steps per channel tile: Kpad == C)
    float *out;
    int ldo, col0;
    // WM == 2 with a work list (round 6, sa_units_*_kernel below): wave w of workgroup g walks items[(4 g + w) * cap + i], i < wg_len[g];
    // nullptr: the static partition of rounds 4-5 (every block of every neighbourhood)
    const int *items, *wg_len;
    int cap;
};

// ---- work list of the lock-step kernel (round 6).  A neighbourhood of S = 64 samples is four 16-row blocks, and ball_query pads it with copies
// of its first hit: a block of padding only reproduces row 0's output, and max pooling does not see it.  On the encoder's third level (r = 0.4
// over 256 points) 31 of the 64 samples are distinct on average, 2.4 of the 4 blocks.  The four waves of a workgroup run in lock step (they
// share every weight fragment through LDS), so they must be given neighbourhoods with the SAME number of live blocks:
//   sa_units_count_kernel   live blocks nb(q) = 1 + the last block holding an index other than the first, per neighbourhood; neighbourhoods
//                           counting-sorted by nb into G lists (one wave ranks 64 of them with ballots, one global atomic per list and wave);
//   sa_units_items_kernel   rounds of four neighbourhoods of equal nb, longest first, dealt round-robin to the workgroups of the main launch
//                           (sorted dealing balances them to within one round) and written out as one item list per wave:
//                           item = block number | first-of-neighbourhood | last | dead (a padding slot of a round / of a workgroup's list).
// The main kernel only replaces its arithmetic block number by a (prefetched) item.  Exact for ANY index list: a block is dropped only if every
// one of its rows carries the neighbourhood's first index, i.e. is the same (source point, centroid) pair as row 0.
constexpr int kItemFirst = 1 << 28, kItemLast = 1 << 29, kItemDead = 1 << 30, kItemBlk = (1 << 28) - 1;

// counts[1 .. G]: neighbourhoods with nb live blocks; order[(nb - 1) * nq_cap + i]: the i-th of them.  One wave ranks 64 neighbourhoods with
// ballots, the 16 waves of a workgroup add up in LDS, and ONE global atomic per list and workgroup reserves the range (agent-scope fetch-adds on one
// cache line run at 88 per microsecond whatever the number of waves -- scripts/micro/atomic_rate.hip: one atomic per wave and list made this
// kernel 16 us long).
constexpr int kCountWaves = 16;
template <int S>
__global__ void __launch_bounds__(64 * kCountWaves) sa_units_count_kernel(int nq, int nq_cap, const int *__restrict__ idx, int *__restrict__ counts, int *__restrict__ order) {
    constexpr int G = S / 16, UPL = 64 / S;          // neighbourhoods per 64-lane load
    __shared__ int s_cnt[kCountWaves][4], s_base[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave = blockIdx.x * kCountWaves + wv;
    const int q0 = wave * 64;                        // this wave's 64 neighbourhoods: lane l keeps nb of neighbourhood q0 + l
    int mine = 0;
    // all of the wave's loads first (64 / UPL independent 256-byte rows in flight), then the ranking
    int vv[64 / UPL];
    const long long last_ = (long long)nq * S - 1;
#pragma unroll
    for (int i = 0; i < 64 / UPL; ++i) {
        const long long e_ = (long long)(q0 + i * UPL) * S + lane;
        vv[i] = idx[e_ < last_ ? e_ : last_];        // (past the end: clamped, dropped below -- a neighbourhood inside the range has all its rows inside)
    }
#pragma unroll
    for (int i = 0; i < 64 / UPL; ++i) asm volatile("" : "+v"(vv[i]));   // (every load issued before the first use)
#pragma unroll
    for (int i = 0; i < 64 / UPL; ++i) {
        const int v = vv[i];
#pragma unroll
        for (int u = 0; u < UPL; ++u) {
            const int h0 = __builtin_amdgcn_readlane(v, u * S);
            unsigned long long m = __builtin_amdgcn_ballot_w64(v != h0);
            if constexpr (UPL == 2) m = u == 0 ? (m & 0xffffffffull) : (m >> 32);
            const int nb = m ? (63 - __builtin_clzll(m)) / 16 + 1 : 1;
            if (lane == i * UPL + u) mine = nb;
        }
    }
    const bool ok = q0 + lane < nq;
    unsigned long long mk[G];
#pragma unroll
    for (int b = 1; b <= G; ++b) {
        mk[b - 1] = __builtin_amdgcn_ballot_w64(ok && mine == b);
        if (lane == 0) s_cnt[wv][b - 1] = (int)__builtin_popcountll(mk[b - 1]);
    }
    __syncthreads();
    if (threadIdx.x < G) {                           // thread b - 1: the workgroup's total of list b, one reservation
        int tot = 0;
        for (int w = 0; w < kCountWaves; ++w) tot += s_cnt[w][threadIdx.x];
        s_base[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x + 1], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int b = 1; b <= G; ++b) {
        if (!(ok && mine == b)) continue;
        int off = s_base[b - 1];
        for (int w = 0; w < wv; ++w) off += s_cnt[w][b - 1];
        order[(size_t)(b - 1) * nq_cap + off + (int)__builtin_popcountll(mk[b - 1] & ((1ull << lane) - 1ull))] = q0 + lane;
    }
}

// thread = (workgroup g, wave w) of the main launch: its item list.  Rounds: all neighbourhoods with G live blocks first, four per round, then
// G - 1, ...; round r goes to workgroup r % nwg.
__global__ void __launch_bounds__(256) sa_units_items_kernel(int G, int nq_cap, int nwg, int cap, const int *__restrict__ counts, const int *__restrict__ order,
                                                             int *__restrict__ items, int *__restrict__ wg_len) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nwg * 4) return;
    const int g = t >> 2, w = t & 3;
    int *mine = items + (size_t)t * cap;
    int n = 0, r0 = 0;                               // items written; first round of the bucket being walked
    for (int b = G; b >= 1; --b) {
        const int c = counts[b], rb = (c + 3) >> 2;
        // this workgroup's rounds of the bucket: r = g, g + nwg, ... within [r0, r0 + rb)
        int r = r0 + ((g - r0 % nwg + nwg) % nwg);
        for (; r < r0 + rb; r += nwg) {
            const int e = (r - r0) * 4 + w;          // entry of the bucket's list
            const int q = e < c ? order[(size_t)(b - 1) * nq_cap + e] : -1;
            for (int j = 0; j < b && n < cap; ++j)
                mine[n++] = q < 0 ? kItemDead : ((q * G + j) | (j == 0 ? kItemFirst : 0) | (j == b - 1 ? kItemLast : 0));
        }
        r0 += rb;
    }
    if (w == 0) wg_len[g] = n;                       // (the same for the four waves: equal nb per round)
    for (; n < cap; ++n) mine[n] = kItemDead;
}


__device__ __forceinline__ float pool4_rows_max_t(float v0, float v1, float v2, float v3) {   // sa_xyz.hip: lane 16 c + fi = max over the 16 rows of tile c
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
    const float m01 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const float m23 = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(m01), __float_as_uint(m23), false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}

constexpr int kStageK = 2;   // WM == 2: k-steps per staged chunk (one barrier per chunk)
#ifndef G4D_SA64_OCC3
#define G4D_SA64_OCC3 0   // experiment: the 64-wide, 32-sample stack (52 KB of LDS) at three workgroups per CU (<= 168 registers, 12 bytes of scratch): within 1 % on one box, off
#endif
#ifndef G4D_SA_TABLE_SPREAD
#define G4D_SA_TABLE_SPREAD 1
#endif
constexpr int kRing = 4;   // weight fragments requested ahead of their MFMAs (LDS: ~130 cycles; L2: see kRingG)
constexpr int kRingG = 6;

// C: table / hidden width (layers C -> C -> 2C), S samples per neighbourhood, MT row tiles of 16 per block.
// WM: where a wave's weight fragments come from
//   1  both matrices resident in LDS (32- / 64-wide stacks: 12 / 48 KB);
//   0  every wave streams its fragments from L2 through a register ring.  At 16 rows per wave a 1 KB fragment feeds 4 MFMAs = 128 cycles
//      of matrix pipe: 8 waves per CU ask the vector-memory path for 64 bytes per cycle, all it has -- the 128-wide stack (192 KB of
//      weights) ran at 0.70 of the matrix pipe this way, the same as the register-chain kernel with 32 rows per wave and half the waves;
//   2  the four waves of a workgroup walk their blocks in lock step and SHARE every fragment: a k-step's fragments (8 or 16 KB) are copied
//      L2 -> LDS once per workgroup (buffer_load ... lds since round 5, each wave a quarter) into a double buffer, one k-step ahead, one barrier per k-step;
//      the waves read them with ds_read_b128.  L2 traffic for weights drops 4x.
template <int C, int S, int MT, int WM>
__global__ void __launch_bounds__(256, (WM == 1 && C == 64 && S == 32 && G4D_SA64_OCC3) ? 3 : 2) sa_table_kernel(const SaTabArgs a) {
    constexpr bool WLDS = WM == 1;
    constexpr int T = C / 16, T3 = 2 * T;
    constexpr int R = 16 * MT;                       // rows per block
    constexpr int G = S > R ? S / R : 1;             // blocks per neighbourhood (running maximum across them)
    constexpr int GP = S < R ? R / S : 1;            // neighbourhoods per block
    static_assert(S == 16 || S == 32 || S == 64, "16 / 32 / 64 samples");
    static_assert(G * R == S || GP * S == R, "a block is a whole number of neighbourhoods or vice versa");
    constexpr int NCONST = 11 * C;                   // wx 3C | ps C | pf C | sc2 C | sh2 C | sc3 2C | sh3 2C
    constexpr int NW2 = C * C, NW3 = 2 * C * C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_wx = smem, *s_ps = smem + 3 * C, *s_pf = s_ps + C, *s_sc2 = s_pf + C, *s_sh2 = s_sc2 + C, *s_sc3 = s_sh2 + C, *s_sh3 = s_sc3 + 2 * C;
    float *s_w2 = smem + NCONST, *s_w3 = s_w2 + NW2;
    float *s_stage = smem + NCONST;                  // WM == 2: [2][kStageK * T3 * 256] floats: two buffers of one chunk's fragments
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * C; i += 256) s_wx[i] = a.wx[i];
    for (int i = tid; i < C; i += 256) { s_ps[i] = a.ps[i]; s_pf[i] = a.pf[i]; s_sc2[i] = a.sc2[i]; s_sh2[i] = a.sh2[i]; }
    for (int i = tid; i < 2 * C; i += 256) { s_sc3[i] = a.sc3[i]; s_sh3[i] = a.sh3[i]; }
    if constexpr (WLDS) {   // fragment order, tiles packed back to back ([tile][T k-steps][64][4]: the host's layout has kst >= T k-steps per tile)
        for (int i = tid; i < NW2 / 4; i += 256) {
            const int frag = i >> 6, l = i & 63, ct = frag / T, ks = frag - ct * T;
            reinterpret_cast<f32x4 *>(s_w2)[i] = *reinterpret_cast<const f32x4 *>(a.W2 + ((size_t)(ct * T + ks) * 64 + l) * 4);
        }
        for (int i = tid; i < NW3 / 4; i += 256) {
            const int frag = i >> 6, l = i & 63, ct = frag / T, ks = frag - ct * T;
            reinterpret_cast<f32x4 *>(s_w3)[i] = *reinterpret_cast<const f32x4 *>(a.W3 + ((size_t)(ct * T + ks) * 64 + l) * 4);
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: block numbers, copy sources and destinations on the scalar unit)
    const int fi = lane & 15, fq = lane >> 4;
    const int nblk = (a.rows + R - 1) / R;           // 16 MT-row blocks of the launch
    const int nunit = (nblk + G - 1) / G;            // a wave's unit of work: G consecutive blocks = whole neighbourhoods
    const int nwaves = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    const int nq = a.rows / S;

    // WM == 2 with a work list: the items of iterations it .. it + 3 (item 3 in flight), shifted at the end of every iteration
    const bool listed = WM == 2 && a.items != nullptr;
    const int *my_items = listed ? a.items + (size_t)wg * a.cap : nullptr;
    int itm0 = 0, itm1 = 0, itm2 = 0, itm3 = 0;
    auto item_at = [&](int i) { return my_items[min(i, a.cap - 1)]; };   // (wave-uniform address: a scalar load)
    if (listed) { itm0 = item_at(0); itm1 = item_at(1); itm2 = item_at(2); itm3 = item_at(3); }
    // block of iteration `it` + d of this wave, d = 0, 1, 2 (past the wave's last block: clamped -- read again, never used)
    auto block_of = [&](int it, int d = 0) {
        if (listed) return min((d == 0 ? itm0 : (d == 1 ? itm1 : itm2)) & kItemBlk, nblk - 1);
        const int unit = wg + ((it + d) / G) * nwaves;
        return min(unit * G + (it + d) % G, nblk - 1);
    };
    auto live = [&](int it) { return WM != 2 || (listed ? !(itm0 & kItemDead) : wg + (it / G) * nwaves < nunit); };   // WM == 2: is this iteration's unit real?
    auto blk_first = [&](int it) { return listed ? (itm0 & kItemFirst) != 0 : it % G == 0; };   // first / last block of its neighbourhood
    auto blk_last = [&](int it) { return listed ? (itm0 & kItemLast) != 0 : it % G == G - 1; };
    struct Rows { f32x4 raw[T][MT]; float px[MT], py[MT], pz[MT], cx[MT], cy[MT], cz[MT]; };
    auto load_idx = [&](int blk, int (&v)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) v[mt] = a.idx[min(blk * R + mt * 16 + fi, a.rows - 1)];
    };
    // The rows of a block: rows_begin() forms the table-row pointers and requests the coordinates (small loads); the T * MT 16-byte table loads --
    // the expensive ones at the CU's 64 B / clk vector-memory path -- are requested ONE AT A TIME by rows_raw(q), which the block loop calls
    // between its MFMA chains (round 5: as one burst of 14 loads they held the wave's in-order issue for ~2k cycles; load_rows() = all at once,
    // for the start-up).
    auto rows_begin = [&](int blk, const int (&v)[MT], Rows &rw, const float *(&tr)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int q = __builtin_amdgcn_readfirstlane(min((blk * R + mt * 16) / S, nq - 1));   // S >= 16: a tile belongs to one neighbourhood
            const int b = q / a.P;
            const size_t pt = (size_t)b * a.N + v[mt];
            tr[mt] = a.tab + pt * a.tab_ld + fq * 4;
            const float *pp = a.xyz + pt * 3;
            rw.px[mt] = pp[0]; rw.py[mt] = pp[1]; rw.pz[mt] = pp[2];
            const float *cc = a.new_xyz + (size_t)q * 3;
            rw.cx[mt] = cc[0]; rw.cy[mt] = cc[1]; rw.cz[mt] = cc[2];
        }
    };
    auto rows_raw = [&](const float *const (&tr)[MT], Rows &rw, int q) {   // q = ks * MT + mt
        const int ks = q / MT, mt = q % MT;
        rw.raw[ks][mt] = *reinterpret_cast<const f32x4u *>(tr[mt] + ks * 16);
    };
    auto load_rows = [&](int blk, const int (&v)[MT], Rows &rw) {
        const float *tr[MT];
        rows_begin(blk, v, rw, tr);
#pragma unroll
        for (int q = 0; q < T * MT; ++q) rows_raw(tr, rw, q);
    };
    // a weight fragment = 1 KB, lane l takes bytes [16 l, 16 l + 16).  From L2 (WLDS false): uniform base (SGPRs, bumped per fragment by scalar
    // adds) + ONE lane offset register; written any other way the 192 fragment addresses of the 128-wide stack are loop invariants that the
    // compiler computes once and keeps (512 registers + scratch)
    unsigned lane16 = (unsigned)lane * 4u;   // (made opaque at the top of every iteration of the block loop, below)
    // (the launcher guarantees T k-steps per channel tile in the host's layout, Kpad == C: fragment offsets are compile-time constants)
    auto wfrag = [&](const float *gw, const float *sw, int ct, int ks) -> f32x4 {
        if constexpr (WLDS) return *reinterpret_cast<const f32x4 *>(sw + ((ct * T + ks) * 64 + lane) * 4);
        else return *reinterpret_cast<const f32x4 *>(gw + (ct * T + ks) * 256 + lane16);
    };
    constexpr int RD = WLDS ? kRing : (WM == 2 ? 1 : kRingG);

    // WM == 2: the waves of a workgroup run the same number of iterations (barriers inside the loop); a wave whose unit lies past the end
    // computes on clamped rows and stores nothing
    const int units_first = WM == 2 ? blockIdx.x * 4 : wg;
    const int my_units = units_first < nunit ? (nunit - units_first + nwaves - 1) / nwaves : 0;
    const int iters = listed ? a.wg_len[blockIdx.x] : my_units * G;
    if (iters == 0) return;
    // kStageK k-steps' fragments (a chunk) into the stage buffer `chunk & 1`: chunks 0 .. T/kStageK-1 are layer 2's, the rest layer 3's; periodic
    // per block (an even number of chunks).  Fragment (kk, ct) of a chunk sits at slot kk * (tiles of the layer) + ct.
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W2), 0, C * C * 4, 0x00020000),
                                 rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W3), 0, 2 * C * C * 4, 0x00020000);
    constexpr bool kSpread = G4D_SA_TABLE_SPREAD;   // the next chunk's copies between this chunk's MFMA chains (A/B: -DG4D_SA_TABLE_SPREAD=0)
    auto stage_issue = [&](int chunk) {
        if constexpr (WM == 2) {
            constexpr int NC2 = T / kStageK, NCH = 2 * NC2;
            chunk = chunk % NCH;
            float *dst = s_stage + (chunk & 1) * (kStageK * T3 * 256);
            // MUBUF copies (round 5; fp_init.hip has the story): a FLAT-encoded global_load_lds marks the wave "flat pending" in the compiler's
            // counter model and every later wait -- for a gathered row, for a ds_read -- becomes vmcnt(0) lgkmcnt(0), i.e. waits for the copy too.
            const int l16b = lane * 16;
            if (chunk < NC2) {
#pragma unroll
                for (int j = 0; j < kStageK * T / 4; ++j) {
                    const int slot = wave + 4 * j, kk = slot / T, ct = slot % T;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (__attribute__((address_space(3))) void *)(dst + slot * 256), 16, l16b,
                                                             (ct * T + chunk * kStageK + kk) * 1024, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < kStageK * T3 / 4; ++j) {
                    const int slot = wave + 4 * j, kk = slot / T3, ct = slot % T3;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW3, (__attribute__((address_space(3))) void *)(dst + slot * 256), 16, l16b,
                                                             (ct * T + (chunk - NC2) * kStageK + kk) * 1024, 0, 0);
                }
            }
        }
    };
    // ONE of a chunk's copies: number j of this wave's share (round 5: the copies of the next chunk go out between the MFMA chains of the current
    // one, one per chain, instead of as a burst behind the barrier -- a wave issues in order, and while its 4-8 copies queued at the CU's
    // vector-memory path, 16 cycles per 1 KB wave-copy and eight waves in line, it issued no MFMA; gemm_tile.hip has the measurement).
    auto stage_issue_one = [&](int chunk, int j) {
        if constexpr (WM == 2) {
            constexpr int NC2 = T / kStageK, NCH = 2 * NC2;
            chunk = chunk % NCH;
            float *dst = s_stage + (chunk & 1) * (kStageK * T3 * 256);
            const int l16b = lane * 16;
            if (chunk < NC2) {
                if (j < kStageK * T / 4) {
                    const int slot = wave + 4 * j, kk = slot / T, ct = slot % T;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (__attribute__((address_space(3))) void *)(dst + slot * 256), 16, l16b,
                                                             (ct * T + chunk * kStageK + kk) * 1024, 0, 0);
                }
            } else {
                if (j < kStageK * T3 / 4) {
                    const int slot = wave + 4 * j, kk = slot / T3, ct = slot % T3;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW3, (__attribute__((address_space(3))) void *)(dst + slot * 256), 16, l16b,
                                                             (ct * T + (chunk - NC2) * kStageK + kk) * 1024, 0, 0);
                }
            }
        }
    };
    // start of a chunk (k-step gks of the block, 0 .. 2T-1; acts on the first k-step of each chunk): this wave's share of the chunk's
    // fragments has landed (vmcnt), so has everybody else's and nobody reads the other buffer any more (barrier); then the next chunk's copy
    // goes out.  (One k-step per barrier: 805 us for the 128-wide stack; a 1k-cycle k-step barely covers the copy's L2 round trip.)
    auto stage_step = [&](int gks) {
        if constexpr (WM == 2) {
            if (gks % kStageK == 0) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (!kSpread) stage_issue(gks / kStageK + 1);
            }
        }
    };
    auto sfrag = [&](int gks, int nt, int ct) -> f32x4 {
        return *reinterpret_cast<const f32x4 *>(s_stage + ((gks / kStageK) & 1) * (kStageK * T3 * 256) + (((gks % kStageK) * nt + ct) * 64 + lane) * 4);
    };
    // DEAD TILES (round 6).  ball_query pads a neighbourhood with copies of its first hit (ball_query_gpu.cu:32-36); a padded row is the same
    // (source point, centroid) pair as row 0, gives the same output, and max pooling does not see it.  On the encoder's third level 5 of the 32
    // samples of the r = 0.2 scale are distinct on average (uniform clouds; fewer on surfaces): the second 16-row tile of every neighbourhood is
    // all padding.  A tile whose rows ALL carry the neighbourhood's first index (and that is not the tile holding row 0) is skipped -- exact for
    // any index list, padded or not; tiles_alive() = how many leading tiles of a block have to be computed (0: a whole trailing block of a
    // 64-sample neighbourhood).  Blocks are whole neighbourhoods or halves of one here (S >= 32), waves are autonomous (no lock step to keep).
    constexpr bool DEDUP = WM != 2 && MT == 2 && GP == 1;
    int h0 = 0;                                      // first neighbour index of the neighbourhood the block being looked at belongs to
    auto tiles_alive = [&](int it_, const int (&v)[MT]) {
        if constexpr (!DEDUP) return MT;
        else {
            const bool head = it_ % G == 0;          // the block starts a neighbourhood
            if (head) h0 = __builtin_amdgcn_readlane(v[0], 0);
            const bool t1 = __builtin_amdgcn_ballot_w64(v[MT - 1] != h0) != 0ull;
            const bool t0 = head || __builtin_amdgcn_ballot_w64(v[0] != h0) != 0ull;
            return t1 ? 2 : (t0 ? 1 : 0);
        }
    };
    stage_issue(0);
    int ivn[MT];
    Rows cur, nxt;
    load_idx(block_of(0), ivn);
    int nt_cur = tiles_alive(0, ivn), nt_nxt = MT;
    load_rows(block_of(0), ivn, cur);
    load_idx(block_of(0, 1), ivn);
    float pm[T3];                                    // running maximum of the neighbourhood across its blocks (G > 1)
    const float *trn[MT];                            // table-row pointers of the next block (rows_begin -> rows_raw)
    for (int it = 0; it < iters; ++it) {
        const int blk = block_of(it);
        if constexpr (WM != 1) asm volatile("" : "+v"(lane16));   // no hoisted fragment addresses
        if constexpr (WM != 2) {
            if constexpr (kSpread) rows_begin(block_of(it, 1), ivn, nxt, trn);   // level 2 of the next block: pointers + coordinates now, the table rows between layer 2's chains
            else load_rows(block_of(it, 1), ivn, nxt);
            nt_nxt = tiles_alive(it + 1, ivn);       // (the indices are in registers: rows_begin has just used them)
            load_idx(block_of(it, 2), ivn);         // level 1 of the one after
        }
        float x[T3];                                 // the block's maximum per channel tile (compute<NT >= 1> sets it)
        auto compute = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;  // leading row tiles of the block that are computed
        // ---- first layer: relu(affine(table row + Wx (x_j - q))), the arithmetic of mlp_chain.hip's table loader, one k-step at a time
        //      in front of that k-step's share of layer 2 (transposed: A = weights, B = activations -- lane (fi, fq) ends with channels
        //      16 ct + 4 fq + r of row fi).  Fragment (ks, ct): every accumulator sees k ascending, as in the chain kernel.
        f32x4 sring[2];
        float gx[MT], gy[MT], gz[MT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) { gx[mt] = cur.px[mt] - cur.cx[mt]; gy[mt] = cur.py[mt] - cur.cy[mt]; gz[mt] = cur.pz[mt] - cur.cz[mt]; }
        f32x4 h2[T][MT];
#pragma unroll
        for (int ct = 0; ct < T; ++ct)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) h2[ct][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            constexpr int F = T * T;
            f32x4 ring[RD];
            if constexpr (WM != 2) {
#pragma unroll
                for (int f = 0; f < RD; ++f)
                    if (f < F) ring[f] = wfrag(a.W2, s_w2, f % T, f / T);
            }
#pragma unroll
            for (int ks = 0; ks < T; ++ks) {
                stage_step(ks);
                const int k0 = ks * 16 + fq * 4;
                const f32x4 wx = *reinterpret_cast<const f32x4 *>(s_wx + k0), wy = *reinterpret_cast<const f32x4 *>(s_wx + C + k0),
                            wz = *reinterpret_cast<const f32x4 *>(s_wx + 2 * C + k0);
                const f32x4 ps = *reinterpret_cast<const f32x4 *>(s_ps + k0), pf = *reinterpret_cast<const f32x4 *>(s_pf + k0);
                f32x4 h1[MT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = cur.raw[ks][mt][e] + __builtin_fmaf(wz[e], gz[mt], __builtin_fmaf(wy[e], gy[mt], wx[e] * gx[mt]));
                        h1[mt][e] = fmaxf(__builtin_fmaf(v, ps[e], pf[e]), 0.f);
                    }
#pragma unroll
                for (int ct = 0; ct < T; ++ct) {
                    const int f = ks * T + ct;
                    f32x4 w;
                    if constexpr (WM == 2) {   // two fragments of the step's buffer ahead of the MFMAs (the ring restarts at every k-step: the next buffer is not ready before its barrier)
                        if (ct == 0) { sring[0] = sfrag(ks, T, 0); sring[1] = sfrag(ks, T, 1); }
                        w = sring[ct & 1];
                        if (ct + 2 < T) sring[ct & 1] = sfrag(ks, T, ct + 2);
                    } else {
                        w = ring[f % RD];
                        if (f + RD < F) ring[f % RD] = wfrag(a.W2, s_w2, (f + RD) % T, (f + RD) / T);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (WM == 2 && kSpread) stage_issue_one(ks / kStageK + 1, (ks % kStageK) * T + ct);   // copy number (chain index) of the next chunk
                    if constexpr (WM != 2 && kSpread) { if (f < T * MT) rows_raw(trn, nxt, f); }                     // one table row of the next block per chain
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < NT; ++mt) h2[ct][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h1[mt][e], h2[ct][mt], 0, 0, 0);
                }
            }
        }
        if constexpr (WM != 2) cur = nxt;
        f32x4 ring3[RD];                             // layer 3's first fragments, requested before the seam
        constexpr int F3 = T * T3;
        if constexpr (WM != 2) {
#pragma unroll
            for (int f = 0; f < RD; ++f)
                if (f < F3) ring3[f] = wfrag(a.W3, s_w3, f % T3, f / T3);
        }
#pragma unroll
        for (int ct = 0; ct < T; ++ct) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc2 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh2 + ct * 16 + fq * 4);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[ct][mt][r] = fmaxf(__builtin_fmaf(h2[ct][mt][r], sc[r], sh[r]), 0.f);
        }
        // ---- layer 3, normal orientation (A = activations, B = weights): lane (fi, fq) holds rows 4 fq + r of channel 16 ct + fi
        f32x4 acc[T3][MT];
#pragma unroll
        for (int ct = 0; ct < T3; ++ct)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) acc[ct][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < F3; ++f) {
            const int ks = f / T3, ct = f % T3;
            if (ct == 0) {
                stage_step(T + ks);
                if constexpr (WM == 2) {
                    if (ks == 0) {   // the next block's rows: requested here, behind a k-step barrier -- every barrier drains this wave's loads (vmcnt(0)),
                                     // so they get the 2k cycles of layer 3's first k-step instead of stalling the block's first one
                        if constexpr (kSpread) rows_begin(block_of(it, 1), ivn, nxt, trn);   // (the table rows: between the chains below)
                        else load_rows(block_of(it, 1), ivn, nxt);
                        load_idx(block_of(it, 2), ivn);
                    }
                }
            }
            f32x4 w;
            if constexpr (WM == 2) {
                if (ct == 0) { sring[0] = sfrag(T + ks, T3, 0); sring[1] = sfrag(T + ks, T3, 1); }
                w = sring[ct & 1];
                if (ct + 2 < T3) sring[ct & 1] = sfrag(T + ks, T3, ct + 2);
            } else {
                w = ring3[f % RD];
                if (f + RD < F3) ring3[f % RD] = wfrag(a.W3, s_w3, (f + RD) % T3, (f + RD) / T3);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WM == 2 && kSpread) {
                stage_issue_one((T + ks) / kStageK + 1, (ks % kStageK) * T3 + ct);
                constexpr int ND = kStageK * T3 / 4;   // this wave's copies per layer-3 chunk: the next block's table rows follow them, one per chain
                if (f >= ND && f - ND < T * MT) rows_raw(trn, nxt, f - ND);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) acc[ct][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(h2[ks][mt][e], w[e], acc[ct][mt], 0, 0, 0);
        }
        // ---- affine, max over the rows, ReLU once per output (max_r relu(y_r) = relu(max_r y_r) exactly)
        float v[T3][MT];
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) {
            const float sc = s_sc3[ct * 16 + fi], sh = s_sh3[ct * 16 + fi];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const float y0 = __builtin_fmaf(acc[ct][mt][0], sc, sh), y1 = __builtin_fmaf(acc[ct][mt][1], sc, sh),
                            y2 = __builtin_fmaf(acc[ct][mt][2], sc, sh), y3 = __builtin_fmaf(acc[ct][mt][3], sc, sh);
                v[ct][mt] = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
            }
        }
        if constexpr (GP > 1) {                      // S == 16, MT == 2: tile mt is neighbourhood 2 blk + mt
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int g = blk * GP + mt;
#pragma unroll
                for (int c4 = 0; c4 < T3 / 4; ++c4) {
                    const float m = fmaxf(pool4_rows_max_t(v[c4 * 4][mt], v[c4 * 4 + 1][mt], v[c4 * 4 + 2][mt], v[c4 * 4 + 3][mt]), 0.f);
                    if (g < nq && live(it)) a.out[(size_t)g * a.ldo + a.col0 + c4 * 64 + lane] = m;
                }
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) {
                x[ct] = v[ct][0];
#pragma unroll
                for (int mt = 1; mt < NT; ++mt) x[ct] = fmaxf(x[ct], v[ct][mt]);
            }
            if constexpr (G > 1) {
                const bool first = blk_first(it);
#pragma unroll
                for (int ct = 0; ct < T3; ++ct) pm[ct] = first ? x[ct] : fmaxf(pm[ct], x[ct]);
            }
        }
        };   // compute
        if (!DEDUP || nt_cur == MT) compute(std::integral_constant<int, MT>{});
        else if (nt_cur == 1) compute(std::integral_constant<int, 1>{});
        else {                                       // a block of padding only: nothing to compute, the pipeline moves on
            if constexpr (WM != 2) {
                if constexpr (kSpread) {
#pragma unroll
                    for (int q = 0; q < T * MT; ++q) rows_raw(trn, nxt, q);
                }
                cur = nxt;
            }
        }
        nt_cur = nt_nxt;
        if constexpr (GP == 1) {
            if (blk_last(it)) {
                const int g = blk / G;
#pragma unroll
                for (int c4 = 0; c4 < T3 / 4; ++c4) {
                    float m;
                    if constexpr (G > 1) m = pool4_rows_max_t(pm[c4 * 4], pm[c4 * 4 + 1], pm[c4 * 4 + 2], pm[c4 * 4 + 3]);
                    else m = pool4_rows_max_t(x[c4 * 4], x[c4 * 4 + 1], x[c4 * 4 + 2], x[c4 * 4 + 3]);
                    if (g < nq && live(it)) a.out[(size_t)g * a.ldo + a.col0 + c4 * 64 + lane] = fmaxf(m, 0.f);
                }
            }
        }
        if constexpr (WM == 2) {
            cur = nxt;
            if (listed) { itm0 = itm1; itm1 = itm2; itm2 = itm3; itm3 = item_at(it + 4); }
        }
    }
}

}  // namespace g4d

using namespace g4d;

// workspace of the lock-step kernel's work list for nq neighbourhoods of G blocks dealt to at most nwg_max workgroups:
// counts[8] | order[G * nq] | wg_len[nwg_max] | items[nwg_max * 4 * cap]
namespace {
constexpr int kListWgMax = 1024;                 // >= resident workgroups of the lock-step kernel (2 per CU)
inline long long list_cap(long long nq, int G, long long nwg) { return (((nq + 3) / 4 + G + nwg - 1) / nwg + 1) * G; }
inline long long list_bytes(long long nq, int G) {
    // (cap shrinks as the grid grows: size for the smallest grid that can occur, min(want, resident) >= min(want, 1), bounded by the list itself)
    const long long want = (nq + 3) / 4;
    const long long nwg_lo = want < kListWgMax ? want : 256;   // a launch uses min(want, resident) workgroups, resident in [256, kListWgMax]
    const long long items = (want < kListWgMax ? want : (long long)kListWgMax) * 4 * list_cap(nq, G, nwg_lo);
    return 32 + 4 * ((long long)G * nq + kListWgMax + items);
}
}  // namespace

template <int C, int S, int MT, int WM>
static int sa_table_launch(SaTabArgs a, hipStream_t st, void *ws = nullptr, long long ws_bytes = 0) {
    constexpr bool WLDS = WM == 1;
    const int lds = (int)sizeof(float) * (11 * C + (WLDS ? 3 * C * C : 0) + (WM == 2 ? 2 * kStageK * (C / 8) * 256 : 0));
    static unsigned long long attr = 0;
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sa_table_kernel<C, S, MT, WM>), lds, attr, "g4d_sa_table");
        if (rc) return rc;
    }
    static const int resident = [] {
        const int lds = (int)sizeof(float) * (11 * C + (WLDS ? 3 * C * C : 0) + (WM == 2 ? 2 * kStageK * (C / 8) * 256 : 0));
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sa_table_kernel<C, S, MT, WM>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return per_cu * 256;
        return per_cu * prop.multiProcessorCount;
    }();
    constexpr int R = 16 * MT, G = S > R ? S / R : 1;
    const long long nunit = ((a.rows + R - 1) / R + G - 1) / G;
    const long long want = (nunit + 3) / 4;
    // (tuning key sa_table_oversub, default 1: more workgroups than fit an empty chip, each with fewer units -- the dispatcher then balances
    //  them over whatever CUs other streams' launches leave free; A/B switch)
    const long long cap = (long long)resident * (long long)tuning("sa_table_oversub", 1);
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    a.items = nullptr; a.wg_len = nullptr; a.cap = 0;
    if constexpr (WM == 2) {
        const long long nq = a.rows / S;
        if (ws && tuning("sa_table_dedup", 1) && a.rows % S == 0 && grid <= (unsigned)kListWgMax && ws_bytes >= list_bytes(nq, G) && nq * G < (1ll << 28)) {
            int *counts = reinterpret_cast<int *>(ws), *order = counts + 8, *wg_len = order + (size_t)G * nq, *items = wg_len + kListWgMax;
            const long long icap = list_cap(nq, G, grid);
            if (hipMemsetAsync(counts, 0, 32, st) != hipSuccess) return check_launch("g4d_sa_table(work list)");
            hipLaunchKernelGGL((sa_units_count_kernel<S>), dim3((unsigned)(((nq + 63) / 64 + kCountWaves - 1) / kCountWaves)), dim3(64 * kCountWaves), 0, st, (int)nq, (int)nq, a.idx, counts, order);
            hipLaunchKernelGGL(sa_units_items_kernel, dim3((grid * 4 + 255) / 256), dim3(256), 0, st, G, (int)nq, (int)grid, (int)icap, counts, order, items, wg_len);
            if (const int rc = check_launch("g4d_sa_table(work list)")) return rc;
            a.items = items; a.wg_len = wg_len; a.cap = (int)icap;
        }
    }
    hipLaunchKernelGGL((sa_table_kernel<C, S, MT, WM>), dim3(grid), dim3(256), lds, st, a);
    return check_launch("g4d_sa_table");
}

// The ONE statement of which launches the persistent kernel takes: the instantiated (Kt, S) pairs below, max pooling, enough rows to pipeline,
// and the calling thread's tuning state (A/B switches).  sa_table_try() and the host-side dispatch (fused.py: the xyz-only route hands over a
// table with row stride 0, which only this kernel reads) both ask here, so they cannot disagree.
extern "C" int g4d_sa_table_supported(long long rows, int Kt, int S, int pool) {
    const int on = (int)g4d::tuning("sa_table_persistent", 1);                 // A/B switch
    const long long min_rows = g4d::tuning("sa_table_min_rows", 262144);       // ~4 blocks of 32 rows per resident wave
    if (!on || pool != 1 || S <= 0 || rows < min_rows || rows >= (1ll << 31) - 64 || rows % S != 0) return 0;
    if ((Kt == 32 && (S == 16 || S == 32)) || (Kt == 64 && (S == 16 || S == 32 || S == 64))) return 1;
    const int wide = (int)g4d::tuning("sa_table_128", 1);                      // the 128-wide stack (weights streamed from L2: 192 KB do not fit LDS)
    return (wide && Kt == 128 && (S == 32 || S == 64)) ? 1 : 0;
}

// Bytes of caller-owned scratch with which a g4d_mlp_chain_group_table_ws_f32 launch of this shape skips the blocks of ball-query padding (the
// lock-step kernel's work list: Kt = 128 only -- the narrower stacks skip dead tiles without one); 0: the shape / tuning state takes none.
extern "C" long long g4d_sa_table_ws_bytes(long long rows, int Kt, int S, int pool) {
    if (Kt != 128 || !(S == 64 || S == 32) || !g4d_sa_table_supported(rows, Kt, S, pool) || !g4d::tuning("sa_table_dedup", 1) || rows % S != 0) return 0;
    return list_bytes(rows / S, S / 16);
}

// Takes the launch if it is one of the instantiated shapes and large enough to pipeline (several row blocks per resident wave);
// returns -1 when it is not (the caller then runs the register-chain kernel), else the launch status.
int g4d::sa_table_try(long long rows, int N, int P, int S, const float *xyz, const float *new_xyz, const int *idx, const float *table, int tab_ld,
                      int Kt, const float *tab_wx, const float *pre_scale, const float *pre_shift, int nlayers, const float *const *W,
                      const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout, const int *relu, int pool, float *out,
                      int ldo, int col0, hipStream_t st, void *ws, long long ws_bytes) {
    if (nlayers != 2 || !g4d_sa_table_supported(rows, Kt, S, pool)) return -1;
    if (Cout[0] != Kt || Cout[1] != 2 * Kt || !relu[0] || !relu[1] || Kpad[0] != Kt || Kpad[1] != Kt) return -1;
    if ((long long)(rows / S / P) * N >= (1ll << 31)) return -1;
    G4D_REQUIRE(xyz && new_xyz && idx && table && tab_wx && pre_scale && pre_shift && out && W[0] && W[1] && scale[0] && scale[1] && shift[0] && shift[1],
                "g4d_mlp_chain_group_table_f32: null pointer");
    G4D_REQUIRE(N > 0 && P > 0 && rows % ((long long)P * S) == 0 && ldo >= col0 + Cout[1] && col0 >= 0 && (tab_ld >= Kt || tab_ld == 0) && tab_ld % 4 == 0,
                "g4d_mlp_chain_group_table_f32: rows must be clouds x P x S, the output window [%d, %d) must fit ldo = %d, the table stride %d must cover %d columns (or be 0)",
                col0, col0 + Cout[1], ldo, tab_ld, Kt);
    SaTabArgs a;
    a.rows = (int)rows; a.N = N; a.P = P; a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.tab = table; a.tab_ld = tab_ld;
    a.wx = tab_wx; a.ps = pre_scale; a.pf = pre_shift;
    a.W2 = W[0]; a.sc2 = scale[0]; a.sh2 = shift[0];
    a.W3 = W[1]; a.sc3 = scale[1]; a.sh3 = shift[1];
    a.out = out; a.ldo = ldo; a.col0 = col0;
    if (Kt == 32 && S == 16) return sa_table_launch<32, 16, 2, 1>(a, st);
    if (Kt == 32 && S == 32) return sa_table_launch<32, 32, 2, 1>(a, st);
    if (Kt == 64 && S == 32) return sa_table_launch<64, 32, 2, 1>(a, st);
    if (Kt == 64 && S == 64) return sa_table_launch<64, 64, 2, 1>(a, st);
    if (Kt == 64 && S == 16) return sa_table_launch<64, 16, 1, 1>(a, st);
    const int wide = (int)tuning("sa_table_128", 1);   // A/B switch: the 128-wide stack (weights streamed from L2: 192 KB do not fit LDS)
    if (wide == 2 && Kt == 128 && S == 64) return sa_table_launch<128, 64, 1, 0>(a, st);   // (A/B: every wave streaming its own fragments)
    if (wide && Kt == 128 && S == 64) return sa_table_launch<128, 64, 1, 2>(a, st, ws, ws_bytes);
    if (wide && Kt == 128 && S == 32) return sa_table_launch<128, 32, 1, 2>(a, st, ws, ws_bytes);
    return -1;
}
