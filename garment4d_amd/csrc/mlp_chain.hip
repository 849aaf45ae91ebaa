// Register-resident shared-MLP stack: the activations never leave the VGPRs between layers -- no LDS, no barrier.
//
// v_mfma_f32_16x16x4_f32 computes D (16x16) += A (16x4) B (4x16) with lane l = (fi = l & 15, fq = l >> 4) supplying
// A[i = fi][k = fq], B[k = fq][j = fi] and holding D[i = 4 fq + r][j = fi], r = 0..3.  A and B have the SAME lane
// layout; D has the transposed one.  So if a layer is evaluated TRANSPOSED,
//
//      out^T (channels x rows) = W (channels x K)  .  h^T (K x rows)            A = weights, B = activations,
//
// lane (fi, fq) ends up with out[row = fi][channel = 16 ct + 4 fq + r] -- exactly the B-fragment
// h[row = fi][k = 16 ks + 4 fq + e] the NEXT layer wants (ct -> ks, r -> e).  The accumulators of one layer ARE the
// operand registers of the next: bias / BN affine / ReLU are applied in place and the chain continues.  The LAST layer
// swaps the operands (A = activations, B = weights): its output comes out row-major (rows on 4 fq + r, channels on fi),
// which is what the max / mean over the S samples (registers + two cross-lane steps) and the coalesced stores want.
//
// One wave owns 16 * MT rows; the first layer streams its input straight from the gather (GROUP: [x_j - q ; f_j], or a
// DIRECT matrix) 16 columns at a time; weights come from L2 in the fragment order the host already packs
// (garment4d_amd/fused.py: [16-channel tile][16-wide k-step][lane][4]), one 16-byte load per 4 * MT MFMAs.
// The LDS-staged kernels (mlp_stack.hip, mlp_wave.hip) pay an LDS round trip plus a fence or barrier per layer for the
// re-layout this kernel gets for free; they remain the route for stacks too wide for the register file (the widest tile
// combination here is 128-128-256 at 32 rows per wave: 173 VGPRs + 200 AGPRs) and for the INTERP / CSR loaders.
#include <cstdlib>

#include "mlp_common.h"
#include "chain_finish.h"

namespace g4d {

struct ChainLayer {
    const float *W;      // fragment order, [CoutPad64 / 16][Kpad / 16][64][4]
    const float *scale;  // [CoutPad64]
    const float *shift;
    int kst;             // Kpad / 16: k-steps per channel tile in W
    int relu, cout;
};

struct ChainArgs {
    LinearArgs in;  // loader + output description (W / scale / shift / Kpad / Cout unused)
    ChainLayer layer[4];
    int tap_layer;   // hidden layer whose activations are also written to HBM (-1: none)
    float *tap_out;
    int tap_ld;
    int dbg_off;     // debug builds: first stamp slot of this role (scripts/dbg_chain_phases2.py)
    int xcd_swz;     // workgroup -> row-block map: 1 = every XCD (blockIdx % 8) takes one CONTIGUOUS eighth of the row blocks
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// affine + ReLU of a TRANSPOSED accumulator tile (lane holds channels 16 ct + 4 fq + r)
template <int TOUT, int MT>
__device__ __forceinline__ void affine_t(const ChainLayer &L, int fq, f32x4 (&acc)[TOUT][MT]) {
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(L.scale + ct * 16 + fq * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(L.shift + ct * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = __builtin_fmaf(acc[ct][mt][r], sc[r], sh[r]);  // one v_fma_f32: packed mul + add pairs beside MFMAs stall the matrix pipe
                if (L.relu) y = fmaxf(y, 0.f);
                acc[ct][mt][r] = y;
            }
    }
}

// affine + ReLU of a ROW-MAJOR accumulator tile (lane holds channel 16 ct + fi)
template <int TOUT, int MT>
__device__ __forceinline__ void affine_r(const ChainLayer &L, int fi, f32x4 (&acc)[TOUT][MT]) {
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) {
        const float sc = L.scale[ct * 16 + fi], sh = L.shift[ct * 16 + fi];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = __builtin_fmaf(acc[ct][mt][r], sc, sh);
                if (L.relu) y = fmaxf(y, 0.f);
                acc[ct][mt][r] = y;
            }
    }
}

template <int TOUT, int MT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TOUT][MT]) {
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Weight fragments are streamed from L2 with explicit software prefetch.  A wave of the wide stacks runs alone on its SIMD
// (173 VGPRs + 200 AGPRs), so nothing else hides an L2 round trip (500+ cycles under load): left to the compiler, a fragment
// load was issued ~1 fragment (8 MFMAs = 256 cycles) ahead of its use -- `s_waitcnt vmcnt(1)` before almost every MFMA group.
// Here fragment f + kWDepth is requested before the MFMAs of fragment f are issued (the empty asm with a memory clobber keeps
// the compiler from sinking the load back down), i.e. kWDepth KB per wave stay in flight.
constexpr int kWDepth = 6;
constexpr int kDeepTab = 8;   // table rows of up to this many 16-column k-steps are requested whole, up front (GROUP table loader)
#ifndef G4D_NO_DEEP_TAB
#define G4D_NO_DEEP_TAB 0
#endif

__device__ __forceinline__ f32x4 load_wfrag(const ChainLayer &L, int ct, int ks, int lane) {
    return *reinterpret_cast<const f32x4 *>(L.W + ((size_t)(ct * L.kst + ks) * 64 + lane) * 4);
}

// the first kWDepth fragments of a chained layer (fragment f = ks * TOUT + ct), requested before the previous layer's epilogue
template <int TIN, int TOUT>
__device__ __forceinline__ void preload_ring(const ChainLayer &L, int lane, f32x4 (&ring)[kWDepth]) {
#pragma unroll
    for (int f = 0; f < kWDepth; ++f)
        if (f < TIN * TOUT) ring[f] = load_wfrag(L, f % TOUT, f / TOUT, lane);
}

#ifdef G4D_CHAIN_DEBUG
__device__ long long g_chain_dbg[8 * 4096];  // per wave (4096 of them, from row block g_chain_dbg_base on): 8 clock stamps (scripts/dbg_chain_phases.py)
__device__ int g_chain_dbg_base = 0;         // first recorded row block: a window in the MIDDLE of a large launch shows the steady state (scripts/dbg_chain_steady.py)
__shared__ int g_dbg_bid;
#define G4D_CSTAMP(i) { const int gw_ = (g_dbg_bid - g_chain_dbg_base) * 4 + (threadIdx.x >> 6); if ((threadIdx.x & 63) == 0 && gw_ >= 0 && gw_ < 4096) g_chain_dbg[gw_ * 8 + (i)] = (long long)__builtin_readcyclecounter(); }
#else
#define G4D_CSTAMP(i)
#endif

// First layer: the input is streamed from the loader, 16 columns (one k-step) at a time.
//
// In-kernel cycle stamps and the SQ counters showed this phase bound by INSTRUCTION issue / fetch, not by MFMA or memory:
// ~6 non-MFMA instructions per MFMA at ~14 cycles each (straight-line code, executed once per wave, fetched cold), most of
// them the per-lane, per-row-tile branching of the loader.  So:
//   * rows past the end are CLAMPED to the last row instead of masked -- an output row depends on its own input row only,
//     and rows >= `rows` are never stored or pooled into a stored group -- which removes every `valid` branch;
//   * whether a k-step can use the 16-byte vector path is decided once per step for the whole wave (all 16 columns inside
//     one source segment), not per lane: the k loop has ONE wave-uniform branch, the common body is base pointer + offset;
//   * weights and inputs of step ks + 1 are requested before the MFMAs of step ks.
//     (Spreading the requests BETWEEN the MFMAs of a step -- one weight fragment per channel tile, two buffer sets alternating so
//     that the double buffer needs no copies -- measured the same within noise: 128-128-256 stack 48.9 -> 48.5 us alone but 0.58 ->
//     0.565 of peak inside bench.py's harness, 16-batch bench +0.5 %.  With no loads at all in this loop the phase still takes 35.3k
//     cycles for 26.6k of MFMA issue at one wave per SIMD: what is left is the wave's start-up chain (kernel arguments -> neighbour
//     index -> row address -> first gather, each a dependent round trip) and the two element-wise seam steps, which only more waves
//     per SIMD or a persistent tile loop can hide.  Round 2.)
//     (Requesting 2-4 steps ahead instead -- the widest stack runs one wave per SIMD -- costs 20-70 registers per instantiation and
//     measured slower everywhere: 128-128-256 stack 49.2 -> 50.4 us alone, the 16-batch bench 27.3k -> 25.7k frames/s; round 2.)
template <int MODE, int TOUT, int MT, bool LAST>
__device__ __forceinline__ void first_layer(const LinearArgs &a, const ChainLayer &L, int lane, int row0, f32x4 (&acc)[TOUT][MT]) {
    const int fi = lane & 15, fq = lane >> 4;
    RowCtx<MODE> ctx[MT];
    int rowc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rowc[mt] = min(row0 + mt * 16 + fi, a.rows - 1);
        ctx[mt] = make_ctx<MODE>(a, rowc[mt]);
    }
    zero_acc<TOUT, MT>(acc);
    if constexpr (MODE == LOAD_INTERP && !LAST) {
        if (a.tab) {   // wave-uniform: conv(sum_i w_i f_i) = sum_i w_i conv(f_i) -- the known-feature part of this layer, per known row, interpolated
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ct = 0; ct < TOUT; ++ct) {   // transposed tile: this lane holds channels 16 ct + 4 fq .. + 3 of row fi
                    const int ch = ct * 16 + fq * 4;
                    const f32x4 t0 = *reinterpret_cast<const f32x4u *>(a.tab + ctx[mt].k0 + ch), t1 = *reinterpret_cast<const f32x4u *>(a.tab + ctx[mt].k1 + ch),
                                t2 = *reinterpret_cast<const f32x4u *>(a.tab + ctx[mt].k2 + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[ct][mt][e] = ctx[mt].w0 * t0[e] + ctx[mt].w1 * t1[e] + ctx[mt].w2 * t2[e];
                }
        }
    }
    const int kst0 = (a.K + 15) >> 4;
    // segment A: columns [a_lo, a_hi) come from pa[mt] + column (GROUP: the feature row behind the 3 xyz columns; DIRECT: the
    // row; INTERP: the three known rows, blended); segment B (INTERP only): columns [C2, K) from the skip row
    const float *pa[MT], *pa1[MT], *pa2[MT], *pb[MT];
    int a_lo, a_hi, b_lo = 0, b_hi = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pa1[mt] = pa2[mt] = pb[mt] = nullptr;
        if constexpr (MODE == LOAD_GROUP) {
            if (a.tab) pa[mt] = a.tab + ctx[mt].pt_base * a.tab_ld;   // pre-contracted table: the whole row is "features"
            else pa[mt] = (a.feats ? a.feats : a.xyz) + ctx[mt].pt_base * a.C - (a.use_xyz ? 3 : 0);  // xyz-only stacks never take the fast path
        }
        else if constexpr (MODE == LOAD_DIRECT) pa[mt] = a.X + (size_t)rowc[mt] * a.ldx;
        else { pa[mt] = a.known_feats + ctx[mt].k0; pa1[mt] = a.known_feats + ctx[mt].k1; pa2[mt] = a.known_feats + ctx[mt].k2;
               pb[mt] = a.skip + ctx[mt].sk - a.C2; }
    }
    if constexpr (MODE == LOAD_GROUP) { a_lo = (a.use_xyz && !a.tab) ? 3 : 0; a_hi = a.K; }
    else if constexpr (MODE == LOAD_DIRECT) { a_lo = 0; a_hi = a.K; }
    else { a_lo = 0; a_hi = a.C2; b_lo = a.C2; b_hi = a.K; }
    float gdx[MT], gdy[MT], gdz[MT];   // table mode: x_j - q of the row (pointnet2_utils.py:254), multiplied by the xyz columns of the weight below
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        gdx[mt] = gdy[mt] = gdz[mt] = 0.f;
        if constexpr (MODE == LOAD_GROUP) {
            if (a.tab) {
                const float *pp = a.xyz + ctx[mt].pt_base * 3;
                gdx[mt] = pp[0] - ctx[mt].cx; gdy[mt] = pp[1] - ctx[mt].cy; gdz[mt] = pp[2] - ctx[mt].cz;
            }
        }
    }
    auto load_b = [&](int ks, f32x4 (&b)[MT]) {
        const int c0 = ks * 16;                       // wave-uniform: the step's 16 columns are [c0, c0 + 16)
        const int k0 = c0 + fq * 4;                   // this lane's 4 consecutive columns
        if (c0 >= a_lo && c0 + 16 <= a_hi) {          // wave-uniform fast path, segment A
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (MODE == LOAD_INTERP) {  // three_interpolate of 4 consecutive channels, load_elem's operation order
                    const f32x4 f0 = *reinterpret_cast<const f32x4u *>(pa[mt] + k0), f1 = *reinterpret_cast<const f32x4u *>(pa1[mt] + k0),
                                f2 = *reinterpret_cast<const f32x4u *>(pa2[mt] + k0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[mt][e] = ctx[mt].w0 * f0[e] + ctx[mt].w1 * f1[e] + ctx[mt].w2 * f2[e];
                    if (a.pre_scale) {   // wave-uniform: the known rows are a pre-contracted table, the layer's affine + ReLU happen here
                        const f32x4 ps = *reinterpret_cast<const f32x4 *>(a.pre_scale + k0), pf = *reinterpret_cast<const f32x4 *>(a.pre_shift + k0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[mt][e] = fmaxf(__builtin_fmaf(b[mt][e], ps[e], pf[e]), 0.f);
                        if (a.in_tap && row0 + mt * 16 + fi < a.rows)
                            *reinterpret_cast<f32x4 *>(a.in_tap + (size_t)out_row(a, row0 + mt * 16 + fi) * a.in_tap_ld + k0) = b[mt];
                    }
                } else {
                    b[mt] = *reinterpret_cast<const f32x4u *>(pa[mt] + k0);
                    if constexpr (MODE == LOAD_GROUP) {
                        if (a.tab) {   // wave-uniform: W [x_j - q ; f_j] = Wx (x_j - q) + (Wf f_j): the second term is the table row
                            const f32x4 wx = *reinterpret_cast<const f32x4 *>(a.tab_wx + k0), wy = *reinterpret_cast<const f32x4 *>(a.tab_wx + a.K + k0),
                                        wz = *reinterpret_cast<const f32x4 *>(a.tab_wx + 2 * a.K + k0);
                            const f32x4 ps = *reinterpret_cast<const f32x4 *>(a.pre_scale + k0), pf = *reinterpret_cast<const f32x4 *>(a.pre_shift + k0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = b[mt][e] + __builtin_fmaf(wz[e], gdz[mt], __builtin_fmaf(wy[e], gdy[mt], wx[e] * gdx[mt]));
                                b[mt][e] = fmaxf(__builtin_fmaf(v, ps[e], pf[e]), 0.f);
                            }
                        }
                    }
                }
            }
        } else if (MODE == LOAD_INTERP && c0 >= b_lo && c0 + 16 <= b_hi) {   // wave-uniform fast path, segment B
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const f32x4u *>(pb[mt] + k0);
        } else {                                      // a step that straddles a seam or the end of the row: element by element
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[mt][e] = load_elem<MODE>(a, ctx[mt], rowc[mt], k0 + e);
        }
    };
    if constexpr (MODE == LOAD_GROUP) {
        if (a.tab && TOUT <= kDeepTab && a.K == 16 * TOUT && !G4D_NO_DEEP_TAB) {   // wave-uniform: table as wide as this layer's output (c, c, 2c stacks)
            constexpr int KS = TOUT <= kDeepTab ? TOUT : 1;   // k-steps, compile-time: the loops below are straight-line code
            // Table loader: this "first layer" is the SECOND layer's contraction fed by relu(affine(table[j] + Wx (x_j - q))).  In-kernel
            // stamps (SA level 3, 128-wide table, scripts/dbg_chain_phases2.py) put 36k of a wave's 82k cycles here for 8k cycles of
            // MFMA issue when gathers and weights were requested one k-step ahead.  A table row is at most 8 k-steps, so ALL of it is
            // requested up front (4 registers per k-step and row tile: one gather latency instead of eight) and turned into operands in
            // place; the k-steps then run as straight-line code off a weight-fragment ring, exactly like a chained layer.
            // Round 3 measurements that bound what is left (same script; 16-batch bench in parentheses):
            //   * no weight loads at all anywhere in the chain kernels (wrong results, timing only): 34.7k -> 35.7k frames/s -- weight
            //     delivery (L2 -> registers, ring depth, an LDS-shared ring) is worth < 3 %; ring depth 12 / 16: 34.3k / 33.4k;
            //   * each workgroup walking 2 / 4 row tiles in a loop (warm instruction cache from the second tile on: 67k -> 57k cycles
            //     per tile alone on a SIMD): 28.1k / 28.9k frames/s -- the waves it takes off the SIMDs cost more than the fetches;
            //   * of two waves sharing a SIMD the older one wins MFMA arbitration (its tile takes 55k cycles, the younger one's 73k):
            //     co-resident waves de-phase by themselves, which is why staggering them by hand measured neutral.
            G4D_CSTAMP(5)
            f32x4 raw[KS][MT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) raw[ks][mt] = *reinterpret_cast<const f32x4u *>(pa[mt] + ks * 16 + fq * 4);
            // weights: the same fragment ring as chain_layer (kWDepth KB per wave in flight all the time); requested a k-step at a time
            // (a burst of TOUT fragments, then a full drain before the step's MFMAs) the same loads took 3-4k cycles per step
            f32x4 wring[kWDepth];
#pragma unroll
            for (int f = 0; f < kWDepth; ++f)
                if (f < KS * TOUT) wring[f] = load_wfrag(L, f % TOUT, f / TOUT, lane);
            // operands: relu(affine(table row + Wx (x_j - q))) in place, the arithmetic of load_b's table branch operation for operation
            f32x4 cwx, cwy, cwz, cps, cpf;
            auto load_small = [&](int ks) {
                const int k0 = ks * 16 + fq * 4;
                cwx = *reinterpret_cast<const f32x4 *>(a.tab_wx + k0); cwy = *reinterpret_cast<const f32x4 *>(a.tab_wx + a.K + k0);
                cwz = *reinterpret_cast<const f32x4 *>(a.tab_wx + 2 * a.K + k0);
                cps = *reinterpret_cast<const f32x4 *>(a.pre_scale + k0); cpf = *reinterpret_cast<const f32x4 *>(a.pre_shift + k0);
            };
            load_small(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f32x4 wx = cwx, wy = cwy, wz = cwz, ps = cps, pf = cpf;
                if (ks + 1 < KS) load_small(ks + 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = raw[ks][mt][e] + __builtin_fmaf(wz[e], gdz[mt], __builtin_fmaf(wy[e], gdy[mt], wx[e] * gdx[mt]));
                        raw[ks][mt][e] = fmaxf(__builtin_fmaf(v, ps[e], pf[e]), 0.f);
                    }
            }
            G4D_CSTAMP(6)
            constexpr int G = (MT == 1 && TOUT % 2 == 0) ? 2 : 1;
            static_assert(kWDepth % 2 == 0, "the ring is consumed G fragments at a time");
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int ct = 0; ct < TOUT; ct += G) {
                    const int f = ks * TOUT + ct;
                    f32x4 w[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        w[g] = wring[(f + g) % kWDepth];
                        if (f + g + kWDepth < KS * TOUT)
                            wring[(f + g) % kWDepth] = load_wfrag(L, (f + g + kWDepth) % TOUT, (f + g + kWDepth) / TOUT, lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[ct + g][mt] = LAST ? mfma4(raw[ks][mt][e], w[g][e], acc[ct + g][mt]) : mfma4(w[g][e], raw[ks][mt][e], acc[ct + g][mt]);
                }
#ifdef G4D_CHAIN_DEBUG
                if (ks == 0) { G4D_CSTAMP(7) }
#endif
            }
            return;
        }
    }
    f32x4 wn[TOUT], bn[MT];
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) wn[ct] = load_wfrag(L, ct, 0, lane);
    load_b(0, bn);
    for (int ks = 0; ks < kst0; ++ks) {
        f32x4 w[TOUT], b[MT];
#pragma unroll
        for (int ct = 0; ct < TOUT; ++ct) w[ct] = wn[ct];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
        if (ks + 1 < kst0) {  // wave-uniform
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct) wn[ct] = load_wfrag(L, ct, ks + 1, lane);
            load_b(ks + 1, bn);
        }
        __builtin_amdgcn_sched_barrier(0);  // nothing is scheduled across: the requests above stay ahead of the MFMAs below
        // consecutive MFMAs must not share an accumulator (40-cycle dependent latency vs 32-cycle issue): with one row tile per
        // wave (MT = 1) two channel tiles are interleaved, with MT = 2 the two row tiles already alternate
        constexpr int G = (MT == 1 && TOUT % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int ct = 0; ct < TOUT; ct += G)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[ct + g][mt] = LAST ? mfma4(b[mt][e], w[ct + g][e], acc[ct + g][mt]) : mfma4(w[ct + g][e], b[mt][e], acc[ct + g][mt]);
    }
}

// later layers: the previous layer's accumulators are the operand fragments; `ring` holds weight fragments 0 .. kWDepth-1
template <int TIN, int TOUT, int MT, bool LAST>
__device__ __forceinline__ void chain_layer(const ChainLayer &L, int lane, const f32x4 (&hin)[TIN][MT], f32x4 (&acc)[TOUT][MT],
                                            f32x4 (&ring)[kWDepth]) {
    zero_acc<TOUT, MT>(acc);
    constexpr int F = TIN * TOUT;
    constexpr int G = (MT == 1 && TOUT % 2 == 0) ? 2 : 1;  // channel tiles interleaved per step (see first_layer)
    static_assert(kWDepth % 2 == 0, "the ring is consumed G fragments at a time");
#pragma unroll
    for (int f = 0; f < F; f += G) {
        const int ks = f / TOUT, ct = f % TOUT;
        f32x4 w[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            w[g] = ring[(f + g) % kWDepth];
            if (f + g + kWDepth < F) ring[(f + g) % kWDepth] = load_wfrag(L, (f + g + kWDepth) % TOUT, (f + g + kWDepth) / TOUT, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[ct + g][mt] = LAST ? mfma4(hin[ks][mt][e], w[g][e], acc[ct + g][mt]) : mfma4(w[g][e], hin[ks][mt][e], acc[ct + g][mt]);
    }
}

// epilogue of a layer whose accumulators feed the next one (transposed) or the pooling / stores (row-major)
template <int TOUT, int MT, bool LAST>
__device__ __forceinline__ void affine(const ChainLayer &L, int lane, f32x4 (&acc)[TOUT][MT]) {
    if (LAST) affine_r<TOUT, MT>(L, lane & 15, acc);
    else affine_t<TOUT, MT>(L, lane >> 4, acc);
}

// hidden-layer tap: the transposed tile holds out[row = 16 mt + fi][channels 16 ct + 4 fq .. + 3]
template <int TOUT, int MT>
__device__ __forceinline__ void tap_store(const ChainArgs &s, int cout, int lane, int row0, const f32x4 (&h)[TOUT][MT]) {
    const int fi = lane & 15, fq = lane >> 4;
    const bool vec = (cout & 3) == 0 && (s.tap_ld & 3) == 0 && (reinterpret_cast<size_t>(s.tap_out) & 15) == 0;  // wave-uniform
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = row0 + mt * 16 + fi;
        if (row >= s.in.rows) continue;
        float *dst = s.tap_out + (size_t)out_row(s.in, row) * s.tap_ld + fq * 4;
        if (vec) {  // a lane's 4 channels are one aligned 16-byte store
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct)
                if (ct * 16 + fq * 4 < cout) *reinterpret_cast<f32x4 *>(dst + ct * 16) = h[ct][mt];
        } else {
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ct * 16 + fq * 4 + r < cout) dst[ct * 16 + r] = h[ct][mt][r];
        }
    }
}

// One role of a launch: workgroup `bid` of `nb` runs the stack described by `s` over its 64 * MT rows.
template <int MODE, int T1, int T2, int T3, int T4, int MT>
__device__ __forceinline__ void chain_body(const ChainArgs &s, int bid, int nb, float *xch) {
    const LinearArgs &a = s.in;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2.  With the identity map every
    // XCD touches every cloud's gather source (tables, features: 4 MB at the last FP level = one whole L2); with the swizzle XCD x owns
    // the x-th contiguous eighth of the row blocks -- at B = 8 one cloud -- and its L2 holds only that eighth of the source.
    // (Measured neutral on the cfg2 step, 32.9k frames/s either way: the gathers are latency-, not capacity-bound.  Kept: it is free.)
    int wg = bid;
    if (s.xcd_swz) {
        const int q = nb >> 3, r = nb & 7, x = wg & 7;
        wg = x * q + min(x, r) + (wg >> 3);
    }
    const int row0 = (wg * 4 + wave) * (16 * MT);
#ifdef G4D_CHAIN_DEBUG
    __syncthreads();
    if (threadIdx.x == 0) g_dbg_bid = bid + s.dbg_off;
    __syncthreads();
#endif
    G4D_CSTAMP(0)
    f32x4 h1[T1][MT];
    f32x4 ring[kWDepth];  // weight fragments in flight for the next chained layer (requested BEFORE the current layer's epilogue)
    if constexpr (T2 == 0) {
        first_layer<MODE, T1, MT, true>(a, s.layer[0], lane, row0, h1);
        G4D_CSTAMP(1)
        affine<T1, MT, true>(s.layer[0], lane, h1);
        finish<T1, MT>(a, s.layer[0].cout, lane, wave, row0, h1, xch);
    } else {
        first_layer<MODE, T1, MT, false>(a, s.layer[0], lane, row0, h1);
        G4D_CSTAMP(1)
        preload_ring<T1, T2>(s.layer[1], lane, ring);
        affine<T1, MT, false>(s.layer[0], lane, h1);
        if (s.tap_layer == 0) tap_store<T1, MT>(s, s.layer[0].cout, lane, row0, h1);
        f32x4 h2[T2][MT];
        if constexpr (T3 == 0) {
            chain_layer<T1, T2, MT, true>(s.layer[1], lane, h1, h2, ring);
            G4D_CSTAMP(3)
            affine<T2, MT, true>(s.layer[1], lane, h2);
            finish<T2, MT>(a, s.layer[1].cout, lane, wave, row0, h2, xch);
        } else {
            chain_layer<T1, T2, MT, false>(s.layer[1], lane, h1, h2, ring);
            G4D_CSTAMP(2)
            preload_ring<T2, T3>(s.layer[2], lane, ring);
            affine<T2, MT, false>(s.layer[1], lane, h2);
            if (s.tap_layer == 1) tap_store<T2, MT>(s, s.layer[1].cout, lane, row0, h2);
            f32x4 h3[T3][MT];
            if constexpr (T4 == 0) {
                chain_layer<T2, T3, MT, true>(s.layer[2], lane, h2, h3, ring);
                G4D_CSTAMP(3)
                affine<T3, MT, true>(s.layer[2], lane, h3);
                finish<T3, MT>(a, s.layer[2].cout, lane, wave, row0, h3, xch);
            } else {
                chain_layer<T2, T3, MT, false>(s.layer[2], lane, h2, h3, ring);
                preload_ring<T3, T4>(s.layer[3], lane, ring);
                affine<T3, MT, false>(s.layer[2], lane, h3);
                if (s.tap_layer == 2) tap_store<T3, MT>(s, s.layer[2].cout, lane, row0, h3);
                f32x4 h4[T4][MT];
                chain_layer<T3, T4, MT, true>(s.layer[3], lane, h3, h4, ring);
                G4D_CSTAMP(3)
                affine<T4, MT, true>(s.layer[3], lane, h4);
                finish<T4, MT>(a, s.layer[3].cout, lane, wave, row0, h4, xch);
            }
        }
    }
    G4D_CSTAMP(4)
}

// (Round 4: a row-block loop around chain_body -- persistent workgroups, arguments re-read per iteration so that nothing is hoisted --
//  measured SLOWER at 240 clouds per launch: SA level 3 scale 1 912 vs 873 us, the last FP level 846 vs 810, SA level 2's pair 746 vs 700, and
//  the loop's codegen cost the single-iteration form 5-8 % as well.  Workgroup turnover is not what these launches wait for; the exposed
//  latencies inside a row block are (scripts/dbg_chain_steady.py) -- sa_table.hip is the form that removes them.)
template <int MODE, int T1, int T2, int T3, int T4, int MT>
__global__ void __launch_bounds__(256) mlp_chain_kernel(const ChainArgs s) {
    __shared__ float xch[4 * 256];  // pooling partials of the 4 waves when a group spans waves (<= 256 channels)
    chain_body<MODE, T1, T2, T3, T4, MT>(s, blockIdx.x, gridDim.x, xch);
}

// Two independent stacks in ONE launch: workgroups [0, nb0) run role A, the rest role B (each a grid of its own).  Every launch costs
// the 16-stream regime ~2.4 us of serialised dispatch on top of its ramp and tail (scripts/exp_dispatch.py: a one-workgroup kernel
// costs 0.9 us per launch with 4-8 streams in flight, 2.35 with 16, 6.5 with 32), so the two scales of an MSG level share one.
template <int MODE, int A1, int A2, int A3, int A4, int AMT, int B1, int B2, int B3, int B4, int BMT>
__global__ void __launch_bounds__(256) mlp_chain_pair_kernel(const ChainArgs sa, const ChainArgs sb, int nb0) {
    __shared__ float xch[4 * 256];
    if ((int)blockIdx.x < nb0) chain_body<MODE, A1, A2, A3, A4, AMT>(sa, blockIdx.x, nb0, xch);
    else chain_body<MODE, B1, B2, B3, B4, BMT>(sb, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, xch);
}

template <int T1, int T2, int T3, int T4, int MT>
static void launch_chain(int mode, const ChainArgs &s, hipStream_t st) {
    const long long rows_per_wg = 4ll * 16 * MT;
    dim3 grid((unsigned)((s.in.rows + rows_per_wg - 1) / rows_per_wg)), block(256);
    if (mode == LOAD_GROUP) hipLaunchKernelGGL((mlp_chain_kernel<LOAD_GROUP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else if (mode == LOAD_INTERP) hipLaunchKernelGGL((mlp_chain_kernel<LOAD_INTERP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else hipLaunchKernelGGL((mlp_chain_kernel<LOAD_DIRECT, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
}

}  // namespace g4d

using namespace g4d;

// tile counts (16 channels each) of the supported stacks; the host mirrors this table (garment4d_amd/fused.py:chain_fits)
static int chain_key(int nlayers, const int *Cout) {
    if (nlayers < 1 || nlayers > 4 || !Cout) return -1;
    int key = 0;
    for (int l = 0; l < 4; ++l) key = key * 100 + (l < nlayers ? (Cout[l] + 15) / 16 : 0);
    return key;
}

#ifdef G4D_CHAIN_DEBUG
extern "C" int g4d_chain_debug_read(long long *host_out) {  // 8 x 4096 cycle stamps of the last launch
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g4d::g_chain_dbg), sizeof(long long) * 8 * 4096);
}
extern "C" int g4d_chain_debug_base(int first_block) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g4d::g_chain_dbg_base), &first_block, sizeof(int));
}
#endif

extern "C" int g4d_mlp_chain_supported(int nlayers, const int *Cout) {
    switch (chain_key(nlayers, Cout)) {
        case 1010200: case 2020400: case 4040800: case 8081600:     // 16-16-32, 32-32-64, 64-64-128, 128-128-256
        case 2020000: case 4040000: case 8080000: case 8040000:     // 32-32, 64-64, 128-128, 128-64
        case 16080000:                                              // 256-128 (the middle feature-propagation level)
        case 16080800:                                              // 256-128-128: ... followed by the next level's first-layer table
        case 1000000: case 2000000: case 4000000: case 8000000:     // single layers up to 128
        case 16000000:                                              // ... and 256 (behind a pre-contracted first layer)
        case 8040201:                                               // 128-64-32-(<=16): last FP level + segmentation head
        case 4020100:                                               // 64-32-(<=16): the head alone, behind a pre-contracted first layer
        case 2040000: case 4080000: case 8160000:                   // 32-64, 64-128, 128-256: SA stacks behind a pre-contracted first layer
            return 1;
        default: return 0;
    }
}

static int chain_launch_one(int mode, int key, int mt, const ChainArgs &s, hipStream_t st) {
#define G4D_CHAIN(T1, T2, T3, T4)                                   \
    if (mt == 2) launch_chain<T1, T2, T3, T4, 2>(mode, s, st);      \
    else launch_chain<T1, T2, T3, T4, 1>(mode, s, st);              \
    break;
    switch (key) {
#ifdef G4D_CHAIN_FEW   // development builds (-S listings): the instantiations of the cfg2 step only
        case 8160000: G4D_CHAIN(8, 16, 0, 0)
        case 4080000: G4D_CHAIN(4, 8, 0, 0)
        case 16080800: G4D_CHAIN(16, 8, 8, 0)
        default: G4D_CHAIN(8, 4, 2, 1)
#else
        case 1010200: G4D_CHAIN(1, 1, 2, 0)
        case 2020400: G4D_CHAIN(2, 2, 4, 0)
        case 4040800: G4D_CHAIN(4, 4, 8, 0)
        case 8081600: G4D_CHAIN(8, 8, 16, 0)
        case 2020000: G4D_CHAIN(2, 2, 0, 0)
        case 4040000: G4D_CHAIN(4, 4, 0, 0)
        case 8080000: G4D_CHAIN(8, 8, 0, 0)
        case 8040000: G4D_CHAIN(8, 4, 0, 0)
        case 16080000: G4D_CHAIN(16, 8, 0, 0)
        case 16080800: G4D_CHAIN(16, 8, 8, 0)
        case 1000000: G4D_CHAIN(1, 0, 0, 0)
        case 2000000: G4D_CHAIN(2, 0, 0, 0)
        case 4000000: G4D_CHAIN(4, 0, 0, 0)
        case 8000000: G4D_CHAIN(8, 0, 0, 0)
        case 16000000: G4D_CHAIN(16, 0, 0, 0)
        case 4020100: G4D_CHAIN(4, 2, 1, 0)
        case 2040000: G4D_CHAIN(2, 4, 0, 0)
        case 4080000: G4D_CHAIN(4, 8, 0, 0)
        case 8160000: G4D_CHAIN(8, 16, 0, 0)
        default: G4D_CHAIN(8, 4, 2, 1)
#endif
    }
#undef G4D_CHAIN
    return check_launch("g4d_mlp_chain_f32");
}

// ---- launch groups: independent stacks recorded between g4d_launch_group_begin() and g4d_launch_group_end() go out as ONE launch
// when a merged kernel is instantiated for the combination (the two scales of an MSG level), one launch each otherwise.  Per host
// thread, like the error text.
constexpr int kMaxRoles = 4;
struct ChainRole { ChainArgs s; int mode, key, mt; };
static thread_local struct { bool active; int n; ChainRole role[kMaxRoles]; } g_group = {false, 0, {}};

static long long role_blocks(const ChainRole &r) { const long long per = 64ll * r.mt; return (r.s.in.rows + per - 1) / per; }

// role A = the heavier stack (its workgroups are dispatched first: the launch's tail is then the light role's)
static bool chain_launch_pair(const ChainRole &A, const ChainRole &B, hipStream_t st) {
    if (A.mode != LOAD_GROUP || B.mode != LOAD_GROUP) return false;
    const long long na = role_blocks(A), nb = role_blocks(B);
    if (na + nb >= (1ll << 31)) return false;
    ChainArgs bs = B.s;
    bs.dbg_off = (int)na;
    const dim3 grid((unsigned)(na + nb)), block(256);
#define G4D_PAIR(KA, MA, KB, MB, ...)                                                                                   \
    if (A.key == KA && A.mt == MA && B.key == KB && B.mt == MB) {                                                       \
        hipLaunchKernelGGL((mlp_chain_pair_kernel<LOAD_GROUP, __VA_ARGS__>), grid, block, 0, st, A.s, bs, (int)na);    \
        return true;                                                                                                    \
    }
    G4D_PAIR(4080000, 2, 2040000, 1, 4, 8, 0, 0, 2, 2, 4, 0, 0, 1)     // SA level 2 of Pointnet2MSGSEG at B = 8: 64-128 on 65536 rows + 32-64 on 32768
    G4D_PAIR(8160000, 1, 4080000, 1, 8, 16, 0, 0, 1, 4, 8, 0, 0, 1)    // SA level 3: 128-256 on 32768 rows + 64-128 on 16384
    G4D_PAIR(4080000, 2, 2040000, 2, 4, 8, 0, 0, 2, 2, 4, 0, 0, 2)     // larger batches (both scales at 32 rows per wave)
    G4D_PAIR(4080000, 1, 2040000, 1, 4, 8, 0, 0, 1, 2, 4, 0, 0, 1)     // smaller ones
#undef G4D_PAIR
    return false;
}

extern "C" int g4d_launch_group_begin(void) {
    G4D_REQUIRE(!g_group.active, "g4d_launch_group_begin: a group is already open on this thread");
    g_group.active = true;
    g_group.n = 0;
    return G4D_OK;
}

// Launches what was recorded (on `stream`) and closes the group; *launches (optional) = the number of kernel launches it took.
extern "C" int g4d_launch_group_end(g4d_stream_t stream, int *launches) {
    G4D_REQUIRE(g_group.active, "g4d_launch_group_end: no open group on this thread");
    g_group.active = false;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int n = g_group.n;
    g_group.n = 0;
    static const int merge = getenv("G4D_LAUNCH_GROUPS") ? atoi(getenv("G4D_LAUNCH_GROUPS")) : 1;   // A/B switch: 0 = one launch per recorded stack
    if (launches) *launches = n;
    if (n == 2 && merge) {
        const ChainRole &r0 = g_group.role[0], &r1 = g_group.role[1];
        const bool heavy0 = role_blocks(r0) * r0.mt * (r0.key / 10000) >= role_blocks(r1) * r1.mt * (r1.key / 10000);
        const ChainRole &A = heavy0 ? r0 : r1, &B = heavy0 ? r1 : r0;
        if (chain_launch_pair(A, B, st)) {
            if (launches) *launches = 1;
            return check_launch("g4d_launch_group_end");
        }
    }
    for (int i = 0; i < n; ++i)
        if (int rc = chain_launch_one(g_group.role[i].mode, g_group.role[i].key, g_group.role[i].mt, g_group.role[i].s, st)) return rc;
    return G4D_OK;
}

// drops an open group without launching anything (error paths of the host layer)
extern "C" int g4d_launch_group_abort(void) {
    g_group.active = false;
    g_group.n = 0;
    return G4D_OK;
}

static int chain_f32_impl(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                          const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                          int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                          int nlayers, const float *const *W, const float *const *scale, const float *const *shift,
                          const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                          int tap_layer, float *tap_out, int tap_ld, const float *pre_scale, const float *pre_shift, float *in_tap,
                          int in_tap_ld, const float *tab, int tab_ld, const float *tab_wx, g4d_stream_t stream, const void *perm_grid = nullptr) {
    G4D_REQUIRE(mode == LOAD_DIRECT || mode == LOAD_GROUP || mode == LOAD_INTERP, "g4d_mlp_chain_f32: mode must be 0 (direct), 1 (group) or 2 (interp)");
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) - 256 && K0 > 0, "g4d_mlp_chain_f32: bad sizes");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "g4d_mlp_chain_f32: null pointer");
    G4D_REQUIRE(g4d_mlp_chain_supported(nlayers, Cout), "g4d_mlp_chain_f32: unsupported layer widths (see g4d_mlp_chain_supported)");
    G4D_REQUIRE(pool >= 0 && pool <= 2, "g4d_mlp_chain_f32: pool must be 0|1|2");
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "g4d_mlp_chain_f32: pooling needs S in {4,8,16,32,64}");
    static const int xcd_swz = getenv("G4D_CHAIN_XCD") ? atoi(getenv("G4D_CHAIN_XCD")) : 1;          // A/B switch
    ChainArgs s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    if (perm_grid) {
        G4D_REQUIRE(mode == LOAD_INTERP && pool == 0 && n > 0 && rows % n == 0, "g4d_mlp_chain_*_cells_f32: cell-ordered rows need the interpolating loader, no pooling, whole clouds");
        size_t off = 0, stride = 0;
        grid_sorted_layout(n, &off, &stride);
        s.in.perm_rec = reinterpret_cast<const unsigned char *>(perm_grid) + off; s.in.perm_stride = stride;
    }
    s.tap_layer = tap_out ? tap_layer : -1; s.tap_out = tap_out; s.tap_ld = tap_ld;
    G4D_REQUIRE(s.tap_layer < nlayers - 1, "g4d_mlp_chain_f32: tap must be a hidden layer");
    if (tab && mode == LOAD_INTERP) {
        G4D_REQUIRE(C2 == 0 && C1 > 0 && K0 == C1 && skip && nlayers >= 2 && Cout[0] % 16 == 0 && tab_ld >= Cout[0] && tab_ld % 4 == 0 &&
                    (reinterpret_cast<size_t>(tab) & 15) == 0 && !pre_scale, "g4d_mlp_chain_interp_init_f32: needs skip features, >= 2 layers, a first-layer width that "
                    "is a multiple of 16 and a 16-byte aligned table at least that wide");
        s.in.tab = tab; s.in.tab_ld = tab_ld;
    } else if (tab) {
        G4D_REQUIRE(mode == LOAD_GROUP && K0 % 16 == 0 && tab_ld >= K0 && tab_ld % 4 == 0 && (reinterpret_cast<size_t>(tab) & 15) == 0 && tab_wx && pre_scale &&
                    pre_shift && xyz && new_xyz && idx, "g4d_mlp_chain_group_table_f32: needs a 16-byte aligned table whose width is a multiple of 16, the xyz "
                    "weights, the affine and the grouping inputs");
        s.in.tab = tab; s.in.tab_ld = tab_ld; s.in.tab_wx = tab_wx; s.in.pre_scale = pre_scale; s.in.pre_shift = pre_shift;
    } else if (pre_scale) {
        G4D_REQUIRE(mode == LOAD_INTERP && C1 == 0 && C2 % 16 == 0 && K0 == C2 && pre_shift, "g4d_mlp_chain_table_f32: needs the interpolating loader, "
                    "no skip features and a table width that is a multiple of 16");
        G4D_REQUIRE(!in_tap || (in_tap_ld >= C2 && in_tap_ld % 4 == 0 && (reinterpret_cast<size_t>(in_tap) & 15) == 0), "g4d_mlp_chain_table_f32: bad input tap");
        s.in.pre_scale = pre_scale; s.in.pre_shift = pre_shift; s.in.in_tap = in_tap; s.in.in_tap_ld = in_tap_ld;
    }
    for (int l = 0; l < nlayers; ++l) {
        G4D_REQUIRE(W[l] && scale[l] && shift[l] && Kpad[l] % 16 == 0 && Cout[l] > 0, "g4d_mlp_chain_f32: bad layer %d", l);
        G4D_REQUIRE(Kpad[l] >= (l == 0 ? K0 : Cout[l - 1]), "g4d_mlp_chain_f32: Kpad of layer %d too small", l);
        s.layer[l].W = W[l]; s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].kst = Kpad[l] / 16; s.layer[l].relu = relu[l]; s.layer[l].cout = Cout[l];
    }
    s.xcd_swz = xcd_swz;
    const int key = chain_key(nlayers, Cout);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // rows per wave (16 * MT).  Measured on the cfg2 stacks (scripts/time_stacks.py, G4D_CHAIN_MT sweep): 32 rows per wave win
    // once the launch still has >= 2048 waves (each weight fragment load then feeds 8 MFMAs); 16 rows per wave otherwise (small
    // launches need the waves); 64 rows per wave never won.  The widest stacks (128-wide first layer) at 1024 waves -- SA3 scale 1 of
    // cfg2 -- are 6 % faster ALONE with 32 rows per wave (37.5 vs 40.1 us behind the table loader), but that instantiation holds 378
    // registers: one wave per SIMD and none on a CU that hosts a sampling workgroup; with 16 batches in flight the 16-row one (212
    // registers, two per SIMD, one beside the FPS waves) gives 30.2k instead of 29.8k frames/s, so it is the default
    // (G4D_CHAIN_MT2_MIN_WAVES_WIDE=1024 restores the other choice).
    const long long waves32 = (rows + 31) / 32;
    static const int mt_env = getenv("G4D_CHAIN_MT") ? atoi(getenv("G4D_CHAIN_MT")) : 0;  // tuning hook: 1 | 2
    const bool wide = key >= 8000000;
    static const long long mt2_min = getenv("G4D_CHAIN_MT2_MIN_WAVES") ? atoll(getenv("G4D_CHAIN_MT2_MIN_WAVES")) : 2048;      // tuning hooks
    static const long long mt2_min_wide = getenv("G4D_CHAIN_MT2_MIN_WAVES_WIDE") ? atoll(getenv("G4D_CHAIN_MT2_MIN_WAVES_WIDE")) : 2048;
    const int mt = mt_env ? (mt_env >= 2 ? 2 : 1) : ((waves32 >= mt2_min || (wide && waves32 >= mt2_min_wide)) ? 2 : 1);
    if (g_group.active) {   // inside g4d_launch_group_begin / _end: the stack becomes a role of the group's launch
        G4D_REQUIRE(g_group.n < kMaxRoles, "g4d_launch_group: more than %d recorded launches", kMaxRoles);
        ChainRole &r = g_group.role[g_group.n++];
        r.s = s; r.mode = mode; r.key = key; r.mt = mt;
        return G4D_OK;
    }
    return chain_launch_one(mode, key, mt, s, st);
}

extern "C" int g4d_mlp_chain_f32(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                 const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                 int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                                 int nlayers, const float *const *W, const float *const *scale, const float *const *shift,
                                 const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                                 int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    return chain_f32_impl(mode, rows, K0, X, ldx, N, P, S, C, use_xyz, xyz, new_xyz, feats, idx, n, m, C2, C1, known_feats, skip, dist2, nn_idx,
                          nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer, tap_out, tap_ld, nullptr, nullptr, nullptr,
                          0, nullptr, 0, nullptr, stream);
}

// Feature propagation without skip features, first layer pre-contracted (pointnet2_modules.py:127-156): the conv of the first
// SharedMLP layer is linear and three_interpolate is a weighted sum of three known rows, so
//   conv(sum_i w_i f_i) = sum_i w_i conv(f_i):
// `table` (B*m, C2) = known features already multiplied by the first layer's weight (m rows per cloud instead of n), and the layer
// itself shrinks to  h = relu(interp(table) * pre_scale + pre_shift)  inside the loader -- no MFMA work for it.  h is written to
// `in_tap` (the FP module's output when its MLP has one layer); `layers` are the REMAINING layers (e.g. the segmentation head).
extern "C" int g4d_mlp_chain_table_f32(long long rows, int n, int m, int C2, const float *table, const float *dist2, const int *nn_idx,
                                       const float *pre_scale, const float *pre_shift, float *in_tap, int in_tap_ld, int nlayers,
                                       const float *const *W, const float *const *scale, const float *const *shift, const int *Kpad,
                                       const int *Cout, const int *relu, float *out, int ldo, int col0, int tap_layer, float *tap_out,
                                       int tap_ld, g4d_stream_t stream) {
    G4D_REQUIRE(table && dist2 && nn_idx && pre_scale && pre_shift, "g4d_mlp_chain_table_f32: null pointer");
    if (W && scale && shift && Kpad && Cout && relu && out) {   // large launches of the last-level stack: fp_table.hip (bit-identical)
        const int rc = fp_table_try(rows, n, m, C2, table, dist2, nn_idx, nullptr, 0, pre_scale, pre_shift, in_tap, nlayers, W, scale, shift, Kpad, Cout, relu,
                                    out, ldo, col0, tap_layer, tap_out, tap_ld, reinterpret_cast<hipStream_t>(stream));
        if (rc != -1) return rc;
    }
    return chain_f32_impl(LOAD_INTERP, rows, C2, nullptr, 0, 0, 0, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, n, m, C2, 0, table, nullptr, dist2,
                          nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, 0, out, ldo, col0, tap_layer, tap_out, tap_ld, pre_scale, pre_shift,
                          in_tap, in_tap_ld, nullptr, 0, nullptr, stream);
}

// g4d_mlp_chain_table_f32 over CELL-ORDERED rows: `unknown_grid` is the ball-grid workspace of the unknown cloud (g4d_ball_grid_build_f32 /
// g4d_fps_gather_grid_f32), dist2 / nn_idx are in that order (g4d_three_nn_cells_sorted_f32), and row p of the launch is the cloud's p-th point
// in cell order: the 64 rows of a workgroup are spatial neighbours whose three nearest known points largely coincide, so the table rows
// they gather (3 x 512 B per row at the last FP level) hit in L1 instead of going to L2 each.  Outputs (out, tap_out, in_tap) land at the
// rows' ORIGINAL positions; every value is bit-identical to g4d_mlp_chain_table_f32 on un-sorted inputs.
extern "C" int g4d_mlp_chain_table_cells_f32(long long rows, int n, int m, int C2, const float *table, const float *dist2, const int *nn_idx,
                                             const void *unknown_grid, const float *pre_scale, const float *pre_shift, float *in_tap,
                                             int in_tap_ld, int nlayers, const float *const *W, const float *const *scale,
                                             const float *const *shift, const int *Kpad, const int *Cout, const int *relu, float *out, int ldo,
                                             int col0, int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    G4D_REQUIRE(table && dist2 && nn_idx && pre_scale && pre_shift && unknown_grid, "g4d_mlp_chain_table_cells_f32: null pointer");
    if (W && scale && shift && Kpad && Cout && relu && out && n > 0) {   // large launches of the last-level stack: fp_table.hip (bit-identical)
        size_t off = 0, stride = 0;
        grid_sorted_layout(n, &off, &stride);
        const int rc = fp_table_try(rows, n, m, C2, table, dist2, nn_idx, reinterpret_cast<const unsigned char *>(unknown_grid) + off, stride, pre_scale,
                                    pre_shift, in_tap, nlayers, W, scale, shift, Kpad, Cout, relu, out, ldo, col0, tap_layer, tap_out, tap_ld,
                                    reinterpret_cast<hipStream_t>(stream));
        if (rc != -1) return rc;
    }
    return chain_f32_impl(LOAD_INTERP, rows, C2, nullptr, 0, 0, 0, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, n, m, C2, 0, table, nullptr, dist2,
                          nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, 0, out, ldo, col0, tap_layer, tap_out, tap_ld, pre_scale, pre_shift,
                          in_tap, in_tap_ld, nullptr, 0, nullptr, stream, unknown_grid);
}

// Set abstraction with the feature part of its first layer pre-contracted (pointnet2_utils.py:232-265 + the first SharedMLP layer):
//   W [x_j - q ; f_j] = Wx (x_j - q) + Wf f_j,   and Wf f_j depends on the SOURCE point j only
// -- `table` row j (stride tab_ld, Kt columns used) = Wf f_j, computed once per level over the N source points instead of once per
// (centroid, sample) pair; tab_wx = Wx transposed, [3][Kt]; the layer itself is relu((table[j] + Wx (x_j - q)) * pre_scale + pre_shift)
// inside the loader.  W / scale / ... describe the REMAINING layers; pooling as in g4d_mlp_chain_f32.  tab_ld = 0: every source point shares
// ONE table row (an xyz-only stack: a row of zeros -- BASELINE config 5's [3, 64, 64, 128] on the persistent kernel of sa_table.hip).
extern "C" int g4d_mlp_chain_group_table_f32(long long rows, int N, int P, int S, const float *xyz, const float *new_xyz, const int *idx,
                                             const float *table, int tab_ld, int Kt, const float *tab_wx, const float *pre_scale,
                                             const float *pre_shift, int nlayers, const float *const *W, const float *const *scale,
                                             const float *const *shift, const int *Kpad, const int *Cout, const int *relu, int pool,
                                             float *out, int ldo, int col0, g4d_stream_t stream) {
    return g4d_mlp_chain_group_table_ws_f32(rows, N, P, S, xyz, new_xyz, idx, table, tab_ld, Kt, tab_wx, pre_scale, pre_shift, nlayers, W, scale, shift, Kpad,
                                            Cout, relu, pool, out, ldo, col0, nullptr, 0, stream);
}

// The same launch with caller-owned scratch (g4d_sa_table_ws_bytes; ws == NULL / too small: every block is computed): the widest stack's
// lock-step kernel then walks a work list without the blocks that hold ball-query padding only.  Same bits either way.
extern "C" int g4d_mlp_chain_group_table_ws_f32(long long rows, int N, int P, int S, const float *xyz, const float *new_xyz, const int *idx,
                                                const float *table, int tab_ld, int Kt, const float *tab_wx, const float *pre_scale,
                                                const float *pre_shift, int nlayers, const float *const *W, const float *const *scale,
                                                const float *const *shift, const int *Kpad, const int *Cout, const int *relu, int pool,
                                                float *out, int ldo, int col0, void *ws, long long ws_bytes, g4d_stream_t stream) {
    G4D_REQUIRE(table && Kt > 0, "g4d_mlp_chain_group_table_f32: null table");
    if (xyz && new_xyz && idx && tab_wx && pre_scale && pre_shift && W && scale && shift && Kpad && Cout && relu && out && (tab_ld >= Kt || tab_ld == 0) && tab_ld % 4 == 0 &&
        (reinterpret_cast<size_t>(table) & 15) == 0 && P > 0 && N > 0) {
        // large launches: the persistent, software-pipelined kernel (sa_table.hip; bit-identical results).  Inside a launch group it simply goes out
        // on its own -- merging launches pays only while they are small.
        const int rc = sa_table_try(rows, N, P, S, xyz, new_xyz, idx, table, tab_ld, Kt, tab_wx, pre_scale, pre_shift, nlayers, W, scale, shift, Kpad,
                                    Cout, relu, pool, out, ldo, col0, reinterpret_cast<hipStream_t>(stream), ws, ws_bytes);
        if (rc != -1) return rc;
    }
    return chain_f32_impl(LOAD_GROUP, rows, Kt, nullptr, 0, N, P, S, 0, 1, xyz, new_xyz, nullptr, idx, 0, 0, 0, 0, nullptr, nullptr, nullptr,
                          nullptr, nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, -1, nullptr, 0, pre_scale, pre_shift, nullptr, 0,
                          table, tab_ld, tab_wx, stream);
}

// Feature propagation WITH skip features, the known-feature part of the first layer pre-contracted (pointnet2_modules.py:127-156):
//   W [interp(f) ; s] = Wa interp(f) + Wb s = interp(Wa f) + Wb s
// -- `table` (B*m rows, stride tab_ld) = known features times Wa^T (m rows per cloud instead of n); the first layer's accumulators start
// from three_interpolate(table) and the matrix pipe adds the C1 skip columns; layer 0 of W / Kpad describes Wb (K = C1), scale / shift /
// relu of layer 0 are the layer's own.  Everything else as g4d_mlp_chain_f32 in its interpolating mode.
extern "C" int g4d_mlp_chain_interp_init_f32(long long rows, int n, int m, int C1, const float *skip, const float *table, int tab_ld,
                                             const float *dist2, const int *nn_idx, int nlayers, const float *const *W,
                                             const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout,
                                             const int *relu, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld,
                                             g4d_stream_t stream) {
    G4D_REQUIRE(table && skip && dist2 && nn_idx, "g4d_mlp_chain_interp_init_f32: null pointer");
    if (W && scale && shift && Kpad && Cout && relu && out) {   // large launches of the middle FP level's stack: fp_init.hip (bit-identical)
        const int rc = fp_init_try(rows, n, m, C1, skip, table, tab_ld, dist2, nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, out, ldo, col0, tap_layer,
                                   tap_out, tap_ld, reinterpret_cast<hipStream_t>(stream));
        if (rc != -1) return rc;
    }
    return chain_f32_impl(LOAD_INTERP, rows, C1, nullptr, 0, 0, 0, 1, 0, 0, nullptr, nullptr, nullptr, nullptr, n, m, 0, C1, nullptr, skip, dist2,
                          nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, 0, out, ldo, col0, tap_layer, tap_out, tap_ld, nullptr, nullptr,
                          nullptr, 0, table, tab_ld, nullptr, stream);
}
