// bf16 variant of mlp_chain.hip (BASELINE config 3: shared-MLP operands in bf16, fp32 accumulation / affine / pooling / I/O).
//
// v_mfma_f32_16x16x32_bf16: lane (fi = l & 15, g = l >> 4) supplies 8 consecutive k of row / column fi for A and B and holds
// D[4 g + r][fi].  Evaluating a hidden layer transposed (A = weights, B = activations) leaves lane (fi, g) with
// out[row = fi][channels 16 ct + 4 g + r]; two neighbouring channel tiles (ct = 2 ks, 2 ks + 1) give the lane 8 values of the
// 32-channel block ks -- channels {4 g .. 4 g + 3} and {16 + 4 g .. 16 + 4 g + 3}.  That is a fixed permutation of k inside the
// block, identical for every lane group, so it is simply baked into the weight packing (garment4d_amd/fused.py: Wc16): the
// fp32 accumulators go through affine + ReLU, are rounded to bf16 in pairs (v_cvt_pk_bf16_f32, RNE) and ARE the next layer's
// B fragment.  Nothing touches LDS.  The last layer swaps the operands, as in the fp32 kernel, for a row-major result.
//
// NSPL = 3 ("bf16x3", fp32-accurate): gfx950's fp32 MFMA peaks at 157 TFLOP/s, its bf16 MFMA at 2.5 PFLOP/s, so an fp32 product is
// cheaper as a sum of bf16 products.  Every fp32 operand is split EXACTLY into three bf16 pieces by truncation,
//     x = hi + mid + lo,   hi = x & 0xffff0000,   mid = (x - hi) & 0xffff0000,   lo = x - hi - mid   (8 + 8 + 8 significand bits),
// weights once on the host, activations in registers (11 VALU ops per pair), and a product is evaluated as the six largest of the
// nine piece products, smallest first, in the fp32 accumulator of the MFMA:
//     w x ~= w_lo x_hi + w_hi x_lo + w_mid x_mid + w_mid x_hi + w_hi x_mid + w_hi x_hi
// Each piece product is exact in fp32 (8 x 8 bits); the three dropped terms are below 2^-23 of |w x|, i.e. the error of ONE fp32
// rounding -- the contraction is as accurate as the fp32 MFMA route (different summation order, same error bound), six bf16 MFMAs
// (96 cycles per 16 x 16 x 32 block) instead of eight fp32 ones (256 cycles).
#include <cstdlib>

#include "mlp_common.h"
#include "chain_finish.h"

namespace g4d {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u_b __attribute__((ext_vector_type(4), aligned(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// RNE, lo -> bits [15:0].  The compiler's own conversion (it emits v_cvt_pk_bf16_f32), NOT inline asm: the packed value feeds an
// MFMA a couple of instructions later, and the hazard recognizer cannot see a VALU write inside an asm block -- with the asm
// version the first MFMA after the conversion read its B operand too early in ~25 % of the waves (garbage in channel tile 0
// of the second layer of the 128-64-32-16 stack at 16 rows per wave; any extra delay, e.g. s_waitcnt vmcnt(0), hid it).
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ uint4 pack8(const f32x4 &a, const f32x4 &b) {
    return make_uint4(cvt_pk_bf16(a[0], a[1]), cvt_pk_bf16(a[2], a[3]), cvt_pk_bf16(b[0], b[1]), cvt_pk_bf16(b[2], b[3]));
}
__device__ __forceinline__ f32x4 mfma32(const uint4 &a, const uint4 &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NSPL>
struct Op {  // operand fragment: [0] alone = RNE-rounded bf16; three = the exact hi / mid / lo pieces
    uint4 p[NSPL];
};

__device__ __forceinline__ void split_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    const float la = ra - __uint_as_float(va & 0xffff0000u), lb = rb - __uint_as_float(vb & 0xffff0000u);
    h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);  // upper halves: a -> bits [15:0], b -> bits [31:16]
    m = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

template <int NSPL>
__device__ __forceinline__ Op<NSPL> pack_op(const f32x4 &a, const f32x4 &b) {
    Op<NSPL> o;
    if constexpr (NSPL == 1) {
        o.p[0] = pack8(a, b);
    } else {
        unsigned h[4], m[4], l[4];
        split_pair(a[0], a[1], h[0], m[0], l[0]);
        split_pair(a[2], a[3], h[1], m[1], l[1]);
        split_pair(b[0], b[1], h[2], m[2], l[2]);
        split_pair(b[2], b[3], h[3], m[3], l[3]);
        o.p[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o.p[1] = make_uint4(m[0], m[1], m[2], m[3]);
        o.p[2] = make_uint4(l[0], l[1], l[2], l[3]);
    }
    return o;
}

// c += w . x  (SWAP: x . w -- the last layer's row-major orientation)
template <int NSPL, bool SWAP>
__device__ __forceinline__ f32x4 mma(const Op<NSPL> &w, const Op<NSPL> &x, f32x4 c) {
    auto one = [&](const uint4 &ww, const uint4 &xx, f32x4 cc) { return SWAP ? mfma32(xx, ww, cc) : mfma32(ww, xx, cc); };
    if constexpr (NSPL == 1) {
        return one(w.p[0], x.p[0], c);
    } else {
        c = one(w.p[2], x.p[0], c);
        c = one(w.p[0], x.p[2], c);
        c = one(w.p[1], x.p[1], c);
        c = one(w.p[1], x.p[0], c);
        c = one(w.p[0], x.p[1], c);
        return one(w.p[0], x.p[0], c);
    }
}

// one piece product of the split: (weight piece, activation piece), smallest first
__device__ constexpr int kPw[6] = {2, 0, 1, 1, 0, 0}, kPx[6] = {0, 2, 1, 0, 1, 0};

struct ChainLayerH {
    const unsigned short *W;  // bf16, chain order [CoutPad64 / 16][Kpad / 32][64 lanes][8]
    const unsigned short *Wm, *Wl;  // bf16x3 only: the mid / lo pieces in the same order (W = hi)
    const float *scale, *shift;
    int kst;                  // Kpad / 32
    int relu, cout;
};

struct ChainArgsH {
    LinearArgs in;
    ChainLayerH layer[4];
    int tap_layer;
    float *tap_out;
    int tap_ld;
};

template <int NSPL>
__device__ __forceinline__ Op<NSPL> load_w(const ChainLayerH &L, size_t frag, int lane) {
    Op<NSPL> o;
    o.p[0] = *reinterpret_cast<const uint4 *>(L.W + (frag * 64 + lane) * 8);
    if constexpr (NSPL == 3) {
        o.p[1] = *reinterpret_cast<const uint4 *>(L.Wm + (frag * 64 + lane) * 8);
        o.p[2] = *reinterpret_cast<const uint4 *>(L.Wl + (frag * 64 + lane) * 8);
    }
    return o;
}

template <int TOUT, int MT>
__device__ __forceinline__ void affine_t_h(const ChainLayerH &L, int g, f32x4 (&acc)[TOUT][MT]) {
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) {
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(L.scale + ct * 16 + g * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(L.shift + ct * 16 + g * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = __builtin_fmaf(acc[ct][mt][r], sc[r], sh[r]);
                if (L.relu) y = fmaxf(y, 0.f);
                acc[ct][mt][r] = y;
            }
    }
}

template <int TOUT, int MT>
__device__ __forceinline__ void affine_r_h(const ChainLayerH &L, int fi, f32x4 (&acc)[TOUT][MT]) {
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

// fp32 transposed tiles -> bf16 B fragments of the next layer (k-step ks = channel tiles 2 ks and 2 ks + 1)
template <int NSPL, int TOUT, int MT>
__device__ __forceinline__ void to_frags(const f32x4 (&acc)[TOUT][MT], Op<NSPL> (&hb)[(TOUT + 1) / 2][MT]) {
#pragma unroll
    for (int ks = 0; ks < (TOUT + 1) / 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 lo = acc[2 * ks][mt];
            const f32x4 hi = (2 * ks + 1 < TOUT) ? acc[(2 * ks + 1 < TOUT) ? 2 * ks + 1 : 0][mt] : f32x4{0.f, 0.f, 0.f, 0.f};
            hb[ks][mt] = pack_op<NSPL>(lo, hi);
        }
}

template <int TOUT, int MT>
__device__ __forceinline__ void zero_acc_h(f32x4 (&acc)[TOUT][MT]) {
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int MODE>
__device__ __forceinline__ f32x4 load4(const LinearArgs &a, const RowCtx<MODE> &c, int row, int k0) {
    if (MODE == LOAD_GROUP && c.valid && a.use_xyz && k0 >= 3 && k0 + 3 < a.K) return *reinterpret_cast<const f32x4u_b *>(a.feats + c.pt_base * a.C + (k0 - 3));
    if (MODE == LOAD_GROUP && c.valid && !a.use_xyz && k0 + 3 < a.K) return *reinterpret_cast<const f32x4u_b *>(a.feats + c.pt_base * a.C + k0);
    if (MODE == LOAD_DIRECT && c.valid && k0 + 3 < a.K) return *reinterpret_cast<const f32x4u_b *>(a.X + (size_t)row * a.ldx + k0);
    if (MODE == LOAD_INTERP && c.valid && k0 + 3 < a.C2) {
        const f32x4 f0 = *reinterpret_cast<const f32x4u_b *>(a.known_feats + c.k0 + k0);
        const f32x4 f1 = *reinterpret_cast<const f32x4u_b *>(a.known_feats + c.k1 + k0);
        const f32x4 f2 = *reinterpret_cast<const f32x4u_b *>(a.known_feats + c.k2 + k0);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c.w0 * f0[e] + c.w1 * f1[e] + c.w2 * f2[e];
        return v;
    }
    if (MODE == LOAD_INTERP && c.valid && k0 >= a.C2 && k0 + 3 < a.K) return *reinterpret_cast<const f32x4u_b *>(a.skip + c.sk + (k0 - a.C2));
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = load_elem<MODE>(a, c, row, k0 + e);
    return v;
}

// First layer (see mlp_chain.hip first_layer for the measurements behind this shape): rows past the end are clamped, not masked;
// a 32-column k-step takes the two-16-byte-loads path when it lies inside one source segment -- decided once per step for the
// whole wave -- and the operands of step ks + 1 are requested before the MFMAs of step ks.
template <int NSPL, int MODE, int TOUT, int MT, bool LAST>
__device__ __forceinline__ void first_layer_h(const LinearArgs &a, const ChainLayerH &L, int lane, int row0, f32x4 (&acc)[TOUT][MT]) {
    const int fi = lane & 15, g = lane >> 4;
    RowCtx<MODE> ctx[MT];
    int rowc[MT];
    const float *pa[MT], *pa1[MT], *pa2[MT], *pb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rowc[mt] = min(row0 + mt * 16 + fi, a.rows - 1);
        ctx[mt] = make_ctx<MODE>(a, rowc[mt]);
        pa1[mt] = pa2[mt] = pb[mt] = nullptr;
        if constexpr (MODE == LOAD_GROUP) pa[mt] = (a.feats ? a.feats : a.xyz) + ctx[mt].pt_base * a.C - (a.use_xyz ? 3 : 0);
        else if constexpr (MODE == LOAD_DIRECT) pa[mt] = a.X + (size_t)rowc[mt] * a.ldx;
        else { pa[mt] = a.known_feats + ctx[mt].k0; pa1[mt] = a.known_feats + ctx[mt].k1; pa2[mt] = a.known_feats + ctx[mt].k2;
               pb[mt] = a.skip + ctx[mt].sk - a.C2; }
    }
    int a_lo, a_hi, b_lo = 0, b_hi = 0;
    if constexpr (MODE == LOAD_GROUP) { a_lo = a.use_xyz ? 3 : 0; a_hi = a.K; }
    else if constexpr (MODE == LOAD_DIRECT) { a_lo = 0; a_hi = a.K; }
    else { a_lo = 0; a_hi = a.C2; b_lo = a.C2; b_hi = a.K; }
    zero_acc_h<TOUT, MT>(acc);
    const int kst0 = (a.K + 31) >> 5;
    auto load_b = [&](int ks, Op<NSPL> (&b)[MT]) {
        const int c0 = ks * 32, k_lo = c0 + g * 4, k_hi = c0 + 16 + g * 4;  // this lane's 8 columns: [k_lo, +4) and [k_hi, +4)
        if (c0 >= a_lo && c0 + 32 <= a_hi) {  // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (MODE == LOAD_INTERP) {
                    f32x4 lo, hi;
                    const f32x4 l0 = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_lo), l1 = *reinterpret_cast<const f32x4u_b *>(pa1[mt] + k_lo),
                                l2 = *reinterpret_cast<const f32x4u_b *>(pa2[mt] + k_lo);
                    const f32x4 h0 = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_hi), h1 = *reinterpret_cast<const f32x4u_b *>(pa1[mt] + k_hi),
                                h2 = *reinterpret_cast<const f32x4u_b *>(pa2[mt] + k_hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = ctx[mt].w0 * l0[e] + ctx[mt].w1 * l1[e] + ctx[mt].w2 * l2[e];
                        hi[e] = ctx[mt].w0 * h0[e] + ctx[mt].w1 * h1[e] + ctx[mt].w2 * h2[e];
                    }
                    b[mt] = pack_op<NSPL>(lo, hi);
                } else {
                    const f32x4 lo = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_lo), hi = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_hi);
                    b[mt] = pack_op<NSPL>(lo, hi);
                }
            }
        } else if (MODE == LOAD_INTERP && c0 >= b_lo && c0 + 32 <= b_hi) {  // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 lo = *reinterpret_cast<const f32x4u_b *>(pb[mt] + k_lo), hi = *reinterpret_cast<const f32x4u_b *>(pb[mt] + k_hi);
                b[mt] = pack_op<NSPL>(lo, hi);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = pack_op<NSPL>(load4<MODE>(a, ctx[mt], rowc[mt], k_lo), load4<MODE>(a, ctx[mt], rowc[mt], k_hi));
        }
    };
    Op<NSPL> bn[MT];
    load_b(0, bn);
    if constexpr (NSPL * TOUT > 24) {  // a double buffer of the weight fragments would not fit the register file: fetch per channel tile
        for (int ks = 0; ks < kst0; ++ks) {
            Op<NSPL> b[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
            if (ks + 1 < kst0) load_b(ks + 1, bn);  // wave-uniform
            // two channel tiles at a time (independent accumulators between consecutive MFMAs); the next pair's weight pieces are
            // requested before this pair's MFMAs (an L2 round trip is ~4x the 24 MFMAs of a pair at 16 rows per wave)
            auto fetch2 = [&](int c0, int kk, Op<NSPL> (&w)[2]) {
                w[0] = load_w<NSPL>(L, (size_t)(c0 * L.kst + kk), lane);
                w[1] = load_w<NSPL>(L, (size_t)(min(c0 + 1, TOUT - 1) * L.kst + kk), lane);
            };
            Op<NSPL> wq[2];
            fetch2(0, ks, wq);
#pragma unroll
            for (int c0 = 0; c0 < TOUT; c0 += 2) {
                Op<NSPL> w[2];
                w[0] = wq[0]; w[1] = wq[1];
                if (c0 + 2 < TOUT) fetch2(c0 + 2, ks, wq);
#pragma unroll
                for (int q = 0; q < (NSPL == 3 ? 6 : 1); ++q)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (c0 + c >= TOUT) continue;
                            f32x4 &a = acc[c0 + c][mt];
                            const uint4 &ww = w[c].p[NSPL == 3 ? kPw[q] : 0], &xx = b[mt].p[NSPL == 3 ? kPx[q] : 0];
                            a = LAST ? mfma32(xx, ww, a) : mfma32(ww, xx, a);
                        }
            }
        }
        if (LAST) affine_r_h<TOUT, MT>(L, fi, acc);
        else affine_t_h<TOUT, MT>(L, g, acc);
        return;
    }
    Op<NSPL> wn[TOUT];
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) wn[ct] = load_w<NSPL>(L, (size_t)(ct * L.kst), lane);
    for (int ks = 0; ks < kst0; ++ks) {
        Op<NSPL> w[TOUT], b[MT];
#pragma unroll
        for (int ct = 0; ct < TOUT; ++ct) w[ct] = wn[ct];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
        if (ks + 1 < kst0) {  // wave-uniform
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct) wn[ct] = load_w<NSPL>(L, (size_t)(ct * L.kst + ks + 1), lane);
            load_b(ks + 1, bn);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < (NSPL == 3 ? 6 : 1); ++q)  // piece-pair loop outermost: TOUT * MT independent accumulators between repeats
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint4 &ww = w[ct].p[NSPL == 3 ? kPw[q] : 0], &xx = b[mt].p[NSPL == 3 ? kPx[q] : 0];
                    acc[ct][mt] = LAST ? mfma32(xx, ww, acc[ct][mt]) : mfma32(ww, xx, acc[ct][mt]);
                }
    }
    if (LAST) affine_r_h<TOUT, MT>(L, fi, acc);
    else affine_t_h<TOUT, MT>(L, g, acc);
}

template <int NSPL, int KS, int TOUT, int MT, bool LAST>
__device__ __forceinline__ void chain_layer_h(const ChainLayerH &L, int lane, const Op<NSPL> (&hb)[KS][MT], f32x4 (&acc)[TOUT][MT]) {
    const int fi = lane & 15, g = lane >> 4;
    zero_acc_h<TOUT, MT>(acc);
    if constexpr (NSPL == 1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct) {
                const Op<NSPL> w = load_w<NSPL>(L, (size_t)(ct * L.kst + ks), lane);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = mma<NSPL, LAST>(w, hb[ks][mt], acc[ct][mt]);
            }
    } else {
        // Split operands: six MFMAs per (channel tile, row tile) and k-step.  Issued tile by tile they would be six DEPENDENT
        // MFMAs on one accumulator (the 16-cycle issue rate needs >= 2-4 independent accumulators in flight), so the channel
        // tiles go in groups of CG, the piece-pair loop is the OUTER one inside a group, and the next group's weight pieces are
        // requested before this group's MFMAs.
        constexpr int CG = TOUT >= 2 ? 2 : 1;
        constexpr int NG = (TOUT + CG - 1) / CG;
        auto fetch = [&](int step, Op<NSPL> (&w)[CG]) {  // step = ks * NG + group
            const int ks = step / NG, c0 = (step - ks * NG) * CG;
#pragma unroll
            for (int c = 0; c < CG; ++c) w[c] = load_w<NSPL>(L, (size_t)(min(c0 + c, TOUT - 1) * L.kst + ks), lane);
        };
        Op<NSPL> w0[CG], w1[CG];   // two groups in flight: an L2 round trip outlasts one group's 12-24 MFMAs
        fetch(0, w0);
        if (KS * NG > 1) fetch(1, w1);
#pragma unroll
        for (int step = 0; step < KS * NG; ++step) {
            const int ks = step / NG, c0 = (step - ks * NG) * CG;
            Op<NSPL> w[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) { w[c] = w0[c]; w0[c] = w1[c]; }
            if (step + 2 < KS * NG) fetch(step + 2, w1);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int c = 0; c < CG; ++c)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (c0 + c >= TOUT) continue;
                        f32x4 &a = acc[c0 + c][mt];
                        a = LAST ? mfma32(hb[ks][mt].p[kPx[q]], w[c].p[kPw[q]], a) : mfma32(w[c].p[kPw[q]], hb[ks][mt].p[kPx[q]], a);
                    }
        }
    }
    if (LAST) affine_r_h<TOUT, MT>(L, fi, acc);
    else affine_t_h<TOUT, MT>(L, g, acc);
}

// output stage: chain_finish.h (shared with the fp32 kernel; same row-major D layout)
template <int TOUT, int MT>
__device__ __forceinline__ void tap_store_h(const ChainArgsH &s, int cout, int lane, int row0, const f32x4 (&h)[TOUT][MT]) {
    const int fi = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = row0 + mt * 16 + fi;
            if (row >= s.in.rows) continue;
            float *dst = s.tap_out + (size_t)row * s.tap_ld + ct * 16 + g * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ct * 16 + g * 4 + r < cout) dst[r] = h[ct][mt][r];
        }
}

template <int NSPL, int MODE, int T1, int T2, int T3, int T4, int MT>
__global__ void __launch_bounds__(256) mlp_chain_bf16_kernel(const ChainArgsH s) {
    __shared__ float xch[4 * 256];
    const LinearArgs &a = s.in;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = (blockIdx.x * 4 + wave) * (16 * MT);
    f32x4 h1[T1][MT];
    if constexpr (T2 == 0) {
        first_layer_h<NSPL, MODE, T1, MT, true>(a, s.layer[0], lane, row0, h1);
        finish<T1, MT>(a, s.layer[0].cout, lane, wave, row0, h1, xch);
    } else {
        first_layer_h<NSPL, MODE, T1, MT, false>(a, s.layer[0], lane, row0, h1);
        if (s.tap_layer == 0) tap_store_h<T1, MT>(s, s.layer[0].cout, lane, row0, h1);
        Op<NSPL> b1[(T1 + 1) / 2][MT];
        to_frags<NSPL, T1, MT>(h1, b1);
        f32x4 h2[T2][MT];
        if constexpr (T3 == 0) {
            chain_layer_h<NSPL, (T1 + 1) / 2, T2, MT, true>(s.layer[1], lane, b1, h2);
            finish<T2, MT>(a, s.layer[1].cout, lane, wave, row0, h2, xch);
        } else {
            chain_layer_h<NSPL, (T1 + 1) / 2, T2, MT, false>(s.layer[1], lane, b1, h2);
            if (s.tap_layer == 1) tap_store_h<T2, MT>(s, s.layer[1].cout, lane, row0, h2);
            Op<NSPL> b2[(T2 + 1) / 2][MT];
            to_frags<NSPL, T2, MT>(h2, b2);
            f32x4 h3[T3][MT];
            if constexpr (T4 == 0) {
                chain_layer_h<NSPL, (T2 + 1) / 2, T3, MT, true>(s.layer[2], lane, b2, h3);
                finish<T3, MT>(a, s.layer[2].cout, lane, wave, row0, h3, xch);
            } else {
                chain_layer_h<NSPL, (T2 + 1) / 2, T3, MT, false>(s.layer[2], lane, b2, h3);
                if (s.tap_layer == 2) tap_store_h<T3, MT>(s, s.layer[2].cout, lane, row0, h3);
                Op<NSPL> b3[(T3 + 1) / 2][MT];
                to_frags<NSPL, T3, MT>(h3, b3);
                f32x4 h4[T4][MT];
                chain_layer_h<NSPL, (T3 + 1) / 2, T4, MT, true>(s.layer[3], lane, b3, h4);
                finish<T4, MT>(a, s.layer[3].cout, lane, wave, row0, h4, xch);
            }
        }
    }
}

template <int NSPL, int T1, int T2, int T3, int T4, int MT>
static void launch_chain_h(int mode, const ChainArgsH &s, hipStream_t st) {
    const long long rows_per_wg = 4ll * 16 * MT;
    dim3 grid((unsigned)((s.in.rows + rows_per_wg - 1) / rows_per_wg)), block(256);
    if (mode == LOAD_GROUP) hipLaunchKernelGGL((mlp_chain_bf16_kernel<NSPL, LOAD_GROUP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else if (mode == LOAD_INTERP) hipLaunchKernelGGL((mlp_chain_bf16_kernel<NSPL, LOAD_INTERP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else hipLaunchKernelGGL((mlp_chain_bf16_kernel<NSPL, LOAD_DIRECT, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_mlp_chain_supported(int nlayers, const int *Cout);

static int chain_bf16_impl(const char *name, int nspl, int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                           const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                           int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                           int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                           const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                           int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    G4D_REQUIRE(mode == LOAD_DIRECT || mode == LOAD_GROUP || mode == LOAD_INTERP, "%s: mode must be 0, 1 or 2", name);
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) - 256 && K0 > 0, "%s: bad sizes", name);
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "%s: null pointer", name);
    G4D_REQUIRE(g4d_mlp_chain_supported(nlayers, Cout), "%s: unsupported layer widths (see g4d_mlp_chain_supported)", name);
    G4D_REQUIRE(pool >= 0 && pool <= 2, "%s: pool must be 0|1|2", name);
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "%s: pooling needs S in {4,8,16,32,64}", name);
    ChainArgsH s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    s.tap_layer = tap_out ? tap_layer : -1; s.tap_out = tap_out; s.tap_ld = tap_ld;
    G4D_REQUIRE(s.tap_layer < nlayers - 1, "%s: tap must be a hidden layer", name);
    int key = 0;
    for (int l = 0; l < 4; ++l) key = key * 100 + (l < nlayers ? (Cout[l] + 15) / 16 : 0);
    for (int l = 0; l < nlayers; ++l) {
        const unsigned short *const *Wl = W + (size_t)l * nspl;   // nspl consecutive pointers per layer: hi[, mid, lo]
        G4D_REQUIRE(Wl[0] && (nspl == 1 || (Wl[1] && Wl[2])) && scale[l] && shift[l] && Kpad[l] % 32 == 0 && Cout[l] > 0, "%s: bad layer %d", name, l);
        G4D_REQUIRE(Kpad[l] >= (l == 0 ? K0 : Cout[l - 1]), "%s: Kpad of layer %d too small", name, l);
        s.layer[l].W = Wl[0]; s.layer[l].Wm = nspl == 3 ? Wl[1] : nullptr; s.layer[l].Wl = nspl == 3 ? Wl[2] : nullptr;
        s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].kst = Kpad[l] / 32; s.layer[l].relu = relu[l]; s.layer[l].cout = Cout[l];
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long waves32 = (rows + 31) / 32;
    static const int mt_env = getenv("G4D_CHAIN_MT") ? atoi(getenv("G4D_CHAIN_MT")) : 0;
    const bool wide = nspl == 3 && key >= 8000000;   // split mode: three weight pieces per fragment -> reuse them over 32 rows earlier
    const int mt = mt_env ? (mt_env >= 2 ? 2 : 1) : ((waves32 >= 2048 || (wide && waves32 >= 1024)) ? 2 : 1);
#define G4D_CHAIN(T1, T2, T3, T4)                                                  \
    if (nspl == 3) {                                                               \
        if (mt == 2) launch_chain_h<3, T1, T2, T3, T4, 2>(mode, s, st);            \
        else launch_chain_h<3, T1, T2, T3, T4, 1>(mode, s, st);                    \
    } else if (mt == 2) launch_chain_h<1, T1, T2, T3, T4, 2>(mode, s, st);         \
    else launch_chain_h<1, T1, T2, T3, T4, 1>(mode, s, st);                        \
    break;
    switch (key) {
        case 1010200: G4D_CHAIN(1, 1, 2, 0)
        case 2020400: G4D_CHAIN(2, 2, 4, 0)
        case 4040800: G4D_CHAIN(4, 4, 8, 0)
        case 8081600: G4D_CHAIN(8, 8, 16, 0)
        case 2020000: G4D_CHAIN(2, 2, 0, 0)
        case 4040000: G4D_CHAIN(4, 4, 0, 0)
        case 8080000: G4D_CHAIN(8, 8, 0, 0)
        case 8040000: G4D_CHAIN(8, 4, 0, 0)
        case 16080000: G4D_CHAIN(16, 8, 0, 0)
        case 1000000: G4D_CHAIN(1, 0, 0, 0)
        case 2000000: G4D_CHAIN(2, 0, 0, 0)
        case 4000000: G4D_CHAIN(4, 0, 0, 0)
        case 8000000: G4D_CHAIN(8, 0, 0, 0)
        default: G4D_CHAIN(8, 4, 2, 1)
    }
#undef G4D_CHAIN
    return check_launch(name);
}

extern "C" int g4d_mlp_chain_bf16(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                  const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                  int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                                  int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                                  const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                                  int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    if (mode == LOAD_INTERP && known_feats && dist2 && nn_idx && W && scale && shift && Kpad && Cout && relu && out) {   // large launches of config 3's last level: fp_head_bf16.hip (bit-identical)
        const int rc = fp_head_bf16_try(rows, n, m, C2, C1, known_feats, dist2, nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer,
                                        tap_out, tap_ld, reinterpret_cast<hipStream_t>(stream));
        if (rc != -1) return rc;
    }
    if (mode == LOAD_GROUP && W && scale && shift && Kpad && Cout && relu && out) {   // large launches of config 3's SA levels: sa_group_bf16.hip (bit-identical)
        const int rc = sa_group_bf16_try(rows, N, P, S, C, use_xyz, xyz, new_xyz, feats, idx, nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0,
                                         tap_out, reinterpret_cast<hipStream_t>(stream));
        if (rc != -1) return rc;
    }
    return chain_bf16_impl("g4d_mlp_chain_bf16", 1, mode, rows, K0, X, ldx, N, P, S, C, use_xyz, xyz, new_xyz, feats, idx, n, m, C2, C1, known_feats, skip,
                           dist2, nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer, tap_out, tap_ld, stream);
}

extern "C" int g4d_mlp_chain_cells_bf16(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                        const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                        int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                                        int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                                        const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                                        int tap_layer, float *tap_out, int tap_ld, const void *unknown_grid, g4d_stream_t stream) {
    if (mode == LOAD_INTERP && unknown_grid && n > 0 && known_feats && dist2 && nn_idx && W && scale && shift && Kpad && Cout && relu && out) {
        size_t off = 0, stride = 0;
        grid_sorted_layout(n, &off, &stride);
        const int rc = fp_head_bf16_try(rows, n, m, C2, C1, known_feats, dist2, nn_idx, nlayers, W, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer,
                                        tap_out, tap_ld, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned char *>(unknown_grid) + off, stride);
        if (rc != -1) return rc;
    }
    return g4d_mlp_chain_bf16(mode, rows, K0, X, ldx, N, P, S, C, use_xyz, xyz, new_xyz, feats, idx, n, m, C2, C1, known_feats, skip, dist2, nn_idx, nlayers, W, scale,
                              shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer, tap_out, tap_ld, stream);
}

extern "C" int g4d_mlp_chain_bf16x3(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                    const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                    int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                                    int nlayers, const unsigned short *const *W3, const float *const *scale, const float *const *shift,
                                    const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                                    int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    return chain_bf16_impl("g4d_mlp_chain_bf16x3", 3, mode, rows, K0, X, ldx, N, P, S, C, use_xyz, xyz, new_xyz, feats, idx, n, m, C2, C1, known_feats, skip,
                           dist2, nn_idx, nlayers, W3, scale, shift, Kpad, Cout, relu, pool, out, ldo, col0, tap_layer, tap_out, tap_ld, stream);
}
