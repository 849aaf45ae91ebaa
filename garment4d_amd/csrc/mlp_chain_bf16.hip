// bf16 variant of mlp_chain.hip (BASELINE config 3: shared-MLP operands in bf16, fp32 accumulation / affine / pooling / I/O).
//
// v_mfma_f32_16x16x32_bf16: lane (fi = l & 15, g = l >> 4) supplies 8 consecutive k of row / column fi for A and B and holds
// D[4 g + r][fi].  Evaluating a hidden layer transposed (A = weights, B = activations) leaves lane (fi, g) with
// out[row = fi][channels 16 ct + 4 g + r]; two neighbouring channel tiles (ct = 2 ks, 2 ks + 1) give the lane 8 values of the
// 32-channel block ks -- channels {4 g .. 4 g + 3} and {16 + 4 g .. 16 + 4 g + 3}.  That is a fixed permutation of k inside the
// block, identical for every lane group, so it is simply baked into the weight packing (garment4d_amd/fused.py: Wc16): the
// fp32 accumulators go through affine + ReLU, are rounded to bf16 in pairs (v_cvt_pk_bf16_f32, RNE) and ARE the next layer's
// B fragment.  Nothing touches LDS.  The last layer swaps the operands, as in the fp32 kernel, for a row-major result.
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

struct ChainLayerH {
    const unsigned short *W;  // bf16, chain order [CoutPad64 / 16][Kpad / 32][64 lanes][8]
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
template <int TOUT, int MT>
__device__ __forceinline__ void to_frags(const f32x4 (&acc)[TOUT][MT], uint4 (&hb)[(TOUT + 1) / 2][MT]) {
#pragma unroll
    for (int ks = 0; ks < (TOUT + 1) / 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 lo = acc[2 * ks][mt];
            const f32x4 hi = (2 * ks + 1 < TOUT) ? acc[(2 * ks + 1 < TOUT) ? 2 * ks + 1 : 0][mt] : f32x4{0.f, 0.f, 0.f, 0.f};
            hb[ks][mt] = pack8(lo, hi);
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
template <int MODE, int TOUT, int MT, bool LAST>
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
    auto load_b = [&](int ks, uint4 (&b)[MT]) {
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
                    b[mt] = pack8(lo, hi);
                } else {
                    const f32x4 lo = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_lo), hi = *reinterpret_cast<const f32x4u_b *>(pa[mt] + k_hi);
                    b[mt] = pack8(lo, hi);
                }
            }
        } else if (MODE == LOAD_INTERP && c0 >= b_lo && c0 + 32 <= b_hi) {  // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 lo = *reinterpret_cast<const f32x4u_b *>(pb[mt] + k_lo), hi = *reinterpret_cast<const f32x4u_b *>(pb[mt] + k_hi);
                b[mt] = pack8(lo, hi);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = pack8(load4<MODE>(a, ctx[mt], rowc[mt], k_lo), load4<MODE>(a, ctx[mt], rowc[mt], k_hi));
        }
    };
    uint4 wn[TOUT], bn[MT];
#pragma unroll
    for (int ct = 0; ct < TOUT; ++ct) wn[ct] = *reinterpret_cast<const uint4 *>(L.W + ((size_t)(ct * L.kst) * 64 + lane) * 8);
    load_b(0, bn);
    for (int ks = 0; ks < kst0; ++ks) {
        uint4 w[TOUT], b[MT];
#pragma unroll
        for (int ct = 0; ct < TOUT; ++ct) w[ct] = wn[ct];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
        if (ks + 1 < kst0) {  // wave-uniform
#pragma unroll
            for (int ct = 0; ct < TOUT; ++ct) wn[ct] = *reinterpret_cast<const uint4 *>(L.W + ((size_t)(ct * L.kst + ks + 1) * 64 + lane) * 8);
            load_b(ks + 1, bn);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < TOUT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = LAST ? mfma32(b[mt], w[ct], acc[ct][mt]) : mfma32(w[ct], b[mt], acc[ct][mt]);
    }
    if (LAST) affine_r_h<TOUT, MT>(L, fi, acc);
    else affine_t_h<TOUT, MT>(L, g, acc);
}

template <int KS, int TOUT, int MT, bool LAST>
__device__ __forceinline__ void chain_layer_h(const ChainLayerH &L, int lane, const uint4 (&hb)[KS][MT], f32x4 (&acc)[TOUT][MT]) {
    const int fi = lane & 15, g = lane >> 4;
    zero_acc_h<TOUT, MT>(acc);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < TOUT; ++ct) {
            const uint4 w = *reinterpret_cast<const uint4 *>(L.W + ((size_t)(ct * L.kst + ks) * 64 + lane) * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = LAST ? mfma32(hb[ks][mt], w, acc[ct][mt]) : mfma32(w, hb[ks][mt], acc[ct][mt]);
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

template <int MODE, int T1, int T2, int T3, int T4, int MT>
__global__ void __launch_bounds__(256) mlp_chain_bf16_kernel(const ChainArgsH s) {
    __shared__ float xch[4 * 256];
    const LinearArgs &a = s.in;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = (blockIdx.x * 4 + wave) * (16 * MT);
    f32x4 h1[T1][MT];
    if constexpr (T2 == 0) {
        first_layer_h<MODE, T1, MT, true>(a, s.layer[0], lane, row0, h1);
        finish<T1, MT>(a, s.layer[0].cout, lane, wave, row0, h1, xch);
    } else {
        first_layer_h<MODE, T1, MT, false>(a, s.layer[0], lane, row0, h1);
        if (s.tap_layer == 0) tap_store_h<T1, MT>(s, s.layer[0].cout, lane, row0, h1);
        uint4 b1[(T1 + 1) / 2][MT];
        to_frags<T1, MT>(h1, b1);
        f32x4 h2[T2][MT];
        if constexpr (T3 == 0) {
            chain_layer_h<(T1 + 1) / 2, T2, MT, true>(s.layer[1], lane, b1, h2);
            finish<T2, MT>(a, s.layer[1].cout, lane, wave, row0, h2, xch);
        } else {
            chain_layer_h<(T1 + 1) / 2, T2, MT, false>(s.layer[1], lane, b1, h2);
            if (s.tap_layer == 1) tap_store_h<T2, MT>(s, s.layer[1].cout, lane, row0, h2);
            uint4 b2[(T2 + 1) / 2][MT];
            to_frags<T2, MT>(h2, b2);
            f32x4 h3[T3][MT];
            if constexpr (T4 == 0) {
                chain_layer_h<(T2 + 1) / 2, T3, MT, true>(s.layer[2], lane, b2, h3);
                finish<T3, MT>(a, s.layer[2].cout, lane, wave, row0, h3, xch);
            } else {
                chain_layer_h<(T2 + 1) / 2, T3, MT, false>(s.layer[2], lane, b2, h3);
                if (s.tap_layer == 2) tap_store_h<T3, MT>(s, s.layer[2].cout, lane, row0, h3);
                uint4 b3[(T3 + 1) / 2][MT];
                to_frags<T3, MT>(h3, b3);
                f32x4 h4[T4][MT];
                chain_layer_h<(T3 + 1) / 2, T4, MT, true>(s.layer[3], lane, b3, h4);
                finish<T4, MT>(a, s.layer[3].cout, lane, wave, row0, h4, xch);
            }
        }
    }
}

template <int T1, int T2, int T3, int T4, int MT>
static void launch_chain_h(int mode, const ChainArgsH &s, hipStream_t st) {
    const long long rows_per_wg = 4ll * 16 * MT;
    dim3 grid((unsigned)((s.in.rows + rows_per_wg - 1) / rows_per_wg)), block(256);
    if (mode == LOAD_GROUP) hipLaunchKernelGGL((mlp_chain_bf16_kernel<LOAD_GROUP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else if (mode == LOAD_INTERP) hipLaunchKernelGGL((mlp_chain_bf16_kernel<LOAD_INTERP, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
    else hipLaunchKernelGGL((mlp_chain_bf16_kernel<LOAD_DIRECT, T1, T2, T3, T4, MT>), grid, block, 0, st, s);
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_mlp_chain_supported(int nlayers, const int *Cout);

extern "C" int g4d_mlp_chain_bf16(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                  const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                  int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx,
                                  int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                                  const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0,
                                  int tap_layer, float *tap_out, int tap_ld, g4d_stream_t stream) {
    G4D_REQUIRE(mode == LOAD_DIRECT || mode == LOAD_GROUP || mode == LOAD_INTERP, "g4d_mlp_chain_bf16: mode must be 0, 1 or 2");
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) - 256 && K0 > 0, "g4d_mlp_chain_bf16: bad sizes");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "g4d_mlp_chain_bf16: null pointer");
    G4D_REQUIRE(g4d_mlp_chain_supported(nlayers, Cout), "g4d_mlp_chain_bf16: unsupported layer widths (see g4d_mlp_chain_supported)");
    G4D_REQUIRE(pool >= 0 && pool <= 2, "g4d_mlp_chain_bf16: pool must be 0|1|2");
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "g4d_mlp_chain_bf16: pooling needs S in {4,8,16,32,64}");
    ChainArgsH s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    s.tap_layer = tap_out ? tap_layer : -1; s.tap_out = tap_out; s.tap_ld = tap_ld;
    G4D_REQUIRE(s.tap_layer < nlayers - 1, "g4d_mlp_chain_bf16: tap must be a hidden layer");
    int key = 0;
    for (int l = 0; l < 4; ++l) key = key * 100 + (l < nlayers ? (Cout[l] + 15) / 16 : 0);
    for (int l = 0; l < nlayers; ++l) {
        G4D_REQUIRE(W[l] && scale[l] && shift[l] && Kpad[l] % 32 == 0 && Cout[l] > 0, "g4d_mlp_chain_bf16: bad layer %d", l);
        G4D_REQUIRE(Kpad[l] >= (l == 0 ? K0 : Cout[l - 1]), "g4d_mlp_chain_bf16: Kpad of layer %d too small", l);
        s.layer[l].W = W[l]; s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].kst = Kpad[l] / 32; s.layer[l].relu = relu[l]; s.layer[l].cout = Cout[l];
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long waves32 = (rows + 31) / 32;
    static const int mt_env = getenv("G4D_CHAIN_MT") ? atoi(getenv("G4D_CHAIN_MT")) : 0;
    const int mt = mt_env ? (mt_env >= 2 ? 2 : 1) : (waves32 >= 2048 ? 2 : 1);
#define G4D_CHAIN(T1, T2, T3, T4)                                     \
    if (mt == 2) launch_chain_h<T1, T2, T3, T4, 2>(mode, s, st);      \
    else launch_chain_h<T1, T2, T3, T4, 1>(mode, s, st);              \
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
    return check_launch("g4d_mlp_chain_bf16");
}
