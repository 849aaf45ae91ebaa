// Set-abstraction scale with bf16 shared-MLP operands (BASELINE config 3), PERSISTENT and software-pipelined (round 4): the large-launch
// form of g4d_mlp_chain_bf16 in its grouping mode for the encoder's three-layer stacks
//     QueryAndGroup(use_xyz) rows [x_j - q ; f_j (CIN)]  ->  C  ->  C  ->  2 C  ->  max over the S samples
// (pointnet2_modules.py:40-53 / pointnet2_utils.py:232-265).  bf16 mode has no pre-contracted table (the first layer is cheap on the bf16
// matrix cores and a table would not be a bf16-operand evaluation of the reference's layer): the rows are gathered whole.
//
// Why: at 240 clouds per call the six SA launches of the register-chain kernel are 1.18 ms of a 3.97 ms call for < 0.15 ms of MFMA: every
// 16- / 32-row tile pays, in sequence, kernel arguments -> neighbour index -> coordinates / feature row, each layer's weights from L2, its
// scale / shift at the seam and a pooling epilogue with four quarter-wave stores per channel tile.  Here
//   * workgroups are resident for the whole launch; all three weight matrices (bf16, chain order: 4 .. 152 KB) and every per-layer
//     constant sit in LDS -- the 128-128-256 stack over 195 inputs fills a CU's LDS with ONE 8-wave workgroup;
//   * a wave walks its tiles with the neighbour indices two tiles ahead, coordinates one tile ahead, and the feature row of tile t + 1
//     requested k-step by k-step as tile t consumes its own (a whole tile of cover: layers 2 and 3);
//   * a pooling group never spans waves (a wave takes whole neighbourhoods: S / 16 tiles with a running maximum): no barrier after the
//     start-up copy; the pooled tiles leave through the four-tile swap reduction of sa_xyz.hip, 64 channels per store, ReLU once per output.
// The arithmetic is mlp_chain_bf16.hip's: fp32 coordinate difference, RNE rounding of the operands to bf16 (the compiler's packed
// conversion), v_mfma_f32_16x16x32_bf16 with k ascending, fp32 affine + ReLU, neighbouring channel tiles packed into the next layer's
// B fragment, last layer with swapped operands -- bit-identical results (max_r relu(y_r) = relu(max_r y_r) exactly).
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4ub __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ unsigned cvt2(float lo, float hi) {   // RNE, lo -> bits [15:0] (NOT inline asm: see mlp_chain_bf16.hip)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ uint4 pack8h(const f32x4 &a, const f32x4 &b) {
    return make_uint4(cvt2(a[0], a[1]), cvt2(a[2], a[3]), cvt2(b[0], b[1]), cvt2(b[2], b[3]));
}
__device__ __forceinline__ f32x4 mfma32h(const uint4 &a, const uint4 &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float pool4_rows_max_h(float v0, float v1, float v2, float v3) {   // sa_xyz.hip: lane 16 c + fi = max over the 16 rows of tile c
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
    const float m01 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const float m23 = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(m01), __float_as_uint(m23), false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
}  // namespace

struct SaGrpHArgs {
    int rows, N, P;                       // rows = B * P * S grouped rows; N source points per cloud, P centroids per cloud
    const float *xyz, *new_xyz, *feats;   // feats (B * N, CIN) point-major or null (CIN == 0)
    const int *idx;
    const unsigned short *W1, *W2, *W3;   // bf16, chain order [tile][k-step of 32][64 lanes][8]
    const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3;
    float *out;
    int ldo, col0;
};

// T1: channel tiles of the first two layers (widths 16 T1, 16 T1, 32 T1); S samples per neighbourhood; CIN feature columns behind the three
// coordinate columns (0 or a multiple of 32: the k-step holding column K - 1 then holds exactly the last three feature columns); NW waves
// per workgroup.
template <int T1, int S, int CIN, int NW>
__global__ void __launch_bounds__(64 * NW) sa_group_bf16_kernel(const SaGrpHArgs a) {
    constexpr int T2 = T1, T3 = 2 * T1;
    constexpr int KS0 = CIN / 32 + 1, KS1 = (T1 + 1) / 2, KS2 = (T2 + 1) / 2;
    constexpr int NW1 = T1 * KS0 * 512, NW2 = T2 * KS1 * 512, NW3 = T3 * KS2 * 512;   // bf16 elements (a fragment = 64 lanes x 8)
    constexpr int G = S / 16;                                                        // tiles per neighbourhood
    static_assert(CIN % 32 == 0 && S % 16 == 0 && (T1 == 1 || T1 % 2 == 0), "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short *s_w1 = reinterpret_cast<unsigned short *>(smem_raw), *s_w2 = s_w1 + NW1, *s_w3 = s_w2 + NW2;
    float *s_sc1 = reinterpret_cast<float *>(s_w3 + NW3), *s_sh1 = s_sc1 + 16 * T1, *s_sc2 = s_sh1 + 16 * T1, *s_sh2 = s_sc2 + 16 * T2,
          *s_sc3 = s_sh2 + 16 * T2, *s_sh3 = s_sc3 + 16 * T3;
    const int tid = threadIdx.x;
    for (int i = tid; i < NW1 / 8; i += 64 * NW) reinterpret_cast<uint4 *>(s_w1)[i] = reinterpret_cast<const uint4 *>(a.W1)[i];
    for (int i = tid; i < NW2 / 8; i += 64 * NW) reinterpret_cast<uint4 *>(s_w2)[i] = reinterpret_cast<const uint4 *>(a.W2)[i];
    for (int i = tid; i < NW3 / 8; i += 64 * NW) reinterpret_cast<uint4 *>(s_w3)[i] = reinterpret_cast<const uint4 *>(a.W3)[i];
    for (int i = tid; i < 16 * T1; i += 64 * NW) { s_sc1[i] = a.sc1[i]; s_sh1[i] = a.sh1[i]; s_sc2[i] = a.sc2[i]; s_sh2[i] = a.sh2[i]; }
    for (int i = tid; i < 16 * T3; i += 64 * NW) { s_sc3[i] = a.sc3[i]; s_sh3[i] = a.sh3[i]; }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, g = lane >> 4;
    const int ntile = a.rows >> 4;                    // rows is a multiple of S (launcher)
    const int nunit = ntile / G;                      // a wave's unit of work: one neighbourhood = G consecutive tiles
    const int nwaves = gridDim.x * NW, wg = blockIdx.x * NW + wave;
    if (wg >= nunit) return;                          // (no barrier below)
    const int iters = ((nunit - wg + nwaves - 1) / nwaves) * G;
    auto tile_of = [&](int it) { return min((wg + (it / G) * nwaves) * G + it % G, ntile - 1); };   // past the wave's last tile: read again, never used

    auto load_idx = [&](int tile) { return a.idx[tile * 16 + fi]; };
    struct Ctx { unsigned f0, f; float dx, dy, dz; };   // feature-row element offsets of this lane (k-step 0 low half | everything else); x_j - q
    auto make = [&](int tile, int j) {
        Ctx c;
        const int q = __builtin_amdgcn_readfirstlane((tile * 16) / S);   // a tile belongs to one neighbourhood
        const int b = q / a.P;
        const unsigned pt = (unsigned)(b * a.N + j);
        const float *pp = a.xyz + (size_t)pt * 3, *cc = a.new_xyz + (size_t)q * 3;
        c.dx = pp[0] - cc[0]; c.dy = pp[1] - cc[1]; c.dz = pp[2] - cc[2];   // pointnet2_utils.py:247 (fp32, before the rounding to bf16)
        c.f = pt * (unsigned)CIN + (unsigned)(g * 4) - 3u;   // column k of the row is feature k - 3
        c.f0 = pt * (unsigned)CIN + (g ? (unsigned)(g * 4) - 3u : 0u);
        return c;
    };
    // this lane's eight columns of one 32-column k-step of a row: [32 ks + 4 g, +4) and [32 ks + 16 + 4 g, +4), as loaded (see fix())
    struct Item { f32x4 lo, hi; };
    auto load_item = [&](const Ctx &c, int ks) {
        Item x;
        if constexpr (CIN > 0) {
            if (ks == KS0 - 1) {   // columns CIN .. CIN + 2 are the last three features; the rest of the step is padding
                x.lo = *reinterpret_cast<const f32x4ub *>(a.feats + (c.f0 - (g ? (unsigned)(g * 4) - 3u : 0u)) + (CIN - 4));
                x.hi = x.lo;
            } else {
                x.lo = *reinterpret_cast<const f32x4ub *>(a.feats + (ks == 0 ? c.f0 : c.f + (unsigned)(ks * 32)));
                x.hi = *reinterpret_cast<const f32x4ub *>(a.feats + (c.f + (unsigned)(ks * 32 + 16)));   // (32-bit wrap: c.f is 'row start - 3')
            }
        } else {
            x.lo = x.hi = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return x;
    };
    auto fix = [&](const Ctx &c, int ks, Item &x) {   // the loaded vectors -> the row's columns (coordinates in front, zero padding behind)
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (CIN == 0) {
            x.lo = g == 0 ? (f32x4){c.dx, c.dy, c.dz, 0.f} : z;
            x.hi = z;
        } else {
            if (ks == 0 && g == 0) x.lo = (f32x4){c.dx, c.dy, c.dz, x.lo[0]};
            if (ks == KS0 - 1) {
                x.lo = g == 0 ? (f32x4){x.lo[1], x.lo[2], x.lo[3], 0.f} : z;
                x.hi = z;
            }
        }
    };
    auto wfrag = [&](const unsigned short *sw, int kst, int ct, int ks) -> uint4 { return *reinterpret_cast<const uint4 *>(sw + ((ct * kst + ks) * 64 + lane) * 8); };

    // DEAD TILES (round 6, as sa_table.hip): a tile whose 16 rows all carry the neighbourhood's first index -- ball_query's padding -- reproduces row
    // 0's output, and max pooling does not see it: it is not computed (the pipeline moves on).  Exact for any index list.
    int jn = load_idx(tile_of(0));
    int h0 = __builtin_amdgcn_readlane(jn, 0);        // first neighbour index of the neighbourhood being looked at
    bool dead_cur = false, dead_nxt = false;
    Ctx cur = make(tile_of(0), jn);
    jn = load_idx(tile_of(1));
    Item item[KS0];
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) item[ks] = load_item(cur, ks);
    float pm[T3];                                     // running maximum of the neighbourhood across its tiles (G > 1)
    for (int it = 0; it < iters; ++it) {
        const int tile = tile_of(it);
        const Ctx nxt = make(tile_of(it + 1), jn);    // from the index loaded one tile ago; its coordinate loads have this whole tile
        if constexpr (G > 1) {
            const bool head = (it + 1) % G == 0;
            if (head) h0 = __builtin_amdgcn_readlane(jn, 0);
            dead_nxt = !head && __builtin_amdgcn_ballot_w64(jn != h0) == 0ull;
        }
        jn = load_idx(tile_of(it + 2));
        const int j = it % G;
        float v[T3];
        if (dead_cur) {                               // (never the first tile of a neighbourhood: pm holds its maximum so far)
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) item[ks] = load_item(nxt, ks);
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) v[ct] = pm[ct];
        } else {
        // ---- layer 1 (3 + CIN -> 16 T1), transposed: lane (fi, g) ends with channels 16 ct + 4 g + r of row fi
        f32x4 a1[T1];
#pragma unroll
        for (int ct = 0; ct < T1; ++ct) a1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            __builtin_amdgcn_sched_barrier(0);   // one scheduling region per k-step (left alone the scheduler requests every fragment of the tile up front)
            Item x = item[ks];
            item[ks] = load_item(nxt, ks);       // the next tile's share of this k-step: in flight across layers 2 and 3
            fix(cur, ks, x);
            const uint4 b = pack8h(x.lo, x.hi);
#pragma unroll
            for (int ct = 0; ct < T1; ++ct) a1[ct] = mfma32h(wfrag(s_w1, KS0, ct, ks), b, a1[ct]);
        }
        __builtin_amdgcn_sched_barrier(0);
        uint4 b1[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            f32x4 t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ct = 2 * ks + h;
                if (ct < T1) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc1 + ct * 16 + g * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh1 + ct * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[h][r] = fmaxf(__builtin_fmaf(a1[ct < T1 ? ct : 0][r], sc[r], sh[r]), 0.f);
                } else {
                    t[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            b1[ks] = pack8h(t[0], t[1]);
        }
        // ---- layer 2 (16 T1 -> 16 T1), transposed
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a2[T2];
#pragma unroll
        for (int ct = 0; ct < T2; ++ct) a2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int ct = 0; ct < T2; ++ct) a2[ct] = mfma32h(wfrag(s_w2, KS1, ct, ks), b1[ks], a2[ct]);
        uint4 b2[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            f32x4 t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ct = 2 * ks + h;
                if (ct < T2) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc2 + ct * 16 + g * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh2 + ct * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[h][r] = fmaxf(__builtin_fmaf(a2[ct < T2 ? ct : 0][r], sc[r], sh[r]), 0.f);
                } else {
                    t[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            b2[ks] = pack8h(t[0], t[1]);
        }
        // ---- layer 3 (16 T1 -> 32 T1), operands swapped: lane (fi, g) holds rows 4 g + r of channel 16 ct + fi
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a3[T3];
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) a3[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) a3[ct] = mfma32h(b2[ks], wfrag(s_w3, KS2, ct, ks), a3[ct]);
        // ---- affine, max over the rows, ReLU once per output
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) {
            const float sc = s_sc3[ct * 16 + fi], sh = s_sh3[ct * 16 + fi];
            const float y0 = __builtin_fmaf(a3[ct][0], sc, sh), y1 = __builtin_fmaf(a3[ct][1], sc, sh), y2 = __builtin_fmaf(a3[ct][2], sc, sh),
                        y3 = __builtin_fmaf(a3[ct][3], sc, sh);
            v[ct] = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
            if constexpr (G > 1) { pm[ct] = j == 0 ? v[ct] : fmaxf(pm[ct], v[ct]); v[ct] = pm[ct]; }
        }
        }   // !dead_cur
        dead_cur = dead_nxt;
        if (j == G - 1) {
            const int q = tile / G;
            float *o = a.out + (size_t)q * a.ldo + a.col0;
            if constexpr (T3 >= 4) {
#pragma unroll
                for (int c4 = 0; c4 < T3 / 4; ++c4)
                    o[c4 * 64 + lane] = fmaxf(pool4_rows_max_h(v[c4 * 4], v[c4 * 4 + 1], v[c4 * 4 + 2], v[c4 * 4 + 3]), 0.f);
            } else {   // 32 output channels: two tiles, the upper half of the wave idles
                const float m = fmaxf(pool4_rows_max_h(v[0], v[1], v[0], v[1]), 0.f);
                if (lane < 32) o[lane] = m;
            }
        }
        cur = nxt;
    }
}

}  // namespace g4d

using namespace g4d;

template <int T1, int S, int CIN, int NW>
static int sa_group_bf16_launch(const SaGrpHArgs &a, hipStream_t st) {
    constexpr int T3 = 2 * T1, KS0 = CIN / 32 + 1, KS1 = (T1 + 1) / 2;
    constexpr int lds = 2 * 512 * (T1 * KS0 + T1 * KS1 + T3 * KS1) + 4 * (4 * 16 * T1 + 2 * 16 * T3);
    static unsigned long long attr = 0;
    if (lds > 64 * 1024) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sa_group_bf16_kernel<T1, S, CIN, NW>), lds, attr, "g4d_sa_group_bf16");
        if (rc) return rc;
    }
    static const int resident = [] {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sa_group_bf16_kernel<T1, S, CIN, NW>, 64 * NW, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return per_cu * 256;
        return per_cu * prop.multiProcessorCount;
    }();
    const long long nunit = a.rows / S;
    const long long want = (nunit + NW - 1) / NW;
    hipLaunchKernelGGL((sa_group_bf16_kernel<T1, S, CIN, NW>), dim3((unsigned)(want < resident ? want : resident)), dim3(64 * NW), lds, st, a);
    return check_launch("g4d_sa_group_bf16");
}

// Takes the launch if it is one of the instantiated stacks and large enough to pipeline; returns -1 when it is not (the caller then runs the
// register-chain kernel), else the launch status.
int g4d::sa_group_bf16_try(long long rows, int N, int P, int S, int C, int use_xyz, const float *xyz, const float *new_xyz, const float *feats,
                           const int *idx, int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                           const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0, float *tap_out, hipStream_t st) {
    const int on = (int)tuning("sa_group_bf16_persistent", 1);                 // A/B switch
    const long long min_rows = tuning("sa_group_bf16_min_rows", 262144);
    if (!on || rows < min_rows || rows >= (1ll << 31) - 64 || nlayers != 3 || pool != 1 || !use_xyz || tap_out || S <= 0 || rows % S != 0 || P <= 0) return -1;
    if (!relu[0] || !relu[1] || !relu[2] || Cout[1] != Cout[0] || Cout[2] != 2 * Cout[0] || (C > 0 && !feats) || !xyz || !new_xyz || !idx) return -1;
    const int T1 = Cout[0] / 16;
    if (Cout[0] % 16 != 0 || Kpad[0] != 32 * (C / 32 + 1) || C % 32 != 0 || Kpad[1] != 32 * ((T1 + 1) / 2) || Kpad[2] != Kpad[1]) return -1;
    if ((rows / S / P) * (long long)N * (C > 3 ? C : 3) >= (1ll << 32)) return -1;   // 32-bit element offsets
    G4D_REQUIRE(out && W[0] && W[1] && W[2] && scale[0] && scale[1] && scale[2] && shift[0] && shift[1] && shift[2], "g4d_mlp_chain_bf16: null pointer");
    G4D_REQUIRE(N > 0 && rows % ((long long)P * S) == 0 && ldo >= col0 + Cout[2] && col0 >= 0, "g4d_mlp_chain_bf16: rows must be clouds x P x S and the output window [%d, %d) must fit ldo = %d",
                col0, col0 + Cout[2], ldo);
    SaGrpHArgs a;
    a.rows = (int)rows; a.N = N; a.P = P; a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.idx = idx;
    a.W1 = W[0]; a.W2 = W[1]; a.W3 = W[2];
    a.sc1 = scale[0]; a.sh1 = shift[0]; a.sc2 = scale[1]; a.sh2 = shift[1]; a.sc3 = scale[2]; a.sh3 = shift[2];
    a.out = out; a.ldo = ldo; a.col0 = col0;
    if (T1 == 1 && S == 16 && C == 0) return sa_group_bf16_launch<1, 16, 0, 4>(a, st);
    if (T1 == 2 && S == 32 && C == 0) return sa_group_bf16_launch<2, 32, 0, 4>(a, st);
    if (T1 == 2 && S == 16 && C == 96) return sa_group_bf16_launch<2, 16, 96, 4>(a, st);
    if (T1 == 4 && S == 32 && C == 96) return sa_group_bf16_launch<4, 32, 96, 4>(a, st);
    if (T1 == 4 && S == 32 && C == 192) return sa_group_bf16_launch<4, 32, 192, 4>(a, st);
    if (T1 == 8 && S == 64 && C == 192) return sa_group_bf16_launch<8, 64, 192, 8>(a, st);
    return -1;
}
