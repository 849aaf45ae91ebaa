// Last feature-propagation level + segmentation head with bf16 shared-MLP operands (BASELINE config 3), PERSISTENT and software-pipelined
// (round 4): the large-launch form of g4d_mlp_chain_bf16 in its interpolating mode for the stack
//     three_interpolate(known features, 128)  ->  128  ->  64 (tapped: the FP module's output)  ->  32  ->  <= 16 (logits)
// (pointnet2_modules.py:127-156 + pointnet2encoder.py:98-107; bf16 mode has no pre-contracted table: the first layer is cheap on the bf16
// matrix cores).
//
// Why: this launch is 872 us of a 4.25 ms bf16 call at 240 clouds (the largest single launch) for ~25 us of MFMA: the register-chain
// kernel pays, per 32-row tile in sequence, kernel arguments -> (index, distance) -> three feature rows per k-step one step ahead, each
// layer's weights from L2 and its scale / shift at the seam.  Here workgroups are resident, all four weight matrices (53 KB of bf16, chain
// order) and every per-layer constant sit in LDS, (index, distance) of tile t + 2 and the interpolation weights / offsets of tile t + 1
// are prepared while tile t computes, and the feature rows stream through a ring that runs across the tile boundary.
// The arithmetic is mlp_chain_bf16.hip's: fp32 blend in load order, RNE rounding of the operands to bf16 (the compiler's packed
// conversion), v_mfma_f32_16x16x32_bf16 with k ascending, fp32 affine + ReLU, neighbouring channel tiles packed into the next layer's
// B fragment -- bit-identical results.
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
constexpr int D0 = 128, D1 = 128, D2 = 64, D3 = 32, D4 = 16;
constexpr int KS0 = D0 / 32, KS1 = D1 / 32, KS2 = D2 / 32, KS3 = D3 / 32;   // 32-wide k-steps of the four layers: 4, 4, 2, 1
constexpr int T1 = D1 / 16, T2 = D2 / 16, T3 = D3 / 16;                     // channel tiles: 8, 4, 2 (+ 1 for the last layer)
constexpr int NW1 = T1 * KS0 * 512, NW2 = T2 * KS1 * 512, NW3 = T3 * KS2 * 512, NW4 = KS3 * 512;   // bf16 elements (a fragment = 64 lanes x 8)
}

struct FpHeadHArgs {
    int rows, n, m;
    const float *feats;                   // (B * m, 128) known features, fp32
    const float *dist2; const int *nn_idx;
    const unsigned short *W1, *W2, *W3, *W4;   // bf16, chain order, Kpad == K
    const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3, *sc4, *sh4;
    int relu1, relu2, relu3, relu4, cout4;
    float *out; int ldo;                  // (rows, cout4)
    float *tap; int tap_ld;               // (rows, 64): output of the second layer
    const unsigned char *perm_rec;        // cell-ordered launch: 16-byte grid records of the unknown cloud, original index in the 4th dword (NULL: rows in place)
    size_t perm_stride;                   // bytes per cloud
};

// PERM: cell-ordered launch.  A template flag (round 5): as a run-time branch around the record load the number of loads in flight was unknown
// at the join and every tile started with s_waitcnt vmcnt(0) behind the loads it had just issued (see fp_table.hip).
template <bool PERM>
__global__ void __launch_bounds__(256) fp_head_bf16_kernel(const FpHeadHArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short s_w1[NW1], s_w2[NW2], s_w3[NW3], s_w4[NW4];
    __shared__ __attribute__((aligned(16))) float s_sc1[D1], s_sh1[D1], s_sc2[D2], s_sh2[D2], s_sc3[D3], s_sh3[D3], s_sc4[D4], s_sh4[D4];
    const int tid = threadIdx.x;
    for (int i = tid; i < NW1 / 8; i += 256) reinterpret_cast<uint4 *>(s_w1)[i] = reinterpret_cast<const uint4 *>(a.W1)[i];
    for (int i = tid; i < NW2 / 8; i += 256) reinterpret_cast<uint4 *>(s_w2)[i] = reinterpret_cast<const uint4 *>(a.W2)[i];
    for (int i = tid; i < NW3 / 8; i += 256) reinterpret_cast<uint4 *>(s_w3)[i] = reinterpret_cast<const uint4 *>(a.W3)[i];
    for (int i = tid; i < NW4 / 8; i += 256) reinterpret_cast<uint4 *>(s_w4)[i] = reinterpret_cast<const uint4 *>(a.W4)[i];
    if (tid < D1) { s_sc1[tid] = a.sc1[tid]; s_sh1[tid] = a.sh1[tid]; }
    if (tid < D2) { s_sc2[tid] = a.sc2[tid]; s_sh2[tid] = a.sh2[tid]; }
    if (tid < D3) { s_sc3[tid] = a.sc3[tid]; s_sh3[tid] = a.sh3[tid]; }
    if (tid < D4) { s_sc4[tid] = a.sc4[tid]; s_sh4[tid] = a.sh4[tid]; }
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: tile and cloud numbers on the scalar unit)
    const int fi = lane & 15, g = lane >> 4;
    const int ntile = (a.rows + 15) >> 4;
    const int nwaves = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    if (wg >= ntile) return;
    const int iters = (ntile - wg + nwaves - 1) / nwaves;
    auto tile_of = [&](int it) { return min(wg + it * nwaves, ntile - 1); };   // past the wave's last tile: read again, never used
    float lo1 = a.relu1 ? 0.f : -__builtin_inff(), lo2 = a.relu2 ? 0.f : -__builtin_inff(), lo3 = a.relu3 ? 0.f : -__builtin_inff(),
          lo4 = a.relu4 ? 0.f : -__builtin_inff();
    asm volatile("" : "+v"(lo1), "+v"(lo2), "+v"(lo3), "+v"(lo4));   // ReLU or not as the floor of one v_max (opaque: else folded back into max + select)

    // Cell-ordered launch (perm_rec): launch row r = (cloud b, position p of the cloud's cell-sorted records) works on ORIGINAL row
    // b n + index(p) -- its search result, its outputs.  Consecutive launch rows are then spatial neighbours and share their nearest known
    // points (the three 512-byte feature rows per row are what this launch is bound by: 3 GB through L2 per 240-cloud call in input order).
    // Every row is computed from its own operands only, so the order changes no bit of any output.
    auto load_orow = [&](int tile) -> int {
        const int row = min(tile * 16 + fi, a.rows - 1);
        if constexpr (!PERM) return row;
        const int b = row / a.n;
        return b * a.n + reinterpret_cast<const int *>(a.perm_rec + (size_t)b * a.perm_stride)[4 * (row - b * a.n) + 3];
    };
    struct Raw { int i0, i1, i2; float d0, d1, d2; int orow; };
    auto load_raw = [&](int orow) {
        Raw r;
        const int *ix = a.nn_idx + (size_t)orow * 3;
        const float *dd = a.dist2 + (size_t)orow * 3;
        r.i0 = ix[0]; r.i1 = ix[1]; r.i2 = ix[2]; r.d0 = dd[0]; r.d1 = dd[1]; r.d2 = dd[2]; r.orow = orow;
        return r;
    };
    struct Ctx { float w0, w1, w2; unsigned k0, k1, k2; int orow; };
    auto make = [&](int tile, const Raw &r) {   // pointnet2_utils.py:98 sqrt; pointnet2_modules.py:140-142 inverse-distance weights (mlp_common.h make_ctx)
        Ctx c;
        const float r0 = 1.0f / (__fsqrt_rn(r.d0) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(r.d1) + 1e-8f), r2 = 1.0f / (__fsqrt_rn(r.d2) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        c.w0 = r0 / norm; c.w1 = r1 / norm; c.w2 = r2 / norm;
        const int row = min(tile * 16 + fi, a.rows - 1);
        const int b0 = __builtin_amdgcn_readfirstlane((tile * 16) / a.n);   // a tile touches at most two clouds (n >= 16)
        const unsigned base = (unsigned)(b0 + (row >= (b0 + 1) * a.n ? 1 : 0)) * (unsigned)a.m;
        c.k0 = (base + (unsigned)r.i0) * D0 + g * 4; c.k1 = (base + (unsigned)r.i1) * D0 + g * 4; c.k2 = (base + (unsigned)r.i2) * D0 + g * 4;
        c.orow = r.orow;
        return c;
    };
    // one 32-column k-step of a row: this lane's columns [32 ks + 4 g, +4) and [32 ks + 16 + 4 g, +4) of the three neighbours' feature rows
    struct Item { f32x4 l0, l1, l2, h0, h1, h2; };
    auto load_item = [&](const Ctx &c, int ks) {
        Item x;
        x.l0 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k0 + ks * 32); x.h0 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k0 + ks * 32 + 16);
        x.l1 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k1 + ks * 32); x.h1 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k1 + ks * 32 + 16);
        x.l2 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k2 + ks * 32); x.h2 = *reinterpret_cast<const f32x4ub *>(a.feats + c.k2 + ks * 32 + 16);
        return x;
    };
    auto wfrag = [&](const unsigned short *sw, int kst, int ct, int ks) -> uint4 { return *reinterpret_cast<const uint4 *>(sw + ((ct * kst + ks) * 64 + lane) * 8); };

    Raw rawn = load_raw(load_orow(tile_of(0)));
    Ctx cur = make(tile_of(0), rawn);
    rawn = load_raw(load_orow(tile_of(1)));
    int oron = load_orow(tile_of(2));     // three levels ahead: original row -> (index, distance) -> feature rows
    Item ring[2] = {load_item(cur, 0), load_item(cur, 1)};
    for (int it = 0; it < iters; ++it) {
        const int tile = tile_of(it);
        const Ctx nxt = make(tile_of(it + 1), rawn);   // from the loads issued one tile ago
        rawn = load_raw(oron);
        oron = load_orow(tile_of(it + 3));
        // ---- layer 1 (128 -> 128), transposed: lane (fi, g) ends with channels 16 ct + 4 g + r of row fi
        f32x4 a1[T1];
#pragma unroll
        for (int ct = 0; ct < T1; ++ct) a1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            __builtin_amdgcn_sched_barrier(0);   // one scheduling region per k-step (left alone the scheduler requests every fragment of the tile up front)
            const Item x = ring[ks & 1];
            ring[ks & 1] = ks + 2 < KS0 ? load_item(cur, ks + 2) : load_item(nxt, ks + 2 - KS0);   // the ring runs across the tile boundary
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // three_interpolate, the chain kernel's operation order
                lo[e] = cur.w0 * x.l0[e] + cur.w1 * x.l1[e] + cur.w2 * x.l2[e];
                hi[e] = cur.w0 * x.h0[e] + cur.w1 * x.h1[e] + cur.w2 * x.h2[e];
            }
            const uint4 b = pack8h(lo, hi);
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
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc1 + ct * 16 + g * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh1 + ct * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[h][r] = fmaxf(__builtin_fmaf(a1[ct][r], sc[r], sh[r]), lo1);
            }
            b1[ks] = pack8h(t[0], t[1]);
        }
        // ---- layer 2 (128 -> 64), transposed; its affine output is the FP module's output (tapped)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a2[T2];
#pragma unroll
        for (int ct = 0; ct < T2; ++ct) a2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int ct = 0; ct < T2; ++ct) a2[ct] = mfma32h(wfrag(s_w2, KS1, ct, ks), b1[ks], a2[ct]);
        const bool row_ok = tile * 16 + fi < a.rows;
        const size_t orow = (size_t)cur.orow;
        uint4 b2[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            f32x4 t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ct = 2 * ks + h;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc2 + ct * 16 + g * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh2 + ct * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[h][r] = fmaxf(__builtin_fmaf(a2[ct][r], sc[r], sh[r]), lo2);
                if (a.tap && row_ok) *reinterpret_cast<f32x4 *>(a.tap + orow * a.tap_ld + ct * 16 + g * 4) = t[h];
            }
            b2[ks] = pack8h(t[0], t[1]);
        }
        // ---- layer 3 (64 -> 32), transposed
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a3[T3];
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) a3[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) a3[ct] = mfma32h(wfrag(s_w3, KS2, ct, ks), b2[ks], a3[ct]);
        uint4 b3;
        {
            f32x4 t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc3 + h * 16 + g * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh3 + h * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[h][r] = fmaxf(__builtin_fmaf(a3[h][r], sc[r], sh[r]), lo3);
            }
            b3 = pack8h(t[0], t[1]);
        }
        // ---- last layer (32 -> <= 16), operands swapped: lane (fi, g) holds rows 4 g + r of channel fi
        __builtin_amdgcn_sched_barrier(0);
        f32x4 o = mfma32h(b3, wfrag(s_w4, KS3, 0, 0), (f32x4){0.f, 0.f, 0.f, 0.f});
        {
            const float sc = s_sc4[fi], sh = s_sh4[fi];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = fmaxf(__builtin_fmaf(o[r], sc, sh), lo4);
                const int row = tile * 16 + g * 4 + r;
                const int dst = __builtin_amdgcn_ds_bpermute((g * 4 + r) << 2, cur.orow);   // lane fi' = 4 g + r holds that row's original index
                if (fi < a.cout4 && row < a.rows) a.out[(size_t)dst * a.ldo + fi] = y;
            }
        }
        cur = nxt;
    }
}

}  // namespace g4d

using namespace g4d;

// Takes the launch if it is the instantiated stack and large enough; returns -1 when it is not (the caller then runs the register-chain kernel).
int g4d::fp_head_bf16_try(long long rows, int n, int m, int C2, int C1, const float *known_feats, const float *dist2, const int *nn_idx, int nlayers,
                          const unsigned short *const *W, const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout,
                          const int *relu, int pool, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st,
                          const void *perm_rec, size_t perm_stride) {
    const int on = (int)tuning("fp_head_bf16_persistent", 1);
    const long long min_rows = tuning("fp_head_bf16_min_rows", 262144);
    if (!on || rows < min_rows || rows >= (1ll << 31) - 64 || C2 != D0 || C1 != 0 || nlayers != 4 || pool != 0 || col0 != 0) return -1;
    if (Cout[0] != D1 || Cout[1] != D2 || Cout[2] != D3 || Cout[3] > D4 || Cout[3] < 1 || Kpad[0] != D0 || Kpad[1] != D1 || Kpad[2] != D2 || Kpad[3] != D3) return -1;
    if (tap_out && (tap_layer != 1 || tap_ld % 4 != 0 || (reinterpret_cast<size_t>(tap_out) & 15) != 0)) return -1;
    if (n < 16 || m <= 0 || rows % n != 0 || (rows / n) * (long long)m * D0 >= (1ll << 32) || (reinterpret_cast<size_t>(known_feats) & 15) != 0) return -1;
    G4D_REQUIRE(known_feats && dist2 && nn_idx && out && W[0] && W[1] && W[2] && W[3] && scale[0] && scale[1] && scale[2] && scale[3] && shift[0] && shift[1] && shift[2] && shift[3],
                "g4d_mlp_chain_bf16: null pointer");
    G4D_REQUIRE(ldo >= Cout[3] && (!tap_out || tap_ld >= Cout[1]), "g4d_mlp_chain_bf16: output row stride %d < %d channels or tap stride %d < %d", ldo, Cout[3], tap_ld, Cout[1]);
    FpHeadHArgs a;
    a.rows = (int)rows; a.n = n; a.m = m; a.feats = known_feats; a.dist2 = dist2; a.nn_idx = nn_idx;
    a.W1 = W[0]; a.W2 = W[1]; a.W3 = W[2]; a.W4 = W[3];
    a.sc1 = scale[0]; a.sh1 = shift[0]; a.sc2 = scale[1]; a.sh2 = shift[1]; a.sc3 = scale[2]; a.sh3 = shift[2]; a.sc4 = scale[3]; a.sh4 = shift[3];
    a.relu1 = relu[0]; a.relu2 = relu[1]; a.relu3 = relu[2]; a.relu4 = relu[3]; a.cout4 = Cout[3];
    a.out = out; a.ldo = ldo; a.tap = tap_out; a.tap_ld = tap_ld;
    a.perm_rec = reinterpret_cast<const unsigned char *>(perm_rec); a.perm_stride = perm_stride;
    typedef void (*Kern)(const FpHeadHArgs);
    const Kern kern = perm_rec ? fp_head_bf16_kernel<true> : fp_head_bf16_kernel<false>;
    static int resident[2] = {0, 0};
    int &res = resident[perm_rec ? 1 : 0];
    if (res == 0) {   // (benign race: every thread computes the same value)
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        res = (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) ? per_cu * 256 : per_cu * prop.multiProcessorCount;
    }
    const long long want = ((rows + 15) / 16 + 3) / 4;
    hipLaunchKernelGGL(kern, dim3((unsigned)(want < res ? want : res)), dim3(256), 0, st, a);
    return check_launch("g4d_fp_head_bf16");
}
