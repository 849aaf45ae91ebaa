// Feature propagation without skip features behind a pre-contracted first layer, with the layers that follow it (the last FP level of
// Pointnet2MSGSEG + the segmentation head: pointnet2_modules.py:127-156, pointnet2encoder.py:98-107), PERSISTENT and software-pipelined
// (round 4): the large-launch form of g4d_mlp_chain_table_f32 / g4d_mlp_chain_table_cells_f32 for the stack
//     h1 = relu(affine(three_interpolate(table)))  (128)  ->  64 (tapped: the FP module's output)  ->  32  ->  <= 16 (logits).
//
// Why: at 240 clouds per launch the register-chain kernel runs this stack at 0.33 of the matrix pipe (27 us per 8 clouds for 9 us of
// MFMA): 5 VALU instructions per MFMA, waves parked in s_waitcnt half of their life (SQ counters, profiles/r04_pmc_sq_B240.csv).  Its
// straight-line row block pays, in sequence, kernel arguments -> (index, distance) -> three table rows per k-step one step ahead, the
// per-layer scale / shift fetched at each seam, and -- for cell-ordered rows -- an integer division and a dependent load PER OUTPUT
// ELEMENT to find the row's original position.  Here
//   * workgroups are resident; all three weight matrices (42 KB) and every per-layer constant sit in LDS, loaded once;
//   * a wave walks 16-row tiles: (index, distance, original row) of tile t + 2 and the interpolation weights / row offsets of tile t + 1
//     are prepared while tile t computes, and the table rows stream through a ring two k-steps deep that runs across the tile boundary;
//   * original rows are looked up once per row (one load), the cloud of a tile once per tile (scalar).
// The arithmetic (inverse-distance weights, blend order, affine, k order of every contraction) is that of mlp_chain.hip's INTERP table
// loader and chained layers: results are bit-identical.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct FpTabArgs {
    int rows, n, m;                       // rows = B * n launch rows (cloud b = row / n), m known points per cloud
    const float *tab;                     // (B * m, 128): known features times the first layer's weight
    const float *dist2;                   // (rows, 3) squared distances, launch-row order
    const int *nn_idx;                    // (rows, 3)
    const unsigned char *perm_rec;        // cell-ordered launches: 16-byte grid records, original index in the 4th dword (NULL: rows in place)
    size_t perm_stride;                   // bytes per cloud
    const float *ps, *pf;                 // [128] affine of the first layer
    const float *W2, *sc2, *sh2, *W3, *sc3, *sh3, *W4, *sc4, *sh4;   // fragment order, Kpad == K
    int cout4, relu4;                     // valid channels of the last layer (<= 16)
    float *out;  int ldo;                 // (rows, cout4) logits
    float *tap;  int tap_ld;              // (rows, 64) output of the 128 -> 64 layer
};

constexpr int kC1 = 128, kC2 = 64, kC3 = 32, kC4 = 16;
constexpr int kT1 = kC1 / 16, kT2 = kC2 / 16, kT3 = kC3 / 16;
// kD: table k-steps in flight ahead of the one being contracted (a divisor of kT1: the ring runs across the tile boundary).
// PERM: cell-ordered launch (the rows' original positions come from the grid records).  A template flag since round 5: as a run-time
// branch around one load it made the number of loads in flight unknown at the join, and the compiler answered with s_waitcnt vmcnt(0) at the
// top of EVERY tile -- behind the loads it had just issued, draining the whole ring.
// HEAD (round 6): false = the FP level ALONE, 128-wide table -> 64 (what PointnetFPModule.forward runs when it is called on its own -- the reference's
// encoder loop over the drop-in modules; the layers behind the tap are not instantiated: out = the tap).
template <bool PERM, int kD, bool HEAD = true>
__global__ void __launch_bounds__(256, kD <= 4 ? 3 : 2) fp_table_head_kernel(const FpTabArgs a) {
    static_assert(kT1 % kD == 0, "ring depth divides the k-steps of a tile");
    constexpr int NW2 = kC1 * kC2, NW3 = kC2 * kC3, NW4 = kC3 * kC4;
    __shared__ __attribute__((aligned(16))) float s_w2[NW2], s_w3[NW3], s_w4[NW4];
    __shared__ __attribute__((aligned(16))) float s_ps[kC1], s_pf[kC1], s_sc2[kC2], s_sh2[kC2], s_sc3[kC3], s_sh3[kC3], s_sc4[kC4], s_sh4[kC4];
    const int tid = threadIdx.x;
    for (int i = tid; i < NW2 / 4; i += 256) reinterpret_cast<f32x4 *>(s_w2)[i] = reinterpret_cast<const f32x4 *>(a.W2)[i];
    if constexpr (HEAD) {
        for (int i = tid; i < NW3 / 4; i += 256) reinterpret_cast<f32x4 *>(s_w3)[i] = reinterpret_cast<const f32x4 *>(a.W3)[i];
        for (int i = tid; i < NW4 / 4; i += 256) reinterpret_cast<f32x4 *>(s_w4)[i] = reinterpret_cast<const f32x4 *>(a.W4)[i];
    }
    if (tid < kC1) { s_ps[tid] = a.ps[tid]; s_pf[tid] = a.pf[tid]; }
    if (tid < kC2) { s_sc2[tid] = a.sc2[tid]; s_sh2[tid] = a.sh2[tid]; }
    if constexpr (HEAD) {
        if (tid < kC3) { s_sc3[tid] = a.sc3[tid]; s_sh3[tid] = a.sh3[tid]; }
        if (tid < kC4) { s_sc4[tid] = a.sc4[tid]; s_sh4[tid] = a.sh4[tid]; }
    }
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: tile numbers, cloud numbers and their divisions run on the scalar unit)
    const int fi = lane & 15, fq = lane >> 4;
    const int ntile = (a.rows + 15) >> 4;
    const int nwaves = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    if (wg >= ntile) return;
    const int iters = (ntile - wg + nwaves - 1) / nwaves;
    auto tile_of = [&](int it) { return min(wg + it * nwaves, ntile - 1); };   // past the wave's last tile: read again, never used

    // level 1: what a row brings along -- three neighbour indices, three squared distances, its original position
    struct Raw { int i0, i1, i2; float d0, d1, d2; int orow; };
    auto load_raw = [&](int tile) {
        Raw r;
        const int row = min(tile * 16 + fi, a.rows - 1);
        const int *ix = a.nn_idx + (size_t)row * 3;
        const float *dd = a.dist2 + (size_t)row * 3;
        r.i0 = ix[0]; r.i1 = ix[1]; r.i2 = ix[2];
        r.d0 = dd[0]; r.d1 = dd[1]; r.d2 = dd[2];
        const int b0 = __builtin_amdgcn_readfirstlane((tile * 16) / a.n);   // a tile touches at most two clouds (n >= 16)
        const int b = b0 + (row >= (b0 + 1) * a.n ? 1 : 0);
        r.orow = row;
        if constexpr (PERM) r.orow = b * a.n + reinterpret_cast<const int *>(a.perm_rec + (size_t)b * a.perm_stride)[4 * (row - b * a.n) + 3];
        return r;
    };
    // level 2: interpolation weights (pointnet2_utils.py:98 sqrt; pointnet2_modules.py:140-142) and the rows' offsets in the table
    struct Ctx { float w0, w1, w2; unsigned k0, k1, k2; int orow; };
    auto make = [&](int tile, const Raw &r) {
        Ctx c;
        const float r0 = 1.0f / (__fsqrt_rn(r.d0) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(r.d1) + 1e-8f), r2 = 1.0f / (__fsqrt_rn(r.d2) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        c.w0 = r0 / norm; c.w1 = r1 / norm; c.w2 = r2 / norm;
        const int row = min(tile * 16 + fi, a.rows - 1);
        const int b0 = __builtin_amdgcn_readfirstlane((tile * 16) / a.n);
        const unsigned base = (unsigned)(b0 + (row >= (b0 + 1) * a.n ? 1 : 0)) * (unsigned)a.m;
        c.k0 = (base + (unsigned)r.i0) * kC1 + fq * 4; c.k1 = (base + (unsigned)r.i1) * kC1 + fq * 4; c.k2 = (base + (unsigned)r.i2) * kC1 + fq * 4;
        c.orow = r.orow;
        return c;
    };
    struct Item { f32x4 t0, t1, t2; };
    auto load_item = [&](const Ctx &c, int ks) {
        Item x;
        x.t0 = *reinterpret_cast<const f32x4u *>(a.tab + c.k0 + ks * 16);
        x.t1 = *reinterpret_cast<const f32x4u *>(a.tab + c.k1 + ks * 16);
        x.t2 = *reinterpret_cast<const f32x4u *>(a.tab + c.k2 + ks * 16);
        return x;
    };

    Raw rawn = load_raw(tile_of(0));
    Ctx cur = make(tile_of(0), rawn);
    rawn = load_raw(tile_of(1));
    Item ring[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) ring[d] = load_item(cur, d);
    for (int it = 0; it < iters; ++it) {
        const int tile = tile_of(it);
        const Ctx nxt = make(tile_of(it + 1), rawn);   // from the level-1 loads issued one tile ago
        rawn = load_raw(tile_of(it + 2));
        // ---- first layer in the loader + layer 2 (128 -> 64), transposed: lane (fi, fq) ends with channels 16 ct + 4 fq + r of row fi
        f32x4 h2[kT2];
#pragma unroll
        for (int ct = 0; ct < kT2; ++ct) h2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kT1; ++ks) {
            // one scheduling region per k-step: left alone, the scheduler requests all 70 weight fragments of the tile up front (386 registers)
            __builtin_amdgcn_sched_barrier(0);
            const Item x = ring[ks % kD];
            ring[ks % kD] = ks + kD < kT1 ? load_item(cur, ks + kD) : load_item(nxt, ks + kD - kT1);   // the ring runs across the tile boundary
            const int k0 = ks * 16 + fq * 4;
            const f32x4 ps = *reinterpret_cast<const f32x4 *>(s_ps + k0), pf = *reinterpret_cast<const f32x4 *>(s_pf + k0);
            f32x4 h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = cur.w0 * x.t0[e] + cur.w1 * x.t1[e] + cur.w2 * x.t2[e];   // three_interpolate, load_elem's operation order
                h1[e] = fmaxf(__builtin_fmaf(v, ps[e], pf[e]), 0.f);
            }
#pragma unroll
            for (int ct = 0; ct < kT2; ++ct) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w2 + ((ct * kT1 + ks) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) h2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h1[e], h2[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool row_ok = tile * 16 + fi < a.rows;
#pragma unroll
        for (int ct = 0; ct < kT2; ++ct) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc2 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh2 + ct * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[ct][r] = fmaxf(__builtin_fmaf(h2[ct][r], sc[r], sh[r]), 0.f);
            if (row_ok) *reinterpret_cast<f32x4 *>(a.tap + (size_t)cur.orow * a.tap_ld + ct * 16 + fq * 4) = h2[ct];   // the FP module's output
        }
        if constexpr (HEAD) {
        // ---- layer 3 (64 -> 32), transposed
        __builtin_amdgcn_sched_barrier(0);
        f32x4 h3[kT3];
#pragma unroll
        for (int ct = 0; ct < kT3; ++ct) h3[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kT2; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < kT3; ++ct) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w3 + ((ct * kT2 + ks) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) h3[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h2[ks][e], h3[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < kT3; ++ct) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc3 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh3 + ct * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) h3[ct][r] = fmaxf(__builtin_fmaf(h3[ct][r], sc[r], sh[r]), 0.f);
        }
        // ---- last layer (32 -> <= 16), normal orientation: lane (fi, fq) holds rows 4 fq + r of channel fi
        __builtin_amdgcn_sched_barrier(0);
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kT3; ++ks) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w4 + (ks * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(h3[ks][e], w[e], o, 0, 0, 0);
        }
        {
            const float sc = s_sc4[fi], sh = s_sh4[fi];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = __builtin_fmaf(o[r], sc, sh);
                if (a.relu4) y = fmaxf(y, 0.f);
                // the row's original position sits in lane (row & 15) of cur.orow: rows 4 fq + r of this lane
                const int orow = __builtin_amdgcn_ds_bpermute((fq * 4 + r) << 2, cur.orow);
                if (fi < a.cout4 && tile * 16 + fq * 4 + r < a.rows) a.out[(size_t)orow * a.ldo + fi] = y;
            }
        }
        }   // HEAD
        cur = nxt;
    }
}

}  // namespace g4d

using namespace g4d;

// Takes the launch if it is the instantiated stack (128-wide table -> 64 -> 32 -> <= 16, ReLU on the first three) and large enough to
// pipeline; returns -1 when it is not (the caller then runs the register-chain kernel), else the launch status.
int g4d::fp_table_try(long long rows, int n, int m, int C2, const float *table, const float *dist2, const int *nn_idx, const void *perm_rec,
                      size_t perm_stride, const float *pre_scale, const float *pre_shift, float *in_tap, int nlayers, const float *const *W,
                      const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout, const int *relu, float *out, int ldo,
                      int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st) {
    const int on = (int)tuning("fp_table_persistent", 1);          // A/B switch
    const long long min_rows = tuning("fp_table_min_rows", 262144);
    // the FP level alone (round 6): one layer behind the table, 128 -> 64, written where the launch's output goes
    const bool alone = nlayers == 1 && !tap_out && !in_tap && col0 == 0 && Cout[0] == kC2 && Kpad[0] == kC1 && relu[0];
    if (alone) { tap_out = out; tap_ld = ldo; }
    if (!on || rows < min_rows || rows >= (1ll << 31) - 64 || C2 != kC1 || (nlayers != 3 && !alone) || in_tap || col0 != 0 || (!alone && tap_layer != 0) || !tap_out) return -1;
    if (!alone && (Cout[0] != kC2 || Cout[1] != kC3 || Cout[2] > kC4 || Cout[2] < 1 || Kpad[0] != kC1 || Kpad[1] != kC2 || Kpad[2] != kC3 || !relu[0] || !relu[1])) return -1;
    if (n < 16 || m <= 0 || rows % n != 0 || (rows / n) * (long long)m * kC1 >= (1ll << 32) || tap_ld % 4 != 0 || (reinterpret_cast<size_t>(tap_out) & 15) != 0 ||
        (reinterpret_cast<size_t>(table) & 15) != 0) return -1;
    G4D_REQUIRE(table && dist2 && nn_idx && pre_scale && pre_shift && out && W[0] && scale[0] && shift[0] && (alone || (W[1] && W[2] && scale[1] && scale[2] && shift[1] && shift[2])),
                "g4d_mlp_chain_table_f32: null pointer");
    G4D_REQUIRE((alone || ldo >= Cout[2]) && tap_ld >= Cout[0], "g4d_mlp_chain_table_f32: output row stride %d or tap stride %d too small", ldo, tap_ld);
    FpTabArgs a;
    a.rows = (int)rows; a.n = n; a.m = m; a.tab = table; a.dist2 = dist2; a.nn_idx = nn_idx;
    a.perm_rec = reinterpret_cast<const unsigned char *>(perm_rec); a.perm_stride = perm_stride;
    a.ps = pre_scale; a.pf = pre_shift;
    a.W2 = W[0]; a.sc2 = scale[0]; a.sh2 = shift[0];
    a.W3 = alone ? nullptr : W[1]; a.sc3 = alone ? nullptr : scale[1]; a.sh3 = alone ? nullptr : shift[1];
    a.W4 = alone ? nullptr : W[2]; a.sc4 = alone ? nullptr : scale[2]; a.sh4 = alone ? nullptr : shift[2];
    a.cout4 = alone ? 0 : Cout[2]; a.relu4 = alone ? 0 : relu[2]; a.out = out; a.ldo = ldo; a.tap = tap_out; a.tap_ld = tap_ld;
    // ring depth 2 (measured at 240 clouds, 1.97 M rows: 2 k-steps ahead 422 us, 4: 436-440, 8: 448 -- the deeper rings only add registers)
    typedef void (*Kern)(const FpTabArgs);
    const Kern kern = alone ? (perm_rec ? fp_table_head_kernel<true, 2, false> : fp_table_head_kernel<false, 2, false>)
                            : (perm_rec ? fp_table_head_kernel<true, 2, true> : fp_table_head_kernel<false, 2, true>);
    static int resident[4] = {0, 0, 0, 0};
    int &res = resident[(perm_rec ? 1 : 0) + (alone ? 2 : 0)];
    if (res == 0) {   // (benign race: every thread computes the same value)
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        res = (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) ? per_cu * 256 : per_cu * prop.multiProcessorCount;
    }
    const long long want = ((rows + 15) / 16 + 3) / 4;
    hipLaunchKernelGGL(kern, dim3((unsigned)(want < res ? want : res)), dim3(256), 0, st, a);
    return check_launch("g4d_fp_table_head");
}
