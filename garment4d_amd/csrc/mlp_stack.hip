// Whole shared-MLP stacks (up to 4 layers) in ONE launch, activations resident in LDS.
//
// The one-layer kernel (mlp.hip) already keeps the grouped tensor out of HBM; what is left of the reference's
// traffic is the hidden activations (B,C,P,S) between the 1x1 convs (pointnet2_modules.py:40 runs Conv2d+BN+ReLU
// as separate cuDNN/elementwise launches, each round-tripping HBM) and one kernel boundary per layer on launches
// that last 5-30 us.  Here a workgroup takes 64 rows (64 / S whole neighbourhoods) through every layer:
//
//   phase 0  the row loader (GROUP / INTERP / DIRECT / CSR, mlp_common.h) gathers the whole [64][K0pad] input
//            tile into LDS buffer 0 -- all global loads of the tile are in flight at once;
//   layer l  A fragments come from the input buffer (ds_read_b128), W fragments straight from global memory
//            (weights are a few hundred KB, L2-resident, each lane reads 16 contiguous bytes of its row),
//            8 waves (4 channel slices x 2 row halves) x 2 accumulator tiles of v_mfma_f32_16x16x4_f32 per 64-channel
//            slab -- the activation buffers bound occupancy (1-4 workgroups per CU), so latency is hidden by waves
//            per workgroup rather than workgroups per CU; the epilogue applies the
//            folded BN affine + ReLU and scatters into the other LDS buffer (ping-pong), or -- last layer -- pools
//            over the S samples in-wave and writes point-major output at a column offset;
//   one barrier per layer (plus one per extra 64-channel slab), none per K chunk.
//
// Buffers are [64][ld] fp32 with ld = widest Kpad read from them + 4 (spreads ds_read_b128 over the banks); a
// hidden layer only writes the columns its successor reads.  Waves whose 16-channel slice lies beyond Cout skip their MFMAs (narrow layers leave the matrix pipe to
// the other resident workgroups).  An optional tap writes one hidden layer to HBM as well (FP1 features feed the
// head AND are returned to the caller).
#include "mlp_common.h"

namespace g4d {

constexpr int kMaxLayers = 4;

struct StackLayer {
    const float *W, *scale, *shift;  // W in FRAGMENT order [CoutPad64/16][Kpad/16][64 lanes][4]; scale/shift [CoutPad64]
    int Kpad, Cout, relu;
};

struct StackArgs {
    LinearArgs in;  // loader description + rows/K/S/pool/out/ldo/col0 (W/scale/shift/Kpad/Cout unused)
    StackLayer layer[kMaxLayers];
    int nlayers;
    int ld0, ld1;  // LDS row strides (floats) of buffer 0 / 1
    int tap_layer; // -1 or index of a hidden layer whose output is also stored to HBM
    float *tap_out;
    int tap_ld;
};

// MT = accumulator tiles per wave: the workgroup takes ROWS = 32*MT rows (64, or 32 to halve the LDS footprint and
// double the resident workgroups when the pool window allows it).
template <int MODE, int MT>
__global__ void __launch_bounds__(512) mlp_stack_kernel(const StackArgs s) {
    constexpr int ROWS = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *buf0 = smem;
    float *buf1 = smem + ROWS * s.ld0;
    const LinearArgs &a = s.in;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cs = wave & 3;   // 16-channel slice of the 64-channel slab
    const int rh = wave >> 2;  // row half: rows [16 MT rh, 16 MT (rh + 1))
    const int row0 = blockIdx.x * ROWS;
    const int fi = lane & 15, fq = lane >> 4;

    // ---- phase 0: gather the input tile [64][K0pad] into buf0 (zeros beyond K and beyond the last row)
    {
        constexpr int TPR = 512 / ROWS;  // threads per row: 8 (64 rows) or 16 (32 rows)
        const int lr = t / TPR;
        const RowCtx<MODE> ctx = make_ctx<MODE>(a, row0 + lr);
        const int K0pad = s.layer[0].Kpad;
        for (int k = (t % TPR) * 4; k < K0pad; k += TPR * 4) {
            f32x4 v;
            if (MODE == LOAD_DIRECT && ctx.valid && k + 3 < a.K && (a.ldx & 3) == 0) {
                v = *reinterpret_cast<const f32x4 *>(a.X + (size_t)(row0 + lr) * a.ldx + k);
            } else {
                v.x = load_elem<MODE>(a, ctx, row0 + lr, k);
                v.y = load_elem<MODE>(a, ctx, row0 + lr, k + 1);
                v.z = load_elem<MODE>(a, ctx, row0 + lr, k + 2);
                v.w = load_elem<MODE>(a, ctx, row0 + lr, k + 3);
            }
            *reinterpret_cast<f32x4 *>(&buf0[lr * s.ld0 + k]) = v;
        }
    }
    // first W fragment of the first job: in flight across the gather barrier
    // W is in fragment order: [16-channel tile][k-step][lane][4]  (one B-fragment load = 1 KB contiguous per wave)
    f32x4 bnext = *reinterpret_cast<const f32x4 *>(s.layer[0].W + (size_t)cs * (s.layer[0].Kpad >> 4) * 256 + lane * 4);
    __syncthreads();

    for (int l = 0; l < s.nlayers; ++l) {
        const StackLayer &L = s.layer[l];
        const float *in = (l & 1) ? buf1 : buf0;
        float *out = (l & 1) ? buf0 : buf1;
        const int ldin = (l & 1) ? s.ld1 : s.ld0, ldout = (l & 1) ? s.ld0 : s.ld1;
        const bool last = l == s.nlayers - 1;
        // a hidden layer only has to produce the columns the next layer reads (its Kpad; zeros beyond Cout)
        const int wcols = last ? L.Cout : s.layer[l + 1].Kpad;
        const int nslab = (wcols + 63) >> 6;
        for (int sl = 0; sl < nslab; ++sl) {
            const int ch = sl * 64 + cs * 16 + fi;
            const bool wave_live = sl * 64 + cs * 16 < L.Cout;   // wave-uniform: has real channels
            const bool wave_writes = sl * 64 + cs * 16 < wcols;  // wave-uniform: columns somebody reads
            f32x4 acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 bf = bnext;
            if (wave_live) {
                const float *wp = L.W + (size_t)(sl * 4 + cs) * (L.Kpad >> 4) * 256 + lane * 4;
                const float *ap = in + (rh * 16 * MT + fi) * ldin + fq * 4;
                for (int kk = 0; kk < L.Kpad; kk += 16) {
                    const f32x4 bcur = bf;
                    if (kk + 16 < L.Kpad) bf = *reinterpret_cast<const f32x4 *>(wp + (kk + 16) * 16);  // next W fragment, ahead of the MFMAs
                    f32x4 av[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4 *>(ap + mt * 16 * ldin + kk);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], bcur[e], acc[mt], 0, 0, 0);
                }
            }
            {   // first W fragment of the NEXT job (next slab, or slab 0 of the next layer): issued before the epilogue
                int nl = l, nsl = sl + 1;
                if (nsl >= nslab) { nl = l + 1; nsl = 0; }
                if (nl < s.nlayers) {
                    const StackLayer &NL = s.layer[nl];
                    bnext = *reinterpret_cast<const f32x4 *>(NL.W + (size_t)(nsl * 4 + cs) * (NL.Kpad >> 4) * 256 + lane * 4);
                }
            }
            const float sc = L.scale[ch], sh = L.shift[ch];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = __builtin_fmaf(acc[mt][r], sc, sh);
                    if (L.relu) y = fmaxf(y, 0.f);
                    acc[mt][r] = y;
                }
            if (!last) {
                // hidden layer: scatter into the other LDS buffer (channels beyond Cout come out as exact zeros:
                // W rows, scale and shift are zero padded), optionally tap to HBM
                if (wave_writes) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[(rh * 16 * MT + mt * 16 + fq * 4 + r) * ldout + ch] = acc[mt][r];
                }
                if (l == s.tap_layer && ch < L.Cout) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row0 + rh * 16 * MT + mt * 16 + fq * 4 + r;
                            if (row < a.rows) s.tap_out[(size_t)row * s.tap_ld + ch] = acc[mt][r];
                        }
                }
                continue;
            }
            // ---- last layer: (pool and) store to HBM
            const bool ch_ok = ch < L.Cout;
            if (a.pool == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + rh * 16 * MT + mt * 16 + fq * 4 + r;
                        if (ch_ok && row < a.rows) a.out[(size_t)row * a.ldo + a.col0 + ch] = acc[mt][r];
                    }
            } else {
                const bool is_max = a.pool == 1;
                if (a.S < 16) {  // S = 4 | 8: 4 | 2 neighbourhoods per accumulator tile (lane owns rows fq*4 .. fq*4+3)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        float x = is_max ? fmaxf(fmaxf(acc[mt][0], acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]))
                                         : ((acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]));
                        if (a.S == 8) {
                            const float y = lane_xor16(x);
                            x = is_max ? fmaxf(x, y) : x + y;
                        }
                        const int first_row = row0 + rh * 16 * MT + mt * 16 + (a.S == 8 ? (fq >> 1) * 8 : fq * 4);
                        const bool writer = a.S == 8 ? (fq & 1) == 0 : true;
                        if (writer && ch_ok && first_row < a.rows)
                            a.out[(size_t)(first_row / a.S) * a.ldo + a.col0 + ch] = is_max ? x : x / (float)a.S;
                    }
                    continue;
                }
                float v[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float x = is_max ? fmaxf(fmaxf(acc[mt][0], acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]))
                                     : ((acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]));
                    const float y = lane_xor16(x);
                    x = is_max ? fmaxf(x, y) : x + y;
                    const float z = lane_xor32(x);
                    x = is_max ? fmaxf(x, z) : x + z;
                    v[mt] = x;
                }
                const float inv = is_max ? 1.f : 1.f / (float)a.S;
                if (a.S == 16) {  // one neighbourhood per accumulator tile
                    if (lane < 16 && ch_ok) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int orow = (row0 >> 4) + rh * MT + mt;
                            if (orow * 16 < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = v[mt] * inv;
                        }
                    }
                } else {
                    float x = v[0];  // this wave's 16*MT rows
                    if constexpr (MT == 2) x = is_max ? fmaxf(v[0], v[1]) : v[0] + v[1];
                    if (a.S == 16 * MT) {
                        const int orow = row0 / a.S + rh;
                        if (lane < 16 && ch_ok && orow * a.S < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = x * inv;
                    } else {  // S == 32*MT == ROWS: the two row halves meet in LDS (`out` is free during the last layer)
                        if (rh == 1 && lane < 16) out[cs * 16 + lane] = x;
                        __syncthreads();
                        if (rh == 0 && lane < 16) {
                            const float y = out[cs * 16 + lane];
                            x = is_max ? fmaxf(x, y) : x + y;
                            const int orow = row0 / a.S;
                            if (ch_ok && orow * a.S < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = x * inv;
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();  // layer l's output complete before layer l+1 reads it (and before buf reuse)
    }
}

}  // namespace g4d

using namespace g4d;

// One C entry point for all loaders: `mode` 0 DIRECT, 1 GROUP, 2 INTERP, 3 CSR; loader pointers that a mode does
// not use are ignored.  Layer descriptors arrive as parallel arrays (host memory) of length nlayers <= 4.
extern "C" int g4d_mlp_stack_f32(int mode, long long rows, int K0,
                                 /* DIRECT / CSR */ const float *X, int ldx,
                                 /* GROUP  */ int N, int P, int S, int C, int use_xyz, const float *xyz, const float *new_xyz,
                                 const float *feats, const int *idx,
                                 /* INTERP */ int n, int m, int C2, int C1, const float *known_feats, const float *skip,
                                 const float *dist2, const int *nn_idx,
                                 /* CSR    */ int Vg, const int *rowptr, const int *colidx, const float *vals,
                                 /* layers */ int nlayers, const float *const *W, const float *const *scale,
                                 const float *const *shift, const int *Kpad, const int *Cout, const int *relu,
                                 /* output */ int pool, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld,
                                 g4d_stream_t stream) {
    G4D_REQUIRE(mode >= 0 && mode <= 3, "g4d_mlp_stack_f32: bad mode");
    G4D_REQUIRE(nlayers >= 1 && nlayers <= kMaxLayers, "g4d_mlp_stack_f32: 1..%d layers", kMaxLayers);
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) && K0 > 0, "g4d_mlp_stack_f32: bad sizes");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "g4d_mlp_stack_f32: null pointer");
    G4D_REQUIRE(pool >= 0 && pool <= 2, "g4d_mlp_stack_f32: pool must be 0|1|2");
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "g4d_mlp_stack_f32: pooling needs S in {4,8,16,32,64}");
    StackArgs s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    s.in.rowptr = rowptr; s.in.colidx = colidx; s.in.vals = vals; s.in.Vg = Vg;
    s.nlayers = nlayers;
    int w0 = 0, w1 = 0;  // widths (floats) buffer 0 / 1 must hold: layer l reads Kpad[l] columns of buffer l&1
    for (int l = 0; l < nlayers; ++l) {
        G4D_REQUIRE(W[l] && scale[l] && shift[l] && Kpad[l] % 32 == 0 && Cout[l] > 0, "g4d_mlp_stack_f32: bad layer %d", l);
        s.layer[l].W = W[l]; s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].Kpad = Kpad[l]; s.layer[l].Cout = Cout[l]; s.layer[l].relu = relu[l];
        int &win = (l & 1) ? w1 : w0;
        win = win > Kpad[l] ? win : Kpad[l];
        if (l > 0) {
            const int prev_pad64 = (Cout[l - 1] + 63) / 64 * 64;  // packed W/scale/shift rows of layer l-1 exist up to here
            G4D_REQUIRE(Kpad[l] <= prev_pad64 && Kpad[l] >= Cout[l - 1], "g4d_mlp_stack_f32: layer %d K does not chain", l);
        }
    }
    G4D_REQUIRE(Kpad[0] >= K0, "g4d_mlp_stack_f32: Kpad[0] < K0");
    s.ld0 = w0 + 8;   // (+ 8, not + 4: conflict-free ds_read_b128 fragment reads for widths that are multiples of 16 -- gemm_tile.hip)
    s.ld1 = w1 + 8;
    // 32-row workgroups (half the LDS, twice the workgroups) are available behind G4D_STACK_MT=1
    static const int mt_env = getenv("G4D_STACK_MT") ? atoi(getenv("G4D_STACK_MT")) : 0;  // tuning hook: 1 | 2 | 0 (auto)
    const size_t lds64 = sizeof(float) * 64 * (size_t)(s.ld0 + s.ld1);
    G4D_REQUIRE(lds64 <= 150 * 1024, "g4d_mlp_stack_f32: stack too wide for LDS (%zu bytes)", lds64);
    const bool can32 = !pool || S <= 32;
    const bool want32 = mt_env == 1;  // measured: no gain from 32-row workgroups on any cfg2 stack, 64 rows stays the default
    const int mt = (can32 && want32) ? 1 : 2;
    const size_t lds = lds64 / 2 * mt;
    s.tap_layer = tap_out ? tap_layer : -1;
    s.tap_out = tap_out; s.tap_ld = tap_ld;
    G4D_REQUIRE(s.tap_layer < nlayers - 1, "g4d_mlp_stack_f32: tap must be a hidden layer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rows_per_wg = 32 * mt;
    dim3 grid((unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), block(512);
#define G4D_LAUNCH_STACK(M)                                                                                        \
    {                                                                                                              \
        static unsigned long long attr2 = 0, attr1 = 0; /* one bit per device */                                   \
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_stack_kernel<M, 2>), 150 * 1024, attr2,   \
                                              "g4d_mlp_stack_f32")) return rc;                                     \
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_stack_kernel<M, 1>), 150 * 1024, attr1,   \
                                              "g4d_mlp_stack_f32")) return rc;                                     \
        if (mt == 2) hipLaunchKernelGGL((mlp_stack_kernel<M, 2>), grid, block, lds, st, s);                        \
        else hipLaunchKernelGGL((mlp_stack_kernel<M, 1>), grid, block, lds, st, s);                                \
    }
    switch (mode) {
        case LOAD_DIRECT: G4D_LAUNCH_STACK(LOAD_DIRECT) break;
        case LOAD_GROUP: G4D_LAUNCH_STACK(LOAD_GROUP) break;
        case LOAD_INTERP: G4D_LAUNCH_STACK(LOAD_INTERP) break;
        default: G4D_LAUNCH_STACK(LOAD_CSR) break;
    }
#undef G4D_LAUNCH_STACK
    return check_launch("g4d_mlp_stack_f32");
}
