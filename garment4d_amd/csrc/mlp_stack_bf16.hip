// bf16 variant of mlp_stack.hip (BASELINE config 3: "bf16 = MLP operands only"): activations and weights are bf16 in
// LDS / registers, v_mfma_f32_16x16x32_bf16 accumulates in fp32, the folded-BN affine + ReLU run in fp32 and the result
// is rounded (RNE) to bf16 for the next layer; coordinates, distances, indices, the pooled output and the optional tap
// stay fp32.  Same structure as the fp32 kernel (64 rows per workgroup, 8 waves = 4 channel slices x 2 row halves, one
// barrier per layer, W fragments straight from L2 in fragment order) but one MFMA covers K=32 and the LDS footprint is
// halved, so even the wide FP stacks ([576,512,256], [352,256,128]) fit: the whole encoder is 9 launches.
#include "mlp_common.h"

namespace g4d {

constexpr int kMaxLayersH = 4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_s __attribute__((ext_vector_type(2)));
typedef float f32x2_s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {  // RNE, lo -> bits [15:0]; compiles to v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_s){lo, hi}, bf16x2_s));  // (no inline asm: see mlp_chain_bf16.hip)
}

struct StackLayerH {
    const unsigned short *W;      // bf16, FRAGMENT order [CoutPad64/16][Kpad/32][64 lanes][8]
    const float *scale, *shift;   // fp32 [CoutPad64]
    int Kpad, Cout, relu;
};

struct StackArgsH {
    LinearArgs in;  // loader description + rows/K/S/pool/out/ldo/col0 (W/scale/shift/Kpad/Cout unused)
    StackLayerH layer[kMaxLayersH];
    int nlayers;
    int ld0, ld1;  // LDS row strides (floats) of buffer 0 / 1
    int tap_layer; // -1 or index of a hidden layer whose output is also stored to HBM
    float *tap_out;
    int tap_ld;
};

template <int MODE>
__global__ void __launch_bounds__(512) mlp_stack_bf16_kernel(const StackArgsH s) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];  // bf16 activations
    unsigned short *buf0 = smem;
    unsigned short *buf1 = smem + 64 * s.ld0;
    const LinearArgs &a = s.in;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cs = wave & 3;   // 16-channel slice of the 64-channel slab
    const int rh = wave >> 2;  // row half: rows [32 rh, 32 rh + 32) = accumulator tiles 2 rh, 2 rh + 1
    const int row0 = blockIdx.x * 64;
    const int fi = lane & 15, fq = lane >> 4;

    // ---- phase 0: gather the input tile [64][K0pad] into buf0 (zeros beyond K and beyond the last row)
    {
        const int lr = t >> 3;  // 8 threads per row
        const RowCtx<MODE> ctx = make_ctx<MODE>(a, row0 + lr);
        const int K0pad = s.layer[0].Kpad;
        for (int k = (t & 7) * 4; k < K0pad; k += 32) {
            f32x4 v;
            if (MODE == LOAD_DIRECT && ctx.valid && k + 3 < a.K && (a.ldx & 3) == 0) {
                v = *reinterpret_cast<const f32x4 *>(a.X + (size_t)(row0 + lr) * a.ldx + k);
            } else {
                v.x = load_elem<MODE>(a, ctx, row0 + lr, k);
                v.y = load_elem<MODE>(a, ctx, row0 + lr, k + 1);
                v.z = load_elem<MODE>(a, ctx, row0 + lr, k + 2);
                v.w = load_elem<MODE>(a, ctx, row0 + lr, k + 3);
            }
            *reinterpret_cast<uint2 *>(&buf0[lr * s.ld0 + k]) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        }
    }
    // first W fragment of the first job: in flight across the gather barrier
    // W is in fragment order: [16-channel tile][k-step of 32][lane][8 bf16]  (one B-fragment load = 1 KB contiguous per wave)
    uint4 bnext = *reinterpret_cast<const uint4 *>(s.layer[0].W + (size_t)cs * (s.layer[0].Kpad >> 5) * 512 + lane * 8);
    __syncthreads();

    for (int l = 0; l < s.nlayers; ++l) {
        const StackLayerH &L = s.layer[l];
        const unsigned short *in = (l & 1) ? buf1 : buf0;
        unsigned short *out = (l & 1) ? buf0 : buf1;
        const int ldin = (l & 1) ? s.ld1 : s.ld0, ldout = (l & 1) ? s.ld0 : s.ld1;
        const bool last = l == s.nlayers - 1;
        // a hidden layer only has to produce the columns the next layer reads (its Kpad; zeros beyond Cout)
        const int wcols = last ? L.Cout : s.layer[l + 1].Kpad;
        const int nslab = (wcols + 63) >> 6;
        for (int sl = 0; sl < nslab; ++sl) {
            const int ch = sl * 64 + cs * 16 + fi;
            const bool wave_live = sl * 64 + cs * 16 < L.Cout;   // wave-uniform: has real channels
            const bool wave_writes = sl * 64 + cs * 16 < wcols;  // wave-uniform: columns somebody reads
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            uint4 bf = bnext;
            if (wave_live) {
                const unsigned short *wp = L.W + (size_t)(sl * 4 + cs) * (L.Kpad >> 5) * 512 + lane * 8;
                const unsigned short *ap = in + (rh * 32 + fi) * ldin + fq * 8;
                for (int kk = 0; kk < L.Kpad; kk += 32) {
                    const uint4 bcur = bf;
                    if (kk + 32 < L.Kpad) bf = *reinterpret_cast<const uint4 *>(wp + (kk + 32) * 16);  // next W fragment
                    const uint4 a0 = *reinterpret_cast<const uint4 *>(ap + kk);
                    const uint4 a1 = *reinterpret_cast<const uint4 *>(ap + 16 * ldin + kk);
                    const bf16x8 fb = __builtin_bit_cast(bf16x8, bcur);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), fb, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), fb, acc[1], 0, 0, 0);
                }
            }
            {   // first W fragment of the NEXT job (next slab, or slab 0 of the next layer): issued before the epilogue
                int nl = l, nsl = sl + 1;
                if (nsl >= nslab) { nl = l + 1; nsl = 0; }
                if (nl < s.nlayers) {
                    const StackLayerH &NL = s.layer[nl];
                    bnext = *reinterpret_cast<const uint4 *>(NL.W + (size_t)(nsl * 4 + cs) * (NL.Kpad >> 5) * 512 + lane * 8);
                }
            }
            const float sc = L.scale[ch], sh = L.shift[ch];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
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
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out[(rh * 32 + mt * 16 + fq * 4 + r) * ldout + ch] = (unsigned short)(pack_bf16(acc[mt][r], 0.f) & 0xffffu);
                }
                if (l == s.tap_layer && ch < L.Cout) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row0 + rh * 32 + mt * 16 + fq * 4 + r;
                            if (row < a.rows) s.tap_out[(size_t)row * s.tap_ld + ch] = acc[mt][r];
                        }
                }
                continue;
            }
            // ---- last layer: (pool and) store to HBM
            const bool ch_ok = ch < L.Cout;
            if (a.pool == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + rh * 32 + mt * 16 + fq * 4 + r;
                        if (ch_ok && row < a.rows) a.out[(size_t)row * a.ldo + a.col0 + ch] = acc[mt][r];
                    }
            } else {
                const bool is_max = a.pool == 1;
                if (a.S < 16) {  // S = 4 | 8: 4 | 2 neighbourhoods per accumulator tile (lane owns rows fq*4 .. fq*4+3)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float x = is_max ? fmaxf(fmaxf(acc[mt][0], acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]))
                                         : ((acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]));
                        if (a.S == 8) {
                            const float y = lane_xor16(x);
                            x = is_max ? fmaxf(x, y) : x + y;
                        }
                        const int first_row = row0 + rh * 32 + mt * 16 + (a.S == 8 ? (fq >> 1) * 8 : fq * 4);
                        const bool writer = a.S == 8 ? (fq & 1) == 0 : true;
                        if (writer && ch_ok && first_row < a.rows)
                            a.out[(size_t)(first_row / a.S) * a.ldo + a.col0 + ch] = is_max ? x : x / (float)a.S;
                    }
                    continue;
                }
                float v[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
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
                        for (int mt = 0; mt < 2; ++mt) {
                            const int orow = (row0 >> 4) + rh * 2 + mt;
                            if (orow * 16 < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = v[mt] * inv;
                        }
                    }
                } else {
                    float x = is_max ? fmaxf(v[0], v[1]) : v[0] + v[1];  // this wave's 32 rows
                    if (a.S == 32) {
                        const int orow = (row0 >> 5) + rh;
                        if (lane < 16 && ch_ok && orow * 32 < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = x * inv;
                    } else {  // S == 64: the two row halves meet in LDS (`out` is free during the last layer)
                        float *scratch = reinterpret_cast<float *>(out);
                        if (rh == 1 && lane < 16) scratch[cs * 16 + lane] = x;
                        __syncthreads();
                        if (rh == 0 && lane < 16) {
                            const float y = scratch[cs * 16 + lane];
                            x = is_max ? fmaxf(x, y) : x + y;
                            const int orow = row0 >> 6;
                            if (ch_ok && orow * 64 < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = x * inv;
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
extern "C" int g4d_mlp_stack_bf16(int mode, long long rows, int K0,
                                 /* DIRECT / CSR */ const float *X, int ldx,
                                 /* GROUP  */ int N, int P, int S, int C, int use_xyz, const float *xyz, const float *new_xyz,
                                 const float *feats, const int *idx,
                                 /* INTERP */ int n, int m, int C2, int C1, const float *known_feats, const float *skip,
                                 const float *dist2, const int *nn_idx,
                                 /* CSR    */ int Vg, const int *rowptr, const int *colidx, const float *vals,
                                 /* layers */ int nlayers, const unsigned short *const *W, const float *const *scale,
                                 const float *const *shift, const int *Kpad, const int *Cout, const int *relu,
                                 /* output */ int pool, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld,
                                 g4d_stream_t stream) {
    G4D_REQUIRE(mode >= 0 && mode <= 3, "g4d_mlp_stack_bf16: bad mode");
    G4D_REQUIRE(nlayers >= 1 && nlayers <= kMaxLayersH, "g4d_mlp_stack_bf16: 1..%d layers", kMaxLayersH);
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) && K0 > 0, "g4d_mlp_stack_bf16: bad sizes");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "g4d_mlp_stack_bf16: null pointer");
    G4D_REQUIRE(pool >= 0 && pool <= 2, "g4d_mlp_stack_bf16: pool must be 0|1|2");
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "g4d_mlp_stack_bf16: pooling needs S in {4,8,16,32,64}");
    StackArgsH s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    s.in.rowptr = rowptr; s.in.colidx = colidx; s.in.vals = vals; s.in.Vg = Vg;
    s.nlayers = nlayers;
    int w0 = 0, w1 = 0;  // widths (floats) buffer 0 / 1 must hold: layer l reads Kpad[l] columns of buffer l&1
    for (int l = 0; l < nlayers; ++l) {
        G4D_REQUIRE(W[l] && scale[l] && shift[l] && Kpad[l] % 32 == 0 && Cout[l] > 0, "g4d_mlp_stack_bf16: bad layer %d", l);
        s.layer[l].W = W[l]; s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].Kpad = Kpad[l]; s.layer[l].Cout = Cout[l]; s.layer[l].relu = relu[l];
        int &win = (l & 1) ? w1 : w0;
        win = win > Kpad[l] ? win : Kpad[l];
        if (l > 0) {
            const int prev_pad64 = (Cout[l - 1] + 63) / 64 * 64;  // packed W/scale/shift rows of layer l-1 exist up to here
            G4D_REQUIRE(Kpad[l] <= prev_pad64 && Kpad[l] >= Cout[l - 1], "g4d_mlp_stack_bf16: layer %d K does not chain", l);
        }
    }
    G4D_REQUIRE(Kpad[0] >= K0, "g4d_mlp_stack_bf16: Kpad[0] < K0");
    s.ld0 = w0 + 8;  // bf16 elements; +16 bytes spreads the ds_read_b128 fragment reads over the banks
    s.ld1 = w1 + 8;
    const size_t lds = sizeof(unsigned short) * 64 * (size_t)(s.ld0 + s.ld1);
    G4D_REQUIRE(lds <= 150 * 1024, "g4d_mlp_stack_bf16: stack too wide for LDS (%zu bytes)", lds);
    s.tap_layer = tap_out ? tap_layer : -1;
    s.tap_out = tap_out; s.tap_ld = tap_ld;
    G4D_REQUIRE(s.tap_layer < nlayers - 1, "g4d_mlp_stack_bf16: tap must be a hidden layer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + 63) / 64)), block(512);
#define G4D_LAUNCH_STACKH(M)                                                                                        \
    {                                                                                                              \
        static unsigned long long attr = 0; /* one bit per device */                                               \
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(mlp_stack_bf16_kernel<M>), 150 * 1024, attr,  \
                                              "g4d_mlp_stack_bf16")) return rc;                                    \
        hipLaunchKernelGGL(mlp_stack_bf16_kernel<M>, grid, block, lds, st, s);                                          \
    }
    switch (mode) {
        case LOAD_DIRECT: G4D_LAUNCH_STACKH(LOAD_DIRECT) break;
        case LOAD_GROUP: G4D_LAUNCH_STACKH(LOAD_GROUP) break;
        case LOAD_INTERP: G4D_LAUNCH_STACKH(LOAD_INTERP) break;
        default: G4D_LAUNCH_STACKH(LOAD_CSR) break;
    }
#undef G4D_LAUNCH_STACKH
    return check_launch("g4d_mlp_stack_bf16");
}
