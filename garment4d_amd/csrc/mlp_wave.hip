// Wave-autonomous shared-MLP stack: ONE WAVE takes 64 rows through every layer, no workgroup barrier anywhere.
//
// For narrow stacks (hidden widths <= 64: SA level 1 [3,16,16,32] / [3,32,32,64], SA level 2 scale 0) the
// workgroup-cooperative kernel (mlp_stack.hip) is latency bound: between two barriers a workgroup has only a handful
// of MFMAs to issue, and the gather -> barrier -> layer -> barrier chain of one 64-row tile takes ~10 us end to end
// with 2-3 workgroups resident per CU.  Here every wave owns its tile outright:
//
//   * the wave gathers its 64 x K0 input tile into a wave-private LDS slab (8 lanes per row, 128-byte row segments,
//     so the loads stay line-shaped), at most 32 columns at a time;
//   * layer l: A fragments by ds_read_b128 from the slab, W fragments from L1/L2 (a narrow stack's weights are a few
//     KB), 4 row tiles x up to 4 channel tiles of v_mfma_f32_16x16x4_f32 accumulators live in registers;
//   * the epilogue (folded BN affine, ReLU) writes the layer's output back into the SAME slab -- the wave has
//     consumed every A fragment by then, and LDS operations of one wave execute in order, so no barrier and no
//     second buffer; the last layer pools over the S samples in registers / DPP and stores point-major output.
//
// Latency is hidden the GPU way: 12-16 independent waves per CU, each at a different point of its own tile.
#include "mlp_common.h"

namespace g4d {

constexpr int kWaveLayers = 4;

struct WaveLayer {
    const float *W, *scale, *shift;
    int Kpad, Cout, relu;
};

struct WaveArgs {
    LinearArgs in;
    WaveLayer layer[kWaveLayers];
    int nlayers;
    int ld;  // slab row stride (floats)
};

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wave is issued and executed in order; this only stops the COMPILER from moving LDS accesses
    // across the point where one lane reads what another lane of the same wave wrote.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) mlp_wave_kernel(const WaveArgs s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LinearArgs &a = s.in;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float *slab = smem + (size_t)wave * 64 * s.ld;
    const int tile = blockIdx.x * 4 + wave;
    const int row0 = tile * 64;
    if (row0 >= a.rows) return;  // whole wave leaves; no barriers in this kernel
    const int fi = lane & 15, fq = lane >> 4;
    const int ld = s.ld;

    f32x4 acc[4][4];  // [row tile][channel tile of the current 64-channel group]

    for (int l = 0; l < s.nlayers; ++l) {
        const WaveLayer &L = s.layer[l];
        const bool last = l == s.nlayers - 1;
        const int ngroups = (L.Cout + 63) >> 6;  // hidden layers: 1 (Cout <= 64, checked on the host)
        for (int g = 0; g < ngroups; ++g) {
            const int nct = min(4, (L.Cout - g * 64 + 15) >> 4);  // live 16-channel tiles in this group (wave-uniform)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[mt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *wbase = L.W + (size_t)(g * 4) * (L.Kpad >> 4) * 256 + lane * 4;  // fragment-order weights
#pragma unroll 1
            for (int kc = 0; kc < L.Kpad; kc += 32) {
                if (l == 0) {
                    // stage columns [kc, kc+32) of the gathered input: 8 lanes per row, 8 rows per pass
                    if (kc > 0 || g > 0) wave_lds_fence();  // previous chunk's A reads are done (their MFMAs consumed them)
                    if (a.K <= 4) {
                        // xyz-only input (first SA level): lane = row, ONE dependent chain idx -> point -> slab row
                        const RowCtx<MODE> ctx = make_ctx<MODE>(a, row0 + lane);
                        f32x4 v;
                        v.x = load_elem<MODE>(a, ctx, row0 + lane, 0);
                        v.y = load_elem<MODE>(a, ctx, row0 + lane, 1);
                        v.z = load_elem<MODE>(a, ctx, row0 + lane, 2);
                        v.w = load_elem<MODE>(a, ctx, row0 + lane, 3);
                        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                        f32x4 *dst = reinterpret_cast<f32x4 *>(&slab[lane * ld]);
                        dst[0] = v;
#pragma unroll
                        for (int c = 1; c < 8; ++c) dst[c] = z;
                    } else {
#pragma unroll 1
                        for (int pass = 0; pass < 8; ++pass) {
                            const int r = pass * 8 + (lane >> 3);
                            const RowCtx<MODE> ctx = make_ctx<MODE>(a, row0 + r);
                            const int k = kc + (lane & 7) * 4;
                            f32x4 v;
                            if (MODE == LOAD_DIRECT && ctx.valid && k + 3 < a.K && (a.ldx & 3) == 0) {
                                v = *reinterpret_cast<const f32x4 *>(a.X + (size_t)(row0 + r) * a.ldx + k);
                            } else {
                                v.x = load_elem<MODE>(a, ctx, row0 + r, k);
                                v.y = load_elem<MODE>(a, ctx, row0 + r, k + 1);
                                v.z = load_elem<MODE>(a, ctx, row0 + r, k + 2);
                                v.w = load_elem<MODE>(a, ctx, row0 + r, k + 3);
                            }
                            *reinterpret_cast<f32x4 *>(&slab[r * ld + (lane & 7) * 4]) = v;
                        }
                    }
                    wave_lds_fence();
                }
                const int koff = (l == 0) ? 0 : kc;  // layer 0 always stages into columns 0..31
#pragma unroll
                for (int kk = 0; kk < 32; kk += 16) {
                    f32x4 af[4], bf[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        af[mt] = *reinterpret_cast<const f32x4 *>(&slab[(mt * 16 + fi) * ld + koff + kk + fq * 4]);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        if (ct < nct) bf[ct] = *reinterpret_cast<const f32x4 *>(wbase + ((size_t)ct * (L.Kpad >> 4) + ((kc + kk) >> 4)) * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            if (ct < nct) {
#pragma unroll
                                for (int mt = 0; mt < 4; ++mt)
                                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e], bf[ct][e], acc[mt][ct], 0, 0, 0);
                            }
                }
            }
            // ---- epilogue of this channel group
            if (!last) wave_lds_fence();  // all A fragments of the layer are in registers/consumed: the slab may be overwritten
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct >= nct && (last || g * 64 + ct * 16 >= s.layer[l + 1].Kpad)) continue;  // nobody reads these columns
                const int ch = g * 64 + ct * 16 + fi;
                const float sc = L.scale[ch], sh = L.shift[ch];  // zero padded up to CoutPad64
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float y = __builtin_fmaf(acc[mt][ct][r], sc, sh);
                        if (L.relu) y = fmaxf(y, 0.f);
                        acc[mt][ct][r] = y;
                    }
                if (!last) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) slab[(mt * 16 + fq * 4 + r) * ld + ch] = acc[mt][ct][r];
                    continue;
                }
                const bool ch_ok = ch < L.Cout;
                if (a.pool == 0) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row0 + mt * 16 + fq * 4 + r;
                            if (ch_ok && row < a.rows) a.out[(size_t)row * a.ldo + a.col0 + ch] = acc[mt][ct][r];
                        }
                } else {
                    const bool is_max = a.pool == 1;
                    if (a.S < 16) {  // S = 4 | 8: 4 | 2 neighbourhoods per accumulator tile (lane owns rows fq*4 .. fq*4+3)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            float x = is_max ? fmaxf(fmaxf(acc[mt][ct][0], acc[mt][ct][1]), fmaxf(acc[mt][ct][2], acc[mt][ct][3]))
                                             : ((acc[mt][ct][0] + acc[mt][ct][1]) + (acc[mt][ct][2] + acc[mt][ct][3]));
                            if (a.S == 8) {
                                const float y = lane_xor16(x);
                                x = is_max ? fmaxf(x, y) : x + y;
                            }
                            const int first_row = row0 + mt * 16 + (a.S == 8 ? (fq >> 1) * 8 : fq * 4);
                            const bool writer = a.S == 8 ? (fq & 1) == 0 : true;
                            if (writer && ch_ok && first_row < a.rows)
                                a.out[(size_t)(first_row / a.S) * a.ldo + a.col0 + ch] = is_max ? x : x / (float)a.S;
                        }
                        continue;
                    }
                    float v[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float x = is_max ? fmaxf(fmaxf(acc[mt][ct][0], acc[mt][ct][1]), fmaxf(acc[mt][ct][2], acc[mt][ct][3]))
                                         : ((acc[mt][ct][0] + acc[mt][ct][1]) + (acc[mt][ct][2] + acc[mt][ct][3]));
                        const float y = lane_xor16(x);
                        x = is_max ? fmaxf(x, y) : x + y;
                        const float z = lane_xor32(x);
                        x = is_max ? fmaxf(x, z) : x + z;
                        v[mt] = x;
                    }
                    const int groups = 64 / a.S;
                    if (groups == 2) {
                        v[0] = is_max ? fmaxf(v[0], v[1]) : v[0] + v[1];
                        v[1] = is_max ? fmaxf(v[2], v[3]) : v[2] + v[3];
                    } else if (groups == 1) {
                        v[0] = is_max ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : ((v[0] + v[1]) + (v[2] + v[3]));
                    }
                    if (lane < 16 && ch_ok) {
                        const float inv = is_max ? 1.f : 1.f / (float)a.S;
                        for (int gg = 0; gg < groups; ++gg) {
                            const int orow = (row0 / a.S) + gg;
                            if (orow * a.S < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = v[gg] * inv;
                        }
                    }
                }
            }
            if (!last) wave_lds_fence();  // next layer's A reads see this layer's output
        }
    }
}

}  // namespace g4d

using namespace g4d;

// Same argument convention as g4d_mlp_stack_f32 (minus the tap); hidden widths must be <= 64.
extern "C" int g4d_mlp_wave_f32(int mode, long long rows, int K0, const float *X, int ldx, int N, int P, int S, int C, int use_xyz,
                                const float *xyz, const float *new_xyz, const float *feats, const int *idx, int n, int m, int C2,
                                int C1, const float *known_feats, const float *skip, const float *dist2, const int *nn_idx, int Vg,
                                const int *rowptr, const int *colidx, const float *vals, int nlayers, const float *const *W,
                                const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout,
                                const int *relu, int pool, float *out, int ldo, int col0, g4d_stream_t stream) {
    G4D_REQUIRE(mode >= 0 && mode <= 3, "g4d_mlp_wave_f32: bad mode");
    G4D_REQUIRE(nlayers >= 1 && nlayers <= kWaveLayers, "g4d_mlp_wave_f32: 1..%d layers", kWaveLayers);
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) && K0 > 0, "g4d_mlp_wave_f32: bad sizes");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && scale && shift && Kpad && Cout && relu && out, "g4d_mlp_wave_f32: null pointer");
    G4D_REQUIRE(pool >= 0 && pool <= 2, "g4d_mlp_wave_f32: pool must be 0|1|2");
    if (pool) G4D_REQUIRE((S == 4 || S == 8 || S == 16 || S == 32 || S == 64) && rows % S == 0, "g4d_mlp_wave_f32: pooling needs S in {4,8,16,32,64}");
    WaveArgs s = {};
    s.in.rows = (int)rows; s.in.K = K0; s.in.out = out; s.in.ldo = ldo; s.in.col0 = col0; s.in.pool = pool; s.in.S = S > 0 ? S : 1;
    s.in.X = X; s.in.ldx = ldx;
    s.in.xyz = xyz; s.in.new_xyz = new_xyz; s.in.feats = feats; s.in.idx = idx; s.in.N = N; s.in.P = P; s.in.C = C; s.in.use_xyz = use_xyz;
    s.in.known_feats = known_feats; s.in.skip = skip; s.in.dist2 = dist2; s.in.nn_idx = nn_idx; s.in.C2 = C2; s.in.C1 = C1; s.in.m = m; s.in.n = n;
    s.in.rowptr = rowptr; s.in.colidx = colidx; s.in.vals = vals; s.in.Vg = Vg;
    s.nlayers = nlayers;
    int width = 32;  // layer 0 stages 32 columns at a time
    for (int l = 0; l < nlayers; ++l) {
        G4D_REQUIRE(W[l] && scale[l] && shift[l] && Kpad[l] % 32 == 0 && Cout[l] > 0, "g4d_mlp_wave_f32: bad layer %d", l);
        s.layer[l].W = W[l]; s.layer[l].scale = scale[l]; s.layer[l].shift = shift[l];
        s.layer[l].Kpad = Kpad[l]; s.layer[l].Cout = Cout[l]; s.layer[l].relu = relu[l];
        if (l > 0) {
            G4D_REQUIRE(Cout[l - 1] <= 64 && Kpad[l] <= 64 && Kpad[l] >= Cout[l - 1], "g4d_mlp_wave_f32: hidden width of layer %d > 64", l - 1);
            width = width > Kpad[l] ? width : Kpad[l];
        }
    }
    G4D_REQUIRE(Kpad[0] >= K0, "g4d_mlp_wave_f32: Kpad[0] < K0");
    s.ld = width + 4;
    const size_t lds = sizeof(float) * 4 * 64 * (size_t)s.ld;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long tiles = (rows + 63) / 64;
    dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    switch (mode) {
        case LOAD_DIRECT: hipLaunchKernelGGL(mlp_wave_kernel<LOAD_DIRECT>, grid, block, lds, st, s); break;
        case LOAD_GROUP: hipLaunchKernelGGL(mlp_wave_kernel<LOAD_GROUP>, grid, block, lds, st, s); break;
        case LOAD_INTERP: hipLaunchKernelGGL(mlp_wave_kernel<LOAD_INTERP>, grid, block, lds, st, s); break;
        default: hipLaunchKernelGGL(mlp_wave_kernel<LOAD_CSR>, grid, block, lds, st, s); break;
    }
    return check_launch("g4d_mlp_wave_f32");
}
