// Per-point shared-MLP layer on the matrix cores (fp32-in / fp32-accumulate MFMA, exact fp32 FMA chains).
//
// One launch = one layer  Y = act((A . W^T) * scale + shift)  over `rows` points, where the A operand is
// never materialised in HBM: a row LOADER builds each 64 x 32 A tile straight into LDS from
//   DIRECT : point-major activations X[row][k]                                   (hidden layers, FC head)
//   GROUP  : ball-query neighbourhoods -- [xyz[idx]-centre (3) | feats[idx] (C)]  (QueryAndGroup + cat,
//            /root/reference/modules/pointnet2/pointnet2/pointnet2_utils.py:242-265; the (B,3+C,P,S) tensor
//            the reference writes to HBM and reads back never exists here)
//   INTERP : [sum_i w_i * known_feats[idx_i] (C2) | skip feats (C1)] with w from the 3-NN distances
//            (PointnetFPModule.forward, pointnet2_modules.py:139-149)
//   CSR    : sum_u Ahat[v,u] * X[f,u,:]  (GraphConvolution, /root/reference/modules/pygcn/layers.py:35-55,
//            evaluated as (Ahat X) W, algebraically equal to Ahat (X W))
// and the epilogue fuses the folded BatchNorm affine, ReLU and the max/avg pool over the S samples of a
// neighbourhood (pointnet2_modules.py:40-53), writing point-major output at a column offset (MSG concat).
//
// Tiling (gfx950): 256 threads = 4 waves; block tile 64 rows x 64 out-channels, K in chunks of 32 staged in
// double-buffered LDS (rows padded to 36 floats so ds_read_b128 fragment reads spread over all banks);
// wave w owns all 64 rows x channels [16w,16w+16): 4 accumulator tiles of v_mfma_f32_16x16x4_f32.  A lane reads
// 4 consecutive k per ds_read_b128 for A and for W and feeds them to 4 MFMAs (same k permutation on both
// operands).  Global loads of chunk c+1 are issued before the MFMAs of chunk c and written to LDS after them:
// one barrier per chunk.  Pool windows (S in {1,16,32,64}) never leave a wave: regs -> accumulator tiles -> DPP.
//
// Roofline: MFMA fp32 (157 TFLOP/s dense) for the contraction; HBM for the gathers.
#include "mlp_common.h"

namespace g4d {

constexpr int BN = 64, KC = 32, LDT = KC + 8;  // LDS row stride in floats: 40 (round 5; 36 put two of the 16-byte slots of a ds_read_b128 lane group on the same banks in 7 of 16 cases -- gemm_tile.hip has the derivation; SQ_LDS_BANK_CONFLICT 0.73 of the busy cycles)

template <int MODE>
__device__ __forceinline__ f32x4 load4(const LinearArgs &a, const RowCtx<MODE> &c, int row, int k) {
    if constexpr (MODE == LOAD_DIRECT) {
        if (c.valid && k + 3 < a.K && (a.ldx & 3) == 0)  // 16-byte aligned fast path
            return *reinterpret_cast<const f32x4 *>(a.X + (size_t)row * a.ldx + k);
    }
    f32x4 v;
    v.x = load_elem<MODE>(a, c, row, k);
    v.y = load_elem<MODE>(a, c, row, k + 1);
    v.z = load_elem<MODE>(a, c, row, k + 2);
    v.w = load_elem<MODE>(a, c, row, k + 3);
    return v;
}

// MT = accumulator tiles per wave: block tile = (16*MT) rows x 64 channels.  MT=4 for big row counts, MT=2 when the
// launch would otherwise leave CUs idle (few rows, wide layers: the FP levels).
template <int MODE, int MT>
__global__ void __launch_bounds__(256) linear_kernel(const LinearArgs a) {
    constexpr int BM = 16 * MT;
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDT];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDT];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int row0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // staging map: thread -> (tile row lr + 32*pass, 4 consecutive k at lk)
    const int lr = t >> 3, lk = (t & 7) * 4;
    constexpr bool TWO = BM == 64;  // a second pass of 32 A rows
    RowCtx<MODE> ctx0 = make_ctx<MODE>(a, row0 + lr), ctx1 = make_ctx<MODE>(a, TWO ? row0 + lr + 32 : a.rows);
    const float *wrow0 = a.W + (size_t)(n0 + lr) * a.Kpad + lk;
    const float *wrow1 = a.W + (size_t)(n0 + lr + 32) * a.Kpad + lk;

    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunk = a.Kpad / KC;  // Kpad is a multiple of 32
    f32x4 ra0 = load4<MODE>(a, ctx0, row0 + lr, lk), ra1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (TWO) ra1 = load4<MODE>(a, ctx1, row0 + lr + 32, lk);
    f32x4 rb0 = *reinterpret_cast<const f32x4 *>(wrow0), rb1 = *reinterpret_cast<const f32x4 *>(wrow1);
    *reinterpret_cast<f32x4 *>(&sA[0][lr * LDT + lk]) = ra0;
    if (TWO) *reinterpret_cast<f32x4 *>(&sA[0][(lr + 32) * LDT + lk]) = ra1;
    *reinterpret_cast<f32x4 *>(&sB[0][lr * LDT + lk]) = rb0;
    *reinterpret_cast<f32x4 *>(&sB[0][(lr + 32) * LDT + lk]) = rb1;
    __syncthreads();

    const int fi = lane & 15, fq = lane >> 4;  // fragment row/col index and k-quarter
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nchunk;
        if (more) {
            const int k = (c + 1) * KC + lk;
            ra0 = load4<MODE>(a, ctx0, row0 + lr, k);
            if (TWO) ra1 = load4<MODE>(a, ctx1, row0 + lr + 32, k);
            rb0 = *reinterpret_cast<const f32x4 *>(wrow0 + (c + 1) * KC);
            rb1 = *reinterpret_cast<const f32x4 *>(wrow1 + (c + 1) * KC);
        }
#pragma unroll
        for (int kk = 0; kk < KC; kk += 16) {
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(&sB[cur][(wave * 16 + fi) * LDT + kk + fq * 4]);
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[mt] = *reinterpret_cast<const f32x4 *>(&sA[cur][(mt * 16 + fi) * LDT + kk + fq * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][e], bf[e], acc[mt], 0, 0, 0);
            }
        }
        if (more) {
            const int nxt = cur ^ 1;
            *reinterpret_cast<f32x4 *>(&sA[nxt][lr * LDT + lk]) = ra0;
            if (TWO) *reinterpret_cast<f32x4 *>(&sA[nxt][(lr + 32) * LDT + lk]) = ra1;
            *reinterpret_cast<f32x4 *>(&sB[nxt][lr * LDT + lk]) = rb0;
            *reinterpret_cast<f32x4 *>(&sB[nxt][(lr + 32) * LDT + lk]) = rb1;
        }
        __syncthreads();
    }

    // epilogue.  C/D layout of 16x16 MFMA: column (channel) = lane & 15, rows = (lane >> 4) * 4 + reg.
    const int ch = n0 + wave * 16 + fi;
    if constexpr (MODE == LOAD_DIRECT) {
        if (a.tab) {   // wave-uniform: + three_interpolate(tab) of the row (g4d_linear_interp_add_f32), added to the finished contraction
            const int chc = min(ch, a.Cout - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const InterpRow c = interp_row(a, min(row0 + mt * 16 + fq * 4 + r, a.rows - 1));
                    acc[mt][r] = acc[mt][r] + interp_at(a, c, chc);
                }
        }
    }
    const float sc = a.scale[ch], sh = a.shift[ch];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = __builtin_fmaf(acc[mt][r], sc, sh);
            if (a.relu) y = fmaxf(y, 0.f);
            acc[mt][r] = y;
        }
    const bool ch_ok = ch < a.Cout;
    if (a.pool == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + mt * 16 + fq * 4 + r;
                if (ch_ok && row < a.rows) a.out[(size_t)row * a.ldo + a.col0 + ch] = acc[mt][r];
            }
        return;
    }
    // pooled: S in {16,32,64}; rows of one neighbourhood are consecutive and tile-aligned (64 % S == 0); MT == 4 only
    if constexpr (MT == 4) {
    const bool is_max = a.pool == 1;
    float v[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float x = is_max ? fmaxf(fmaxf(acc[mt][0], acc[mt][1]), fmaxf(acc[mt][2], acc[mt][3]))
                         : ((acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]));
        // fold the four 16-lane groups (rows 0-3,4-7,8-11,12-15 of the tile)
        const float y = lane_xor16(x);
        x = is_max ? fmaxf(x, y) : x + y;
        const float z = lane_xor32(x);
        x = is_max ? fmaxf(x, z) : x + z;
        v[mt] = x;
    }
    const int groups = 64 / a.S;  // 4, 2 or 1 neighbourhoods per tile
    if (groups == 2) {
        v[0] = is_max ? fmaxf(v[0], v[1]) : v[0] + v[1];
        v[1] = is_max ? fmaxf(v[2], v[3]) : v[2] + v[3];
    } else if (groups == 1) {
        v[0] = is_max ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : ((v[0] + v[1]) + (v[2] + v[3]));
    }
    if (lane < 16 && ch_ok) {
        const float inv = is_max ? 1.f : 1.f / (float)a.S;
        for (int g = 0; g < groups; ++g) {
            const int orow = (row0 / a.S) + g;
            if (orow * a.S < a.rows) a.out[(size_t)orow * a.ldo + a.col0 + ch] = v[g] * inv;
        }
    }
    }
}

// max / mean over S consecutive rows of a point-major matrix (any S) -- used when S is not 16/32/64.
__global__ void __launch_bounds__(256) pool_rows_kernel(int groups, int S, int C, const float *__restrict__ in, int ldi,
                                                       float *__restrict__ out, int ldo, int col0, int is_max) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)groups * C) return;
    const int g = (int)(gid / C), c = (int)(gid - (long long)g * C);
    const float *p = in + (size_t)g * S * ldi + c;
    float acc = p[0];
    for (int s = 1; s < S; ++s) {
        const float x = p[(size_t)s * ldi];
        acc = is_max ? fmaxf(acc, x) : acc + x;
    }
    out[(size_t)g * ldo + col0 + c] = is_max ? acc : acc / (float)S;
}

// The same transposition with 16-byte accesses on both sides (round 6: the drop-in route converts five feature tensors per call to the reference's
// (B, C, N) layout -- 1.4 GB moved at 240 clouds; the 4-byte form above ran at 4 TB/s): 64 x 64 tiles, a thread reads four float4 along the source
// rows and writes four float4 along the destination rows; LDS row stride 65 (column reads conflict-free).  R and Cc multiples of 4, both matrices
// 16-byte aligned (launcher); partial tiles are predicated per float4.
__global__ void __launch_bounds__(256) transpose4_kernel(int R, int Cc, const float *__restrict__ in, float *__restrict__ out) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const float *src = in + (size_t)b * R * Cc;
    float *dst = out + (size_t)b * R * Cc;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int q = (threadIdx.x & 15) * 4, p = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + p + 16 * i, c = c0 + q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R && c < Cc) v = *reinterpret_cast<const f32x4 *>(src + (size_t)r * Cc + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[p + 16 * i][q + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + p + 16 * i, r = r0 + q;
        if (c < Cc && r < R) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tile[q + e][p + 16 * i];
            *reinterpret_cast<f32x4 *>(dst + (size_t)c * R + r) = v;
        }
    }
}

// (B,C,N) channel-major <-> (B,N,C) point-major, 32x32 LDS tiles, coalesced both ways.
__global__ void __launch_bounds__(256) transpose_kernel(int R, int Cc, const float *__restrict__ in, float *__restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float *src = in + (size_t)b * R * Cc;
    float *dst = out + (size_t)b * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Cc) tile[i][tx] = src[(size_t)(r0 + i) * Cc + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

static int launch_linear(int mode, const LinearArgs &a, hipStream_t s) {
    if (mode == LOAD_DIRECT) {
        int rc = G4D_OK;
        // (round 5: the tile kernel first -- since its loads sit between the MFMA chains it is level with the row-streaming kernel at
        //  K = 128 -> 128 and 8 % ahead at 128 -> 384 (983k rows: 1128 vs 1230 us); the streaming kernel keeps the shapes the tile kernel declines)
        if (gemm_narrow_try(a, s, &rc)) return rc;   // tall, un-pooled, K = Cout = 96: weights in registers (gemm_narrow.hip, round 6)
        if (gemm_tile_try(a, s, &rc)) return rc;     // tall, un-pooled, K >= 128, Cout a multiple of 128: 128 x 128 tiles (gemm_tile.hip)
        if (gemm_stream_try(a, s, &rc)) return rc;   // tall, un-pooled, Cout a multiple of 128 after padding, K <= 128: the row-streaming GEMM
    }
    const int nb = (a.Cout + BN - 1) / BN;
    // 32-row tiles when 64-row tiles would not even give two workgroups per CU (and no fused pooling is asked for)
    const bool small = a.pool == 0 && (long long)((a.rows + 63) / 64) * nb < 2 * 256;
    dim3 block(256);
    if (small) {
        dim3 grid((a.rows + 31) / 32, nb);
        switch (mode) {
            case LOAD_DIRECT: hipLaunchKernelGGL((linear_kernel<LOAD_DIRECT, 2>), grid, block, 0, s, a); break;
            case LOAD_GROUP: hipLaunchKernelGGL((linear_kernel<LOAD_GROUP, 2>), grid, block, 0, s, a); break;
            case LOAD_INTERP: hipLaunchKernelGGL((linear_kernel<LOAD_INTERP, 2>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((linear_kernel<LOAD_CSR, 2>), grid, block, 0, s, a); break;
        }
    } else {
        dim3 grid((a.rows + 63) / 64, nb);
        switch (mode) {
            case LOAD_DIRECT: hipLaunchKernelGGL((linear_kernel<LOAD_DIRECT, 4>), grid, block, 0, s, a); break;
            case LOAD_GROUP: hipLaunchKernelGGL((linear_kernel<LOAD_GROUP, 4>), grid, block, 0, s, a); break;
            case LOAD_INTERP: hipLaunchKernelGGL((linear_kernel<LOAD_INTERP, 4>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((linear_kernel<LOAD_CSR, 4>), grid, block, 0, s, a); break;
        }
    }
    return check_launch("g4d_linear");
}

static int check_common(const char *name, long long rows, int K, int Kpad, int Cout, const float *W, const float *scale,
                        const float *shift, float *out, int ldo, int col0, int pool, int S) {
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) && K > 0 && Cout > 0, "%s: bad sizes", name);
    G4D_REQUIRE(Kpad % KC == 0 && Kpad >= K, "%s: Kpad must be a multiple of %d and >= K", name, KC);
    G4D_REQUIRE(W && scale && shift && out, "%s: null pointer", name);
    G4D_REQUIRE(ldo >= col0 + Cout && col0 >= 0, "%s: output row too narrow", name);
    G4D_REQUIRE(pool >= 0 && pool <= 2, "%s: pool must be 0|1|2", name);
    if (pool) G4D_REQUIRE(S == 16 || S == 32 || S == 64, "%s: fused pooling needs S in {16,32,64} (got %d)", name, S);
    return G4D_OK;
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_linear_f32(long long rows, int K, int Kpad, int Cout, const float *X, int ldx, const float *W,
                              const float *scale, const float *shift, int relu, int pool, int s_pool, float *out, int ldo,
                              int col0, g4d_stream_t stream) {
    if (int rc = check_common("g4d_linear_f32", rows, K, Kpad, Cout, W, scale, shift, out, ldo, col0, pool, s_pool)) return rc;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(X && ldx >= K, "g4d_linear_f32: bad X");
    G4D_REQUIRE(!pool || rows % s_pool == 0, "g4d_linear_f32: rows must be a multiple of the pool window");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = K; a.Kpad = Kpad; a.Cout = Cout; a.W = W; a.scale = scale; a.shift = shift; a.relu = relu;
    a.out = out; a.ldo = ldo; a.col0 = col0; a.pool = pool; a.S = pool ? s_pool : 1; a.X = X; a.ldx = ldx;
    return launch_linear(LOAD_DIRECT, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_linear_interp_add_f32(long long rows, int n, int m, int K, int Kpad, int Cout, const float *X, int ldx, const float *W, const float *table,
                                         int tab_ld, const float *dist2, const int *nn_idx, const float *scale, const float *shift, int relu, float *out, int ldo,
                                         int col0, g4d_stream_t stream) {
    if (int rc = check_common("g4d_linear_interp_add_f32", rows, K, Kpad, Cout, W, scale, shift, out, ldo, col0, 0, 1)) return rc;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(X && ldx >= K && table && dist2 && nn_idx && n > 0 && m > 0 && rows % n == 0 && tab_ld >= Cout,
                "g4d_linear_interp_add_f32: needs X, a table at least as wide as the output, the 3-NN results and whole clouds");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = K; a.Kpad = Kpad; a.Cout = Cout; a.W = W; a.scale = scale; a.shift = shift; a.relu = relu;
    a.out = out; a.ldo = ldo; a.col0 = col0; a.pool = 0; a.S = 1; a.X = X; a.ldx = ldx;
    a.tab = table; a.tab_ld = tab_ld; a.dist2 = dist2; a.nn_idx = nn_idx; a.n = n; a.m = m;
    return launch_linear(LOAD_DIRECT, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_group_linear_f32(int b, int n, int p, int s, int c, int use_xyz, const float *xyz, const float *new_xyz,
                                    const float *feats, const int *idx, int Kpad, int Cout, const float *W,
                                    const float *scale, const float *shift, int relu, int pool, float *out, int ldo,
                                    int col0, g4d_stream_t stream) {
    const long long rows = (long long)b * p * s;
    const int K = (use_xyz ? 3 : 0) + c;
    if (int rc = check_common("g4d_group_linear_f32", rows, K, Kpad, Cout, W, scale, shift, out, ldo, col0, pool, s)) return rc;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(idx && (c == 0 || feats) && (!use_xyz || (xyz && new_xyz)), "g4d_group_linear_f32: null pointer");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = K; a.Kpad = Kpad; a.Cout = Cout; a.W = W; a.scale = scale; a.shift = shift; a.relu = relu;
    a.out = out; a.ldo = ldo; a.col0 = col0; a.pool = pool; a.S = s;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.idx = idx; a.N = n; a.P = p; a.C = c; a.use_xyz = use_xyz;
    return launch_linear(LOAD_GROUP, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_interp_linear_f32(int b, int n, int m, int c2, int c1, const float *known_feats, const float *skip,
                                     const float *dist2, const int *nn_idx, int Kpad, int Cout, const float *W,
                                     const float *scale, const float *shift, int relu, float *out, int ldo, int col0,
                                     g4d_stream_t stream) {
    const long long rows = (long long)b * n;
    if (int rc = check_common("g4d_interp_linear_f32", rows, c2 + c1, Kpad, Cout, W, scale, shift, out, ldo, col0, 0, 1)) return rc;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(known_feats && dist2 && nn_idx && (c1 == 0 || skip), "g4d_interp_linear_f32: null pointer");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = c2 + c1; a.Kpad = Kpad; a.Cout = Cout; a.W = W; a.scale = scale; a.shift = shift; a.relu = relu;
    a.out = out; a.ldo = ldo; a.col0 = col0; a.pool = 0; a.S = 1;
    a.known_feats = known_feats; a.skip = skip; a.dist2 = dist2; a.nn_idx = nn_idx; a.C2 = c2; a.C1 = c1; a.m = m; a.n = n;
    return launch_linear(LOAD_INTERP, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_gcn_linear_f32(int frames, int vg, int fin, const float *X, int ldx, const int *rowptr,
                                  const int *colidx, const float *vals, int Kpad, int Cout, const float *W,
                                  const float *scale, const float *shift, int relu, float *out, int ldo, int col0,
                                  g4d_stream_t stream) {
    const long long rows = (long long)frames * vg;
    if (int rc = check_common("g4d_gcn_linear_f32", rows, fin, Kpad, Cout, W, scale, shift, out, ldo, col0, 0, 1)) return rc;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(X && rowptr && colidx && vals && ldx >= fin, "g4d_gcn_linear_f32: null pointer");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = fin; a.Kpad = Kpad; a.Cout = Cout; a.W = W; a.scale = scale; a.shift = shift; a.relu = relu;
    a.out = out; a.ldo = ldo; a.col0 = col0; a.pool = 0; a.S = 1;
    a.X = X; a.ldx = ldx; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.Vg = vg;
    return launch_linear(LOAD_CSR, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int g4d_pool_rows_f32(int groups, int s, int c, const float *in, int ldi, float *out, int ldo, int col0,
                                 int is_max, g4d_stream_t stream) {
    G4D_REQUIRE(groups >= 0 && s > 0 && c >= 0, "g4d_pool_rows_f32: bad sizes");
    if (groups == 0 || c == 0) return G4D_OK;
    G4D_REQUIRE(in && out, "g4d_pool_rows_f32: null pointer");
    hipLaunchKernelGGL(pool_rows_kernel, dim3((unsigned)(((long long)groups * c + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       groups, s, c, in, ldi, out, ldo, col0, is_max);
    return check_launch("g4d_pool_rows_f32");
}

extern "C" int g4d_transpose_f32(int b, int r, int c, const float *in, float *out, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && r >= 0 && c >= 0 && b <= 65535, "g4d_transpose_f32: bad sizes");
    if ((long long)b * r * c == 0) return G4D_OK;
    G4D_REQUIRE(in && out && (r + 31) / 32 <= 65535, "g4d_transpose_f32: bad args");
    if (r % 4 == 0 && c % 4 == 0 && ((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 15) == 0) {
        hipLaunchKernelGGL(transpose4_kernel, dim3((c + 63) / 64, (r + 63) / 64, b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), r, c, in, out);
        return check_launch("g4d_transpose_f32");
    }
    hipLaunchKernelGGL(transpose_kernel, dim3((c + 31) / 32, (r + 31) / 32, b), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), r, c, in, out);
    return check_launch("g4d_transpose_f32");
}
