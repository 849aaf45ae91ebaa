// Positional encoders of the refinement loop (/root/reference/modules/mesh_encoder.py:452-464): per garment vertex q,
// QueryAndGroup(radius, nsample, use_xyz=True) rows [x_j - q ; f_j] -> Linear(3+C, 32) -> ReLU -> Linear(32, 32) -> max
// over the nsample rows.  18 such encoders run per forward over Vg * nsample rows each (31 M rows at Vg = 4096, 240
// frames), with 6- to 387-wide inputs and 32-wide layers -- far too narrow for the LDS-staged stack kernel to pay off.
//
// Wave-autonomous, no LDS, no barriers.  A wave owns 64 consecutive rows (= 64 / nsample queries):
//   layer 1 on the VALU, produced DIRECTLY in the MFMA A-fragment layout: lane (fi = lane & 15, fq = lane >> 4) computes
//     channels {16 ks + 4 fq + e} of rows {16 mt + fi}:  h = relu(t_j[c] + b1[c] + sum_i W1[c][i] * in_i), where
//     in = [x_j - q (3) ; extra_j (E <= 5, e.g. the body normals)] and t_j is an optional per-SOURCE-point table
//     (the feature part Wf f_j + b of the first Linear, computed once per source point by the caller: the first layer
//     is linear, and the garment features do not change over the refinement rounds).  Six FMAs per value instead of a
//     K = 32-padded MFMA contraction; the coordinate difference is formed first in fp32 like the reference does.
//   layer 2 (32 x 32) on v_mfma_f32_16x16x4_f32: 64 MFMAs per 64 rows, B fragments resident in registers.
//   max over the nsample rows in registers / cross-lane, + bias (max(a + b) == max(a) + b), point-major store.
// The 48 + 16 weight registers are loaded once per wave; waves stride over the row chunks.
#include "g4d_common.h"

namespace g4d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PeArgs {
    int n, p, S, logS;
    long long rows;
    const float *xyz, *new_xyz, *extra, *table;
    const int *idx;
    const float *W1, *b1, *W2f, *b2;
    float *out;
    int ldo, col0;
};

template <int E, bool TABLE>
__global__ void __launch_bounds__(256, 2) pos_encode_kernel(const PeArgs a) {
    constexpr int KX = 3 + E;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, fq = lane >> 4;
    float w1[8][KX], bb1[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = (c8 >> 2) * 16 + fq * 4 + (c8 & 3);
#pragma unroll
        for (int i = 0; i < KX; ++i) w1[c8][i] = a.W1[c * KX + i];
        bb1[c8] = a.b1 ? a.b1[c] : 0.f;
    }
    f32x4 bf[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bf[ct][ks] = *reinterpret_cast<const f32x4 *>(a.W2f + ((size_t)(ct * 2 + ks) * 64 + lane) * 4);
    const float bias2[2] = {a.b2[fi], a.b2[16 + fi]};

    const long long nchunks = (a.rows + 63) >> 6;
    for (long long chunk = (long long)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (long long)gridDim.x * 4) {
        const long long row0 = chunk << 6;
        f32x4 af[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            long long row = row0 + mt * 16 + fi;
            if (row >= a.rows) row = a.rows - 1;
            const long long qi = row >> a.logS;                 // global query (f * P + p)
            const long long f = qi / a.p;
            const size_t src = (size_t)f * a.n + a.idx[row];
            float in[KX];
            in[0] = a.xyz[src * 3 + 0] - a.new_xyz[qi * 3 + 0];
            in[1] = a.xyz[src * 3 + 1] - a.new_xyz[qi * 3 + 1];
            in[2] = a.xyz[src * 3 + 2] - a.new_xyz[qi * 3 + 2];
#pragma unroll
            for (int e = 0; e < E; ++e) in[3 + e] = a.extra[src * E + e];
            f32x4 t[2];
            if (TABLE) {
                t[0] = *reinterpret_cast<const f32x4 *>(a.table + src * 32 + fq * 4);
                t[1] = *reinterpret_cast<const f32x4 *>(a.table + src * 32 + 16 + fq * 4);
            }
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                float h = TABLE ? t[c8 >> 2][c8 & 3] + bb1[c8] : bb1[c8];
#pragma unroll
                for (int i = 0; i < KX; ++i) h = __builtin_fmaf(w1[c8][i], in[i], h);
                af[mt][c8 >> 2][c8 & 3] = fmaxf(h, 0.f);
            }
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][ks][e], bf[ct][ks][e], acc[mt][ct], 0, 0, 0);
            }
        // max over the S rows of each query; acc[mt][ct][r] = row 16 mt + 4 fq + r, channel 16 ct + fi
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int ch = ct * 16 + fi;
            float v[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) v[mt] = fmaxf(fmaxf(acc[mt][ct][0], acc[mt][ct][1]), fmaxf(acc[mt][ct][2], acc[mt][ct][3]));
            if (a.S <= 8) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float x = v[mt];
                    if (a.S == 8) x = fmaxf(x, __shfl_xor(x, 16));
                    const long long first_row = row0 + mt * 16 + (a.S == 8 ? (fq >> 1) * 8 : fq * 4);
                    const bool writer = a.S == 8 ? (fq & 1) == 0 : true;
                    if (writer && first_row < a.rows) a.out[(size_t)(first_row >> a.logS) * a.ldo + a.col0 + ch] = x + bias2[ct];
                }
                continue;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                v[mt] = fmaxf(v[mt], __shfl_xor(v[mt], 16));
                v[mt] = fmaxf(v[mt], __shfl_xor(v[mt], 32));
            }
            const int groups = 64 >> a.logS;  // 4 | 2 | 1
            if (groups == 2) {
                v[0] = fmaxf(v[0], v[1]);
                v[1] = fmaxf(v[2], v[3]);
            } else if (groups == 1) {
                v[0] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            }
            if (lane < 16) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long long first_row = row0 + (long long)g * a.S;
                    if (g < groups && first_row < a.rows) a.out[(size_t)(first_row >> a.logS) * a.ldo + a.col0 + ch] = v[g] + bias2[ct];
                }
            }
        }
    }
}

template <bool TABLE>
static void launch_pe(int E, dim3 grid, hipStream_t st, const PeArgs &a) {
    switch (E) {
        case 0: hipLaunchKernelGGL((pos_encode_kernel<0, TABLE>), grid, dim3(256), 0, st, a); break;
        case 1: hipLaunchKernelGGL((pos_encode_kernel<1, TABLE>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((pos_encode_kernel<2, TABLE>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((pos_encode_kernel<3, TABLE>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((pos_encode_kernel<4, TABLE>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((pos_encode_kernel<5, TABLE>), grid, dim3(256), 0, st, a); break;
    }
}

}  // namespace g4d

extern "C" int g4d_pos_encode_f32(int frames, int n, int p, int nsample, int n_extra, const float *xyz, const float *new_xyz,
                                  const float *extra, const float *table, const int *idx, const float *W1, const float *b1,
                                  const float *W2_frag, const float *b2, float *out, int ldo, int col0, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(frames >= 0 && n > 0 && p >= 0 && n_extra >= 0 && n_extra <= 5, "g4d_pos_encode_f32: bad sizes (n_extra <= 5)");
    G4D_REQUIRE(nsample == 4 || nsample == 8 || nsample == 16 || nsample == 32 || nsample == 64,
                "g4d_pos_encode_f32: nsample must be 4|8|16|32|64 (got %d)", nsample);
    const long long rows = (long long)frames * p * nsample;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(xyz && new_xyz && idx && W1 && W2_frag && b2 && out && (extra || n_extra == 0) && (b1 || table),
                "g4d_pos_encode_f32: null pointer");
    G4D_REQUIRE(ldo >= col0 + 32 && col0 >= 0, "g4d_pos_encode_f32: output window out of range");
    PeArgs a;
    a.n = n; a.p = p; a.S = nsample; a.logS = __builtin_ctz((unsigned)nsample); a.rows = rows;
    a.xyz = xyz; a.new_xyz = new_xyz; a.extra = extra; a.table = table; a.idx = idx;
    a.W1 = W1; a.b1 = b1; a.W2f = W2_frag; a.b2 = b2; a.out = out; a.ldo = ldo; a.col0 = col0;
    const long long nchunks = (rows + 63) / 64;
    const long long want = (nchunks + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);  // persistent waves: weights are loaded once
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (table) launch_pe<true>(n_extra, dim3(grid), st, a);
    else launch_pe<false>(n_extra, dim3(grid), st, a);
    return check_launch("g4d_pos_encode_f32");
}
