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
#include <cstdlib>

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

// 32-bit byte offsets from uniform base pointers (every gathered tensor is < 4 GB, checked by the launcher): the loads
// become `global_load v, v_off, s[base]` with one VALU op per address instead of a 64-bit multiply-add chain.
__device__ __forceinline__ float ldf(const float *base, unsigned elem) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (elem << 2));
}
typedef float f32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 ldf3(const float *base, unsigned elem) {  // one 12-byte load: one cache-line lookup per row instead of three
    return *reinterpret_cast<const F3 *>(reinterpret_cast<const char *>(base) + (elem << 2));
}
__device__ __forceinline__ f32x4 ldf4(const float *base, unsigned elem) {
    return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + (elem << 2));
}

// L1M (round 5): layer 1 on the matrix pipe too.  D1 = W1 (16 channels x K) . in^T (K x 16 rows) per (row tile, channel tile), K = 3 + E <= 8 padded
// to one or two k-steps of v_mfma_f32_16x16x4_f32, the accumulator started from table row + bias: lane (row fi, fq) ends with channels 16 ks + 4 fq + r
// -- exactly the A-operand layout layer 2 wants (above), and the same k-ascending FMA chain as the VALU form (bit-identical).  A lane then loads only
// the ONE input component per k-step it feeds (4 bytes instead of three 12-byte rows), holds 4 weight registers instead of 8 KX, and the layer costs
// 8-16 MFMAs + 32 v_max per 64 rows instead of 32 (KX + 2) VALU instructions (VALU work is ADDED to the matrix pipe's time on this chip).
#ifndef G4D_PE_PIPE_OCC4
#define G4D_PE_PIPE_OCC4 1
#endif
template <int E, bool TABLE, bool SBIG, bool PIPE, bool L1M>  // PIPE: software-pipelined gathers (pays for the table rows, costs occupancy otherwise); SBIG: nsample >= 16, a 16-row tile belongs to ONE query -> its centre is wave-uniform
__global__ void __launch_bounds__(256, (PIPE && !TABLE && L1M && G4D_PE_PIPE_OCC4) ? 4 : 2) pos_encode_kernel(const PeArgs a) {
    constexpr int KX = 3 + E;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, fq = lane >> 4;
    constexpr int KK = L1M ? (KX + 3) / 4 : 1;   // k-steps of layer 1 on the matrix pipe
    float w1[L1M ? 1 : 8][KX], bb1[8];
    float w1f[2][KK];                           // L1M: A fragments of W1: lane (channel fi, fq) holds W1[16 ks + fi][4 kk + fq] (0 beyond KX)
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = (c8 >> 2) * 16 + fq * 4 + (c8 & 3);
        if constexpr (!L1M) {
#pragma unroll
            for (int i = 0; i < KX; ++i) w1[c8][i] = a.W1[c * KX + i];
        }
        bb1[c8] = a.b1 ? a.b1[c] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) w1f[ks][kk] = (L1M && 4 * kk + fq < KX) ? a.W1[(ks * 16 + fi) * KX + min(4 * kk + fq, KX - 1)] : 0.f;
    f32x4 bf[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bf[ct][ks] = *reinterpret_cast<const f32x4 *>(a.W2f + ((size_t)(ct * 2 + ks) * 64 + lane) * 4);
    const float bias2[2] = {a.b2[fi], a.b2[16 + fi]};

    const int rows = (int)a.rows;  // < 2^31 (launcher)
    const int nchunks = (rows + 63) >> 6;
    const int stride = gridDim.x * 4;
    // Software pipeline over the wave's chunks: the gathers are two dependent levels (index -> coordinates / table row).
    // While chunk k is on the VALU / MFMA, the rows of chunk k+1 are in flight and so are the indices of chunk k+2.
    // (The conditional loads make the compiler wait for the prefetch where the branches join, so within a wave little overlaps; a
    // branch-free, two-buffer version of this loop -- unconditional clamped loads, unrolled by two -- measured the same 31-37 G rows/s:
    // the 2-3 waves per SIMD already cover each other's gathers.  Round 2.)
    struct Raw {
        F3 px[4], pq[4], pe[4];
        float ex[4][E == 3 ? 1 : (E ? E : 1)];
        f32x4 t[4][2];
        float c0[4], q0[4], e0[4], c1[4];   // L1M: the lane's own input component of k-step 0 (coordinate / centre / first extra) and of k-step 1
    };
    // (a chunk past the wave's last one is clamped onto the launch's last chunk -- loaded again, never used: a conditional load makes the
    //  compiler wait for the whole prefetch where the branches join, i.e. at once, and the gathers of a wave then overlap nothing)
    auto load_idx = [&](int chunk, int (&iv)[4]) {
        const int row0 = min(chunk, nchunks - 1) << 6;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) iv[mt] = a.idx[row0 + min(mt * 16 + fi, rows - 1 - row0)];
    };
    // The rows of a chunk, one 16-row tile (mt) at a time: the pipelined form requests tile mt of the NEXT chunk between the MFMA groups of
    // layer 2's column steps (round 5: issued as one burst of up to 24 loads at the top of a chunk they held the wave's in-order issue while the
    // 12-16 waves of a CU queued at its vector-memory path; gemm_tile.hip has the measurement).
    auto load_rows_mt = [&](int chunk, const int (&iv)[4], Raw &rw, int mt) {
        const int row0 = __builtin_amdgcn_readfirstlane(min(chunk, nchunks - 1) << 6);
        const int q0 = row0 >> a.logS;                // first query of the chunk (uniform)
        const int f0 = q0 / a.p;                      // its frame: one scalar division per chunk
        const int qnext = (f0 + 1) * a.p;             // first query of the next frame (a chunk touches <= 2 frames, see launcher)
        const int r = min(mt * 16 + fi, rows - 1 - row0);  // clamp the tail chunk onto the last valid row
        int qi = (row0 + r) >> a.logS;
        if (SBIG) qi = __builtin_amdgcn_readfirstlane(qi);
        const int f = f0 + (qi >= qnext ? 1 : 0);
        const unsigned src = (unsigned)(f * a.n + iv[mt]);
        if constexpr (L1M) {
            const unsigned fq2 = (unsigned)min(fq, 2);
            rw.c0[mt] = ldf(a.xyz, src * 3 + fq2);               // in[4 kk + fq]: kk = 0 -> dx | dy | dz | extra 0
            rw.q0[mt] = ldf(a.new_xyz, (unsigned)qi * 3 + fq2);
            if (E > 0) rw.e0[mt] = ldf(a.extra, src * E);
            if (E > 1) rw.c1[mt] = ldf(a.extra, src * E + (unsigned)min(1 + fq, E - 1));   // kk = 1 -> extra 1 + fq (0 beyond E)
        } else {
            rw.px[mt] = ldf3(a.xyz, src * 3);
            rw.pq[mt] = ldf3(a.new_xyz, (unsigned)qi * 3);
            if (E == 3) rw.pe[mt] = ldf3(a.extra, src * 3);
            else {
#pragma unroll
                for (int e = 0; e < E; ++e) rw.ex[mt][e] = ldf(a.extra, src * E + e);
            }
        }
        if (TABLE) {
            rw.t[mt][0] = ldf4(a.table, src * 32 + fq * 4);
            rw.t[mt][1] = ldf4(a.table, src * 32 + 16 + fq * 4);
        }
    };
    auto load_rows = [&](int chunk, const int (&iv)[4], Raw &rw) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) load_rows_mt(chunk, iv, rw, mt);
    };
    int chunk0 = blockIdx.x * 4 + wave;
    int iv_a[4], iv_b[4];
    Raw ra, rb;
    if (PIPE) {
        load_idx(chunk0, iv_a);
        load_rows(chunk0, iv_a, ra);
        load_idx(chunk0 + stride, iv_a);
    }
    // One chunk.  PIPE: rows `cur` (requested a chunk ago) through the layers while the next chunk's rows arrive in `nxt` (through the indices
    // iv_next, requested two chunks ago) and the indices of the chunk after that in iv_nn.  The loop below runs chunks in PAIRS with the two row
    // buffers and the two index arrays swapping roles (round 5: `cur = nxt` at the end of a chunk needed the prefetched rows a few hundred cycles
    // after the last of them had been requested).  A chunk past the wave's last one computes on clamped rows and stores nothing (its first_row
    // tests fail).
    auto do_chunk = [&](int chunk, Raw &cur, Raw &nxt, int (&iv_next)[4], int (&iv_nn)[4]) {
        const int row0 = __builtin_amdgcn_readfirstlane(chunk << 6);
        if (!PIPE) {
            load_idx(chunk, iv_next);
            load_rows(chunk, iv_next, cur);
        }
        // Layer 1 (VALU) and layer 2 (MFMA) as ONE software pipeline over the 8 k-columns of layer 2: column c (= hidden channel 16 ks + 4 fq + e,
        // c = 4 ks + e) of all four row tiles is produced while the 8 MFMAs of column c - 1 run.  Issued as two phases (all of layer 1, then
        // 64 MFMAs) the matrix pipe was 0.48 busy and the VALU 0.4: a wave cannot issue its own VALU work behind a queued MFMA, and the other
        // wave of the SIMD was as often in the same phase as not.  Interleaved, ~4 VALU instructions fit in the 32 cycles of each MFMA.
        float in[4][KX];
        f32x4 d1[4][2];
        if constexpr (L1M) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float dif = cur.c0[mt] - cur.q0[mt];   // the coordinate difference first, in fp32, like the reference
                const float b0 = fq < 3 ? dif : (E > 0 ? cur.e0[mt] : 0.f);
                const float b1_ = (E > 1 && 1 + fq < E) ? cur.c1[mt] : 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d1[mt][ks][e] = TABLE ? cur.t[mt][ks][e] + bb1[ks * 4 + e] : bb1[ks * 4 + e];
                    d1[mt][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[ks][0], b0, d1[mt][ks], 0, 0, 0);
                    if constexpr (KK > 1) d1[mt][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[ks][KK - 1], b1_, d1[mt][ks], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if constexpr (L1M) continue;
            in[mt][0] = cur.px[mt].x - cur.pq[mt].x;
            in[mt][1] = cur.px[mt].y - cur.pq[mt].y;
            in[mt][2] = cur.px[mt].z - cur.pq[mt].z;
            if (E == 3) {
                in[mt][3] = cur.pe[mt].x; in[mt][4] = cur.pe[mt].y; in[mt][5] = cur.pe[mt].z;
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) in[mt][3 + e] = cur.ex[mt][e];
            }
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        float afc[2][4];                                    // [parity of the column][row tile]
        auto column = [&](int c8, float (&o)[4]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if constexpr (L1M) { o[mt] = fmaxf(d1[mt][c8 >> 2][c8 & 3], 0.f); continue; }
                float h = TABLE ? cur.t[mt][c8 >> 2][c8 & 3] + bb1[c8] : bb1[c8];
#pragma unroll
                for (int i = 0; i < KX; ++i) h = __builtin_fmaf(w1[c8][i], in[mt][i], h);
                o[mt] = fmaxf(h, 0.f);
            }
        };
        column(0, afc[0]);
        // k ascending per accumulator (ks, e) as before; eight independent accumulators between two MFMAs on the same one
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                if ((c8 & 1) == 0) load_rows_mt(chunk + stride, iv_next, nxt, c8 >> 1);
                if (c8 == 7) load_idx(chunk + 2 * stride, iv_nn);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c8 + 1 < 8) column(c8 + 1, afc[(c8 + 1) & 1]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afc[c8 & 1][mt], bf[ct][c8 >> 2][c8 & 3], acc[mt][ct], 0, 0, 0);
            if (c8 + 1 < 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {              // one MFMA, then its share of the next column's VALU work
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, L1M ? 1 : (4 * (KX + 2) + 7) / 8, 0);
                }
            }
        }

        // max over the S rows of each query; acc[mt][ct][r] = row 16 mt + 4 fq + r, channel 16 ct + fi
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int ch = ct * 16 + fi;
            float v[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) v[mt] = fmaxf(fmaxf(acc[mt][ct][0], acc[mt][ct][1]), fmaxf(acc[mt][ct][2], acc[mt][ct][3]));
            if (!SBIG) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float x = v[mt];
                    if (a.S == 8) x = max_xor16(x);
                    const int first_row = row0 + mt * 16 + (a.S == 8 ? (fq >> 1) * 8 : fq * 4);
                    const bool writer = a.S == 8 ? (fq & 1) == 0 : true;
                    if (writer && first_row < rows) a.out[(size_t)(first_row >> a.logS) * a.ldo + a.col0 + ch] = x + bias2[ct];
                }
                continue;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                v[mt] = max_xor32(max_xor16(v[mt]));
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
                    const int first_row = row0 + g * a.S;
                    if (g < groups && first_row < rows) a.out[(size_t)(first_row >> a.logS) * a.ldo + a.col0 + ch] = v[g] + bias2[ct];
                }
            }
        }
    };
    if constexpr (PIPE) {
        for (int chunk = chunk0; chunk < nchunks; chunk += 2 * stride) {
            do_chunk(chunk, ra, rb, iv_a, iv_b);                 // iv_a: indices of chunk + stride; iv_b <- indices of chunk + 2 stride
            do_chunk(chunk + stride, rb, ra, iv_b, iv_a);        // (past the wave's last chunk: nothing stored)
        }
    } else {
        for (int chunk = chunk0; chunk < nchunks; chunk += stride) do_chunk(chunk, ra, rb, iv_a, iv_b);
    }
}

template <bool TABLE, bool SBIG>
static void launch_pe(int E, dim3 grid, hipStream_t st, const PeArgs &a) {
    static const int pipe_env = getenv("G4D_PE_PIPE") ? atoi(getenv("G4D_PE_PIPE")) : -1;  // tuning hook: 0 | 1, default by variant
    static const int l1m_env = getenv("G4D_PE_L1_MFMA") ? atoi(getenv("G4D_PE_L1_MFMA")) : 1;   // A/B switch: layer 1 on the matrix pipe (round 5)
    // round 4 measured the pipelined gathers at +10 % with a table and -17 % without (157 registers: three waves per SIMD instead of four).
    // With layer 1 on the matrix pipe the pipelined form without a table fits 128 registers (launch bounds: four waves per SIMD) and wins
    // there too: body encoders 901 / 446 / 241 -> 852 / 422 / 230 us at 240 frames x 4096 garment vertices, S = 32 / 16 / 8.
    const bool pipe = pipe_env >= 0 ? pipe_env != 0 : (l1m_env ? true : (TABLE && E == 0));
#define G4D_PE_LAUNCH(EE, PP, LL) hipLaunchKernelGGL((pos_encode_kernel<EE, TABLE, SBIG, PP, LL>), grid, dim3(256), 0, st, a)
#define G4D_PE_CASE(EE)                                                                                     \
    case EE:                                                                                                \
        if (l1m_env) { if (pipe) G4D_PE_LAUNCH(EE, true, true); else G4D_PE_LAUNCH(EE, false, true); }      \
        else { if (pipe) G4D_PE_LAUNCH(EE, true, false); else G4D_PE_LAUNCH(EE, false, false); }            \
        break;
    switch (E) {
        G4D_PE_CASE(0) G4D_PE_CASE(1) G4D_PE_CASE(2) G4D_PE_CASE(3) G4D_PE_CASE(4)
        default:
            if (l1m_env) { if (pipe) G4D_PE_LAUNCH(5, true, true); else G4D_PE_LAUNCH(5, false, true); }
            else { if (pipe) G4D_PE_LAUNCH(5, true, false); else G4D_PE_LAUNCH(5, false, false); }
    }
#undef G4D_PE_CASE
#undef G4D_PE_LAUNCH
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
    G4D_REQUIRE(rows < (1ll << 31) - 64 && (long long)frames * n * 32 * 4 < (1ll << 32) && (long long)p * nsample >= 64,
                "g4d_pos_encode_f32: needs rows < 2^31, frames*n*128 B < 4 GB (32-bit gather offsets) and p*nsample >= 64");
    PeArgs a;
    a.n = n; a.p = p; a.S = nsample; a.logS = __builtin_ctz((unsigned)nsample); a.rows = rows;
    a.xyz = xyz; a.new_xyz = new_xyz; a.extra = extra; a.table = table; a.idx = idx;
    a.W1 = W1; a.b1 = b1; a.W2f = W2_frag; a.b2 = b2; a.out = out; a.ldo = ldo; a.col0 = col0;
    const long long nchunks = (rows + 63) / 64;
    const long long want = (nchunks + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);  // persistent waves: weights are loaded once
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool sbig = nsample >= 16;
    if (table) { if (sbig) launch_pe<true, true>(n_extra, dim3(grid), st, a); else launch_pe<true, false>(n_extra, dim3(grid), st, a); }
    else { if (sbig) launch_pe<false, true>(n_extra, dim3(grid), st, a); else launch_pe<false, false>(n_extra, dim3(grid), st, a); }
    return check_launch("g4d_pos_encode_f32");
}
