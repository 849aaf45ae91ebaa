// Shared pieces of the MFMA shared-MLP kernels (mlp.hip: one layer per launch; mlp_stack.hip: up to four layers
// per launch with activations resident in LDS): argument block, per-row loader state and the element loader of
// the four A-operand sources (DIRECT / GROUP / INTERP / CSR).
#pragma once
#include "g4d_common.h"

namespace g4d {

enum { LOAD_DIRECT = 0, LOAD_GROUP = 1, LOAD_INTERP = 2, LOAD_CSR = 3 };

struct LinearArgs {
    // contraction
    int rows, K, Kpad, Cout;
    const float *W;      // [CoutPad64][Kpad] row-major, zero padded (packed once on the host side)
    const float *scale;  // [CoutPad64]
    const float *shift;  // [CoutPad64]
    int relu;
    // output
    float *out;
    int ldo, col0;
    int pool;  // 0 none, 1 max, 2 avg  over S consecutive rows
    int S;
    // DIRECT / CSR source
    const float *X;
    int ldx;
    // GROUP
    const float *xyz, *new_xyz, *feats;
    const int *idx;
    int N, P, C, use_xyz;
    // INTERP
    const float *known_feats, *skip, *dist2;
    const int *nn_idx;
    int C2, C1, m, n;
    // CSR
    const int *rowptr, *colidx;
    const float *vals;
    int Vg;
    // INTERP over a pre-contracted table (register-chain kernel only): the loader's row becomes relu(row * pre_scale + pre_shift)
    // and, when in_tap is set, is also written to in_tap[row * in_tap_ld + column]
    const float *pre_scale, *pre_shift;
    float *in_tap;
    int in_tap_ld;
    // INTERP WITH skip features (register-chain kernel only, C2 == 0 in these arguments): `tab` (stride tab_ld) holds the known features
    // already multiplied by the known-feature columns of the first layer's weight; the first layer's accumulators START from
    // three_interpolate(tab) and the matrix pipe adds the skip columns (K = C1).
    // GROUP over a pre-contracted table (register-chain kernel only): row j of `tab` (stride tab_ld) holds the feature part of the
    // first layer for source point j; the loader's row is relu((tab[j] + tab_wx . (x_j - q)) * pre_scale + pre_shift), tab_wx = [3][K]
    const float *tab, *tab_wx;
    int tab_ld;
    // INTERP over CELL-ORDERED rows (register-chain kernel only): row p of the launch is the p-th point of its cloud's ball-grid order
    // (dist2 / nn_idx are stored in that order); its skip features are read from, and its outputs written to, row perm(p) = the
    // original index kept in the 4th dword of the cloud's 16-byte grid records (perm_rec + cloud * perm_stride bytes).  NULL: rows in place.
    const unsigned char *perm_rec;
    size_t perm_stride;
};

// output / skip row of launch row `row` (see LinearArgs::perm_rec)
__device__ __forceinline__ int out_row(const LinearArgs &a, int row) {
    if (!a.perm_rec) return row;
    const int b = row / a.n, p = row - b * a.n;
    return b * a.n + reinterpret_cast<const int *>(a.perm_rec + (size_t)b * a.perm_stride)[4 * p + 3];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// DIRECT launches with `tab` set (g4d_linear_interp_add_f32): the GEMM result of a row gets three_interpolate(tab) of that row added before the
// affine -- the known-feature part of a feature-propagation level's first layer, pre-contracted over the m known rows (conv(sum_i w_i f_i) =
// sum_i w_i conv(f_i); pointnet2_modules.py:139-149).  Weights as make_ctx<LOAD_INTERP> computes them; offsets in floats into `tab`.
struct InterpRow { float w0, w1, w2; size_t k0, k1, k2; };
__device__ __forceinline__ InterpRow interp_row(const LinearArgs &a, int row) {
    InterpRow c;
    const int b = row / a.n;
    const int *ix = a.nn_idx + (size_t)row * 3;
    const float *d2 = a.dist2 + (size_t)row * 3;
    const float r0 = 1.0f / (__fsqrt_rn(d2[0]) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(d2[1]) + 1e-8f), r2 = 1.0f / (__fsqrt_rn(d2[2]) + 1e-8f);
    const float norm = (r0 + r1) + r2;
    c.w0 = r0 / norm; c.w1 = r1 / norm; c.w2 = r2 / norm;
    c.k0 = ((size_t)b * a.m + ix[0]) * a.tab_ld; c.k1 = ((size_t)b * a.m + ix[1]) * a.tab_ld; c.k2 = ((size_t)b * a.m + ix[2]) * a.tab_ld;
    return c;
}
__device__ __forceinline__ float interp_at(const LinearArgs &a, const InterpRow &c, int ch) {
    return c.w0 * a.tab[c.k0 + ch] + c.w1 * a.tab[c.k1 + ch] + c.w2 * a.tab[c.k2 + ch];
}

// gemm_stream.hip: persistent row-streaming GEMM for tall DIRECT launches; returns false when the launch is not its kind
bool gemm_stream_try(const LinearArgs &a, hipStream_t s, int *rc);
bool gemm_narrow_try(const LinearArgs &a, hipStream_t s, int *rc);   // gemm_narrow.hip: K = Cout = 96, weights in registers
// gemm_tile.hip: 128 x 128-tile GEMM for tall DIRECT launches with a deep contraction; same contract
bool gemm_tile_try(const LinearArgs &a, hipStream_t s, int *rc);

// fp_table.hip: persistent, software-pipelined form of g4d_mlp_chain_table(_cells)_f32 for the 128 -> 64 -> 32 -> <= 16 stack; -1 = not its kind
int fp_table_try(long long rows, int n, int m, int C2, const float *table, const float *dist2, const int *nn_idx, const void *perm_rec,
                 size_t perm_stride, const float *pre_scale, const float *pre_shift, float *in_tap, int nlayers, const float *const *W,
                 const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout, const int *relu, float *out, int ldo,
                 int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st);

// fp_init.hip: persistent form of g4d_mlp_chain_interp_init_f32 for the skip 96 -> 256 -> 128 -> 128 stack, weights shared through LDS; -1 = not its kind
int fp_init_try(long long rows, int n, int m, int C1, const float *skip, const float *table, int tab_ld, const float *dist2, const int *nn_idx, int nlayers,
                const float *const *W, const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout, const int *relu, float *out,
                int ldo, int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st);

// fp_head_bf16.hip: persistent form of g4d_mlp_chain_bf16 (interpolating mode) for the 128 -> 128 -> 64 -> 32 -> <= 16 stack of config 3; -1 = not its kind
int fp_head_bf16_try(long long rows, int n, int m, int C2, int C1, const float *known_feats, const float *dist2, const int *nn_idx, int nlayers,
                     const unsigned short *const *W, const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout,
                     const int *relu, int pool, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st,
                     const void *perm_rec = nullptr, size_t perm_stride = 0);   // perm_rec: walk the rows in the cell order of the unknown cloud's grid records

// sa_group_bf16.hip: persistent form of g4d_mlp_chain_bf16 (grouping mode) for the encoder's three-layer SA stacks
int sa_group_bf16_try(long long rows, int N, int P, int S, int C, int use_xyz, const float *xyz, const float *new_xyz, const float *feats,
                      const int *idx, int nlayers, const unsigned short *const *W, const float *const *scale, const float *const *shift,
                      const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0, float *tap_out, hipStream_t st);

// sa_table.hip: persistent, software-pipelined form of g4d_mlp_chain_group_table_f32 for large launches (same arguments); -1 = not its kind
int sa_table_try(long long rows, int N, int P, int S, const float *xyz, const float *new_xyz, const int *idx, const float *table, int tab_ld, int Kt,
                 const float *tab_wx, const float *pre_scale, const float *pre_shift, int nlayers, const float *const *W, const float *const *scale,
                 const float *const *shift, const int *Kpad, const int *Cout, const int *relu, int pool, float *out, int ldo, int col0, hipStream_t st,
                 void *ws = nullptr, long long ws_bytes = 0);   // ws: scratch for the work list that skips blocks of ball-query padding (g4d_sa_table_ws_bytes)

template <int MODE>
struct RowCtx {  // per-thread, per-row state reused across K chunks
    bool valid;
    // GROUP
    size_t pt_base;   // (b*N + j)
    float cx, cy, cz;
    // INTERP
    size_t k0, k1, k2, sk;
    float w0, w1, w2;
    // CSR
    int f, beg, end;
};

template <int MODE>
__device__ __forceinline__ RowCtx<MODE> make_ctx(const LinearArgs &a, int row) {
    RowCtx<MODE> c;
    c.valid = row < a.rows;
    if (!c.valid) return c;
    if constexpr (MODE == LOAD_GROUP) {
        const int q = row / a.S;  // (b*P + p)
        const int b = q / a.P;
        const int j = a.idx[row];
        c.pt_base = (size_t)b * a.N + j;
        const float *ctr = a.new_xyz + (size_t)q * 3;
        c.cx = ctr[0]; c.cy = ctr[1]; c.cz = ctr[2];
    } else if constexpr (MODE == LOAD_INTERP) {
        const int b = row / a.n;
        const int *ix = a.nn_idx + (size_t)row * 3;
        const float *d2 = a.dist2 + (size_t)row * 3;
        // pointnet2_utils.py:98 sqrt; pointnet2_modules.py:140-142 inverse-distance weights
        const float r0 = 1.0f / (__fsqrt_rn(d2[0]) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(d2[1]) + 1e-8f),
                    r2 = 1.0f / (__fsqrt_rn(d2[2]) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        c.w0 = r0 / norm; c.w1 = r1 / norm; c.w2 = r2 / norm;
        const size_t ks = (a.tab && a.C2 == 0) ? (size_t)a.tab_ld : (size_t)a.C2;   // register-chain kernel with an accumulator-init table: offsets into it
        c.k0 = ((size_t)b * a.m + ix[0]) * ks;
        c.k1 = ((size_t)b * a.m + ix[1]) * ks;
        c.k2 = ((size_t)b * a.m + ix[2]) * ks;
        c.sk = (size_t)(a.C1 ? out_row(a, row) : row) * a.C1;
    } else if constexpr (MODE == LOAD_CSR) {
        c.f = row / a.Vg;
        const int v = row - c.f * a.Vg;
        c.beg = a.rowptr[v];
        c.end = a.rowptr[v + 1];
    }
    return c;
}

template <int MODE>
__device__ __forceinline__ float load_elem(const LinearArgs &a, const RowCtx<MODE> &c, int row, int k) {
    if (!c.valid || k >= a.K) return 0.f;
    if constexpr (MODE == LOAD_DIRECT) {
        return a.X[(size_t)row * a.ldx + k];
    } else if constexpr (MODE == LOAD_GROUP) {
        if (a.use_xyz) {
            if (k < 3) {
                const float g = a.xyz[c.pt_base * 3 + k];
                return g - (k == 0 ? c.cx : (k == 1 ? c.cy : c.cz));
            }
            return a.feats[c.pt_base * a.C + (k - 3)];
        }
        return a.feats[c.pt_base * a.C + k];
    } else if constexpr (MODE == LOAD_INTERP) {
        if (k < a.C2) return c.w0 * a.known_feats[c.k0 + k] + c.w1 * a.known_feats[c.k1 + k] + c.w2 * a.known_feats[c.k2 + k];
        return a.skip[c.sk + (k - a.C2)];
    } else {
        float s = 0.f;
        for (int e = c.beg; e < c.end; ++e) s += a.vals[e] * a.X[((size_t)c.f * a.Vg + a.colidx[e]) * a.ldx + k];
        return s;
    }
}

}  // namespace g4d
