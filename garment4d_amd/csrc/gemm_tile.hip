// Y = act((X . W^T) * scale + shift) for TALL direct launches with a deep contraction (the wide feature-propagation level of a
// coalesced call: 61440 rows x [576 -> 512 -> 256]; the per-source-point tables of the SA / FP levels) -- pytorch_utils.py:5-32 as one
// GEMM per layer.  Round 4: mlp.hip's linear_kernel tiles 64 rows x 64 channels; every 32-deep k chunk then moves 16 KB through L2
// for 262 kFLOP (the A rows are re-read by every 64-channel column block, W by every row block: 9 TB/s of L2 traffic at the rate the
// matrix pipe could run) and every wave re-reads the whole A tile from LDS (5 ds_read_b128 per 16 MFMAs): 60 % of the fp32 MFMA peak at
// 240 clouds per call.  Here the block tile is 128 x 128 (half the L2 traffic per flop), a wave owns 64 x 64 of it (16 accumulator tiles:
// 8 ds_read_b128 per 64 MFMAs), and workgroups that share a row block run next to each other on ONE XCD (its L2 serves the A rows to all
// column blocks).  Products, k order and epilogue arithmetic are linear_kernel's (the MFMAs run with the operands swapped, which changes
// neither): results are bit-identical.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TM = 128, TN = 128, TK = 32, TLD = TK + 4;   // LDS row stride 36 floats: ds_read_b128 fragment reads spread over all banks
}

// FULLK: K == Kpad (no zero-filled tail columns): the loads carry no predicate at all.  TAB: g4d_linear_interp_add_f32 -- the rows'
// interpolation contexts (index / distance loads, three divisions) are formed BEFORE the contraction and only the table rows themselves are
// fetched in the epilogue (round 5: with K = 192 the epilogue's dependent loads were a sixth of the launch).
template <bool FULLK, bool TAB>
__global__ void __launch_bounds__(256, 2) gemm_tile_kernel(const LinearArgs a, int nrow_blk, int ncol_blk, int cpad) {
    extern __shared__ __attribute__((aligned(16))) float g_smem[];
    float *sA = g_smem;                         // [2][TM * TLD]
    float *sB = g_smem + 2 * TM * TLD;          // [2][TN * TLD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // workgroup -> (row block, column block): blocks are dispatched round-robin over the 8 XCDs; XCD x takes row blocks x, x + 8, ... and
    // runs all column blocks of a row block back to back
    int rb, cb;
    {
        const int b = blockIdx.x, x = b & 7, slot = b >> 3;
        rb = (slot / ncol_blk) * 8 + x;
        cb = slot % ncol_blk;
        if (rb >= nrow_blk) return;
    }
    const int row0 = rb * TM, n0 = cb * TN;
    // staging map: thread -> rows lr + 32 p (p = 0..3), 4 consecutive k at lk
    const int lr = t >> 3, lk = (t & 7) * 4;
    const float *xrow[4], *wrow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        xrow[p] = a.X + (size_t)min(row0 + lr + 32 * p, a.rows - 1) * a.ldx + lk;     // rows past the end: clamped, never stored
        wrow[p] = a.W + (size_t)min(n0 + lr + 32 * p, cpad - 1) * a.Kpad + lk;        // channels past the padded width: clamped, never stored
    }
    // K and ldx are multiples of 4 and X is 16-byte aligned (launcher): a lane's four columns are all inside the row or all past its end.
    // Branch-free on purpose: a load inside a conditional block makes the number of loads in flight unknown at the join and the compiler
    // then waits for the prefetch of the NEXT chunk (vmcnt(0)) in front of this chunk's MFMAs.
    // (round 5: written as `in ? v : 0` the compiler turned the select into a load under an exec mask -- four conditional blocks per chunk,
    //  each followed by s_waitcnt vmcnt(3): the eight prefetches of a chunk went out in three round trips instead of one.  The tail columns
    //  are now cleared with a bit mask, and launches whose K is a whole number of chunks -- every GEMM of the benched path -- have no
    //  predicate.)
    auto load_x = [&](int p, int k) -> f32x4 {
        if constexpr (FULLK) return *reinterpret_cast<const f32x4 *>(xrow[p] + k);
        const bool in = k + lk < a.K;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(xrow[p] + (in ? k : 0));
        const unsigned m = in ? 0xffffffffu : 0u;
        const u32x4 r = {v[0] & m, v[1] & m, v[2] & m, v[3] & m};
        return __builtin_bit_cast(f32x4, r);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    InterpRow ctx[4] = {};
    if constexpr (TAB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ctx[i] = interp_row(a, min(row0 + (wave >> 1) * 64 + i * 16 + (lane & 15), a.rows - 1));
    }
    const int nchunk = a.Kpad / TK;
    f32x4 ra[4], rb4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { ra[p] = load_x(p, 0); rb4[p] = *reinterpret_cast<const f32x4 *>(wrow[p]); }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<f32x4 *>(&sA[(lr + 32 * p) * TLD + lk]) = ra[p];
        *reinterpret_cast<f32x4 *>(&sB[(lr + 32 * p) * TLD + lk]) = rb4[p];
    }
    __syncthreads();
    const int fi = lane & 15, fq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;    // the wave's 64 x 64 quadrant
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nchunk;
        if (more) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { ra[p] = load_x(p, (c + 1) * TK); rb4[p] = *reinterpret_cast<const f32x4 *>(wrow[p] + (c + 1) * TK); }
        }
        const float *cA = sA + cur * TM * TLD + (wr * 64 + fi) * TLD + fq * 4;
        const float *cB = sB + cur * TN * TLD + (wc * 64 + fi) * TLD + fq * 4;
#pragma unroll
        for (int kk = 0; kk < TK; kk += 16) {
            f32x4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { af[i] = *reinterpret_cast<const f32x4 *>(cA + i * 16 * TLD + kk); bf[i] = *reinterpret_cast<const f32x4 *>(cB + i * 16 * TLD + kk); }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // (weights as the A operand: see the epilogue)
        }
        if (more) {
            const int nxt = cur ^ 1;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<f32x4 *>(&sA[nxt * TM * TLD + (lr + 32 * p) * TLD + lk]) = ra[p];
                *reinterpret_cast<f32x4 *>(&sB[nxt * TN * TLD + (lr + 32 * p) * TLD + lk]) = rb4[p];
            }
        }
        __syncthreads();
    }
    // epilogue.  The MFMAs ran transposed (A = weights, B = activations: the same products summed in the same k order, i.e. the same bits as
    // linear_kernel's orientation), so lane (fi, fq) holds channels 4 fq .. 4 fq + 3 of tile j for row fi of tile i: ONE interpolation context
    // per (lane, row tile) instead of one per output row of four, 16-byte table loads and 16-byte stores instead of dword ones.
    const bool vec = (a.ldo & 3) == 0 && (a.col0 & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0;   // launch-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + wr * 64 + i * 16 + fi;
        const InterpRow c = ctx[i];   // + three_interpolate(tab) of the row (g4d_linear_interp_add_f32): as linear_kernel, added to the finished contraction
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch0 = n0 + wc * 64 + j * 16 + fq * 4;
            const int chc = min(ch0, cpad - 4);            // the packed scale / shift are padded to 64 channels
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + chc), sh = *reinterpret_cast<const f32x4 *>(a.shift + chc);
            f32x4 y = acc[i][j];
            if constexpr (TAB) {   // (the launcher guarantees Cout % 128 == 0, tab_ld % 4 == 0 and a 16-byte aligned table: whole 16-byte groups inside the row)
                const f32x4 t0 = *reinterpret_cast<const f32x4 *>(a.tab + c.k0 + ch0), t1 = *reinterpret_cast<const f32x4 *>(a.tab + c.k1 + ch0),
                            t2 = *reinterpret_cast<const f32x4 *>(a.tab + c.k2 + ch0);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = y[r] + (c.w0 * t0[r] + c.w1 * t1[r] + c.w2 * t2[r]);   // interp_at(), element by element
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = __builtin_fmaf(y[r], sc[r], sh[r]);
                if (a.relu) y[r] = fmaxf(y[r], 0.f);
            }
            if (row < a.rows) {
                float *o = a.out + (size_t)row * a.ldo + a.col0 + ch0;
                if (vec && ch0 + 3 < a.Cout) *reinterpret_cast<f32x4 *>(o) = y;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ch0 + r < a.Cout) o[r] = y[r];
                }
            }
        }
    }
}

// Used by launch_linear (mlp.hip) for DIRECT launches without pooling, after gemm_stream_try: true when the launch was taken.
// Measured at 61440 rows (scripts/time_gemm.py; linear_kernel / this kernel / torch.mm = hipBLASLt, TFLOP/s): 576 -> 512: 85 / 91 / 98;
// 512 -> 256: 84 / 88 / 114; 256 -> 256: 82 / 82 / 106; 192 -> 192: 78 / 58 / 97 (a half-empty second column block) -- taken only where it wins.
bool gemm_tile_try(const LinearArgs &a, hipStream_t s, int *rc) {
    const int enabled = (int)tuning("gemm_tile", 1);
    const long long min_rows = tuning("gemm_tile_min_rows", 32768);
    const int min_cout = (int)tuning("gemm_tile_min_cout", 128), min_kpad = (int)tuning("gemm_tile_min_kpad", 128);   // (A/B switches; 256 / 256 until the epilogue stored 16 bytes per lane)
    if (!enabled || a.pool != 0 || a.rows < min_rows || a.Kpad < (a.tab ? 128 : min_kpad) || a.Cout < min_cout || a.Cout % TN != 0 || (a.K & 3) || (a.ldx & 3) || (reinterpret_cast<size_t>(a.X) & 15)) return false;
    if (a.tab && ((a.tab_ld & 3) || (reinterpret_cast<size_t>(a.tab) & 15))) return false;
    const int lds = 2 * (TM + TN) * TLD * (int)sizeof(float);   // 73728 bytes: two workgroups per CU
    typedef void (*Kern)(const LinearArgs, int, int, int);
    static const Kern kerns[4] = {gemm_tile_kernel<false, false>, gemm_tile_kernel<false, true>, gemm_tile_kernel<true, false>, gemm_tile_kernel<true, true>};
    const int which = (a.K == a.Kpad ? 2 : 0) + (a.tab ? 1 : 0);
    static unsigned long long attr[4] = {0, 0, 0, 0};
    *rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kerns[which]), lds, attr[which], "g4d_linear_f32(tile)");
    if (*rc) return true;
    const int cpad = (a.Cout + 63) / 64 * 64;   // the packed weight / scale / shift are padded to 64 channels
    const int nrow = (a.rows + TM - 1) / TM, ncol = (a.Cout + TN - 1) / TN;
    const long long blocks = (long long)((nrow + 7) / 8) * 8 * ncol;   // XCD-major numbering: row blocks rounded up to a multiple of 8
    if (blocks >= (1ll << 31)) return false;
    hipLaunchKernelGGL(kerns[which], dim3((unsigned)blocks), dim3(256), lds, s, a, nrow, ncol, cpad);
    *rc = check_launch("g4d_linear_f32(tile)");
    return true;
}

}  // namespace g4d
