// Fused "aggregate, then the next layer's contraction" for the GCN stacks of the refinement loop
// (/root/reference/modules/mesh_encoder.py:477-481 calling modules/pygcn/layers.py:35-55 four times per round).
//
// A GraphConvolution is  h = act(Ahat (X W) + b).  Chained, the aggregation of layer i and the contraction of layer i+1 are
//     h_i = act(Ahat S_i + b_i)        (SpMM: reads S_i seven times through L2 -- once per neighbour -- writes h_i)
//     S_{i+1} = h_i W_{i+1}            (reads h_i back, writes S_{i+1})
// At 240 frames x 4096 vertices x 128 channels each tensor is 503 MB, and the two launches took 344 + 410 us.  This kernel does
// both for a tile of 128 consecutive vertices of one frame, in the REFERENCE's operation order (aggregate the already contracted
// rows, add the bias, activate, then contract with the next weight), so h_i never exists in HBM unless the caller asks for it
// (`tap`: the third layer's output feeds the next round's attention):
//   * Mesh numberings are local: the neighbours of 128 consecutive vertices lie in a WINDOW of a few hundred consecutive
//     vertices (64 x 64 quad cylinder: 128 + 2 * 65).  The block finds the window from the CSR column indices of its rows and
//     stages it in LDS, 32 channels at a time -- S_i is read ~2x instead of 7x.  A tile whose window exceeds kWin rows (an
//     arbitrary numbering) gathers from global memory instead: same result, the speed of the old SpMM.
//   * aggregation: thread = (row, 4 channels), fmaf over the row's CSR entries in CSR order, + bias, ReLU -- the exact
//     arithmetic of spmm_rows_kernel, so h is bit-identical to the two-launch route; the 128 x 32 slice goes to LDS in fp32.
//   * contraction: v_mfma_f32_16x16x4_f32, wave w owns output channels [32 w, 32 w + 32) x all 128 rows (16 accumulator
//     tiles), A fragments from the LDS slice (ds_read_b128, 4 consecutive k per lane feeding 4 MFMAs), B fragments of the next
//     weight pre-packed in fragment order (one 16-byte load per lane per 16 k) and read once per block.
//     For a narrow next layer (Cout <= 16: the 128 -> 3 regressor) wave w owns rows [32 w, 32 w + 32) x one channel tile.
// LDS: window slice 288 x 32 floats + A slice 128 x 36 floats + 128 x 8 padded CSR pairs = 63 KB -> two blocks per CU.
// Roofline: fp32 MFMA (2 * rows * 128 * Cout flop; 205 us at 983k rows, Cout = 128) over HBM (rows * 128 * 4 B read + rows * Cout * 4 B
// written [+ the tap]).  Measured phase split per tile (in-kernel stamps, scripts/dbg_gcn_phases.py, -DG4D_GCN_DEBUG): of ~83k cycles
// only 19k are MFMA issue; 20k aggregation (LDS chains), 6k staging, 6k load issue, ~25k tile prologue / epilogue (dependent global
// loads for the window bounds and the padded rows, output store) -- with two 4-wave blocks per CU these latencies are exposed.
// What did NOT matter: the block mix per CU (3 blocks of 64 rows: same), a start stagger between co-resident blocks, LDS bank
// conflicts of the window reads (stride 36 -> 32), the order of the loads against the in-order vmcnt.  Next: per-tile metadata
// precomputed once per mesh (window bounds, padded rows) and a persistent block that prefetches the next TILE.
#include <cstdlib>

#include "g4d_common.h"

namespace g4d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Tile geometry.  TILE = rows (vertices) per block; WIN = window rows staged in LDS.
//   TILE 128 / WIN 288: 63 KB of LDS, two blocks per CU, window read amplification ~2x on a 64-wide grid mesh
//   TILE  64 / WIN 224: 42 KB of LDS, three blocks per CU, amplification ~3x
template <int TILE> struct Geo;
template <> struct Geo<128> { static constexpr int kWin = 288; };
template <> struct Geo<64> { static constexpr int kWin = 224; };
constexpr int kEll = 8;      // CSR rows of a tile are kept in LDS padded to this many (column, weight) pairs (zero weights); longer rows: slow path
constexpr int kLd = 36;      // A-slice row stride in floats (32 channels + 4: the MFMA fragment reads -- 16 rows, same k -- spread over the banks)
constexpr int kLdW = 32;     // window row stride: the aggregation reads 8 lanes x 16 B of DIFFERENT rows per 16-lane LDS group ({row a: ch 0-15,
                             // row b: 16-31, row c: 16-31, row d: 0-15}); neighbour columns of consecutive rows are consecutive, so a 32-float
                             // stride alternates the bank halves and the group is conflict-free (stride 36: two-way conflicts, measured)
constexpr int kC = 128;      // support width (hidden_dim of the refinement GCNs)

#ifdef G4D_GCN_DEBUG
__device__ long long g_gcn_dbg[8 * 4096];  // per block (first 4096), wave 0: cycles per phase (scripts/dbg_gcn_phases.py)
#define G4D_GSTAMP(i) { if (threadIdx.x == 0 && blockIdx.x < 4096) { const long long now_ = (long long)__builtin_readcyclecounter(); g_gcn_dbg[blockIdx.x * 8 + (i)] += now_ - dbg_last; dbg_last = now_; } }
#else
#define G4D_GSTAMP(i)
#endif

struct GcnFusedArgs {
    int vg, frames, tpf;     // tpf = tiles per frame
    const float *S;          // (frames, vg, 128)
    const int *rowptr, *colidx;
    const float *vals, *bias;
    int relu;
    float *tap;              // (frames, vg, 128) or null: h itself
    const float *Wp;         // next weight (Cout_pad x 128) in the fragment order of the LDS-resident MLP kernels: [16-channel tile][k-step of 16 (8)][lane = fq*16+fi][4 consecutive k]
    int cout;                // real output channels (<= 16 * NT)
    float *out;              // (frames, vg, cout)
    const int *meta;         // per-tile metadata built once per mesh by g4d_gcn_tile_meta_build (NULL: every workgroup derives its own)
};

// Per-tile metadata (round 5): the window and the padded (column offset, weight) rows of a tile depend on the MESH only -- the same for all
// frames of a launch, all four layers of a regressor, all three refinement rounds -- but every workgroup used to derive them from the CSR
// arrays through three dependent round trips (row pointers -> column range -> atomics -> entries): ~25k of a tile's ~83k cycles.
// Layout per tile, in ints: [0] lo, [1] hi, [2] longest row, [3] fast (window and valence fit), then TILE * kEll (offset, weight-bits) pairs.
template <int TILE> constexpr int gcn_meta_ints() { return 4 + TILE * 8 * 2; }

// NT = channel tiles of the next layer: 8 (Cout = 128: wave owns 2 channel tiles x 8 row tiles) or 1 (Cout <= 16: wave owns
// 2 row tiles x the one channel tile).
// FAST: the tile's window fits kWin rows and no row has more than kEll entries -> window slices and the padded (column, weight) rows live in LDS, the
// next slice's window rows are prefetched into registers while this slice is aggregated and contracted.  !FAST: any numbering, any
// valence -- neighbour rows and CSR entries straight from global memory (the old SpMM's access pattern).
template <int NT, int TILE, bool FAST>
__device__ __forceinline__ void gcn_fused_tile(const GcnFusedArgs &a, int f, int r0, float *win, float *asl, int2 *ent, int lo, int wrows, int e0, int ell) {
#ifdef G4D_GCN_DEBUG
    long long dbg_last = (long long)__builtin_readcyclecounter();
#endif
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fi = lane & 15, fq = lane >> 4;
    constexpr int kTile = TILE, kWin = Geo<TILE>::kWin, NP = TILE / 32;   // NP = rows per thread in the aggregation
    const int nrows = min(kTile, a.vg - r0);
    const float *S = a.S + (size_t)f * a.vg * kC;
    constexpr int NPRE = kWin / 32;   // window rows per thread and slice

    // aggregation map: thread -> (row ar + 32 p, channels ac .. ac + 3 of the slice)
    const int ar = t >> 3, ac = (t & 7) * 4;
    int beg[NP], len[NP], maxlen = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = ar + 32 * p;
        const bool ok = r < nrows;
        const int b = ok ? a.rowptr[r0 + r] : 0, e = ok ? a.rowptr[r0 + r + 1] : 0;
        beg[p] = b;
        len[p] = e - b;
        maxlen = max(maxlen, len[p]);
    }
    if constexpr (FAST) {
        if (a.meta) {   // the pairs were built once per mesh: one coalesced copy, no dependent loads
            const int2 *src = reinterpret_cast<const int2 *>(a.meta + (size_t)(r0 / kTile) * gcn_meta_ints<TILE>() + 4);
#pragma unroll
            for (int k = 0; k < kTile * kEll / 256; ++k) ent[t + 256 * k] = src[t + 256 * k];
        } else
        // the tile's CSR rows, padded to `ell` (column offset into `win`, weight) pairs each: the aggregation loop then has a
        // block-uniform trip count and no per-entry predicate.  A padded pair is (the ZERO row behind the window, weight 0): it adds
        // 0 * 0 whatever the activations hold -- pointing it at a real neighbour row would turn an inf / NaN there into NaN for rows
        // that do not reference it (an isolated vertex must come out as bias only, as g4d_spmm_rows_f32 gives).  Thread t fills
        // pairs t, t + 256, ...
        for (int k = t; k < kTile * kEll; k += 256) {
            const int r = k / kEll, j = k - r * kEll;
            int off = kWin * kLdW;
            float w = 0.f;
            if (r < nrows) {
                const int b = a.rowptr[r0 + r], n = a.rowptr[r0 + r + 1] - b;
                if (j < n) {
                    off = (a.colidx[b + j] - lo) * kLdW;
                    w = a.vals[b + j];
                }
            }
            ent[k] = make_int2(off, __float_as_int(w));
        }
        maxlen = ell;
    } else {
        // the longest row of the WAVE bounds the entry loop: a wave-uniform trip count (a per-lane one makes the loop divergent)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
        maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    }
    f32x4 pre[NPRE];
    auto prefetch = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int r = ar + 32 * i;
            // unconditional (rows past the window: its last row again, never stored): a load under a condition makes the number of loads in
            // flight unknown where the branches join, and the wait for this slice's B fragments then drains the prefetch too
            pre[i] = *reinterpret_cast<const f32x4 *>(S + (size_t)(lo + min(r, wrows - 1)) * kC + ks * 32 + ac);
        }
    };
    if constexpr (FAST) prefetch(0);

    constexpr int MT = NT == 8 ? TILE / 16 : (TILE / 64 > 0 ? TILE / 64 : 1);   // row tiles per wave
    constexpr int NW = NT == 8 ? 2 : 1;   // channel tiles per wave
    f32x4 acc[MT][NW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    G4D_GSTAMP(0)   // tile prologue: row pointers, CSR entries to LDS, first prefetch issued
    for (int ks = 0; ks < 4; ++ks) {
        const int c0 = ks * 32;
        // 1. the window slice (rows lo .. lo + wrows, channels c0 .. c0 + 31) goes from the prefetch registers to LDS
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int r = ar + 32 * i;
                if (r < wrows) *reinterpret_cast<f32x4 *>(&win[r * kLdW + ac]) = pre[i];
            }
        }
        lds_barrier();   // window (and, first time, the CSR entries) visible; the previous slice's MFMAs are done with `asl`
        G4D_GSTAMP(1)   // prefetched rows arrived + stored + barrier
        // Issue ORDER matters: vector-memory loads retire in order (s_waitcnt vmcnt counts from the oldest), so whatever this slice
        // itself waits for -- the B fragments, the bias -- must be requested BEFORE the next slice's window rows; behind them, the
        // first use of `bv` would wait for the whole HBM round trip of the prefetch (measured: 6k cycles per slice in the aggregation
        // phase, and the MFMAs started only after the prefetch had landed).
        f32x4 bf[2][NW];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int nt = NT == 8 ? wave * 2 + n : 0;
                bf[h2][n] = *reinterpret_cast<const f32x4 *>(a.Wp + ((size_t)(nt * 8 + ks * 2 + h2) * 64 + lane) * 4);
            }
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + c0 + ac) : (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FAST) {
            if (ks < 3) prefetch(ks + 1);   // in flight during the aggregation and the MFMAs of this slice
        }
        __builtin_amdgcn_sched_barrier(0);
        G4D_GSTAMP(2)   // next prefetch + B fragments issued
        f32x4 h[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) h[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (FAST) {
            // padded rows in LDS: per (row, entry) one 8-byte read, one address add, one 16-byte read, two packed FMAs; the pair of
            // entry j + 1 is requested before entry j's row is read
            int2 cvn[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) cvn[p] = ent[(ar + 32 * p) * kEll];
            for (int j = 0; j < maxlen; ++j) {
                int2 cv[NP];
                f32x4 x[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    cv[p] = cvn[p];
                    x[p] = *reinterpret_cast<const f32x4 *>(&win[cv[p].x + ac]);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) cvn[p] = ent[(ar + 32 * p) * kEll + min(j + 1, kEll - 1)];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float w = __int_as_float(cv[p].y);
                    h[p].x = __builtin_fmaf(w, x[p].x, h[p].x); h[p].y = __builtin_fmaf(w, x[p].y, h[p].y);
                    h[p].z = __builtin_fmaf(w, x[p].z, h[p].z); h[p].w = __builtin_fmaf(w, x[p].w, h[p].w);
                }
            }
        } else {
            // any numbering / valence: entries and neighbour rows straight from global memory, loads unconditional (a lane past the
            // end of its row re-reads its first entry and discards the product) so that the NP rows of a thread advance together
            for (int j = 0; j < maxlen; ++j) {
                float w[NP];
                f32x4 x[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int ee = len[p] > 0 ? beg[p] + (j < len[p] ? j : 0) : e0;   // e0: any valid entry
                    w[p] = a.vals[ee];
                    x[p] = *reinterpret_cast<const f32x4 *>(S + (size_t)a.colidx[ee] * kC + c0 + ac);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x4 n;
                    n.x = __builtin_fmaf(w[p], x[p].x, h[p].x); n.y = __builtin_fmaf(w[p], x[p].y, h[p].y);
                    n.z = __builtin_fmaf(w[p], x[p].z, h[p].z); n.w = __builtin_fmaf(w[p], x[p].w, h[p].w);
                    h[p] = j < len[p] ? n : h[p];
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = ar + 32 * p;
            f32x4 y = h[p] + bv;
            if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
            if (r >= nrows) y = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(&asl[r * kLd + ac]) = y;
            if (a.tap && r < nrows) *reinterpret_cast<f32x4 *>(a.tap + ((size_t)f * a.vg + r0 + r) * kC + c0 + ac) = y;
        }
        G4D_GSTAMP(3)   // aggregation loop + stores
        lds_barrier();   // A slice visible; everybody is done reading `win`
        G4D_GSTAMP(4)   // barrier
        // 3. contract the slice with the next weight
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int mt = NT == 8 ? m : wave * MT + m;
                const f32x4 af = *reinterpret_cast<const f32x4 *>(&asl[(mt * 16 + fi) * kLd + h2 * 16 + fq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[h2][n][e], acc[m][n], 0, 0, 0);
            }
        }
        G4D_GSTAMP(5)   // MFMAs issued
    }
    // store S_next: C/D layout of the 16x16 MFMA -- column = lane & 15, rows = (lane >> 4) * 4 + reg
    float *out = a.out + ((size_t)f * a.vg + r0) * a.cout;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int mt = NT == 8 ? m : wave * MT + m;
            const int ch = (NT == 8 ? wave * 2 + n : 0) * 16 + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + fq * 4 + r;
                if (row < nrows && ch < a.cout) out[(size_t)row * a.cout + ch] = acc[m][n][r];
            }
        }
}

template <int NT, int TILE>
__global__ void __launch_bounds__(256, TILE == 64 ? 3 : 2) gcn_fused_kernel(const GcnFusedArgs a) {
    constexpr int kTile = TILE, kWin = Geo<TILE>::kWin;
    __shared__ __attribute__((aligned(16))) float win[(kWin + 1) * kLdW];   // + one row of zeros: the target of padded CSR pairs
    __shared__ __attribute__((aligned(16))) float asl[kTile * kLd];
    __shared__ int2 ent[kTile * kEll];
    __shared__ int s_lohi[3];   // window [lo, hi], longest row
    const int t = threadIdx.x, lane = t & 63;
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD has its own L2, and the windows
    // of neighbouring tiles overlap (the halo rows).  XCD x therefore walks a CONTIGUOUS range of the (frame, tile) list: the
    // halo a tile shares with its predecessor is an L2 hit instead of a second HBM read.
    const int total = a.tpf * a.frames;
    const int per_xcd = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= total) return;   // block-uniform
    const int f = logical / a.tpf;
    const int r0 = (logical - f * a.tpf) * kTile;
    const int nrows = min(kTile, a.vg - r0);
    if (t < kLdW) win[kWin * kLdW + t] = 0.f;
    if (a.meta) {   // per-mesh metadata: window, longest row and fast-path flag of the tile, read (uniformly) instead of derived
        const int *hd = a.meta + (size_t)(r0 / kTile) * gcn_meta_ints<TILE>();
        const int lo = hd[0], hi = hd[1], ell = hd[2], fast = hd[3];
        lds_barrier();   // (the zero row)
        if (fast) gcn_fused_tile<NT, TILE, true>(a, f, r0, win, asl, ent, lo, hi - lo, 0, ell);
        else gcn_fused_tile<NT, TILE, false>(a, f, r0, win, asl, ent, 0, 0, a.rowptr[r0], ell);
        return;
    }
    // window of the tile: [lo, hi) over the column indices of its rows (a contiguous CSR range)
    if (t == 0) { s_lohi[0] = 0x7fffffff; s_lohi[1] = -1; s_lohi[2] = 0; }
    lds_barrier();
    const int e0 = a.rowptr[r0], e1 = a.rowptr[r0 + nrows];
    {
        int lo = 0x7fffffff, hi = -1;
        for (int e = e0 + t; e < e1; e += 256) {
            const int c = a.colidx[e];
            lo = min(lo, c);
            hi = max(hi, c);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if (lane == 0 && hi >= 0) { atomicMin(&s_lohi[0], lo); atomicMax(&s_lohi[1], hi); }
        int rl = t < nrows ? a.rowptr[r0 + t + 1] - a.rowptr[r0 + t] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rl = max(rl, __shfl_xor(rl, o));
        if (lane == 0) atomicMax(&s_lohi[2], rl);
    }
    lds_barrier();
    const int lo = s_lohi[0], hi = s_lohi[1] + 1;
    const int ell = s_lohi[2];
    if (ell <= kEll && hi - lo <= kWin && hi > lo)   // block-uniform
        gcn_fused_tile<NT, TILE, true>(a, f, r0, win, asl, ent, lo, hi - lo, e0, ell);
    else
        gcn_fused_tile<NT, TILE, false>(a, f, r0, win, asl, ent, 0, 0, e0, ell);
}

// One workgroup per tile of the mesh: what gcn_fused_kernel's prologue derives, written once (layout: gcn_meta_ints above).
template <int TILE>
__global__ void __launch_bounds__(256) gcn_meta_kernel(int vg, const int *rowptr, const int *colidx, const float *vals, int *meta) {
    constexpr int kTile = TILE, kWin = Geo<TILE>::kWin;
    __shared__ int s_lohi[3];
    const int t = threadIdx.x, lane = t & 63;
    const int r0 = (int)blockIdx.x * kTile;
    const int nrows = min(kTile, vg - r0);
    if (t == 0) { s_lohi[0] = 0x7fffffff; s_lohi[1] = -1; s_lohi[2] = 0; }
    __syncthreads();
    const int e0 = rowptr[r0], e1 = rowptr[r0 + nrows];
    int lo = 0x7fffffff, hi = -1;
    for (int e = e0 + t; e < e1; e += 256) {
        const int c = colidx[e];
        lo = min(lo, c);
        hi = max(hi, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if (lane == 0 && hi >= 0) { atomicMin(&s_lohi[0], lo); atomicMax(&s_lohi[1], hi); }
    int rl = t < nrows ? rowptr[r0 + t + 1] - rowptr[r0 + t] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rl = max(rl, __shfl_xor(rl, o));
    if (lane == 0) atomicMax(&s_lohi[2], rl);
    __syncthreads();
    const int wlo = s_lohi[0], whi = s_lohi[1] + 1, ell = s_lohi[2];
    const bool fast = ell <= kEll && whi - wlo <= kWin && whi > wlo;
    int *hd = meta + (size_t)blockIdx.x * gcn_meta_ints<TILE>();
    if (t == 0) { hd[0] = wlo; hd[1] = whi; hd[2] = ell; hd[3] = fast ? 1 : 0; }
    int2 *pairs = reinterpret_cast<int2 *>(hd + 4);
    for (int k = t; k < kTile * kEll; k += 256) {   // (the pairs of gcn_fused_tile<FAST>: same rule, same padding)
        const int r = k / kEll, j = k - r * kEll;
        int off = kWin * kLdW;
        float w = 0.f;
        if (fast && r < nrows) {
            const int b = rowptr[r0 + r], n = rowptr[r0 + r + 1] - b;
            if (j < n) {
                off = (colidx[b + j] - wlo) * kLdW;
                w = vals[b + j];
            }
        }
        pairs[k] = make_int2(off, __float_as_int(w));
    }
}

}  // namespace g4d

using namespace g4d;

static int gcn_tile_rows() {
    static const int tile = [] { const char *e = getenv("G4D_GCN_TILE"); return e && atoi(e) == 64 ? 64 : 128; }();   // measured: 2.12 ms (128) vs 2.20 ms (64) per 4-layer stack at 240 x 4096 rows
    return tile;
}

extern "C" long long g4d_gcn_tile_meta_bytes(int vg) {
    const int tile = gcn_tile_rows();
    const long long tiles = (vg + tile - 1) / tile;
    return tiles * (tile == 128 ? gcn_meta_ints<128>() : gcn_meta_ints<64>()) * (long long)sizeof(int);
}

extern "C" int g4d_gcn_tile_meta_build(int vg, const int *rowptr, const int *colidx, const float *vals, void *meta, g4d_stream_t stream) {
    G4D_REQUIRE(vg >= 0, "g4d_gcn_tile_meta_build: bad size");
    if (vg == 0) return G4D_OK;
    G4D_REQUIRE(rowptr && colidx && vals && meta, "g4d_gcn_tile_meta_build: null pointer");
    const int tile = gcn_tile_rows();
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (tile == 128) hipLaunchKernelGGL(gcn_meta_kernel<128>, dim3((vg + 127) / 128), dim3(256), 0, st, vg, rowptr, colidx, vals, static_cast<int *>(meta));
    else hipLaunchKernelGGL(gcn_meta_kernel<64>, dim3((vg + 63) / 64), dim3(256), 0, st, vg, rowptr, colidx, vals, static_cast<int *>(meta));
    return check_launch("g4d_gcn_tile_meta_build");
}

static int gcn_agg_linear(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx, const float *vals, const float *bias, int relu,
                          float *tap, const float *Wp, int cout, float *out, const void *meta, g4d_stream_t stream);

extern "C" int g4d_gcn_agg_linear_f32(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx, const float *vals,
                                      const float *bias, int relu, float *tap, const float *Wp, int cout, float *out,
                                      g4d_stream_t stream) {
    return gcn_agg_linear(frames, vg, c, S, rowptr, colidx, vals, bias, relu, tap, Wp, cout, out, nullptr, stream);
}

extern "C" int g4d_gcn_agg_linear_meta_f32(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx, const float *vals,
                                           const float *bias, int relu, float *tap, const float *Wp, int cout, float *out, const void *meta,
                                           g4d_stream_t stream) {
    return gcn_agg_linear(frames, vg, c, S, rowptr, colidx, vals, bias, relu, tap, Wp, cout, out, meta, stream);
}

static int gcn_agg_linear(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx, const float *vals, const float *bias, int relu,
                          float *tap, const float *Wp, int cout, float *out, const void *meta, g4d_stream_t stream) {
    G4D_REQUIRE(frames >= 0 && vg >= 0, "g4d_gcn_agg_linear_f32: bad sizes");
    G4D_REQUIRE(c == kC, "g4d_gcn_agg_linear_f32: the support width must be %d (got %d)", kC, c);
    G4D_REQUIRE(cout == 128 || (cout >= 1 && cout <= 16), "g4d_gcn_agg_linear_f32: Cout must be 128 or <= 16 (got %d)", cout);
    if (frames == 0 || vg == 0) return G4D_OK;
    G4D_REQUIRE(S && rowptr && colidx && vals && Wp && out, "g4d_gcn_agg_linear_f32: null pointer");
    const int tile = gcn_tile_rows();
    const int tpf = (vg + tile - 1) / tile;
    G4D_REQUIRE((long long)tpf * frames < (1ll << 30), "g4d_gcn_agg_linear_f32: too many tiles");
    GcnFusedArgs a = {vg, frames, tpf, S, rowptr, colidx, vals, bias, relu, tap, Wp, cout, out, static_cast<const int *>(meta)};
    const int per_xcd = (tpf * frames + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (tile == 128) {
        if (cout == 128) hipLaunchKernelGGL((gcn_fused_kernel<8, 128>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gcn_fused_kernel<1, 128>), grid, dim3(256), 0, st, a);
    } else {
        if (cout == 128) hipLaunchKernelGGL((gcn_fused_kernel<8, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gcn_fused_kernel<1, 64>), grid, dim3(256), 0, st, a);
    }
    return check_launch("g4d_gcn_agg_linear_f32");
}

#ifdef G4D_GCN_DEBUG
extern "C" int g4d_gcn_debug_read(long long *host_out, int reset) {
    const int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g4d::g_gcn_dbg), sizeof(long long) * 8 * 4096);
    if (reset) {
        static long long zeros[8 * 4096];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g4d::g_gcn_dbg), zeros, sizeof(zeros));
    }
    return rc;
}
#endif
