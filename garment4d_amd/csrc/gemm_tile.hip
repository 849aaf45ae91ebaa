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
#include <type_traits>

#include "mlp_common.h"

namespace g4d {

#ifdef G4D_GEMM_DEBUG
__device__ long long g_gemm_dbg[8 * 1024];   // per workgroup (first 1024), wave 0: cycles per phase summed over its chunks (scripts/dbg_gemm_phases.py)
#define G4D_MSTAMP(i) { if (threadIdx.x == 0 && blockIdx.x < 1024) { const long long now_ = (long long)__builtin_readcyclecounter(); g_gemm_dbg[blockIdx.x * 8 + (i)] += now_ - dbg_last; dbg_last = now_; } }
#else
#define G4D_MSTAMP(i)
#endif

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef G4D_TK
#define G4D_TK 32
#endif
constexpr int TM = 128, TN = 128, TK = G4D_TK, TLD = TK + 8;   // LDS row stride 40 floats.  ds_read_b128 is served in four 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... over 64 banks (MI355X_MICROARCH.md): a group mixes k-offsets 0 and 4 of different rows, and at stride 36 (round 4) seven of its sixteen 16-byte slots collided -- every fragment read took two LDS cycles per group.  Strides 40 / 56 / 72 are conflict-free.
}

// FULLK: K == Kpad (no zero-filled tail columns): the loads carry no predicate at all.  TAB: g4d_linear_interp_add_f32 -- the rows'
// interpolation contexts (index / distance loads, three divisions) are formed BEFORE the contraction and only the table rows themselves are
// fetched in the epilogue (round 5: with K = 192 the epilogue's dependent loads were a sixth of the launch).
// WN: waves along the 128 channels of the tile (2: four waves of 64 x 64, 16 accumulator tiles each; 4: eight waves of 64 x 32).
// D2: the global prefetch runs TWO k chunks ahead through a second register set (hipBLASLt's kernels for these shapes do: PGR2 in their names).
//     An A/B switch, off by default: measured equal to one chunk ahead, 40 registers more.
// What bounded this kernel, and what fixed it (round 5; profiles/r05_gemm_tile_phases.txt = scripts/dbg_gemm_phases.py on a -DG4D_GEMM_DEBUG
// build, scripts/exp_clock_gemm.py for the sustained rate): it sat at 0.72-0.74 of the matrix pipe sustained (0.60-0.65 in 7 ms bursts from an idle
// chip) whatever was changed AROUND the MFMAs -- persistent tiles, eight waves instead of four, one or two workgroups per CU, a conflict-free LDS
// stride, unpredicated loads, two chunks of look-ahead, 64-deep chunks, the four k-steps of a fragment chained on one accumulator: each within
// 2 %.  Cycle stamps: per 32-deep chunk a wave spent ~2.0k cycles ISSUING its 16 global loads as one burst -- the eight waves of a CU queue at
// its one 64 B / clk vector-memory path, 16 cycles per 1 KB wave-load -- 1.1k staging the older chunk, 1.8k in the barrier and 5.1k in its 128
// MFMAs (4.1k of pipe); a wave issues in order, so it issued no MFMA while its loads queued, and two waves per SIMD cannot cover 5 k cycles of
// each other's non-MFMA time with 4.1k of MFMAs.  Now the loads and the LDS stores sit BETWEEN the MFMA chains, one per two chains (8 MFMAs =
// 256 cycles of pipe each): 313 -> 292 us at 61440 x 576 -> 512, 143 -> 132 us at 512 -> 256 sustained = 0.79 / 0.78 of the fp32 MFMA peak,
// hipBLASLt's kernels (MT64x128x64 / MT256x256x32, PGR2) 291 / 128 us on the same box.
template <bool FULLK, bool TAB, int WN, bool D2>
__global__ void __launch_bounds__(128 * WN, WN) gemm_tile_kernel(const LinearArgs a, int nrow_blk, int ncol_blk, int cpad, int nblocks) {
    constexpr int NTH = 128 * WN, TPR = TK / 4, P = TM / (NTH / TPR), JT = TN / WN / 16;   // threads; threads per staged row; staging passes per operand; channel tiles per wave
    extern __shared__ __attribute__((aligned(16))) float g_smem[];
    float *sA = g_smem;                         // [2][TM * TLD]
    float *sB = g_smem + 2 * TM * TLD;          // [2][TN * TLD]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // PERSISTENT (round 5): a workgroup walks blocks blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8 or the whole launch, so
    // a workgroup stays on the row blocks of ONE XCD) and treats the k chunks of all its tiles as ONE stream: the prefetch runs across tile
    // boundaries and the result stores of tile i drain under the contraction of tile i + 1.
    // block -> (row block, column block): blocks are numbered round-robin over the 8 XCDs; XCD x takes row blocks x, x + 8, ... and runs all
    // column blocks of a row block back to back.  The numbering rounds the row blocks up to a multiple of 8: a block past the last row block is
    // contracted on clamped rows like any other and stores nothing.
    const int G = gridDim.x;
    auto place = [&](int b, int &r0, int &c0) {
        const int x = b & 7, slot = b >> 3;
        r0 = ((slot / ncol_blk) * 8 + x) * TM;
        c0 = (slot % ncol_blk) * TN;
    };
    const int ntile = (nblocks - (int)blockIdx.x + G - 1) / G;   // >= 1 (grid <= nblocks)
    const int nchunk = a.Kpad / TK;
    const int total = ntile * nchunk;
    // staging map: thread -> rows lr + RS p, 4 consecutive k at lk
    constexpr int RS = NTH / TPR;
    const int lr = t / TPR, lk = (t % TPR) * 4;
    // K and ldx are multiples of 4 and X is 16-byte aligned (launcher): a lane's four columns are all inside the row or all past its end.
    // Every load is UNCONDITIONAL (clamped rows / tiles, the tail columns cleared with a bit mask, no predicate at all when K is a whole number
    // of chunks): a load inside a conditional block -- or a select the compiler turns into one -- makes the number of loads in flight unknown at
    // the join, and the wait for the OLDER prefetch in front of its LDS stores becomes vmcnt(0).
    auto load_x = [&](const float *xr, int k) -> f32x4 {
        if constexpr (FULLK) return *reinterpret_cast<const f32x4 *>(xr + k);
        const bool in = k + lk < a.K;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(xr + (in ? k : 0));
        const unsigned m = in ? 0xffffffffu : 0u;
        const u32x4 r = {v[0] & m, v[1] & m, v[2] & m, v[3] & m};
        return __builtin_bit_cast(f32x4, r);
    };
    // the fetch cursor: chunk fc of the workgroup's tile fj (past the last tile: the last tile again, never used)
    const float *fx[P], *fw[P];
    int fc = 0, fj = 0;
    auto set_cursor = [&](int j) {
        int r0, c0;
        place((int)blockIdx.x + min(j, ntile - 1) * G, r0, c0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            fx[p] = a.X + (size_t)min(r0 + lr + RS * p, a.rows - 1) * a.ldx + lk;     // rows past the end: clamped, never stored
            fw[p] = a.W + (size_t)min(c0 + lr + RS * p, cpad - 1) * a.Kpad + lk;      // channels past the padded width: clamped, never stored
        }
    };
    set_cursor(0);
    auto fetch = [&](f32x4 (&rx)[P], f32x4 (&rw)[P]) {
        const int k = fc * TK;
#pragma unroll
        for (int p = 0; p < P; ++p) { rx[p] = load_x(fx[p], k); rw[p] = *reinterpret_cast<const f32x4 *>(fw[p] + k); }
        if (++fc == nchunk) { fc = 0; set_cursor(++fj); }
    };
    auto stage = [&](int buf, const f32x4 (&rx)[P], const f32x4 (&rw)[P]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            *reinterpret_cast<f32x4 *>(&sA[buf * TM * TLD + (lr + RS * p) * TLD + lk]) = rx[p];
            *reinterpret_cast<f32x4 *>(&sB[buf * TN * TLD + (lr + RS * p) * TLD + lk]) = rw[p];
        }
    };
    f32x4 ax[P], aw[P], bx[D2 ? P : 1], bw[D2 ? P : 1];
    fetch(ax, aw);              // chunk 0
    stage(0, ax, aw);
    if constexpr (D2) fetch(ax, aw);   // chunk 1: staged during step 0 (one register set: requested AND staged during step 0)
    lds_barrier();
    const int fi = lane & 15, fq = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;   // the wave's 64 x (128 / WN) part of the tile
    constexpr int WCOLS = TN / WN;
    const bool vec = (a.ldo & 3) == 0 && (a.col0 & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0;   // launch-uniform
    int cur = 0, c = 0, tj = 0;                 // stage buffer / chunk / tile being contracted
    int row0, n0;
    place((int)blockIdx.x, row0, n0);
    f32x4 acc[4][JT];
    struct Ctx { float w0, w1, w2; unsigned k0, k1, k2; };   // (32-bit table offsets: the launcher checks the table is < 2^32 floats)
    Ctx ctx[4] = {};
    auto begin_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (TAB) {   // g4d_linear_interp_add_f32: the rows' interpolation contexts, formed ahead of the contraction
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const InterpRow r = interp_row(a, min(row0 + wr * 64 + i * 16 + fi, a.rows - 1));
                ctx[i] = Ctx{r.w0, r.w1, r.w2, (unsigned)r.k0, (unsigned)r.k1, (unsigned)r.k2};
            }
        }
    };
    begin_tile();
    // epilogue.  The MFMAs ran transposed (A = weights, B = activations: the same products summed in the same k order, i.e. the same bits as
    // linear_kernel's orientation), so lane (fi, fq) holds channels 4 fq .. 4 fq + 3 of tile j for row fi of tile i: ONE interpolation context
    // per (lane, row tile), 16-byte table loads and 16-byte stores.  Tiles that lie inside the matrix take a path without the per-row predicate:
    // under it every (i, j) step was its own exec-masked block with the scale / shift loads sunk into it, each pair followed by vmcnt(0).
    auto epilogue = [&](auto full_tag, auto vec_tag) {
        constexpr bool FULL = decltype(full_tag)::value, VEC = decltype(vec_tag)::value;   // VEC: 16-byte stores (launch-uniform; as a run-time test every store sat in its own basic block)
        // steps s = (channel tile j, row tile i), j outermost: one scale / shift pair is live at a time
        f32x4 sc[JT], sh[JT];
        auto ld_affine = [&](int j) {
            const int chc = min(n0 + wc * WCOLS + j * 16 + fq * 4, cpad - 4);            // the packed scale / shift are padded to 64 channels
            sc[j] = *reinterpret_cast<const f32x4 *>(a.scale + chc); sh[j] = *reinterpret_cast<const f32x4 *>(a.shift + chc);
        };
        ld_affine(0);
        // the three table rows of step s + 1 are requested before step s is finished (TAB)
        struct T3 { f32x4 t0, t1, t2; };
        auto tab_ld3 = [&](int s_) {
            const int j = s_ >> 2, i = s_ & 3;
            const unsigned ch0 = (unsigned)(n0 + wc * WCOLS + j * 16 + fq * 4);
            return T3{*reinterpret_cast<const f32x4 *>(a.tab + (ctx[i].k0 + ch0)), *reinterpret_cast<const f32x4 *>(a.tab + (ctx[i].k1 + ch0)),
                      *reinterpret_cast<const f32x4 *>(a.tab + (ctx[i].k2 + ch0))};
        };
        T3 tn = {};
        if constexpr (TAB) tn = tab_ld3(0);
#pragma unroll
        for (int s_ = 0; s_ < 4 * JT; ++s_) {
            const int j = s_ >> 2, i = s_ & 3;
            if (i == 0 && j + 1 < JT) ld_affine(j + 1);
            const int row = row0 + wr * 64 + i * 16 + fi;
            const int ch0 = n0 + wc * WCOLS + j * 16 + fq * 4;
            const T3 tc = tn;
            if constexpr (TAB) { if (s_ + 1 < 4 * JT) tn = tab_ld3(s_ + 1); }
            f32x4 y = acc[i][j];
            if constexpr (TAB) {   // (the launcher guarantees Cout % 128 == 0, tab_ld % 4 == 0 and a 16-byte aligned table: whole 16-byte groups inside the row)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = y[r] + (ctx[i].w0 * tc.t0[r] + ctx[i].w1 * tc.t1[r] + ctx[i].w2 * tc.t2[r]);   // interp_at(), element by element
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = __builtin_fmaf(y[r], sc[j][r], sh[j][r]);
                if (a.relu) y[r] = fmaxf(y[r], 0.f);
            }
            if (FULL || row < a.rows) {
                float *o = a.out + (size_t)row * a.ldo + a.col0 + ch0;
                if constexpr (VEC) *reinterpret_cast<f32x4 *>(o) = y;          // (Cout is a multiple of 128: the four channels are inside the row)
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = y[r];
                }
            }
        }
    };
    // one chunk: request a chunk further down the stream into `fill`, contract the staged chunk, stage the chunk held in `drain` (the wait in
    // front of its LDS stores carries the exact count of the younger loads), LDS-only barrier (a __syncthreads() would drain the prefetch and
    // the result stores), and at a tile's last chunk its epilogue
#ifdef G4D_GEMM_DEBUG
    long long dbg_last = (long long)__builtin_readcyclecounter();
#endif
    auto step = [&](f32x4 (&fill_x)[P], f32x4 (&fill_w)[P], const f32x4 (&drain_x)[P], const f32x4 (&drain_w)[P], bool same) {
        G4D_MSTAMP(7)   // (loop overhead / tile setup since the last stamp)
        // The 2 P loads of chunk s + 2 (one register set: of chunk s + 1) and the 2 P LDS stores of chunk s + 1 are spread BETWEEN the chunk's MFMA
        // chains (one per two chains: loads in the first half, stores in the second).  Issued as a burst at the top of the chunk they took the wave ~2k cycles -- eight waves
        // queue at the CU's one 64 B / clk vector-memory path -- during which it issued no MFMA (profiles/r05_gemm_tile_phases.txt).
        constexpr int NCH = (TK / 16) * 4 * JT;      // MFMA chains per chunk
        static_assert(NCH >= 8 * P, "room for 2 P loads and 2 P stores between the chains");
        const int kf = fc * TK;
        (void)same;
        G4D_MSTAMP(0)
        const float *cA = sA + cur * TM * TLD + (wr * 64 + fi) * TLD + fq * 4;
        const float *cB = sB + cur * TN * TLD + (wc * WCOLS + fi) * TLD + fq * 4;
        float *dA = sA + (cur ^ 1) * TM * TLD + lr * TLD + lk, *dB = sB + (cur ^ 1) * TN * TLD + lr * TLD + lk;
#pragma unroll
        for (int kk = 0; kk < TK; kk += 16) {
            f32x4 af[4], bf[JT];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x4 *>(cA + i * 16 * TLD + kk);
#pragma unroll
            for (int j = 0; j < JT; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(cB + j * 16 * TLD + kk);
            // the four k-steps of a fragment pair back to back on ONE accumulator
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    __builtin_amdgcn_sched_barrier(0);   // (keeps the chain -- and the memory operation in front of it -- where it is written)
                    {
                        const int ch = (kk / 16) * 4 * JT + i * JT + j;
                        if (ch % 2 == 0 && ch / 2 < 2 * P) {                      // request n of chunk s + 2
                            constexpr int dummy = 0; (void)dummy;
                            const int n = ch / 2;
                            if (n < P) fill_x[n] = load_x(fx[n], kf);
                            else fill_w[n - P] = *reinterpret_cast<const f32x4 *>(fw[n - P] + kf);
                        }
                        if (ch % 2 == 1 && ch >= NCH - 4 * P) {                  // LDS store n of chunk s + 1
                            const int n = (ch - (NCH - 4 * P)) / 2;
                            if (n < P) *reinterpret_cast<f32x4 *>(dA + RS * n * TLD) = drain_x[n];
                            else *reinterpret_cast<f32x4 *>(dB + RS * (n - P) * TLD) = drain_w[n - P];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);   // (weights as the A operand: see the epilogue)
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        G4D_MSTAMP(1)   // fragments read, MFMAs issued (D2: with the requests and the staging in between)
        if (++fc == nchunk) { fc = 0; set_cursor(++fj); }   // the cursor moves on (fetch() does it for the burst form of the prologue)
        G4D_MSTAMP(2)
        lds_barrier();
        G4D_MSTAMP(3)   // barrier
        cur ^= 1;
        if (++c == nchunk) {
            if (vec) {
                if (row0 + TM <= a.rows) epilogue(std::true_type{}, std::true_type{});
                else epilogue(std::false_type{}, std::true_type{});
            } else epilogue(std::false_type{}, std::false_type{});
            c = 0;
            place((int)blockIdx.x + min(++tj, ntile - 1) * G, row0, n0);
            begin_tile();
            G4D_MSTAMP(4)   // epilogue + next tile's setup
        }
    };
    if constexpr (D2) {
        for (int s_ = 0; s_ < total; s_ += 2) {
            step(bx, bw, ax, aw, false);                       // chunk s: fill B with chunk s + 2, stage chunk s + 1 from A
            if (s_ + 1 < total) step(ax, aw, bx, bw, false);   // chunk s + 1: fill A with chunk s + 3, stage chunk s + 2 from B
        }
    } else {
        for (int s_ = 0; s_ < total; ++s_) step(ax, aw, ax, aw, true);
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
    if (a.tab && ((a.tab_ld & 3) || (reinterpret_cast<size_t>(a.tab) & 15) || a.n <= 0 || (a.rows / a.n + 1) * (long long)a.m * a.tab_ld >= (1ll << 32))) return false;
    static const int lds_extra = getenv("G4D_GEMM_TILE_LDS_EXTRA") ? atoi(getenv("G4D_GEMM_TILE_LDS_EXTRA")) : 0;   // experiment: > 0 forces one workgroup per CU
    const int lds = 2 * (TM + TN) * TLD * (int)sizeof(float) + lds_extra;   // 81920 bytes: two workgroups per CU fill its 160 KB exactly
    typedef void (*Kern)(const LinearArgs, int, int, int, int);
    static const int d2 = getenv("G4D_GEMM_TILE_D2") ? atoi(getenv("G4D_GEMM_TILE_D2")) : 0;   // A/B switch: prefetch two chunks ahead (measured: no gain, 32 more registers -- off; never for the interp-add form)
    static const Kern kerns[8] = {gemm_tile_kernel<false, false, 2, false>, gemm_tile_kernel<false, true, 2, false>, gemm_tile_kernel<true, false, 2, false>, gemm_tile_kernel<true, true, 2, false>,
                                  gemm_tile_kernel<false, false, 2, true>, gemm_tile_kernel<false, true, 2, false>, gemm_tile_kernel<true, false, 2, true>, gemm_tile_kernel<true, true, 2, false>};
    const int wn = 2;
    const int which = (d2 ? 4 : 0) + (a.K == a.Kpad ? 2 : 0) + (a.tab ? 1 : 0);
    static unsigned long long attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    *rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kerns[which]), lds, attr[which], "g4d_linear_f32(tile)");
    if (*rc) return true;
    const int cpad = (a.Cout + 63) / 64 * 64;   // the packed weight / scale / shift are padded to 64 channels
    const int nrow = (a.rows + TM - 1) / TM, ncol = (a.Cout + TN - 1) / TN;
    const long long blocks = (long long)((nrow + 7) / 8) * 8 * ncol;   // XCD-major numbering: row blocks rounded up to a multiple of 8
    if (blocks >= (1ll << 31)) return false;
    // persistent: two workgroups per CU by LDS; the grid a multiple of 8 so that a workgroup's blocks stay on one XCD's row blocks
    static const int resident = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        return (2 * cus) / 8 * 8;
    }();
    static const int persist = getenv("G4D_GEMM_TILE_PERSIST") ? atoi(getenv("G4D_GEMM_TILE_PERSIST")) : 1;   // A/B switch
    const long long grid = persist && blocks > resident ? resident : blocks;
    hipLaunchKernelGGL(kerns[which], dim3((unsigned)grid), dim3(wn == 4 ? 512 : 256), lds, s, a, nrow, ncol, cpad, (int)blocks);
    *rc = check_launch("g4d_linear_f32(tile)");
    return true;
}

}  // namespace g4d

#ifdef G4D_GEMM_DEBUG
extern "C" int g4d_gemm_debug_read(long long *host_out, int clear) {   // 8 x 1024 phase sums since the last clear
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g4d::g_gemm_dbg), sizeof(long long) * 8 * 1024);
    if (clear) { static long long z[8 * 1024]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g4d::g_gemm_dbg), z, sizeof(z)); }
    return rc;
}
#endif
