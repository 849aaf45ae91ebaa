// Feature propagation WITH skip features, the known-feature part of its first layer pre-contracted (g4d_mlp_chain_interp_init_f32:
// pointnet2_modules.py:127-156 with  W [interp(f) ; s] = interp(Wa f) + Wb s), PERSISTENT, software-pipelined and with the weights SHARED
// by the waves of a workgroup (round 4) -- the large-launch form for the middle FP level of Pointnet2MSGSEG:
//     acc = three_interpolate(table 256)  +  skip (96) . Wb   -> affine, ReLU (256)  ->  128 (tapped: the FP module's output)  ->  128 (the
//     next level's first-layer table, or any third layer)
//
// Why: 295 KB of weights per 16-row tile.  The register-chain kernel streams them from L2 per wave: at 16 rows per wave the vector-memory
// path is the bound (64 B / cycle / CU), at 32 rows per wave the 348 registers leave one wave per SIMD and every latency is exposed; both
// run at 0.53 of the matrix pipe (447 us per 240-cloud call for 231 us of MFMA).  Here the four waves of a workgroup walk their 16-row tiles
// in lock step and two k-steps' weight fragments at a time are copied L2 -> LDS once per workgroup (buffer_load ... lds since round 5, double buffer, one
// chunk ahead, one barrier per chunk), as in sa_table.hip; per-layer constants sit in LDS; (index, distance) of tile t + 2, the
// interpolation weights / offsets and skip rows of tile t + 1 are prepared while tile t computes, and tile t + 1's starting accumulators
// (the interpolated table) are built during tile t's middle layer, one channel tile per k-step.  Arithmetic and k order are the chain
// kernel's: bit-identical results.  Round 4 measured 432 us (one k-step per barrier: 449) and could not say what bounded it -- ruled out by
// A/B builds: stores in front of the chunk barriers (behind them: 444 us), the weight traffic itself (8 waves sharing a copy: 484 us), L2
// locality of the table gathers (XCD-contiguous tile ranges: 450 us).  Round 5 read the ISA (the comment at stage_issue_one below): scratch
// spills behind every early gather, vmcnt(0) between a copy and the next ds_read, vmcnt(0) after every FLAT lds copy -- 428 -> 324 us, the
// matrix pipe 0.52 -> 0.72 busy.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

#ifdef G4D_FPINIT_DEBUG
__device__ long long g_fpinit_dbg[8 * 1024];   // per workgroup (first 1024), wave 0: cycles per phase summed over its tiles (scripts/dbg_fp_init_phases.py)
#define G4D_FSTAMP(i) { if (threadIdx.x == 0 && blockIdx.x < 1024) { const long long now_ = (long long)__builtin_readcyclecounter(); g_fpinit_dbg[blockIdx.x * 8 + (i)] += now_ - dbg_last; dbg_last = now_; } }
#else
#define G4D_FSTAMP(i)
#endif

struct FpInitArgs {
    int rows, n, m;
    const float *skip;                    // (rows, C1 = 96)
    const float *tab; int tab_ld;         // (B * m, tab_ld >= 256): known features times Wa
    const float *dist2; const int *nn_idx;
    const float *W1, *sc1, *sh1;          // Wb: 96 -> 256, fragment order, Kpad == 96
    const float *W2, *sc2, *sh2;          // 256 -> 128
    const float *W3, *sc3, *sh3;          // 128 -> 128
    int relu1, relu2, relu3;
    float *out; int ldo;                  // last layer (rows, 128)
    float *tap; int tap_ld;               // output of the 256 -> 128 layer (rows, 128); may be NULL
};

namespace {
constexpr int C0 = 96, C1 = 256, C2 = 128, C3 = 128;
constexpr int K0 = C0 / 16, T1 = C1 / 16, T2 = C2 / 16, T3 = C3 / 16;   // 6 k-steps of skip columns; 16 / 8 / 8 channel tiles
constexpr int KSC = 2;                                                   // k-steps per staged chunk (one barrier per chunk)
constexpr int STEPS = (K0 + T1 + T2) / KSC;                              // chunks of a tile: 3 (layer 1) + 8 (layer 2) + 4 (layer 3)
constexpr int STAGE = KSC * 16 * 256;                                    // floats per stage buffer: the largest chunk (layer 1: 2 x 16 fragments)
constexpr int kStageBytes = 2 * STAGE * (int)sizeof(float);               // the double buffer: dynamic shared memory (see the kernel)
static_assert(K0 % KSC == 0 && T1 % KSC == 0 && T2 % KSC == 0, "whole chunks per layer");
}

__global__ void __launch_bounds__(256, 2) fp_init_kernel(const FpInitArgs a) {
    // DYNAMIC shared memory on purpose (round 5): for a static __shared__ array the compiler knows that the L2 -> LDS copies and the fragment
    // reads touch the same object and puts s_waitcnt vmcnt(0) between a chunk's copies and the first ds_read behind them -- the copy that was
    // issued one chunk AHEAD was waited for at once, in every chunk (0.40 of the wave cycles parked).  The ordering that is needed is the
    // explicit vmcnt(0) + barrier of stage_step().
    extern __shared__ __attribute__((aligned(16))) float fp_init_smem[];
    float (*s_stage)[STAGE] = reinterpret_cast<float (*)[STAGE]>(fp_init_smem);
    __shared__ __attribute__((aligned(16))) float s_sc1[C1], s_sh1[C1], s_sc2[C2], s_sh2[C2], s_sc3[C3], s_sh3[C3];
    const int tid = threadIdx.x;
    s_sc1[tid] = a.sc1[tid]; s_sh1[tid] = a.sh1[tid];
    if (tid < C2) { s_sc2[tid] = a.sc2[tid]; s_sh2[tid] = a.sh2[tid]; s_sc3[tid] = a.sc3[tid]; s_sh3[tid] = a.sh3[tid]; }
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the copy destinations below go through M0)
    const int fi = lane & 15, fq = lane >> 4;
    const int ntile = (a.rows + 15) >> 4;
    const int nwaves = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    if ((int)blockIdx.x * 4 >= ntile) return;
    const int iters = (ntile - (int)blockIdx.x * 4 + nwaves - 1) / nwaves;   // the same for the four waves (barriers inside the loop)
    auto tile_of = [&](int it) { return min(wg + it * nwaves, ntile - 1); };
    auto live = [&](int it) { return wg + it * nwaves < ntile; };
    const int lane16 = lane * 16;                    // byte offset of the lane's 16 bytes inside a 1 KB fragment

    // a chunk's weight fragments (KSC k-steps x all channel tiles; fragment (kk, ct) at slot kk * nt + ct) into one of the two stage buffers:
    // chunks 0..2 layer 1 (16 tiles), 3..10 layer 2 (8 tiles), 11..14 layer 3 (8 tiles).  One k-step per barrier measured no gain over the
    // register-chain kernel (449 vs 447 us): a 1k-cycle k-step does not cover the copy's L2 round trip, so every barrier waited for it.
    //
    // Round 5, three findings of reading the ISA (scripts/isa_events.py prints a kernel as runs of loads / waits / MFMAs):
    //  * `step` (the chunk's index inside its tile) is now a compile-time constant at every call site and only the buffer parity is run-time
    //    state.  With round 4's run-time `abs_chunk % STEPS` every chunk carried a dozen scalar selects and eight conditional branches around
    //    its copies; the branches cut the tile loop into basic blocks, the compiler sank the blends of the next tile's first table rows into the
    //    loop latch and kept the RAW rows alive instead -- 74 dwords of scratch, each spill store right behind its gather with s_waitcnt
    //    vmcnt(0) in between: the gathers this kernel exists to hide were waited for one by one.
    //  * the copies are MUBUF (buffer_load ... lds), not global_load_lds: a FLAT-encoded instruction with an LDS operand marks the wave's
    //    counters "flat pending" in the compiler's scoreboard and EVERY later vmcnt wait becomes vmcnt(0) -- the table gathers could not stay
    //    in flight past a copy.  With MUBUF copies the waits carry exact counts.
    //  * the stage buffers are dynamic shared memory (see above).
    const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W1), 0, C0 * C1 * 4, 0x00020000),
                                 rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W2), 0, C1 * C2 * 4, 0x00020000),
                                 rW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W3), 0, C2 * C3 * 4, 0x00020000);
    // copy number j of this wave's share of chunk `step` (fragments wave, wave + 4, ...).  Round 5: a chunk's copies go out ONE PER MFMA CHAIN of
    // the chunk before it, not as a burst behind the barrier: a wave issues in order, and while its 4-8 copies queued at the CU's 64 B / clk
    // vector-memory path (16 cycles per 1 KB wave-copy, eight waves in line) it issued no MFMA (gemm_tile.hip has the measurement).
    auto stage_issue_one = [&](int buf, int step, int j) {
        float *dst = s_stage[buf];
        constexpr int c1 = K0 / KSC, c2 = c1 + T1 / KSC;
        const int nt = step < c1 ? T1 : (step < c2 ? T2 : T3), kst = step < c1 ? K0 : (step < c2 ? T1 : T2);
        const int ks0 = (step < c1 ? step : (step < c2 ? step - c1 : step - c2)) * KSC;
        if (j < KSC * nt / 4) {
            const int slot = wave + 4 * j;
            const int kk = slot / nt, ct = slot - kk * nt;
            const int soff = (ct * kst + ks0 + kk) * 1024;
            __attribute__((address_space(3))) void *d = (__attribute__((address_space(3))) void *)(dst + slot * 256);
            if (step < c1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, d, 16, lane16, soff, 0, 0);
            else if (step < c2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, d, 16, lane16, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rW3, d, 16, lane16, soff, 0, 0);
        }
    };
    auto stage_issue = [&](int buf, int step) {      // all of them (the start-up)
#pragma unroll
        for (int j = 0; j < KSC * 4; ++j) stage_issue_one(buf, step, j);
    };
    int par0 = 0;                                    // stage buffer of the current tile's first chunk (STEPS is odd: it flips per tile)
    static_assert(STEPS % 2 == 1, "buffer parity flips per tile");
    // chunk c of the tile: this wave's share has landed, everybody's has, nobody reads the other buffer any more; chunk c + 1 (the next tile's
    // first chunk after the last one) goes out.  `younger` = the vector-memory instructions this wave issued AFTER chunk c's copies (vmcnt
    // counts in issue order on gfx9): they may stay in flight across the barrier.  Only the chunk boundaries INSIDE layer 2 use it -- there the
    // only younger instructions are the unconditional table gathers of its two k-steps (requested BEHIND the wave's four copies of the chunk).
#ifdef G4D_FPINIT_DEBUG
    long long dbg_last = (long long)__builtin_readcyclecounter();
#endif
    // The count is tied to the code that issues those gathers: load_item() is kItemLoads (= 3: one 16-byte gather per neighbour, struct Item) vector-
    // memory instructions and kYoungerItems (= 2) of them are younger than the copies.  A change of Item's layout breaks the static_asserts below
    // instead of silently shortening the wait; a compiler that merged or reordered these loads would be caught by the bit-identity test against
    // fp_init_persistent = 0 (tests/test_large_launch_gpu.py), which runs for every build.
    constexpr int kItemLoads = 3, kYoungerItems = 2, kYounger = kItemLoads * kYoungerItems;
    static_assert(kYounger == 6, "stage_step's s_waitcnt immediate below is written for 6 younger gathers");
    auto stage_step = [&](int c, int younger) {
        G4D_FSTAMP(0)   // compute since the last stamp
        if (younger == kYounger) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G4D_FSTAMP(1)   // own loads / copies landed
        asm volatile("s_barrier" ::: "memory");
        G4D_FSTAMP(2)   // barrier
    };
    // in front of MFMA chain `ct` of k-step gks (a layer with nt channel tiles): the next chunk's copy number (chain index within the chunk)
    auto copy_at = [&](int gks, int nt, int ct) {
        const int c = gks / KSC;
        stage_issue_one(par0 ^ ((c + 1) & 1), (c + 1) % STEPS, (gks % KSC) * nt + ct);
    };
    // fragment (global k-step gks of the tile, channel tile ct) of a layer with nt tiles
    auto sfrag = [&](int gks, int nt, int ct) -> f32x4 {
        return *reinterpret_cast<const f32x4 *>(&s_stage[par0 ^ ((gks / KSC) & 1)][(((gks % KSC) * nt + ct) * 64 + lane) * 4]);
    };
    auto stage_at = [&](int gks, int younger = 0) { if (gks % KSC == 0) stage_step(gks / KSC, younger); };

    struct Raw { int i0, i1, i2; float d0, d1, d2; };
    auto load_raw = [&](int tile) {
        Raw r;
        const int row = min(tile * 16 + fi, a.rows - 1);
        const int *ix = a.nn_idx + (size_t)row * 3;
        const float *dd = a.dist2 + (size_t)row * 3;
        r.i0 = ix[0]; r.i1 = ix[1]; r.i2 = ix[2]; r.d0 = dd[0]; r.d1 = dd[1]; r.d2 = dd[2];
        return r;
    };
    struct Ctx { float w0, w1, w2; unsigned k0, k1, k2; };
    auto make = [&](int tile, const Raw &r) {   // pointnet2_utils.py:98 sqrt; pointnet2_modules.py:140-142 inverse-distance weights
        Ctx c;
        const float r0 = 1.0f / (__fsqrt_rn(r.d0) + 1e-8f), r1 = 1.0f / (__fsqrt_rn(r.d1) + 1e-8f), r2 = 1.0f / (__fsqrt_rn(r.d2) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        c.w0 = r0 / norm; c.w1 = r1 / norm; c.w2 = r2 / norm;
        const int row = min(tile * 16 + fi, a.rows - 1);
        const int b0 = __builtin_amdgcn_readfirstlane((tile * 16) / a.n);   // a tile touches at most two clouds (n >= 16)
        const unsigned base = (unsigned)(b0 + (row >= (b0 + 1) * a.n ? 1 : 0)) * (unsigned)a.m;
        c.k0 = (base + (unsigned)r.i0) * (unsigned)a.tab_ld + fq * 4; c.k1 = (base + (unsigned)r.i1) * (unsigned)a.tab_ld + fq * 4;
        c.k2 = (base + (unsigned)r.i2) * (unsigned)a.tab_ld + fq * 4;
        return c;
    };
    struct Skip { f32x4 s[K0]; };
    auto load_skip = [&](int tile) {
        Skip k;
        const float *p = a.skip + (size_t)min(tile * 16 + fi, a.rows - 1) * C0 + fq * 4;
#pragma unroll
        for (int ks = 0; ks < K0; ++ks) k.s[ks] = *reinterpret_cast<const f32x4u *>(p + ks * 16);
        return k;
    };
    struct Item { f32x4 t0, t1, t2; };
    static_assert(sizeof(Item) == kItemLoads * sizeof(f32x4), "one 16-byte gather per member: stage_step counts them (vmcnt)");
    auto load_item = [&](const Ctx &c, int ct) {
        Item x;
        x.t0 = *reinterpret_cast<const f32x4u *>(a.tab + c.k0 + ct * 16);
        x.t1 = *reinterpret_cast<const f32x4u *>(a.tab + c.k1 + ct * 16);
        x.t2 = *reinterpret_cast<const f32x4u *>(a.tab + c.k2 + ct * 16);
        return x;
    };

    auto blend = [&](const Ctx &c, const Item &x) -> f32x4 {   // three_interpolate of 4 consecutive channels, the chain kernel's operation order
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c.w0 * x.t0[e] + c.w1 * x.t1[e] + c.w2 * x.t2[e];
        return v;
    };
    // ReLU or not as the floor of one v_max (no select per element): exact for every finite / infinite input
    float lo1 = a.relu1 ? 0.f : -__builtin_inff(), lo2 = a.relu2 ? 0.f : -__builtin_inff(), lo3 = a.relu3 ? 0.f : -__builtin_inff();
    asm volatile("" : "+v"(lo1), "+v"(lo2), "+v"(lo3));   // (opaque: the compiler otherwise turns each max back into max + select)
    stage_issue(0, 0);
    Raw rawn = load_raw(tile_of(0));
    Ctx cur = make(tile_of(0), rawn);
    Skip sk = load_skip(tile_of(0));
    rawn = load_raw(tile_of(1));
    // accumulators of a tile start from three_interpolate(table): lane (fi, fq) holds channels 16 ct + 4 fq + e of row fi (transposed tile).
    // The first tile's are built here (latency exposed once); every later tile's are built DURING the previous tile's layer 2, one
    // channel tile per k-step, its three table rows requested a k-step earlier.
    f32x4 h1[T1];
    {
        Item ring[2] = {load_item(cur, 0), load_item(cur, 1)};
#pragma unroll
        for (int ct = 0; ct < T1; ++ct) {
            const Item x = ring[ct & 1];
            if (ct + 2 < T1) ring[ct & 1] = load_item(cur, ct + 2);
            h1[ct] = blend(cur, x);
        }
    }
    for (int it = 0; it < iters; ++it) {
        G4D_FSTAMP(3)   // tile epilogue (last layer's stores) + loop
        const int tile = tile_of(it);
        par0 = it & 1;
        const bool row_ok = tile * 16 + fi < a.rows && live(it);
        // ---- layer 1: the skip columns on the matrix pipe, transposed (A = weights, B = the skip row's 16 columns of the k-step)
        f32x4 sring[2];
#pragma unroll
        for (int ks = 0; ks < K0; ++ks) {
            stage_at(ks);
            sring[0] = sfrag(ks, T1, 0); sring[1] = sfrag(ks, T1, 1);
#pragma unroll
            for (int ct = 0; ct < T1; ++ct) {
                const f32x4 w = sring[ct & 1];
                if (ct + 2 < T1) sring[ct & 1] = sfrag(ks, T1, ct + 2);
                __builtin_amdgcn_sched_barrier(0);
                copy_at(ks, T1, ct);
#pragma unroll
                for (int e = 0; e < 4; ++e) h1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], sk.s[ks][e], h1[ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < T1; ++ct) {
            if ((ct & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler requests all 32 scale / shift vectors up front: 128 registers)
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc1 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh1 + ct * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h1[ct][r] = fmaxf(__builtin_fmaf(h1[ct][r], sc[r], sh[r]), lo1);
            }
        }
        // ---- layer 2 (256 -> 128), transposed; the next tile's inputs are requested behind its first k-step barrier
        f32x4 h2[T2];
#pragma unroll
        for (int ct = 0; ct < T2; ++ct) h2[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Ctx nxt;
        Skip skn;
        f32x4 h1n[T1];
        constexpr int RING = 3;                      // the table rows of k-steps ks - 3 .. ks - 1 are in flight: a gather is consumed three k-steps after its request
        Item item[RING];
#pragma unroll
        for (int ks = 0; ks < T1; ++ks) {
            if (ks == 0) {                           // (ahead of the barrier's copies: older than them in the wave's memory queue)
                nxt = make(tile_of(it + 1), rawn);
                rawn = load_raw(tile_of(it + 2));
            }
            // inner barriers (ks = 2, 4, ..): the copies of this chunk went out in chains 0..3 of k-step ks - 2; YOUNGER than them in the wave's
            // memory queue are only the gathers of k-steps ks - 2 (requested in chain 4, behind the copies) and ks - 1: vmcnt(6)
            stage_at(K0 + ks, ks >= 2 ? kYounger : 0);
            if (ks >= RING) {
                h1n[ks - RING] = blend(nxt, item[ks % RING]);
                asm volatile("" : "+v"(h1n[ks - RING]));   // the blend happens HERE (not sunk towards the loop latch with the raw rows kept alive)
            }
            sring[0] = sfrag(K0 + ks, T2, 0); sring[1] = sfrag(K0 + ks, T2, 1);
#pragma unroll
            for (int ct = 0; ct < T2; ++ct) {
                const f32x4 w = sring[ct & 1];
                if (ct + 2 < T2) sring[ct & 1] = sfrag(K0 + ks, T2, ct + 2);
                __builtin_amdgcn_sched_barrier(0);
                copy_at(K0 + ks, T2, ct);
                if (ct == KSC * T2 / 4) item[ks % RING] = load_item(nxt, ks);   // behind this wave's copies of the chunk (chains 0 .. 3 of its first k-step)
#pragma unroll
                for (int e = 0; e < 4; ++e) h2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h1[ks][e], h2[ct], 0, 0, 0);
            }
        }
        const size_t orow = (size_t)min(tile * 16 + fi, a.rows - 1);
#pragma unroll
        for (int ct = 0; ct < T2; ++ct) {
            if ((ct & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc2 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh2 + ct * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h2[ct][r] = fmaxf(__builtin_fmaf(h2[ct][r], sc[r], sh[r]), lo2);
            }
            if (a.tap && row_ok) *reinterpret_cast<f32x4 *>(a.tap + orow * a.tap_ld + ct * 16 + fq * 4) = h2[ct];
        }
        // ---- layer 3 (128 -> 128), transposed like the two before it (round 5; the same products in the same k order as the normal orientation
        //      of round 4, i.e. the same bits): lane (fi, fq) ends with channels 16 ct + 4 fq + r of row fi = ONE 16-byte store per channel
        //      tile.  In the normal orientation the tile left through 32 dword stores per lane -- 12.6k of a tile's 110k cycles by the phase
        //      stamps (scripts/dbg_fp_init_phases.py), the wave issuing nothing else meanwhile.
        f32x4 o[T3];
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < T2; ++ks) {
            stage_at(K0 + T1 + ks);
            if (ks == 0) {                            // the last RING gathers of layer 2 (complete: the barrier above waited for everything)
#pragma unroll
                for (int q = T1 - RING; q < T1; ++q) { h1n[q] = blend(nxt, item[q % RING]); asm volatile("" : "+v"(h1n[q])); }
                skn = load_skip(tile_of(it + 1));
            }
            sring[0] = sfrag(K0 + T1 + ks, T3, 0); sring[1] = sfrag(K0 + T1 + ks, T3, 1);
#pragma unroll
            for (int ct = 0; ct < T3; ++ct) {
                const f32x4 w = sring[ct & 1];
                if (ct + 2 < T3) sring[ct & 1] = sfrag(K0 + T1 + ks, T3, ct + 2);
                __builtin_amdgcn_sched_barrier(0);
                copy_at(K0 + T1 + ks, T3, ct);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h2[ks][e], o[ct], 0, 0, 0);
            }
        }
        const bool tile_live = live(it);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < T3; ++ct) {
            if ((ct & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc3 + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh3 + ct * 16 + fq * 4);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(__builtin_fmaf(o[ct][r], sc[r], sh[r]), lo3);
            if (tile_live && row_ok) *reinterpret_cast<f32x4 *>(a.out + orow * a.ldo + ct * 16 + fq * 4) = y;   // (16-byte aligned: checked by the launcher)
        }
        cur = nxt;
        sk = skn;
#pragma unroll
        for (int ct = 0; ct < T1; ++ct) h1[ct] = h1n[ct];
    }
}

}  // namespace g4d

#ifdef G4D_FPINIT_DEBUG
extern "C" int g4d_fpinit_debug_read(long long *host_out, int clear) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g4d::g_fpinit_dbg), sizeof(long long) * 8 * 1024);
    if (clear) { static long long z[8 * 1024]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g4d::g_fpinit_dbg), z, sizeof(z)); }
    return rc;
}
#endif

using namespace g4d;

// Takes the launch if it is the instantiated stack (skip 96 -> 256 -> 128 -> 128) and large enough; -1 when it is not.
int g4d::fp_init_try(long long rows, int n, int m, int C1_, const float *skip, const float *table, int tab_ld, const float *dist2, const int *nn_idx,
                     int nlayers, const float *const *W, const float *const *scale, const float *const *shift, const int *Kpad, const int *Cout,
                     const int *relu, float *out, int ldo, int col0, int tap_layer, float *tap_out, int tap_ld, hipStream_t st) {
    const int on = (int)tuning("fp_init_persistent", 1);
    const long long min_rows = tuning("fp_init_min_rows", 131072);
    if (!on || rows < min_rows || rows >= (1ll << 31) - 64 || C1_ != C0 || nlayers != 3 || col0 != 0) return -1;
    if (Cout[0] != C1 || Cout[1] != C2 || Cout[2] != C3 || Kpad[0] != C0 || Kpad[1] != C1 || Kpad[2] != C2) return -1;
    if (tap_out && (tap_layer != 1 || tap_ld % 4 != 0 || (reinterpret_cast<size_t>(tap_out) & 15) != 0)) return -1;
    if (n < 16 || m <= 0 || rows % n != 0 || (rows / n) * (long long)m * tab_ld >= (1ll << 32) || tab_ld < C1 || (reinterpret_cast<size_t>(table) & 15) != 0) return -1;
    G4D_REQUIRE(skip && table && dist2 && nn_idx && out && W[0] && W[1] && W[2] && scale[0] && scale[1] && scale[2] && shift[0] && shift[1] && shift[2],
                "g4d_mlp_chain_interp_init_f32: null pointer");
    if (ldo % 4 != 0 || (reinterpret_cast<size_t>(out) & 15) != 0) return -1;   // (the last layer leaves through 16-byte stores)
    G4D_REQUIRE(ldo >= Cout[2] && (!tap_out || tap_ld >= Cout[1]) && tab_ld % 4 == 0, "g4d_mlp_chain_interp_init_f32: output row stride %d < %d channels, tap stride %d < %d or table stride %d not a multiple of 4",
                ldo, Cout[2], tap_ld, Cout[1], tab_ld);
    FpInitArgs a;
    a.rows = (int)rows; a.n = n; a.m = m; a.skip = skip; a.tab = table; a.tab_ld = tab_ld; a.dist2 = dist2; a.nn_idx = nn_idx;
    a.W1 = W[0]; a.sc1 = scale[0]; a.sh1 = shift[0]; a.W2 = W[1]; a.sc2 = scale[1]; a.sh2 = shift[1]; a.W3 = W[2]; a.sc3 = scale[2]; a.sh3 = shift[2];
    a.relu1 = relu[0]; a.relu2 = relu[1]; a.relu3 = relu[2];
    a.out = out; a.ldo = ldo; a.tap = tap_out; a.tap_ld = tap_ld;
    static unsigned long long attr = 0;
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(fp_init_kernel), kStageBytes, attr, "g4d_fp_init")) return rc;
    static const int resident = [] {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fp_init_kernel, 256, kStageBytes) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return per_cu * 256;
        return per_cu * prop.multiProcessorCount;
    }();
    const long long want = ((rows + 15) / 16 + 3) / 4;
    hipLaunchKernelGGL(fp_init_kernel, dim3((unsigned)(want < resident ? want : resident)), dim3(256), kStageBytes, st, a);
    return check_launch("g4d_fp_init");
}
