// Row-streaming fp32 GEMM for the tall contractions of the refinement model (BASELINE config 4) with K <= 128:
// Y = act((X . W^T) * scale + shift) over ~10^6 rows -- the qkv projection of the temporal attention (128 -> 384 on frames x Vg rows,
// /root/reference/modules/mesh_encoder.py:469).
//
// The LDS-tiled kernel of mlp.hip stages K in chunks of 32 with ONE chunk of look-ahead: 32 MFMAs (0.4 us) of cover against an HBM
// round trip of 2+ us, so every chunk of every 64 x 64 tile waits for its data (64-66 TFLOP/s on these shapes).  Here a workgroup
// (8 waves) is persistent and owns a 128-row x 128-channel tile at a time:
//   * K is walked in super-chunks of 64: the whole 128 x 64 block of X and of W for the NEXT super-chunk (or the next tile) is requested
//     into registers -- all loads issued together -- before the current one's 128 MFMAs per wave start: one memory round trip per
//     1.7-3.4 us of matrix work instead of one per 0.4 us;
//   * wave w owns all 128 rows x channels [16 w, 16 w + 16): 8 accumulator tiles, A fragments by ds_read_b128 (4 consecutive k per
//     lane feeding 4 MFMAs, same k permutation on both operands -- the LDS-tiled kernel's order, so results are bit-identical to it);
//   * 2 x 128 x 72 floats of LDS = 74 KB: two workgroups per CU (row stride 72 since round 5: conflict-free fragment reads).
// Measured at 983k rows (scripts/time_gemm_stream.py): 128 -> 128 491 -> 385-400 us (81-84 TFLOP/s), 128 -> 384 1510 -> 1210-1290 us;
// with K = 195 / 323 (W re-streamed from L2 per tile and super-chunk, ragged tail) it LOSES to the register-chain kernel (842 / 1226 vs
// 713 / 1093 us), so launch_linear takes this route only for Kpad <= 128.  What still separates it from the matrix pipe's 157: skipping
// 3/4 of the MFMAs leaves 201 us -- the time to stream 1 GB at ~5 TB/s -- and the MFMAs add their 190 us on top.  Two things that
// should have let them overlap and measured the same: (1) results leave through BUFFER stores with an out-of-range offset for idle
// lanes, so the stores are unconditional, their count is static and the wait for the prefetched operands (loads and stores share
// gfx9's in-order vmcnt) does not include them; (2) the barriers are LDS-only (lds_barrier(), g4d_common.h) -- __syncthreads() drains
// every global load and store in flight.  Both are kept: they are the right shape.  What still serialises the two phases (ISA, round 2):
// the compiler's own wait in front of the LDS stores of a tile's first super-chunk is `s_waitcnt vmcnt(3..0)` -- it does not credit the
// 32 result stores issued after the operand loads, so the write round trip of tile t still sits in front of tile t + 1.  A variant with
// every load unconditional (clamped rows / tiles, K templated, 32 no-op stores in the prologue so that every path into that wait carries
// the same 32 younger stores) got the same waits from hipcc and the same 365-380 us at 128 -> 128; only hand-placed waits around
// asm loads (or LDS-direct loads) would change it.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

namespace {
constexpr int GM = 128, GN = 128, GK = 64, GLD = GK + 8;   // (72: conflict-free fragment reads; 68 was two-way -- gemm_tile.hip)
constexpr int GT = 512;   // threads
}  // namespace

template <bool BUF>
__global__ void __launch_bounds__(GT, 2) gemm_stream_kernel(const LinearArgs a, int ntiles, unsigned out_bytes) {
    extern __shared__ __attribute__((aligned(16))) float gs_smem[];
    float *sA = gs_smem;
    float *sW = gs_smem + GM * GLD;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fi = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.y * GN;
    const int nsc = (a.Kpad + GK - 1) / GK;          // super-chunks along K
    const bool w_resident = nsc == 1;
    // staging map: thread -> rows (t >> 4) + 32 i (i < 4), 4 consecutive k at (t & 15) * 4
    const int sr = t >> 4, sk = (t & 15) * 4;
    constexpr int NS = GM * GK / 4 / GT;   // 16-byte pieces per thread and operand
    f32x4 ra[NS], rw[NS];
    auto load_a = [&](int tile, int sc) {
        const int k = sc * GK + sk;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int row = tile * GM + sr + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < a.rows && k < a.K) {
                const float *p = a.X + (size_t)row * a.ldx + k;
                if (k + 3 < a.K && (a.ldx & 3) == 0) v = *reinterpret_cast<const f32x4 *>(p);
                else { v.x = p[0]; if (k + 1 < a.K) v.y = p[1]; if (k + 2 < a.K) v.z = p[2]; if (k + 3 < a.K) v.w = p[3]; }
            }
            ra[i] = v;
        }
    };
    auto load_w = [&](int sc) {
        const int k = sc * GK + sk;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            rw[i] = k < a.Kpad ? *reinterpret_cast<const f32x4 *>(a.W + (size_t)(n0 + sr + 32 * i) * a.Kpad + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i) *reinterpret_cast<f32x4 *>(&sA[(sr + 32 * i) * GLD + sk]) = ra[i];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i) *reinterpret_cast<f32x4 *>(&sW[(sr + 32 * i) * GLD + sk]) = rw[i];
    };
    const int ch = n0 + wave * 16 + fi;
    const float sc_ = a.scale[ch], sh_ = a.shift[ch];
    const bool ch_ok = ch < a.Cout;

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    load_w(0);
    load_a(tile, 0);
    bool w_in_lds = false;
    while (tile < ntiles) {
        f32x4 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int sc = 0; sc < nsc; ++sc) {
            lds_barrier();   // the previous super-chunk's MFMAs are done with sA / sW
            store_a();
            if (!w_in_lds) store_w();
            if (w_resident) w_in_lds = true;
            lds_barrier();
            // the next super-chunk (or the next tile's first one): in flight during this super-chunk's MFMAs
            const bool last = sc + 1 == nsc;
            const int ntile = last ? tile + (int)gridDim.x : tile;
            const int nsc_i = last ? 0 : sc + 1;
            if (ntile < ntiles) {
                if (!w_resident) load_w(nsc_i);
                load_a(ntile, nsc_i);
            }
            for (int kk = 0; kk < GK; kk += 16) {
                const f32x4 bf = *reinterpret_cast<const f32x4 *>(&sW[(wave * 16 + fi) * GLD + kk + fq * 4]);
                f32x4 af[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const f32x4 *>(&sA[(m * 16 + fi) * GLD + kk + fq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][e], bf[e], acc[m], 0, 0, 0);
            }
        }
        // epilogue: C/D layout of the 16x16 MFMA -- column (channel) = lane & 15, rows = (lane >> 4) * 4 + reg.
        // (Writing a tile's results one tile LATER, so that their write round trip hides behind the next MFMAs -- loads and stores
        //  share the in-order vmcnt on gfx9, and the wait for the prefetched operands also waits for these stores -- needs 32 more
        //  registers, drops the kernel to one workgroup per CU and measured slower: 497 vs 389 us at 128 -> 128.)
        const int row0 = tile * GM;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + m * 16 + fq * 4 + r;
                float y = __builtin_fmaf(acc[m][r], sc_, sh_);
                if (a.relu) y = fmaxf(y, 0.f);
                const bool ok = ch_ok && row < a.rows;
                const size_t elem = (size_t)row * a.ldo + a.col0 + ch;
                // BUFFER stores, out-of-range offset for the lanes with nothing to write: unconditional instructions, so the number of
                // stores per tile is a compile-time constant and the wait for the next tile's operands (older than these stores;
                // loads and stores share gfx9's in-order vmcnt) is vmcnt(32), not vmcnt(0) -- the write round trip of a tile's results
                // no longer sits between its MFMAs and the next tile's
                if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orsrc, ok ? (unsigned)(elem * 4) : 0x80000000u, 0, 0);
                else if (ok) a.out[elem] = y;
            }
        tile += gridDim.x;
    }
}

// Used by launch_linear (mlp.hip) for DIRECT launches without pooling: true when the launch was taken.
bool gemm_stream_try(const LinearArgs &a, hipStream_t s, int *rc) {
    static const int enabled = [] { const char *e = getenv("G4D_GEMM_STREAM"); return e ? atoi(e) : 1; }();
    static const long long min_rows = [] { const char *e = getenv("G4D_GEMM_STREAM_MIN_ROWS"); return e ? atoll(e) : 65536ll; }();
    const int cpad = (a.Cout + 63) / 64 * 64;   // the packed weight / scale / shift are padded to 64 channels
    if (!enabled || a.pool != 0 || a.rows < min_rows || cpad % GN != 0 || a.Kpad > 128 || a.tab) return false;   // (tab: the interpolate-add epilogue lives in mlp.hip / gemm_tile.hip)
    static unsigned long long attr = 0;
    const int lds = (GM + GN) * GLD * (int)sizeof(float);
    static unsigned long long attr2 = 0;
    *rc = ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_stream_kernel<true>), lds, attr, "g4d_linear_f32(stream)");
    if (*rc) return true;
    *rc = ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_stream_kernel<false>), lds, attr2, "g4d_linear_f32(stream)");
    if (*rc) return true;
    const int ntiles = (a.rows + GM - 1) / GM;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int ncol = cpad / GN;
    int gx = 2 * cus / ncol > 0 ? 2 * cus / ncol : 1;   // two persistent workgroups per CU in total (70 KB of LDS each)
    if (gx > ntiles) gx = ntiles;
    const unsigned long long ob = (unsigned long long)a.rows * a.ldo * sizeof(float);
    static const int use_buf = [] { const char *e = getenv("G4D_GEMM_BUFFER_STORES"); return e ? atoi(e) : 1; }();
    if (use_buf && ob < 0x7fffffffull) hipLaunchKernelGGL(gemm_stream_kernel<true>, dim3((unsigned)gx, (unsigned)ncol), dim3(GT), lds, s, a, ntiles, (unsigned)ob);
    else hipLaunchKernelGGL(gemm_stream_kernel<false>, dim3((unsigned)gx, (unsigned)ncol), dim3(GT), lds, s, a, ntiles, 0u);
    *rc = check_launch("g4d_linear_f32(stream)");
    return true;
}

}  // namespace g4d
