// Y = act((X . W^T) * scale + shift) for TALL, NARROW direct launches: K == Cout == 96 (the first-layer table of the encoder's second
// set-abstraction level: 245760 source points x [32 + 64 features] -> [32 + 64 first-layer channels], pytorch_utils.py:5-32 as one GEMM; round 6).
//
// Why: gemm_stream.hip pads this shape to 128 channels (two of its eight waves compute padding) and to two 64-deep super-chunks (the second
// half empty): 0.56 of its MFMAs are useful and the launch ran at 0.34 of the fp32 MFMA peak (86 us; the shape moves 189 MB: 24 us at 8 TB/s,
// 29 us of matrix pipe).  Here the whole 96 x 96 weight matrix lives in LDS in MFMA operand (fragment) order, loaded once per workgroup (36 KB: four
// workgroups per CU; held in registers instead -- 144 VGPRs, two waves per SIMD -- one tile of prefetch did not cover the memory round trip: 74 us), waves
// are persistent and autonomous (no LDS tile, no barrier in the loop), a wave takes 16 rows at a time -- its six 16-byte row pieces per lane are the
// MFMA operands as they arrive -- with the next tile's rows in flight behind the current tile's 144 MFMAs, and every lane stores 16 bytes per
// channel tile (the MFMAs run with A = weights, B = activations: lane (row fi, fq) ends with channels 16 ct + 4 fq + r).
// Products, k order (within a 16-wide k-step: k = 4 fq + e, fq inside the MFMA, e across the four MFMAs; k-steps ascending) and epilogue
// arithmetic are those of linear_kernel / gemm_stream / gemm_tile: bit-identical results.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

// NK 16-wide k-steps, NC 16-channel tiles (K = 16 NK, Cout = 16 NC), NW waves per workgroup, OCC workgroups per CU (by LDS)
template <int NK, int NC, int NW, int OCC>
__global__ void __launch_bounds__(64 * NW, OCC) gemm_narrow_kernel(const LinearArgs a, int ntile) {
    extern __shared__ __attribute__((aligned(16))) float gn_smem[];
    float *s_w = gn_smem;                            // [ct][ks][lane][4]
    float *s_sc = gn_smem + NC * NK * 256, *s_sh = s_sc + 16 * NC;
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * NC; i += 64 * NW) { s_sc[i] = a.scale[i]; s_sh[i] = a.shift[i]; }
    const int lane = tid & 63, fi = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
    // weights: A fragment of (channel tile ct, k-step ks): lane l = (channel fi, k = 16 ks + 4 fq + e) -> s_w[((ct NK + ks) 64 + l) 4 + e]
    for (int i = tid; i < NC * NK * 64; i += 64 * NW) {
        const int frag = i >> 6, l = i & 63, ct = frag / NK, ks = frag - ct * NK;
        reinterpret_cast<f32x4 *>(s_w)[i] = *reinterpret_cast<const f32x4 *>(a.W + (size_t)(ct * 16 + (l & 15)) * a.Kpad + ks * 16 + (l >> 4) * 4);
    }
    __syncthreads();
    auto load_x = [&](int tile, f32x4 (&x)[NK]) {
        const float *p = a.X + (size_t)min(min(tile, ntile - 1) * 16 + fi, a.rows - 1) * a.ldx + fq * 4;   // past the end: clamped, never stored
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) x[ks] = *reinterpret_cast<const f32x4 *>(p + ks * 16);
    };
    f32x4 xa[NK], xb[NK];
    auto do_tile = [&](int tile, const f32x4 (&x)[NK], f32x4 (&nxt)[NK]) {
        load_x(tile + nwaves, nxt);                  // the next tile's rows: in flight behind this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w + ((ct * NK + ks) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], x[ks][e], acc[ct], 0, 0, 0);
            }
        const int row = tile * 16 + fi;
        float *o = a.out + (size_t)min(row, a.rows - 1) * a.ldo + a.col0 + fq * 4;
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_sc + ct * 16 + fq * 4), sh = *reinterpret_cast<const f32x4 *>(s_sh + ct * 16 + fq * 4);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = __builtin_fmaf(acc[ct][r], sc[r], sh[r]);
                if (a.relu) y[r] = fmaxf(y[r], 0.f);
            }
            if (row < a.rows) *reinterpret_cast<f32x4 *>(o + ct * 16) = y;
        }
    };
    int tile = wg;
    if (tile >= ntile) return;
    load_x(tile, xa);
    for (; tile < ntile; tile += 2 * nwaves) {       // tiles in pairs: the two row buffers swap roles (no copy behind the prefetch)
        do_tile(tile, xa, xb);
        if (tile + nwaves < ntile) do_tile(tile + nwaves, xb, xa);
    }
}

template <int NK, int NC, int NW, int OCC>
static int gemm_narrow_launch(const LinearArgs &a, hipStream_t s) {
    const int lds = (int)sizeof(float) * (NC * NK * 256 + 32 * NC);
    static unsigned long long attr = 0;
    if (lds > 64 * 1024) {
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_narrow_kernel<NK, NC, NW, OCC>), lds, attr, "g4d_linear_f32(narrow)")) return rc;
    }
    const int ntile = (a.rows + 15) / 16;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long want = ((long long)ntile + NW - 1) / NW, cap = (long long)OCC * cus;   // persistent: OCC workgroups per CU
    hipLaunchKernelGGL((gemm_narrow_kernel<NK, NC, NW, OCC>), dim3((unsigned)(want < cap ? want : cap)), dim3(64 * NW), lds, s, a, ntile);
    return check_launch("g4d_linear_f32(narrow)");
}

// Used by launch_linear (mlp.hip) ahead of the wider GEMM forms: true when the launch was taken.
bool gemm_narrow_try(const LinearArgs &a, hipStream_t s, int *rc) {
    static const int enabled = [] { const char *e = getenv("G4D_GEMM_NARROW"); return e ? atoi(e) : 1; }();   // A/B switch
    if (!enabled || a.pool != 0 || a.tab || a.rows < 32768 || a.K != a.Kpad || (a.ldx & 3) || (a.ldo & 3) || (a.col0 & 3) ||
        (reinterpret_cast<size_t>(a.X) & 15) || (reinterpret_cast<size_t>(a.out) & 15))
        return false;
    // 96 -> 96 (SA level 2's first-layer table): 36 KB of weights, four workgroups per CU.  64 -> 32 (the first block of the segmentation head when
    // pytorch_utils.Conv1d is called on its own -- the drop-in module route: 1.97 M rows, 226 us on linear_kernel): 8 KB, eight workgroups per CU.
    // (192 -> 192 -- SA level 3's table, 144 KB of weights, one 8-wave workgroup per CU at 256 registers -- measured 58 us against linear_kernel's
    //  56, and as two column blocks of 96 channels with 72 KB each 56: at 61440 rows a wave gets four tiles, the start-up copy is the launch.  Not
    //  instantiated.)
    if (a.K == 96 && a.Cout == 96) { *rc = gemm_narrow_launch<6, 6, 4, 4>(a, s); return true; }
    if (a.K == 64 && a.Cout == 32) { *rc = gemm_narrow_launch<4, 2, 4, 8>(a, s); return true; }
    return false;
}

}  // namespace g4d
