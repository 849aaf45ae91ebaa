// Output stage shared by the register-chain MLP kernels (mlp_chain.hip fp32, mlp_chain_bf16.hip): the last layer's row-major
// accumulator tiles -- same 16x16 fp32 D layout for v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_bf16 -- are stored or pooled.
#pragma once
#include "mlp_common.h"

namespace g4d {

// Output stage of the row-major (last-layer) tiles: t[mt][r] = out[row0 + 16 mt + 4 fq + r][16 ct + fi].  Stores them, or pools
// them over S consecutive rows (max / mean); when a pooling group spans several waves (S > 16 MT) the wave's partial goes to
// xch[wave][channel] and finish_cross() completes the groups.
//
// In-kernel cycle stamps put 8-14k cycles per wave here when pool mode and S were run-time values: every channel tile went through
// a maze of wave-uniform branches (max AND mean computed, then selected; the S cases; guarded stores) in cold straight-line code.
// The mode is therefore resolved ONCE, outside the tile loop: PS = 0 stores, PS = 4 | 8 | 16 | 32 | 64 is a max pool over PS rows with
// everything folded at compile time, PS = -1 is the generic run-time path (mean pooling, other windows).
template <int MT, int PS>
__device__ __forceinline__ void finish_tile(const LinearArgs &a, int cout, int lane, int wave, int row0, int ct, const f32x4 (&t)[MT],
                                            float *xch, int xld) {
    const int fi = lane & 15, fq = lane >> 4;
    const int ch = ct * 16 + fi;
    const bool ch_ok = ch < cout;
    constexpr int R = 16 * MT;  // rows per wave
    if constexpr (PS == 0) {
        if (a.perm_rec) {   // wave-uniform: cell-ordered launch rows, outputs go to their original rows
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + mt * 16 + fq * 4 + r;
                    if (ch_ok && row < a.rows) a.out[(size_t)out_row(a, row) * a.ldo + a.col0 + ch] = t[mt][r];
                }
            return;
        }
        // one exec-mask region per tile instead of one per element: the row guard is needed in the last workgroup only
        float *o = a.out + (size_t)(row0 + fq * 4) * a.ldo + a.col0 + ch;
        if (row0 + R <= a.rows) {  // wave-uniform
            if (ch_ok) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(size_t)(mt * 16 + r) * a.ldo] = t[mt][r];
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ch_ok && row0 + mt * 16 + fq * 4 + r < a.rows) o[(size_t)(mt * 16 + r) * a.ldo] = t[mt][r];
        }
    } else if constexpr (PS > 0) {  // max pool, window PS
        float v[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) v[mt] = fmaxf(fmaxf(t[mt][0], t[mt][1]), fmaxf(t[mt][2], t[mt][3]));
        if constexpr (PS < 16) {  // 4 | 8 rows: 4 | 2 groups per tile
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float x = v[mt];
                if constexpr (PS == 8) x = max_xor16(x);
                const int first_row = row0 + mt * 16 + (PS == 8 ? (fq >> 1) * 8 : fq * 4);
                const bool writer = PS == 8 ? (fq & 1) == 0 : true;
                if (writer && ch_ok && first_row < a.rows) a.out[(size_t)(first_row / PS) * a.ldo + a.col0 + ch] = x;
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {  // the 16 rows of a tile: across the four fq groups
                v[mt] = max_xor32(max_xor16(v[mt]));
            }
            if constexpr (PS <= R) {
                constexpr int TPG = PS / 16;  // tiles per group
#pragma unroll
                for (int g = 0; g < MT / TPG; ++g) {
                    float x = v[g * TPG];
#pragma unroll
                    for (int tt = 1; tt < TPG; ++tt) x = fmaxf(x, v[g * TPG + tt]);
                    const int first_row = row0 + g * PS;
                    if (lane < 16 && ch_ok && first_row < a.rows) a.out[(size_t)(first_row / PS) * a.ldo + a.col0 + ch] = x;
                }
            } else {  // a group spans PS / R consecutive waves of the workgroup: partials meet in LDS
                float x = v[0];
#pragma unroll
                for (int tt = 1; tt < MT; ++tt) x = fmaxf(x, v[tt]);
                if (lane < 16) xch[wave * xld + ch] = x;
            }
        }
    } else {  // generic: run-time pool mode and window
        const bool is_max = a.pool == 1;
        const float inv = is_max ? 1.f : 1.f / (float)a.S;
        const int S = a.S;
        float v[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            v[mt] = is_max ? fmaxf(fmaxf(t[mt][0], t[mt][1]), fmaxf(t[mt][2], t[mt][3])) : ((t[mt][0] + t[mt][1]) + (t[mt][2] + t[mt][3]));
        if (S < 16) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float x = v[mt];
                if (S == 8) {
                    const float y = lane_xor16(x);
                    x = is_max ? fmaxf(x, y) : x + y;
                }
                const int first_row = row0 + mt * 16 + (S == 8 ? (fq >> 1) * 8 : fq * 4);
                const bool writer = S == 8 ? (fq & 1) == 0 : true;
                if (writer && ch_ok && first_row < a.rows) a.out[(size_t)(first_row / S) * a.ldo + a.col0 + ch] = x * inv;
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float y = lane_xor16(v[mt]);
            v[mt] = is_max ? fmaxf(v[mt], y) : v[mt] + y;
            const float z = lane_xor32(v[mt]);
            v[mt] = is_max ? fmaxf(v[mt], z) : v[mt] + z;
        }
        if (S <= R) {
            const int tiles_per_group = S >> 4;  // 1 | 2 | 4
#pragma unroll
            for (int g = 0; g < MT; ++g) {
                if (g * tiles_per_group >= MT) break;
                float x = v[g * tiles_per_group];
#pragma unroll
                for (int tt = 1; tt < MT; ++tt)
                    if (tt < tiles_per_group) x = is_max ? fmaxf(x, v[g * tiles_per_group + tt]) : x + v[g * tiles_per_group + tt];
                const int first_row = row0 + g * S;
                if (lane < 16 && ch_ok && first_row < a.rows) a.out[(size_t)(first_row / S) * a.ldo + a.col0 + ch] = x * inv;
            }
        } else {
            float x = v[0];
#pragma unroll
            for (int tt = 1; tt < MT; ++tt) x = is_max ? fmaxf(x, v[tt]) : x + v[tt];
            if (lane < 16) xch[wave * xld + ch] = x;
        }
    }
}

// second half of the pooling when a group spans waves; wave-uniform, every wave of the workgroup calls it
template <int MT>
__device__ __forceinline__ void finish_cross(const LinearArgs &a, int cout, int lane, int wave, int row0, const float *xch, int xld) {
    constexpr int R = 16 * MT;
    if (a.pool == 0 || a.S <= R) return;
    const bool is_max = a.pool == 1;
    const float inv = is_max ? 1.f : 1.f / (float)a.S;
    __syncthreads();
    const int span = a.S / R;  // 2 | 4 waves per group
    if ((wave % span) == 0) {
        for (int ch = lane; ch < xld; ch += 64) {
            float x = xch[wave * xld + ch];
            for (int w = 1; w < span; ++w) {
                const float y = xch[(wave + w) * xld + ch];
                x = is_max ? fmaxf(x, y) : x + y;
            }
            if (ch < cout && row0 < a.rows) a.out[(size_t)(row0 / a.S) * a.ldo + a.col0 + ch] = x * inv;
        }
    }
}

// all tiles of the last layer; the pool mode / window is dispatched here, once, to a fully specialised tile loop
template <int TOUT, int MT>
__device__ __forceinline__ void finish(const LinearArgs &a, int cout, int lane, int wave, int row0, f32x4 (&acc)[TOUT][MT], float *xch) {
#define G4D_FIN(PSV)                                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int ct = 0; ct < TOUT; ++ct) finish_tile<MT, PSV>(a, cout, lane, wave, row0, ct, acc[ct], xch, TOUT * 16); \
    }
    if (a.pool == 0) G4D_FIN(0)
    else if (a.pool == 1 && a.S == 32) G4D_FIN(32)
    else if (a.pool == 1 && a.S == 16) G4D_FIN(16)
    else if (a.pool == 1 && a.S == 64) G4D_FIN(64)
    else if (a.pool == 1 && a.S == 8) G4D_FIN(8)
    else if (a.pool == 1 && a.S == 4) G4D_FIN(4)
    else G4D_FIN(-1)
#undef G4D_FIN
    finish_cross<MT>(a, cout, lane, wave, row0, xch, TOUT * 16);
}

}  // namespace g4d
