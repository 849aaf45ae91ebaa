// Shared device/host helpers for libg4d_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/g4d.h"

namespace g4d {

void set_error(const char *fmt, ...);

// Every launcher ends with this: report, never exit (the reference exits: sampling_gpu.cu:248-252).
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return G4D_OK;
}

#define G4D_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            g4d::set_error(__VA_ARGS__); \
            return G4D_EINVAL;          \
        }                               \
    } while (0)

constexpr int kWave = 64;

// Raise a kernel's dynamic-LDS limit above the 64 KB default.  The attribute is PER DEVICE: `done` holds one bit per device
// ordinal so that a process driving several GPUs sets it on each of them (idempotent, so a race between host threads is benign).
inline int ensure_dynamic_lds(const void *kernel, int bytes, unsigned long long &done, const char *what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (__atomic_load_n(&done, __ATOMIC_RELAXED) & bit)) return G4D_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return (int)e;
    }
    if (dev < 64) __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
    return G4D_OK;
}

// ---- wave-level primitives (DPP, no LDS) ------------------------------------------------------
// dpp_ctrl encodings: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f(float v) {
    // old = v: lanes whose source is outside the row / masked keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// max over the 64 lanes, returned wave-uniform (SGPR).  No NaNs expected in v.
// Hand-written DPP chain: hipcc does not fold the DPP move into v_max_f32 (it emits mov_dpp + a
// canonicalising max + max + copy + s_nop per step); here each step is ONE v_max_f32_dpp.  Lanes whose DPP
// source is outside the row are write-disabled (bound_ctrl off) and keep their value.  `s_nop 1` = the two
// wait states a DPP read needs after a VALU write of the same VGPR.
__device__ __forceinline__ float wave_max_f32(float v) {
    int out;
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %1, %0, 63\n\t"
        "s_nop 3"
        : "+v"(v), "=s"(out));
    return __int_as_float(out);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, dpp_mov_u<0x111, 0xf>(v));
    v = min(v, dpp_mov_u<0x112, 0xf>(v));
    v = min(v, dpp_mov_u<0x114, 0xf>(v));
    v = min(v, dpp_mov_u<0x118, 0xf>(v));
    v = min(v, dpp_mov_u<0x142, 0xa>(v));
    v = min(v, dpp_mov_u<0x143, 0xc>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

}  // namespace g4d
