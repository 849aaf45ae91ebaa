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

// ---- squared distance under the three contraction contracts (include/g4d.h: G4D_CONTRACT_*) ---------------------------
// Every .hip file is built with -ffp-contract=off, so the only fused operations are the ones written here.
//   FM = 0  (x*x + y*y) + z*z, every product and sum rounded              (nvcc -fmad=false; the CPU-portable contract)
//   FM = 1  fma(z, z, fma(x, x, y*y))   the contraction of `x*x + y*y + z*z` by the LLVM / NVVM DAG combiner: of the two
//           products feeding the first add the LEFT one is fused (fold (fadd (fmul a, b), c) -> fma a, b, c), then the
//           third product is fused into the second add -- the reference's kernels as nvcc -O2 (fmad on) compiles them
//   FM = 2  fma(z, z, fma(y, y, x*x))   the accumulate-loop shape `d = 0; d += x*x; d += y*y; d += z*z` (pytorch3d /
//           chamferdist knn) -- also the alternative pairing for the expression above
// All three are monotone non-decreasing in |x|, |y|, |z| (each fp32 rounding is monotone), which is what the exact box
// pruning of fps_bucket.hip / ball_query.hip relies on: a box gap evaluated with the SAME FM bounds every point inside.
template <int FM>
__device__ __forceinline__ float dist2(float x, float y, float z) {
    if constexpr (FM == 0) return x * x + y * y + z * z;
    else if constexpr (FM == 1) return __builtin_fmaf(z, z, __builtin_fmaf(x, x, y * y));
    else return __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
}
typedef float g4d_f32x2 __attribute__((ext_vector_type(2)));
template <int FM>
__device__ __forceinline__ g4d_f32x2 dist2(g4d_f32x2 x, g4d_f32x2 y, g4d_f32x2 z) {  // two points on the packed fp32 pipe
    if constexpr (FM == 0) return x * x + y * y + z * z;
    else if constexpr (FM == 1) return __builtin_elementwise_fma(z, z, __builtin_elementwise_fma(x, x, y * y));
    else return __builtin_elementwise_fma(z, z, __builtin_elementwise_fma(y, y, x * x));
}

// ball_grid.hip: counting sort of a cloud into the cells of a uniform grid (cell edge 1.01 * rmax); `ws` holds, per cloud, a
// 64-byte header, the cell start table and the records (x, y, z, original index) in cell order
int grid_build(int b, int n, float rmax, const float *xyz, void *ws, hipStream_t st, int budget = 0);
void grid_sorted_layout(int n, size_t *offset_bytes, size_t *stride_bytes);   // where a cloud's (x, y, z, index) records in cell order sit in that workspace  // rmax <= 0: finest grid within `budget` cells
size_t grid_bytes_per_cloud(int n);
size_t grid_records_offset(int n);  // byte offset of the float4 records inside a cloud's workspace

long long tuning(const char *key, long long dflt);   // api.hip: run-time switch (g4d_tuning_set) > environment G4D_<KEY> > default
int distance_contraction();  // api.hip: the process-wide G4D_CONTRACT_* mode (0, 1 or 2)
inline int knn_shape(int mode) { return mode == 0 ? 0 : 2; }  // the accumulate loop contracts to the chain shape

// run `body` with a compile-time FM equal to the run-time `fm`
#define G4D_WITH_FM(fm, ...)                                   \
    switch (fm) {                                              \
        case 0: { constexpr int FM = 0; __VA_ARGS__; } break;  \
        case 1: { constexpr int FM = 1; __VA_ARGS__; } break;  \
        default: { constexpr int FM = 2; __VA_ARGS__; } break; \
    }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier: the compiler drains EVERY
// outstanding memory operation in front of it (s_waitcnt vmcnt(0) lgkmcnt(0)), global loads and stores included.  In a kernel that
// has global prefetches in flight across the barrier (operands of the next tile / slice) or has just issued its result stores, that
// puts an HBM round trip on the critical path at every barrier.  Kernels whose waves exchange data through LDS only use this one:
// the wave's own LDS operations are complete (lgkmcnt(0)) when it signals, global memory is left alone.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exchanges between lanes l and l ^ 16 / l ^ 32 on gfx950's v_permlane16_swap / v_permlane32_swap (VALU) instead of __shfl_xor's
// ds_bpermute (an LDS-pipe round trip + address arithmetic): the pooling epilogues of the MFMA kernels reduce across the four
// 16-lane rows of the C/D layout.  swap(a, a) leaves {row 0, row 0, row 2, row 2} / {row 1, row 1, row 3, row 3} (16) or
// {lower half twice} / {upper half twice} (32) in its two results (scripts/micro/permlane_swap.hip checks both against __shfl_xor).
__device__ __forceinline__ float lane_xor16(float x) {
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    return __uint_as_float((__lane_id() & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float x) {
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float((__lane_id() & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float max_xor16(float x) {   // max(x, x of lane ^ 16)
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float x) {
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

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
