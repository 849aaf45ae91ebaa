// Wide feature-propagation level with bf16 shared-MLP operands (BASELINE config 3: [384 + 192 -> 512 -> 256] over 61440 rows per 240-cloud
// call) as two tiled bf16 GEMMs behind an interpolation pre-pass -- the large-launch form of g4d_mlp_stack_bf16 in its interpolating mode
// (pointnet2_modules.py:127-156).  The LDS stack kernel keeps a 64-row tile's activations on chip and streams every layer's weights (850 KB)
// from L2 once per 64 rows with a barrier per layer: 358 us = 0.06 of the bf16 matrix pipe.  Here
//   1. interp_frag_bf16_kernel writes [three_interpolate(known features) ; skip features], rounded to bf16 (RNE, the rounding the stack
//      kernel applies when it fills its LDS tile), in MFMA A-FRAGMENT order: [16-row tile][32-column k-step][lane = (k / 8 % 4) * 16 + row % 16][8];
//   2. gemm_frag_bf16_kernel<true>: 128 x 128 block tile, a wave owns 64 x 64 (16 accumulator tiles), A and W fragments (both 1 KB
//      contiguous) copied L2 -> LDS by global_load_lds two k-steps at a time into a double buffer, one barrier per two k-steps; epilogue
//      affine + ReLU in fp32, rounded to bf16 and transposed through LDS into the NEXT layer's A-fragment order;
//   3. gemm_frag_bf16_kernel<false>: the same with an fp32 row-major epilogue (the level's output).
// Same operands, same rounding points, same v_mfma_f32_16x16x32_bf16 with k ascending per accumulator as the stack kernel: bit-identical
// results, so the route may be chosen by launch size.
#include <cstdlib>

#include "mlp_common.h"

namespace g4d {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt2(float lo, float hi) {   // RNE, lo -> bits [15:0] (NOT inline asm: see mlp_chain_bf16.hip)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
constexpr int GM = 128, GN = 128, KC = 2;       // block tile; k-steps (of 32) per staged chunk
constexpr int kFragU16 = 512;                   // a fragment = 64 lanes x 8 bf16 = 1 KB
constexpr int kBufU16 = (GM / 16 + GN / 16) * KC * kFragU16;   // one stage buffer: 16 A + 16 W fragments = 32 KB
}

// One thread = 8 consecutive columns of one row.  Rows beyond `rows` (the fragment buffer is padded to whole 128-row blocks) are zeros.
__global__ void __launch_bounds__(256) interp_frag_bf16_kernel(LinearArgs a, int kst, long long nthreads, unsigned short *__restrict__ x16) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    const int lane = (int)(t & 63);
    const long long frag = t >> 6;                      // (row tile, k-step)
    const int ks = (int)(frag % kst);
    const long long rt = frag / kst;
    const long long row = rt * 16 + (lane & 15);
    const int k0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < a.rows) {
        const RowCtx<LOAD_INTERP> c = make_ctx<LOAD_INTERP>(a, (int)row);
        if (k0 + 7 < a.C2) {                           // whole group inside the interpolated part (C2 % 8 == 0: launcher)
            const f32x4 f0a = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k0 + k0), f0b = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k0 + k0 + 4);
            const f32x4 f1a = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k1 + k0), f1b = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k1 + k0 + 4);
            const f32x4 f2a = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k2 + k0), f2b = *reinterpret_cast<const f32x4 *>(a.known_feats + c.k2 + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {               // load_elem<LOAD_INTERP>'s expression, operation by operation
                v[e] = c.w0 * f0a[e] + c.w1 * f1a[e] + c.w2 * f2a[e];
                v[4 + e] = c.w0 * f0b[e] + c.w1 * f1b[e] + c.w2 * f2b[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = load_elem<LOAD_INTERP>(a, c, (int)row, k0 + e);   // skip columns; zeros beyond K
        }
    }
    *reinterpret_cast<uint4 *>(x16 + t * 8) = make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7]));
}

struct GemmHArgs {
    int rows, kst, Cout, cpad, relu;
    const unsigned short *A;        // fragment order [row tile][kst][64][8], row tiles padded to whole 128-row blocks
    const unsigned short *W;        // fragment order [channel tile][kst][64][8] (PackedLayer.Wf16)
    const float *scale, *shift;
    unsigned short *out16; int kst_out;   // OUT_FRAG: the next layer's A operand, [row tile][kst_out][64][8]
    float *out; int ldo, col0;            // else: fp32 row-major
};

template <bool OUT_FRAG>
__global__ void __launch_bounds__(256, 2) gemm_frag_bf16_kernel(const GemmHArgs a, int nrow_blk, int ncol_blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned short h_smem[];   // [2][kBufU16]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int rb, cb;
    {   // XCD x takes row blocks x, x + 8, ... and runs all column blocks of a row block back to back (its L2 serves the A rows to all of them)
        const int b = blockIdx.x, x = b & 7, slot = b >> 3;
        rb = (slot / ncol_blk) * 8 + x;
        cb = slot % ncol_blk;
        if (rb >= nrow_blk) return;
    }
    const int fi = lane & 15, fq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int nchunk = a.kst / KC;
    const unsigned short *Ab = a.A + (size_t)rb * (GM / 16) * a.kst * kFragU16;
    const unsigned short *Wb = a.W + (size_t)cb * (GN / 16) * a.kst * kFragU16;
    // chunk -> stage buffer: slot s < 16: A fragment (row tile s / KC, k-step s % KC), s >= 16: W fragment likewise; each wave copies 8 slots
    auto stage = [&](int chunk) {
        unsigned short *dst = h_smem + (chunk & 1) * kBufU16;
        unsigned l8 = (unsigned)lane * 8u;
        asm volatile("" : "+v"(l8));   // (no hoisted copy-source addresses)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = wave * 8 + j;                 // wave-uniform
            const int tile = (s & 15) / KC, kk = (s & 15) % KC;
            const unsigned short *src = (s < 16 ? Ab : Wb) + ((size_t)tile * a.kst + chunk * KC + kk) * kFragU16 + l8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(dst + s * kFragU16), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage(0);
    for (int c = 0; c < nchunk; ++c) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // chunk c has landed (everybody's share) and nobody reads the other buffer any more
        if (c + 1 < nchunk) stage(c + 1);
        const unsigned short *buf = h_smem + (c & 1) * kBufU16;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            uint4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const uint4 *>(buf + (((wr * 4 + i) * KC + kk) * 64 + lane) * 8);
                bf[i] = *reinterpret_cast<const uint4 *>(buf + ((16 + (wc * 4 + i) * KC + kk) * 64 + lane) * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        }
    }
    // epilogue.  D layout: lane (fi, fq) holds rows 4 fq + r of channel 16 j' + fi
    const int row0 = rb * GM + wr * 64, n0 = cb * GN + wc * 64;
    if constexpr (OUT_FRAG) {
        __syncthreads();                                 // the stage buffers become the transpose scratch: [wave][64 rows][64 + 8] bf16
        constexpr int LDT = 72;
        unsigned short *scr = h_smem + wave * 64 * LDT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = n0 + j * 16 + fi;             // < cpad (launcher: the padded width is a multiple of 128)
            const float sc = a.scale[ch], sh = a.shift[ch];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = __builtin_fmaf(acc[i][j][r], sc, sh);
                    if (a.relu) y = fmaxf(y, 0.f);
                    scr[(i * 16 + fq * 4 + r) * LDT + j * 16 + fi] = (unsigned short)(cvt2(y, 0.f) & 0xffffu);   // channels beyond Cout: exact zeros (padded W, scale, shift)
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {             // the wave's 64 channels = two k-steps of the next layer
                const uint4 v = *reinterpret_cast<const uint4 *>(scr + (i * 16 + fi) * LDT + ks * 32 + fq * 8);
                const size_t rt = (size_t)(row0 >> 4) + i;
                const int kso = (n0 >> 5) + ks;
                if (kso < a.kst_out) *reinterpret_cast<uint4 *>(a.out16 + ((rt * a.kst_out + kso) * 64 + lane) * 8) = v;
            }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = n0 + j * 16 + fi;
            const int chc = min(ch, a.cpad - 1);
            const float sc = a.scale[chc], sh = a.shift[chc];
            const bool ch_ok = ch < a.Cout;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = __builtin_fmaf(acc[i][j][r], sc, sh);
                    if (a.relu) y = fmaxf(y, 0.f);
                    const int row = row0 + i * 16 + fq * 4 + r;
                    if (ch_ok && row < a.rows) a.out[(size_t)row * a.ldo + a.col0 + ch] = y;
                }
        }
    }
}

}  // namespace g4d

using namespace g4d;

// bf16 elements of a fragment-order buffer for `rows` rows of `kpad` columns (rows padded to whole 128-row blocks)
extern "C" long long g4d_frag_bf16_elems(long long rows, int kpad) {
    if (rows < 0 || kpad <= 0 || kpad % 32) return -1;
    return ((rows + GM - 1) / GM) * GM * (long long)kpad;
}

extern "C" int g4d_interp_concat_frag_bf16(int b, int n, int m, int C2, int C1, const float *known_feats, const float *skip, const float *dist2,
                                           const int *nn_idx, int kpad, unsigned short *x16, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && n > 0 && m > 0 && C2 > 0 && C1 >= 0 && kpad % 32 == 0 && kpad >= C2 + C1 && C2 % 8 == 0, "g4d_interp_concat_frag_bf16: bad sizes (C2 % 8 == 0, kpad % 32 == 0)");
    const long long rows = (long long)b * n;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(known_feats && dist2 && nn_idx && x16 && (skip || C1 == 0), "g4d_interp_concat_frag_bf16: null pointer");
    G4D_REQUIRE(rows < (1ll << 31) - 256 && (reinterpret_cast<size_t>(known_feats) & 15) == 0 && (reinterpret_cast<size_t>(x16) & 15) == 0,
                "g4d_interp_concat_frag_bf16: rows < 2^31, 16-byte aligned buffers");
    LinearArgs a = {};
    a.rows = (int)rows; a.K = C2 + C1; a.n = n; a.m = m; a.C2 = C2; a.C1 = C1; a.known_feats = known_feats; a.skip = skip; a.dist2 = dist2; a.nn_idx = nn_idx;
    const int kst = kpad / 32;
    const long long nthreads = ((rows + GM - 1) / GM) * (GM / 16) * (long long)kst * 64;
    hipLaunchKernelGGL(interp_frag_bf16_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, kst, nthreads, x16);
    return check_launch("g4d_interp_concat_frag_bf16");
}

extern "C" int g4d_gemm_frag_bf16(long long rows, int kpad, const unsigned short *A16, const unsigned short *W16, const float *scale, const float *shift, int relu,
                                  int Cout, unsigned short *out16, int kpad_out, float *out, int ldo, int col0, g4d_stream_t stream) {
    G4D_REQUIRE(rows >= 0 && rows < (1ll << 31) - 256 && kpad > 0 && kpad % (32 * KC) == 0 && Cout > 0, "g4d_gemm_frag_bf16: bad sizes (kpad %% 64 == 0)");
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(A16 && W16 && scale && shift && ((out16 != nullptr) != (out != nullptr)), "g4d_gemm_frag_bf16: null pointer / exactly one of out16, out");
    const int cpad = (Cout + 63) / 64 * 64;     // PackedLayer pads weights, scale and shift to 64 channels
    G4D_REQUIRE(cpad % GN == 0, "g4d_gemm_frag_bf16: the padded output width must be a multiple of 128");
    if (out16) G4D_REQUIRE(kpad_out % 32 == 0 && kpad_out >= Cout && kpad_out <= cpad, "g4d_gemm_frag_bf16: kpad_out must hold the outputs and lie inside the padded width");
    else G4D_REQUIRE(ldo >= col0 + Cout && col0 >= 0, "g4d_gemm_frag_bf16: output window out of range");
    GemmHArgs a;
    a.rows = (int)rows; a.kst = kpad / 32; a.Cout = Cout; a.cpad = cpad; a.relu = relu; a.A = A16; a.W = W16; a.scale = scale; a.shift = shift;
    a.out16 = out16; a.kst_out = kpad_out / 32; a.out = out; a.ldo = ldo; a.col0 = col0;
    const int lds = 2 * kBufU16 * (int)sizeof(unsigned short);   // 65536 bytes: two workgroups per CU
    static unsigned long long attr_t = 0, attr_f = 0;
    const int rc = out16 ? ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_frag_bf16_kernel<true>), lds, attr_t, "g4d_gemm_frag_bf16")
                         : ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_frag_bf16_kernel<false>), lds, attr_f, "g4d_gemm_frag_bf16");
    if (rc) return rc;
    const int nrow = (int)((rows + GM - 1) / GM), ncol = cpad / GN;
    const long long blocks = (long long)((nrow + 7) / 8) * 8 * ncol;
    G4D_REQUIRE(blocks < (1ll << 31), "g4d_gemm_frag_bf16: too many blocks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (out16) hipLaunchKernelGGL(gemm_frag_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, a, nrow, ncol);
    else hipLaunchKernelGGL(gemm_frag_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, a, nrow, ncol);
    return check_launch("g4d_gemm_frag_bf16");
}
