// Temporal attention of the refinement rounds (/root/reference/modules/mesh_encoder.py:467-476): per clip,
//   q, k, v = chunk(temporal_qkv(last_feat), 3)            each (T, Vg*C), flattened over the garment vertices
//   att = softmax(q k^T / sqrt(T))                          (T, T)
//   out = att v                                             (T, Vg*C)
// T is the clip length (30), D = Vg*C is 524288 at Vg = 4096: two skinny contractions that read 3 * T * D floats and do
// almost no arithmetic -- HBM-bound.  (A library GEMM called with M = N = 30, K = 524288 ran at 340 GB/s here.)
//   scores_partial : each wave owns a slice of D, streams q / k rows straight from the (F, Vg, 3C) qkv buffer into MFMA
//                    A / B fragments (16 consecutive channels of one vertex = one float4 per lane), accumulates a 32x32
//                    tile pair on v_mfma_f32_16x16x4_f32, and writes its partial (deterministic: no atomics);
//   scores_softmax : one workgroup per clip sums the partials in a fixed order, scales, soft-maxes the rows;
//   mix            : one thread per (vertex, channel) column reads the T values of v once, forms the T outputs with the
//                    (uniform) attention weights and writes them into the caller's feature buffer at a column offset
//                    (the torch.cat of :476 never materialises).
#include "g4d_common.h"

namespace g4d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kAttKsteps = 32;  // 16-wide k-steps per wave in scores_partial
constexpr int kAttMaxT = 32;

__global__ void __launch_bounds__(256) att_scores_partial_kernel(int T, int vg, int C, const float *__restrict__ qkv, float *__restrict__ partial,
                                                                int slices) {
    const int lane = threadIdx.x & 63, fi = lane & 15, fq = lane >> 4;
    const int slice = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = blockIdx.y;
    const bool active = slice < slices;   // inactive waves keep zero tiles and still join the reduction
    const long long total_ksteps = (long long)vg * C / 16;
    const long long s0 = (long long)slice * kAttKsteps;
    const int t0 = min(fi, T - 1), t1 = min(16 + fi, T - 1);
    const size_t ld = (size_t)vg * 3 * C;  // floats per frame
    const float *base = qkv + (size_t)c * T * ld + fq * 4;
    const float *q0 = base + (size_t)t0 * ld, *q1 = base + (size_t)t1 * ld;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < kAttKsteps && active; ++s) {
        const long long ks = s0 + s;
        if (ks >= total_ksteps) break;
        const long long d = ks * 16;
        const size_t off = (size_t)(d / C) * 3 * C + (size_t)(d % C);
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(q0 + off), a1 = *reinterpret_cast<const f32x4 *>(q1 + off);
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(q0 + off + C), b1 = *reinterpret_cast<const f32x4 *>(q1 + off + C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
        }
    }
    // the 4 waves of the workgroup meet in LDS: one partial tile per WORKGROUP (4x fewer to reduce in the softmax kernel)
    __shared__ float red[4][kAttMaxT * kAttMaxT];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(rt * 16 + fq * 4 + r) * kAttMaxT + ct * 16 + fi] = acc[rt][ct][r];
    __syncthreads();
    float *p = partial + ((size_t)c * gridDim.x + blockIdx.x) * (kAttMaxT * kAttMaxT);
    for (int i = threadIdx.x; i < kAttMaxT * kAttMaxT; i += 256) p[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

__global__ void __launch_bounds__(1024) att_scores_softmax_kernel(int T, int slices, const float *__restrict__ partial, float *__restrict__ att) {
    __shared__ float sc[kAttMaxT][kAttMaxT + 1];
    const int t = threadIdx.x >> 5, u = threadIdx.x & 31;
    const int c = blockIdx.x;
    const float *p = partial + (size_t)c * slices * (kAttMaxT * kAttMaxT) + threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains: the loads of 4 partials are in flight together
    int w = 0;
    for (; w + 3 < slices; w += 4) {
        s0 += p[(size_t)w * (kAttMaxT * kAttMaxT)];
        s1 += p[(size_t)(w + 1) * (kAttMaxT * kAttMaxT)];
        s2 += p[(size_t)(w + 2) * (kAttMaxT * kAttMaxT)];
        s3 += p[(size_t)(w + 3) * (kAttMaxT * kAttMaxT)];
    }
    for (; w < slices; ++w) s0 += p[(size_t)w * (kAttMaxT * kAttMaxT)];
    const float s = (s0 + s1) + (s2 + s3);
    sc[t][u] = s / (float)sqrt((double)T);  // qk / np.sqrt(T)
    __syncthreads();
    if (t < T && u < T) {
        float m = sc[t][0];
        for (int j = 1; j < T; ++j) m = fmaxf(m, sc[t][j]);
        float z = 0.f;
        for (int j = 0; j < T; ++j) z += expf(sc[t][j] - m);
        att[((size_t)c * T + t) * T + u] = expf(sc[t][u] - m) / z;
    }
}

// One thread per FOUR consecutive channels of a vertex (C % 16 == 0): 16-byte loads of v -- a wave reads 1 KB contiguous per frame instead of
// 256 B -- and 16-byte stores (dword-aligned: the caller's column offset need not be a multiple of 4).  Round 4: 481 -> ~200 us at 8 clips x
// 30 frames x 4096 vertices x 128 channels (1 GB moved).  Same products, same summation order per output as before.
typedef float f32x4u_att __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256) att_mix_kernel(int T, int vg, int C, const float *__restrict__ qkv, const float *__restrict__ att,
                                                     float *__restrict__ out, int ldo, int col0) {
    const long long col4 = (long long)blockIdx.x * 256 + threadIdx.x;  // (vertex, group of 4 channels)
    const int c = blockIdx.y;
    const int c4 = C >> 2;
    if (col4 >= (long long)vg * c4) return;
    const int v = (int)(col4 / c4), ch = (int)(col4 - (long long)v * c4) * 4;
    const size_t ld = (size_t)vg * 3 * C;
    const float *vp = qkv + (size_t)c * T * ld + (size_t)v * 3 * C + 2 * C + ch;
    f32x4 val[kAttMaxT];
#pragma unroll
    for (int u = 0; u < kAttMaxT; ++u) val[u] = u < T ? (f32x4)*reinterpret_cast<const f32x4u_att *>(vp + (size_t)u * ld) : (f32x4){0.f, 0.f, 0.f, 0.f};   // dword-aligned 16-byte loads: a qkv view at any 4-byte offset stays valid
    const float *a = att + (size_t)c * T * T;
    for (int t = 0; t < T; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kAttMaxT; ++u)
            if (u < T) {
                const float w = a[t * T + u];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(w, val[u][e], acc[e]);
            }
        *reinterpret_cast<f32x4u_att *>(out + ((size_t)(c * T + t) * vg + v) * ldo + col0 + ch) = acc;
    }
}

}  // namespace g4d

extern "C" size_t g4d_temporal_attention_scratch_floats(int nclips, int vg, int c) {
    using namespace g4d;
    const long long ksteps = (long long)vg * c / 16;
    const long long slices = (ksteps + kAttKsteps - 1) / kAttKsteps;
    return (size_t)nclips * slices * kAttMaxT * kAttMaxT;
}

extern "C" int g4d_temporal_attention_f32(int nclips, int t, int vg, int c, const float *qkv, float *scratch, float *att, float *out, int ldo,
                                          int col0, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(nclips >= 0 && t >= 1 && t <= kAttMaxT && vg >= 0 && c > 0 && c % 16 == 0,
                "g4d_temporal_attention_f32: need 1 <= T <= %d and C %% 16 == 0", kAttMaxT);
    if (nclips == 0 || vg == 0) return G4D_OK;
    G4D_REQUIRE(qkv && scratch && att && out && ldo >= col0 + c && col0 >= 0 && nclips <= 65535, "g4d_temporal_attention_f32: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long ksteps = (long long)vg * c / 16;
    const int slices = (int)((ksteps + kAttKsteps - 1) / kAttKsteps);
    hipLaunchKernelGGL(att_scores_partial_kernel, dim3((slices + 3) / 4, nclips), dim3(256), 0, st, t, vg, c, qkv, scratch, slices);
    hipLaunchKernelGGL(att_scores_softmax_kernel, dim3(nclips), dim3(1024), 0, st, t, (slices + 3) / 4, scratch, att);
    hipLaunchKernelGGL(att_mix_kernel, dim3((unsigned)(((long long)vg * (c / 4) + 255) / 256), nclips), dim3(256), 0, st, t, vg, c, qkv, att, out, ldo,
                       col0);
    return check_launch("g4d_temporal_attention_f32");
}
