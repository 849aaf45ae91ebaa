// Kernels of the garment skinning around the hot path (/root/reference/modules/mesh_encoder.py:337-390):
//   knn_blend_weights : inverse-distance blend of the K nearest body vertices' skinning weights (:339-347, :374-382)
//   spmm_axpy_rows    : one Jacobi smoothing step  W <- W + coeff * (adj . W)  over the garment mesh (:385-390, x100)
// Both HBM/L2-bound gathers with point-major rows; nothing is repeated/expanded the way the reference does it
// (`.repeat(1, 1, K, 1)` of the (V, J) weight table followed by torch.gather materialises B*T x Vg x K x J floats).
#include "g4d_common.h"

namespace g4d {

// out[f,v,:] = sum_k w_k * W[f, idx[c,v,k], :],  c = f / frames_per_clip,
// w_k = (1/d_k with inf -> 0) / sum, again inf -> 0   (the reference's two isinf fix-ups, :342-345)
// One wave per output row.  The K <= 256 (index, weight) pairs of the row live in registers (lane l holds pairs l, l+64,
// ...), are normalised once, and are handed out with v_readlane -- no dependent global load in the gather loop.  With
// J <= 32 joints the wave walks TWO neighbours per step (lanes 0-31 the even k, lanes 32-63 the odd k; a 24-joint row is
// 96 B, so a step gathers two rows) and the halves meet in one cross-lane add at the end: sum over even k + sum over odd k.
template <int HALVES>
__global__ void __launch_bounds__(256) knn_blend_weights_kernel(long long rows, int vg, int v, int K, int J, int frames_per_clip,
                                                               const float *__restrict__ W, const int *__restrict__ idx,
                                                               const float *__restrict__ dists, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long f = row / vg;
    const int gv = (int)(row - f * vg);
    const long long c = f / frames_per_clip;
    const int *ix = idx + ((size_t)c * vg + gv) * K;
    const float *dd = dists + ((size_t)c * vg + gv) * K;
    int iv[4];
    float wv[4];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = q * 64 + lane;
        iv[q] = k < K ? ix[k] : 0;
        float w = k < K ? 1.0f / dd[k] : 0.f;
        if (__builtin_isinf(w)) w = 0.f;
        wv[q] = w;
        s += w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float w = wv[q] / s;
        if (__builtin_isinf(w)) w = 0.f;
        wv[q] = w;
    }
    const float *Wf = W + (size_t)f * v * J;
    const int half = HALVES == 2 ? lane >> 5 : 0;
    const int j = HALVES == 2 ? lane & 31 : lane;
    float acc = 0.f;
    auto walk = [&](const int ivq, const float wvq, const int kend) {  // kend <= 0 past the end of the list
        for (int kk = 0; kk < kend; kk += HALVES) {
            int i;
            float w;
            if (HALVES == 2) {
                const int k1 = min(kk + 1, 63);
                const int i0 = __builtin_amdgcn_readlane(ivq, kk), i1 = __builtin_amdgcn_readlane(ivq, k1);
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wvq), kk));
                const float w1 = kk + 1 < kend ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wvq), k1)) : 0.f;
                i = half ? i1 : i0;
                w = half ? w1 : w0;
            } else {
                i = __builtin_amdgcn_readlane(ivq, kk);
                w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wvq), kk));
            }
            if (j < J) acc += Wf[(size_t)i * J + j] * w;
        }
    };
    walk(iv[0], wv[0], min(64, K));
    walk(iv[1], wv[1], min(64, K - 64));
    walk(iv[2], wv[2], min(64, K - 128));
    walk(iv[3], wv[3], min(64, K - 192));
    if (HALVES == 2) acc += __shfl_xor(acc, 32);
    if (lane < J) out[(size_t)row * J + lane] = acc;
}

// out[f,v,:] = S[f,v,:] + coeff * sum_u adj[v,u] * S[f,u,:]
__global__ void __launch_bounds__(256) spmm_axpy_rows_kernel(long long rows, int vg, int c, const float *__restrict__ S,
                                                            const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                            const float *__restrict__ vals, float coeff, float *__restrict__ out) {
    const int per_row = (c + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * per_row) return;
    const long long row = gid / per_row;
    const int c0 = (int)(gid - row * per_row) * 4;
    const long long f = row / vg;
    const int v = (int)(row - f * vg);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        const float a = vals[e];
        const float *src = S + ((size_t)f * vg + colidx[e]) * c + c0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < c) acc[j] += a * src[j];
    }
    const float *self = S + (size_t)row * c + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (c0 + j < c) out[(size_t)row * c + c0 + j] = self[j] + coeff * acc[j];
}

// ALL `iters` Jacobi steps  S <- S + coeff * (adj . S)  of the garment-weight smoothing (mesh_encoder.py:385-390: 100 torch.spmm
// + add round trips through HBM in the reference) in ONE launch: a workgroup owns a (frame, 4-column slab) -- the (vg x 4) slab
// lives in LDS, double buffered, for the whole iteration; a thread owns vg / 1024 vertices and keeps their adjacency rows
// (column, value pairs, <= kJDeg each: a quad mesh has 4-7 incl. the diagonal) in registers, so a step touches no global memory:
// kJDeg 16-byte LDS gathers + FMAs per vertex, one barrier.  HBM traffic = the slab once in, once out (2 * 4 * vg * c bytes per
// frame instead of 100 x that); arithmetic order = spmm_axpy_rows_kernel's (acc in CSR order, then self + coeff * acc), so the
// result is bit-identical to the step-by-step route.
constexpr int kJThreads = 1024, kJCols = 4, kJDeg = 8, kJVerts = 5;  // up to 5 * 1024 = 5120 vertices per workgroup

__global__ void __launch_bounds__(kJThreads) jacobi_smooth_kernel(int vg, int c, int iters, float coeff, const float *__restrict__ S,
                                                                 const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                                 const float *__restrict__ vals, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float jbuf[];  // [2][vg][4]
    const int t = threadIdx.x;
    const int slabs = (c + kJCols - 1) / kJCols;
    const int f = blockIdx.x / slabs, c0 = (blockIdx.x - f * slabs) * kJCols;
    const float *src = S + (size_t)f * vg * c;
    float *dst = out + (size_t)f * vg * c;
    float4 *cur = reinterpret_cast<float4 *>(jbuf), *nxt = cur + vg;
    // adjacency rows of the owned vertices -> registers (vertex v = t + i * 1024)
    int col[kJVerts][kJDeg];
    float val[kJVerts][kJDeg];
#pragma unroll
    for (int i = 0; i < kJVerts; ++i) {
        const int v = t + i * kJThreads;
        const int beg = v < vg ? rowptr[v] : 0, end = v < vg ? rowptr[v + 1] : 0;
#pragma unroll
        for (int e = 0; e < kJDeg; ++e) {
            const bool ok = beg + e < end;
            col[i][e] = ok ? colidx[beg + e] : 0;   // padding entries: value 0 at column 0 (adds +0 * x: x + 0.0 * y keeps x bit for bit
            val[i][e] = ok ? vals[beg + e] : 0.f;   //  unless y is inf / NaN -- weights are finite)
        }
        if (v < vg) {
            float4 x;
            x.x = c0 + 0 < c ? src[(size_t)v * c + c0 + 0] : 0.f; x.y = c0 + 1 < c ? src[(size_t)v * c + c0 + 1] : 0.f;
            x.z = c0 + 2 < c ? src[(size_t)v * c + c0 + 2] : 0.f; x.w = c0 + 3 < c ? src[(size_t)v * c + c0 + 3] : 0.f;
            cur[v] = x;
        }
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kJVerts; ++i) {
            const int v = t + i * kJThreads;
            if (v >= vg) break;
            float4 acc = {0.f, 0.f, 0.f, 0.f};
            const int deg = rowptr[v + 1] - rowptr[v];  // L1-resident; only the loop bound, the entries are in registers
#pragma unroll
            for (int e = 0; e < kJDeg; ++e) {
                if (e >= deg) break;
                const float4 y = cur[col[i][e]];
                const float a = val[i][e];
                acc.x += a * y.x; acc.y += a * y.y; acc.z += a * y.z; acc.w += a * y.w;
            }
            const float4 self = cur[v];
            nxt[v] = make_float4(self.x + coeff * acc.x, self.y + coeff * acc.y, self.z + coeff * acc.z, self.w + coeff * acc.w);
        }
        __syncthreads();
        float4 *tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int v = t; v < vg; v += kJThreads) {
        const float4 x = cur[v];
        if (c0 + 0 < c) dst[(size_t)v * c + c0 + 0] = x.x;
        if (c0 + 1 < c) dst[(size_t)v * c + c0 + 1] = x.y;
        if (c0 + 2 < c) dst[(size_t)v * c + c0 + 2] = x.z;
        if (c0 + 3 < c) dst[(size_t)v * c + c0 + 3] = x.w;
    }
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_knn_blend_weights_f32(int frames, int frames_per_clip, int vg, int v, int k, int j, const float *W, const int *idx,
                                         const float *dists, float *out, g4d_stream_t stream) {
    G4D_REQUIRE(frames >= 0 && frames_per_clip >= 1 && vg >= 0 && v > 0 && k >= 1 && k <= 256 && j >= 1 && j <= 64,
                "g4d_knn_blend_weights_f32: bad sizes (K <= 256, J <= 64)");
    const long long rows = (long long)frames * vg;
    if (rows == 0) return G4D_OK;
    G4D_REQUIRE(W && idx && dists && out, "g4d_knn_blend_weights_f32: null pointer");
    G4D_REQUIRE((rows + 3) / 4 < (1ll << 31), "g4d_knn_blend_weights_f32: too large");
    if (j <= 32)
        hipLaunchKernelGGL(knn_blend_weights_kernel<2>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           rows, vg, v, k, j, frames_per_clip, W, idx, dists, out);
    else
        hipLaunchKernelGGL(knn_blend_weights_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           rows, vg, v, k, j, frames_per_clip, W, idx, dists, out);
    return check_launch("g4d_knn_blend_weights_f32");
}

extern "C" int g4d_spmm_axpy_rows_f32(int frames, int vg, int c, const float *S, const int *rowptr, const int *colidx,
                                      const float *vals, float coeff, float *out, g4d_stream_t stream) {
    G4D_REQUIRE(frames >= 0 && vg >= 0 && c >= 0, "g4d_spmm_axpy_rows_f32: negative size");
    const long long rows = (long long)frames * vg;
    if (rows == 0 || c == 0) return G4D_OK;
    G4D_REQUIRE(S && rowptr && colidx && vals && out && S != out, "g4d_spmm_axpy_rows_f32: null or aliased pointer");
    const long long work = rows * ((c + 3) / 4);
    G4D_REQUIRE((work + 255) / 256 < (1ll << 31), "g4d_spmm_axpy_rows_f32: too large");
    hipLaunchKernelGGL(spmm_axpy_rows_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows,
                       vg, c, S, rowptr, colidx, vals, coeff, out);
    return check_launch("g4d_spmm_axpy_rows_f32");
}

// All `iters` smoothing steps in one launch (see jacobi_smooth_kernel).  Requirements: vg <= 5120 and every adjacency row holds
// <= 8 entries (checked on the HOST copy `max_row_entries` the caller passes: a quad / triangle garment mesh has 5-8 incl. the
// diagonal); otherwise G4D_EINVAL -- the caller falls back to `iters` x g4d_spmm_axpy_rows_f32.  S and out may alias.
extern "C" int g4d_jacobi_smooth_f32(int frames, int vg, int c, int iters, float coeff, int max_row_entries, const float *S,
                                     const int *rowptr, const int *colidx, const float *vals, float *out, g4d_stream_t stream) {
    G4D_REQUIRE(frames >= 0 && vg >= 0 && c >= 0 && iters >= 0, "g4d_jacobi_smooth_f32: negative size");
    if (frames == 0 || vg == 0 || c == 0) return G4D_OK;
    G4D_REQUIRE(S && rowptr && colidx && vals && out, "g4d_jacobi_smooth_f32: null pointer");
    G4D_REQUIRE(vg <= kJVerts * kJThreads && max_row_entries <= kJDeg,
                "g4d_jacobi_smooth_f32: needs vg <= %d and <= %d entries per adjacency row (vg=%d, max row=%d)", kJVerts * kJThreads, kJDeg, vg,
                max_row_entries);
    const size_t lds = (size_t)2 * vg * sizeof(float4);
    G4D_REQUIRE(lds <= 160 * 1024 - 512, "g4d_jacobi_smooth_f32: vg = %d does not fit the LDS-resident slab (max 5104)", vg);
    static unsigned long long attr = 0;  // one bit per device
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(jacobi_smooth_kernel), 160 * 1024 - 512, attr, "g4d_jacobi_smooth_f32")) return rc;
    const long long blocks = (long long)frames * ((c + kJCols - 1) / kJCols);
    G4D_REQUIRE(blocks < (1ll << 31), "g4d_jacobi_smooth_f32: too large");
    hipLaunchKernelGGL(jacobi_smooth_kernel, dim3((unsigned)blocks), dim3(kJThreads), lds, reinterpret_cast<hipStream_t>(stream), vg, c, iters, coeff, S,
                       rowptr, colidx, vals, out);
    return check_launch("g4d_jacobi_smooth_f32");
}
