// Two small HBM-bound kernels of the model around the hot path:
//   segment_select / segment_take : `PCAGarmentEncoderSeg.calc_segmentation_results`
//       (/root/reference/modules/mesh_encoder.py:109-125): per frame, the points whose arg-max class is the garment's,
//       in index order, first n_out of them, zero padded.  The reference runs a Python loop over frames with boolean
//       indexing (one device sync per frame); here one workgroup per frame does an ordered compaction with ballot
//       prefix sums, then a row gather writes coordinates and point-major features.
//   vertex_normals : `compute_vnorms` (/root/reference/utils/mesh_utils.py:116-134): unit face normals (norm clamped at
//       1e-6) summed per vertex over the incident faces (torch_scatter sum; order = the CSR order given by the caller),
//       re-normalised with the same clamp.  One thread per (frame, vertex); faces are recomputed per incident vertex
//       (3x the flops, no (F, nf, 3) intermediate and no atomics -> deterministic).
#include "g4d_common.h"

namespace g4d {

__global__ void __launch_bounds__(1024) segment_select_kernel(int n, int classes, int target, int n_out, const float *__restrict__ logits_all,
                                                             int *__restrict__ sel_all, int *__restrict__ counts) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int f = blockIdx.x;
    const float *logits = logits_all + (size_t)f * n * classes;
    int *sel = sel_all + (size_t)f * n_out;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + t;
        bool hit = false;
        if (k < n) {
            const float *l = logits + (size_t)k * classes;
            float best = l[0];
            int arg = 0;
            for (int c = 1; c < classes; ++c) {  // first maximum wins (torch.argmax)
                const float v = l[c];
                if (v > best || (v != v && best == best)) { best = v; arg = c; }  // NaN counts as the maximum, like torch
            }
            hit = arg == target;
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) wave_cnt[wave] = __builtin_popcountll(mask);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        const int slot = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (hit && slot < n_out) sel[slot] = k;
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            base_s += tot;
        }
        __syncthreads();
    }
    const int total = base_s;
    for (int j = total + t; j < n_out; j += 1024) sel[j] = -1;
    if (t == 0 && counts) counts[f] = total;
}

// out[f,j,:] = sel[f,j] >= 0 ? in[f,sel[f,j],:] : 0
__global__ void __launch_bounds__(256) segment_take_kernel(long long total, int n, int n_out, int c, const float *__restrict__ in,
                                                          const int *__restrict__ sel, float *__restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const long long row = gid / c;
    const int ch = (int)(gid - row * c);
    const long long f = row / n_out;
    const int s = sel[row];
    out[gid] = s >= 0 ? in[((size_t)f * n + s) * c + ch] : 0.f;
}

__device__ __forceinline__ void unit_clamped(float &x, float &y, float &z) {
    float d = sqrtf(x * x + y * y + z * z);
    d = fmaxf(d, 1.e-6f);
    x /= d; y /= d; z /= d;
}

__global__ void __launch_bounds__(256) vertex_normals_kernel(long long total, int v, const float *__restrict__ verts_all,
                                                            const int *__restrict__ faces, const int *__restrict__ rowptr,
                                                            const int *__restrict__ fid, float *__restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const long long f = gid / v;
    const int vid = (int)(gid - f * v);
    const float *verts = verts_all + (size_t)f * v * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int e = rowptr[vid]; e < rowptr[vid + 1]; ++e) {
        const int *tri = faces + (size_t)fid[e] * 3;
        const float *p0 = verts + (size_t)tri[0] * 3, *p1 = verts + (size_t)tri[1] * 3, *p2 = verts + (size_t)tri[2] * 3;
        const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;  // cross(e01, e02)
        unit_clamped(nx, ny, nz);
        sx += nx; sy += ny; sz += nz;
    }
    unit_clamped(sx, sy, sz);
    out[gid * 3 + 0] = sx; out[gid * 3 + 1] = sy; out[gid * 3 + 2] = sz;
}

}  // namespace g4d

extern "C" int g4d_segment_select_f32(int frames, int n, int classes, int target, int n_out, const float *logits, int *sel, int *counts,
                                      g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(frames >= 0 && n >= 0 && classes >= 1 && n_out >= 0, "g4d_segment_select_f32: bad sizes");
    if (frames == 0 || n_out == 0) return G4D_OK;
    G4D_REQUIRE(sel && (logits || n == 0), "g4d_segment_select_f32: null pointer");
    hipLaunchKernelGGL(segment_select_kernel, dim3(frames), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), n, classes, target, n_out,
                       logits, sel, counts);
    return check_launch("g4d_segment_select_f32");
}

extern "C" int g4d_segment_take_f32(int frames, int n, int n_out, int c, const float *in, const int *sel, float *out, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(frames >= 0 && n >= 0 && n_out >= 0 && c >= 0, "g4d_segment_take_f32: bad sizes");
    const long long total = (long long)frames * n_out * c;
    if (total == 0) return G4D_OK;
    G4D_REQUIRE(sel && out && (in || n == 0), "g4d_segment_take_f32: null pointer");
    hipLaunchKernelGGL(segment_take_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), total, n,
                       n_out, c, in, sel, out);
    return check_launch("g4d_segment_take_f32");
}

extern "C" int g4d_vertex_normals_f32(int frames, int v, const float *verts, const int *faces, const int *vf_rowptr, const int *vf_fid,
                                      float *out, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(frames >= 0 && v >= 0, "g4d_vertex_normals_f32: bad sizes");
    const long long total = (long long)frames * v;
    if (total == 0) return G4D_OK;
    G4D_REQUIRE(verts && faces && vf_rowptr && vf_fid && out, "g4d_vertex_normals_f32: null pointer");
    hipLaunchKernelGGL(vertex_normals_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), total, v,
                       verts, faces, vf_rowptr, vf_fid, out);
    return check_launch("g4d_vertex_normals_f32");
}

// Interpenetration penalty of `calc_interpenetration_loss` (/root/reference/smplx/loss/temporal_loss.py:20-46), forward:
// per garment vertex g with nearest body vertex b (index from the nearest-neighbour search) and its normal n,
//   pen = relu(-(n . (g - b)))     -- positive when the garment vertex lies behind the body surface.
namespace g4d {
__global__ void __launch_bounds__(256) interpenetration_kernel(long long total, int vg, int v, const float *__restrict__ garment,
                                                              const float *__restrict__ body, const float *__restrict__ normals,
                                                              const int *__restrict__ nn_idx, int idx_stride, float *__restrict__ pen) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const long long f = gid / vg;
    const size_t b = (size_t)f * v + nn_idx[gid * idx_stride];
    const float dx = garment[gid * 3 + 0] - body[b * 3 + 0], dy = garment[gid * 3 + 1] - body[b * 3 + 1], dz = garment[gid * 3 + 2] - body[b * 3 + 2];
    const float d = normals[b * 3 + 0] * dx + normals[b * 3 + 1] * dy + normals[b * 3 + 2] * dz;  // torch.mul(...).sum(-1), left to right
    pen[gid] = fmaxf(-d, 0.f);
}
}  // namespace g4d

extern "C" int g4d_interpenetration_f32(int frames, int vg, int v, const float *garment, const float *body, const float *normals,
                                        const int *nn_idx, int idx_stride, float *pen, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(frames >= 0 && vg >= 0 && v > 0 && idx_stride >= 1, "g4d_interpenetration_f32: bad sizes");
    const long long total = (long long)frames * vg;
    if (total == 0) return G4D_OK;
    G4D_REQUIRE(garment && body && normals && nn_idx && pen, "g4d_interpenetration_f32: null pointer");
    hipLaunchKernelGGL(interpenetration_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), total, vg, v,
                       garment, body, normals, nn_idx, idx_stride, pen);
    return check_launch("g4d_interpenetration_f32");
}
