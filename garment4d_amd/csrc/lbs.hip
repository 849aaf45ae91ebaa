// SMPL linear blend skinning for gfx950 -- the reference runs this as ~40 small torch ops per call, on the CPU,
// batch 1, three times per frame inside its DataLoader workers (/root/reference/smplx/smplx/lbs.py:152-248,
// called from utils/dataloader.py:199-212).  Here it is four kernels over the whole batch of frames:
//
//   lbs_shape_kernel   v_shaped = v_template + shapedirs . betas                       (lbs.py:205, 288-309)
//   joint_regress      J = J_regressor . v_shaped  (shared or per-sample regressor)    (lbs.py:209, 251-286)
//   lbs_rigid_kernel   Rodrigues, kinematic chain, rel. transforms A, pose feature     (lbs.py:312-419, 215-222)
//   lbs_pose_blend     v_posed = v_shaped + pose_feature . posedirs                     (lbs.py:223-229)
//   lbs_skin           T = W . A;  verts = T [v_posed;1]                                (lbs.py:233-246)
//
// All HBM-bound: the model constants (posedirs 17 MB, shapedirs 0.8 MB, J_regressor 0.66 MB, weights 0.66 MB)
// are read once per batch of up to 8 frames; per-frame algorithmic traffic is 24 B/vertex + 64 B/joint.
// lbs_pose_blend splits the 207 pose features over 4 k-slices x 64 columns per workgroup so that ~1300 waves stream
// posedirs; lbs_skin runs one thread per (frame, vertex) with the frame's 24 joint transforms in LDS.
#include <cstdlib>

#include "g4d_common.h"

namespace g4d {

constexpr int kFB = 8;  // frames per thread in shape / pose-skin kernels

__global__ void __launch_bounds__(256) lbs_shape_kernel(int B, int V3, int NB, int betas_bstride, const float *__restrict__ betas,
                                                       const float *__restrict__ v_template,
                                                       const float *__restrict__ shapedirs, float *__restrict__ v_shaped) {
    extern __shared__ float sb[];  // [B][NB]
    for (int i = threadIdx.x; i < B * NB; i += 256) sb[i] = betas[(size_t)(i / NB) * betas_bstride + (i % NB)];
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= V3) return;
    const float vt = v_template[e];
    const float *sd = shapedirs + (size_t)e * NB;
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int l = 0; l < NB; ++l) acc = fmaf(sb[b * NB + l], sd[l], acc);
        v_shaped[(size_t)b * V3 + e] = vt + acc;  // lbs.py:205  v_template + blend_shapes
    }
}

// joints[b,j,:] = sum_v jreg[(b),j,v] * verts[b,v,:]
__global__ void __launch_bounds__(256) joint_regress_kernel(int J, int V, long long jreg_bstride, int jreg_group, const float *__restrict__ jreg,
                                                           const float *__restrict__ verts, float *__restrict__ joints) {
    __shared__ float red[4][3];
    const int j = blockIdx.x, b = blockIdx.y;
    const float *w = jreg + (size_t)(b / jreg_group) * jreg_bstride + (size_t)j * V;
    const float *vb = verts + (size_t)b * V * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float wv = w[v];
        sx = fmaf(wv, vb[v * 3 + 0], sx);
        sy = fmaf(wv, vb[v * 3 + 1], sy);
        sz = fmaf(wv, vb[v * 3 + 2], sz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        joints[((size_t)b * J + j) * 3 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// lbs.py:312-346  batch_rodrigues: angle = ||r + 1e-8||, dir = r / angle, R = I + sin K + (1 - cos) K K
__device__ __forceinline__ void rodrigues(float rx, float ry, float rz, float *R) {
    const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    const float angle = __fsqrt_rn(ax * ax + ay * ay + az * az);
    const float x = rx / angle, y = ry / angle, z = rz / angle;
    const float s = sinf(angle), c1 = 1.0f - cosf(angle);
    // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  K K = [[-(y2+z2), xy, xz],[xy, -(x2+z2), yz],[xz, yz, -(x2+y2)]]
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = 1.0f + c1 * (-(zz + yy));  // row-major 3x3; K K[0][0] = -z*z - y*y (bmm order: (-z)(z) + (y)(-y))
    R[1] = s * (-z) + c1 * xy;
    R[2] = s * y + c1 * xz;
    R[3] = s * z + c1 * xy;
    R[4] = 1.0f + c1 * (-(zz + xx));
    R[5] = s * (-x) + c1 * yz;
    R[6] = s * (-y) + c1 * xz;
    R[7] = s * x + c1 * yz;
    R[8] = 1.0f + c1 * (-(yy + xx));
}

__global__ void __launch_bounds__(256) rodrigues_kernel(int n, const float *__restrict__ rv, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float R[9];
    rodrigues(rv[i * 3 + 0], rv[i * 3 + 1], rv[i * 3 + 2], R);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[(size_t)i * 9 + k] = R[k];
}

// One wave per frame.  lanes j < J: per-joint work; the chain runs 23 dependent steps with lanes 0..11 each
// owning one entry of the 3x4 transform.  lbs.py:362-419.
__global__ void __launch_bounds__(64) lbs_rigid_kernel(int J, int pose2rot, const float *__restrict__ pose /* (B,J,3) or (B,J,9) */,
                                                      const float *__restrict__ joints /* (B,J,3) */,
                                                      const int *__restrict__ parents, float *__restrict__ rot_out /* (B,J,9)|null */,
                                                      float *__restrict__ posed_joints /* (B,J,3)|null */,
                                                      float *__restrict__ rel_transforms /* (B,J,16) */,
                                                      float *__restrict__ pose_feature /* (B,pf_ld)|null, written at column pf_col0 */,
                                                      int pf_ld, int pf_col0,
                                                      const float *__restrict__ Jt /* (J,3)|null: joints of the template */,
                                                      const float *__restrict__ Js /* (J,3,NB): d joints / d betas */,
                                                      const float *__restrict__ betas, int nb, int betas_bstride) {
    __shared__ float sR[64][9], sJ[64][3], sL[64][12], sG[64][12];
    __shared__ int sP[64];
    const int b = blockIdx.x, l = threadIdx.x;
    if (l < J) {
        float R[9];
        if (pose2rot) {
            const float *p = pose + ((size_t)b * J + l) * 3;
            rodrigues(p[0], p[1], p[2], R);
        } else {
            const float *p = pose + ((size_t)b * J + l) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = p[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) sR[l][k] = R[k];
        if (Jt) {
            // fused front end: J_regressor (v_template + shapedirs beta) == (J_regressor v_template) + (J_regressor shapedirs) beta;
            // the two products are constants of the model, so the joints cost 3 * NB FMAs here instead of a V-long reduction
            const float *be = betas + (size_t)b * betas_bstride;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float acc = Jt[l * 3 + r];
                for (int k = 0; k < nb; ++k) acc = fmaf(Js[(l * 3 + r) * nb + k], be[k], acc);
                sJ[l][r] = acc;
            }
        } else {
            const float *jp = joints + ((size_t)b * J + l) * 3;
            sJ[l][0] = jp[0]; sJ[l][1] = jp[1]; sJ[l][2] = jp[2];
        }
        sP[l] = parents[l];
        if (rot_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) rot_out[((size_t)b * J + l) * 9 + k] = R[k];
        }
        if (pose_feature && l > 0) {  // lbs.py:217 / :222  (R[1:] - I).view(B, -1)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                pose_feature[(size_t)b * pf_ld + pf_col0 + (l - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
        }
    }
    if (Jt && pose_feature && l < nb) pose_feature[(size_t)b * pf_ld + l] = betas[(size_t)b * betas_bstride + l];  // [betas | R - I]
    __syncthreads();
    if (l < J) {  // local transform [R | J - J_parent]  (lbs.py:390-396)
        const int p = sP[l];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            sL[l][r * 4 + 0] = sR[l][r * 3 + 0];
            sL[l][r * 4 + 1] = sR[l][r * 3 + 1];
            sL[l][r * 4 + 2] = sR[l][r * 3 + 2];
            sL[l][r * 4 + 3] = (l > 0) ? (sJ[l][r] - sJ[p][r]) : sJ[l][r];
        }
    }
    __syncthreads();
    if (l < 12) sG[0][l] = sL[0][l];
    __syncthreads();
    for (int i = 1; i < J; ++i) {  // G_i = G_parent(i) . L_i   (lbs.py:399-405), 4x4 product with implicit [0 0 0 1]
        if (l < 12) {
            const int p = sP[i], r = l >> 2, c = l & 3;
            float acc = sG[p][r * 4 + 0] * sL[i][0 * 4 + c];
            acc = fmaf(sG[p][r * 4 + 1], sL[i][1 * 4 + c], acc);
            acc = fmaf(sG[p][r * 4 + 2], sL[i][2 * 4 + c], acc);
            if (c == 3) acc += sG[p][r * 4 + 3];
            sG[i][l] = acc;
        }
        __syncthreads();
    }
    if (l < J) {
        float *A = rel_transforms + ((size_t)b * J + l) * 16;
        const float jx = sJ[l][0], jy = sJ[l][1], jz = sJ[l][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float g0 = sG[l][r * 4 + 0], g1 = sG[l][r * 4 + 1], g2 = sG[l][r * 4 + 2], g3 = sG[l][r * 4 + 3];
            A[r * 4 + 0] = g0; A[r * 4 + 1] = g1; A[r * 4 + 2] = g2;
            A[r * 4 + 3] = g3 - fmaf(g2, jz, fmaf(g1, jy, g0 * jx));  // lbs.py:414-417
            if (posed_joints) posed_joints[((size_t)b * J + l) * 3 + r] = g3;  // lbs.py:410
        }
        A[12] = 0.f; A[13] = 0.f; A[14] = 0.f; A[15] = 1.f;
    }
}

// v_posed[b,e] = v_shaped[b,e] + sum_k pose_feature[b,k] * posedirs[k,e]     (lbs.py:223-229), e over V*3.
// 256 threads = 64 columns x 4 k-slices: posedirs (17 MB) streams once per <= 8 frames with 256-byte coalesced row
// segments and ~1300 waves in flight (a thread-per-vertex layout leaves the chip at 27 workgroups and is latency
// bound); the four k-slices meet in LDS.
__global__ void __launch_bounds__(256) lbs_pose_blend_kernel(int B, int E, int PF, const float *__restrict__ v_shaped, long long v_bstride,
                                                            const float *__restrict__ pose_feature,
                                                            const float *__restrict__ posedirs, float *__restrict__ v_posed) {
    extern __shared__ float smem[];
    float *sPF = smem;                 // [kFB][PF]
    float *sRed = smem + kFB * PF;     // [3][64][kFB]
    const int b0 = blockIdx.y * kFB;
    const int nf = min(kFB, B - b0);
    for (int i = threadIdx.x; i < kFB * PF; i += 256) sPF[i] = (i < nf * PF) ? pose_feature[(size_t)b0 * PF + i] : 0.f;
    __syncthreads();
    const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + col;
    const int kper = (PF + 3) / 4;
    const int k0 = ks * kper, k1 = min(PF, k0 + kper);
    float acc[kFB];
#pragma unroll
    for (int f = 0; f < kFB; ++f) acc[f] = 0.f;
    if (e < E) {
        const float *pd = posedirs + e;
#pragma unroll 4
        for (int k = k0; k < k1; ++k) {
            const float d = pd[(size_t)k * E];
#pragma unroll
            for (int f = 0; f < kFB; ++f) acc[f] = fmaf(sPF[f * PF + k], d, acc[f]);
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int f = 0; f < kFB; ++f) sRed[((ks - 1) * 64 + col) * kFB + f] = acc[f];
    }
    __syncthreads();
    if (ks == 0 && e < E) {
#pragma unroll
        for (int f = 0; f < kFB; ++f) {
            if (f < nf) {
                const float o = ((acc[f] + sRed[(0 * 64 + col) * kFB + f]) + sRed[(1 * 64 + col) * kFB + f]) + sRed[(2 * 64 + col) * kFB + f];
                v_posed[(size_t)(b0 + f) * E + e] = o + v_shaped[(size_t)(b0 + f) * v_bstride + e];
            }
        }
    }
}

// verts[b,v,:] = (sum_j W[(b),v,j] A[b,j]) . [v_in[b,v,:]; 1]   -- one thread per (frame, vertex), A[b] in LDS.
__global__ void __launch_bounds__(256) lbs_skin_kernel(int V, int J, const float *__restrict__ v_in,
                                                      const float *__restrict__ weights, long long w_bstride, int w_group,
                                                      const float *__restrict__ A, float *__restrict__ verts) {
    extern __shared__ float sA[];  // [J][12]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < J * 12; i += 256) sA[i] = A[((size_t)b * J + i / 12) * 16 + (i % 12)];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float *w = weights + (size_t)(b / w_group) * w_bstride + (size_t)v * J;
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int j = 0; j < J; ++j) {
        const float wj = w[j];
        const float *a = sA + j * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(wj, a[e], T[e]);  // lbs.py:238  T = W . A
    }
    const float *p = v_in + ((size_t)b * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float *o = verts + ((size_t)b * V + v) * 3;
    o[0] = fmaf(T[2], z, fmaf(T[1], y, T[0] * x)) + T[3];  // lbs.py:244  T . [v;1]
    o[1] = fmaf(T[6], z, fmaf(T[5], y, T[4] * x)) + T[7];
    o[2] = fmaf(T[10], z, fmaf(T[9], y, T[8] * x)) + T[11];
}


// ---- lbs() in ONE launch ------------------------------------------------------------------------------------------------------
// Workgroup = 64 vertices x up to 8 frames, 8 waves.  Wave w (1) requests its slice of the blend matrix rows for the tile -- every
// row of [shapedirs^T ; posedirs] the tile needs is in flight at once, 12 bytes per lane and row -- then, while those loads fly,
// (2) does frame w's per-frame work: Rodrigues, the joints from betas, the blend coefficients [betas | R - I] and the 23-step
// kinematic chain, all wave-local in LDS (every workgroup repeats this per-frame work: ~3k cycles against a launch of its own);
// (3) multiplies its slice into 8 frames x 3 coordinates of partial sums; (4) the slices meet in LDS and wave w skins frame w.
// Against the three-launch route (rigid -> blend -> skin: 10 + 15 + 5 us alone) there is no v_posed / coefficient round trip and
// the 17.9 MB of blend rows are requested up front instead of 4 dependent loads per lane at a time.
namespace {
constexpr int kOneKS = 8;     // k-slices = waves per workgroup
constexpr int kOneRows = 28;  // blend rows per slice held in registers (8 x 28 = 224 >= 10 + 207)
constexpr int kOneJ = 32;     // joints
typedef float lbs_f4 __attribute__((ext_vector_type(4)));
}  // namespace

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(64 * kOneKS) lbs_one_kernel(int B, int V, int J, int NB, int pose2rot, const float *__restrict__ betas,
                                                              int betas_bstride, const float *__restrict__ pose,
                                                              const float *__restrict__ v_template, const float *__restrict__ blend_dirs,
                                                              const float *__restrict__ Jt, const float *__restrict__ Js,
                                                              const int *__restrict__ parents, const float *__restrict__ weights,
                                                              float *__restrict__ A_out, float *__restrict__ posed_joints,
                                                              float *__restrict__ verts) {
    extern __shared__ __attribute__((aligned(16))) float one_smem[];
    float *sC = one_smem;                                  // [kOneKS * kOneRows][kFB]  blend coefficients, zero beyond NC / nf
    float *sR = sC + kOneKS * kOneRows * kFB;              // [kFB][kOneJ][9]
    float *sJ = sR + kFB * kOneJ * 9;                      // [kFB][kOneJ][3]
    float *sL = sJ + kFB * kOneJ * 3;                      // [kFB][kOneJ][12]   local transforms, later A (3x4)
    float *sG = sL + kFB * kOneJ * 12;                     // [kFB][kOneJ][12]
    float *sRed = sG + kFB * kOneJ * 12;                   // [kOneKS][kFB * 3][64]
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NC = NB + (J - 1) * 9;
    const size_t E = (size_t)V * 3;
    const int v = blockIdx.x * 64 + l;
    const int vc = min(v, V - 1);                          // lanes past the end read the last vertex and never write

    // (1) this wave's blend rows: all requested now
    float d[kOneRows][3];
    const int k0 = w * kOneRows;
#pragma unroll
    for (int r = 0; r < kOneRows; ++r) {
        const float *src = blend_dirs + (size_t)min(k0 + r, NC - 1) * E + (size_t)vc * 3;   // rows >= NC: coefficient 0
        d[r][0] = src[0]; d[r][1] = src[1]; d[r][2] = src[2];
    }
    float wt[kOneJ];
    {
        const float *wp = weights + (size_t)vc * J;
        if ((J & 3) == 0) {   // a lane's weight row is 16-byte aligned
#pragma unroll
            for (int j4 = 0; j4 < kOneJ / 4; ++j4) {
                const lbs_f4 q = j4 * 4 < J ? *reinterpret_cast<const lbs_f4 *>(wp + j4 * 4) : (lbs_f4){0.f, 0.f, 0.f, 0.f};
                wt[j4 * 4 + 0] = q.x; wt[j4 * 4 + 1] = q.y; wt[j4 * 4 + 2] = q.z; wt[j4 * 4 + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kOneJ; ++j) wt[j] = j < J ? wp[j] : 0.f;
        }
    }
    const float t0 = v_template[(size_t)vc * 3 + 0], t1 = v_template[(size_t)vc * 3 + 1], t2 = v_template[(size_t)vc * 3 + 2];

    // Frame groups of 8: blockIdx.y takes groups blockIdx.y, blockIdx.y + gridDim.y, ... with the blend rows, weights and template of its 64
    // vertices held in registers throughout (many frames per call -- the executor's 240: every group used to be a workgroup of its own that
    // re-read the 17.9 MB of blend rows; per frame the arithmetic is the same whatever the grouping, so results do not depend on the batch size)
    const int f = w;
    const int ngroups = (B + kFB - 1) / kFB;
  for (int grp = blockIdx.y; grp < ngroups; grp += gridDim.y) {
    const int b0 = grp * kFB, nf = min(kFB, B - b0);
    // (2) frame w: rotations, joints, coefficients, chain (wave-local)
    for (int i = l; i < kOneRows * kFB; i += 64) sC[(size_t)w * kOneRows * kFB + i] = 0.f;   // 1/8 of the table each
    lds_barrier();   // LDS only: __syncthreads() would wait for the loads above
    float *R_ = sR + (size_t)f * kOneJ * 9, *J_ = sJ + (size_t)f * kOneJ * 3, *L_ = sL + (size_t)f * kOneJ * 12, *G_ = sG + (size_t)f * kOneJ * 12;
    if (f < nf) {
        const int b = b0 + f;
        if (l < J) {
            float R[9];
            if (pose2rot) {
                const float *pp = pose + ((size_t)b * J + l) * 3;
                rodrigues(pp[0], pp[1], pp[2], R);
            } else {
                const float *pp = pose + ((size_t)b * J + l) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = pp[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) R_[l * 9 + k] = R[k];
            const float *be = betas + (size_t)b * betas_bstride;
#pragma unroll
            for (int r = 0; r < 3; ++r) {   // J_regressor (v_template + shapedirs beta) = Jt + Js beta  (lbs_rigid_kernel)
                float acc = Jt[l * 3 + r];
                for (int k = 0; k < NB; ++k) acc = fmaf(Js[(l * 3 + r) * NB + k], be[k], acc);
                J_[l * 3 + r] = acc;
            }
            if (l > 0) {                    // lbs.py:217 / :222  (R[1:] - I)
#pragma unroll
                for (int k = 0; k < 9; ++k) sC[(size_t)(NB + (l - 1) * 9 + k) * kFB + f] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
            }
        }
        for (int i = l; i < NB; i += 64) sC[(size_t)i * kFB + f] = betas[(size_t)b * betas_bstride + i];   // (NB may exceed the 64 lanes: few joints, many betas)
        wave_sync_lds();
        if (l < J) {  // local transform [R | J - J_parent]  (lbs.py:390-396)
            const int p = parents[l];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                L_[l * 12 + r * 4 + 0] = R_[l * 9 + r * 3 + 0];
                L_[l * 12 + r * 4 + 1] = R_[l * 9 + r * 3 + 1];
                L_[l * 12 + r * 4 + 2] = R_[l * 9 + r * 3 + 2];
                L_[l * 12 + r * 4 + 3] = (l > 0) ? (J_[l * 3 + r] - J_[p * 3 + r]) : J_[l * 3 + r];
            }
        }
        wave_sync_lds();
        if (l < 12) G_[l] = L_[l];
        wave_sync_lds();
        for (int i = 1; i < J; ++i) {  // G_i = G_parent(i) . L_i   (lbs.py:399-405)
            if (l < 12) {
                const int p = parents[i], r = l >> 2, c = l & 3;
                float acc = G_[p * 12 + r * 4 + 0] * L_[i * 12 + 0 * 4 + c];
                acc = fmaf(G_[p * 12 + r * 4 + 1], L_[i * 12 + 1 * 4 + c], acc);
                acc = fmaf(G_[p * 12 + r * 4 + 2], L_[i * 12 + 2 * 4 + c], acc);
                if (c == 3) acc += G_[p * 12 + r * 4 + 3];
                G_[i * 12 + l] = acc;
            }
            wave_sync_lds();
        }
        if (l < J) {  // A = G with the rest-pose joint removed (lbs.py:414-417); kept in L_ for the skinning below
            const float jx = J_[l * 3 + 0], jy = J_[l * 3 + 1], jz = J_[l * 3 + 2];
            float *Ag = (blockIdx.x == 0) ? A_out + ((size_t)b * J + l) * 16 : nullptr;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float g0 = G_[l * 12 + r * 4 + 0], g1 = G_[l * 12 + r * 4 + 1], g2 = G_[l * 12 + r * 4 + 2], g3 = G_[l * 12 + r * 4 + 3];
                const float tr = g3 - fmaf(g2, jz, fmaf(g1, jy, g0 * jx));
                if (Ag) { Ag[r * 4 + 0] = g0; Ag[r * 4 + 1] = g1; Ag[r * 4 + 2] = g2; Ag[r * 4 + 3] = tr; }
                if (Ag && posed_joints) posed_joints[((size_t)b * J + l) * 3 + r] = g3;  // lbs.py:410
                L_[l * 12 + r * 4 + 0] = g0; L_[l * 12 + r * 4 + 1] = g1; L_[l * 12 + r * 4 + 2] = g2; L_[l * 12 + r * 4 + 3] = tr;
            }
            if (Ag) { Ag[12] = 0.f; Ag[13] = 0.f; Ag[14] = 0.f; Ag[15] = 1.f; }
        }
    }
    __syncthreads();   // coefficients of all frames complete; also drains this wave's outstanding loads, which are needed now

    // (3) partial sums of this slice
    float acc[kFB][3];
#pragma unroll
    for (int g = 0; g < kFB; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0.f;
#pragma unroll
    for (int r = 0; r < kOneRows; ++r) {
        const lbs_f4 c0 = *reinterpret_cast<const lbs_f4 *>(&sC[(size_t)(k0 + r) * kFB]), c1 = *reinterpret_cast<const lbs_f4 *>(&sC[(size_t)(k0 + r) * kFB + 4]);
#pragma unroll
        for (int g = 0; g < kFB; ++g) {
            const float c = g < 4 ? c0[g] : c1[g - 4];
            acc[g][0] = fmaf(c, d[r][0], acc[g][0]); acc[g][1] = fmaf(c, d[r][1], acc[g][1]); acc[g][2] = fmaf(c, d[r][2], acc[g][2]);
        }
    }
#pragma unroll
    for (int g = 0; g < kFB; ++g)
#pragma unroll
        for (int c = 0; c < 3; ++c) sRed[((size_t)w * kFB * 3 + g * 3 + c) * 64 + l] = acc[g][c];
    __syncthreads();

    // (4) wave w finishes frame w: v_posed = v_template + sum of the slices (fixed order), then the skinning
    if (f < nf && v < V) {
        float vp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < kOneKS; ++sl)
#pragma unroll
            for (int c = 0; c < 3; ++c) vp[c] += sRed[((size_t)sl * kFB * 3 + f * 3 + c) * 64 + l];
        const float x = vp[0] + t0, y = vp[1] + t1, z = vp[2] + t2;     // lbs.py:223-229
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < kOneJ; ++j) {
            if (j < J) {
                const lbs_f4 a0 = *reinterpret_cast<const lbs_f4 *>(&L_[j * 12]), a1 = *reinterpret_cast<const lbs_f4 *>(&L_[j * 12 + 4]),
                             a2 = *reinterpret_cast<const lbs_f4 *>(&L_[j * 12 + 8]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { T[e] = fmaf(wt[j], a0[e], T[e]); T[4 + e] = fmaf(wt[j], a1[e], T[4 + e]); T[8 + e] = fmaf(wt[j], a2[e], T[8 + e]); }   // lbs.py:238
            }
        }
        float *o = verts + ((size_t)(b0 + f) * V + v) * 3;
        o[0] = fmaf(T[2], z, fmaf(T[1], y, T[0] * x)) + T[3];  // lbs.py:244
        o[1] = fmaf(T[6], z, fmaf(T[5], y, T[4] * x)) + T[7];
        o[2] = fmaf(T[10], z, fmaf(T[9], y, T[8] * x)) + T[11];
    }
    __syncthreads();   // the next group overwrites the coefficient table, the transforms and the partial sums
  }
}


// ---- lbs() on the matrix pipe (round 5) -----------------------------------------------------------------------------------------------
// The pose / shape blend of B frames is a (V*3 x NC) . (NC x B) GEMM (NC = NB + 9 (J - 1) = 217 for SMPL) and the blend of the joint
// transforms T = W . A a (V x J) . (J x 12 B) one; lbs_one_kernel runs both on the VALU and repeats the per-frame Rodrigues + kinematic
// chain in every vertex tile (189 us per 240 frames, 0.04 of the HBM roofline).  Here:
//   lbs_frame_kernel   one wave per frame, ONCE: rotations, joints from betas, the 23-step chain (the code of lbs_one_kernel's step 2),
//                      A / posed joints to their outputs, and the two B operands of the GEMMs in fragment order to a workspace:
//                      coefficients C[f][q][ks] = c_f[4 ks + q] (c = [betas | R - I], zero padded to 4 KS) and transforms
//                      At[f][q][e][js] = A_f[4 js + q][e] (e = 0..11 of the 3x4 part, zero beyond J);
//   lbs_mfma_kernel    a workgroup owns 32 vertices: their blend rows (NC x 96 floats = 86 KB, read from HBM once per launch) sit in
//                      LDS in A-operand fragment order, their skinning weights in registers; its 8 waves walk the 16-frame tiles.  Per
//                      (frame tile, 16-vertex half): v_posed^T (vertices x frames, one accumulator tile per coordinate, started from
//                      v_template) = 3 x KS v_mfma_f32_16x16x4_f32, T^T (one tile per transform entry) = 12 x JS, and because both are
//                      (vertices x frames) tiles a lane ends with the posed vertex AND the complete 3x4 transform of the same four
//                      (vertex, frame) pairs: the final T . [v; 1] is 9 FMAs per pair in registers, 48 contiguous bytes stored per lane.
// Every output element is one accumulator chain over k ascending, whatever the batch: a frame's bits do not depend on the frames it is
// launched with (tests/test_pipeline_gpu.py, test_large_launch_gpu.py).
namespace {
constexpr int kMfKS = 56;     // k-steps of 4 blend rows (224 >= 217)
constexpr int kMfVT = 32;     // vertices per workgroup (two 16-vertex halves)
constexpr int kMfWaves = 8;
typedef float mf_f4 __attribute__((ext_vector_type(4)));
typedef float mf_f2 __attribute__((ext_vector_type(2)));
}  // namespace

__global__ void __launch_bounds__(256) lbs_frame_kernel(int B, int J, int NB, int JS, int pose2rot, const float *__restrict__ betas, int betas_bstride,
                                                        const float *__restrict__ pose, const float *__restrict__ Jt, const float *__restrict__ Js,
                                                        const int *__restrict__ parents, float *__restrict__ A_out, float *__restrict__ posed_joints,
                                                        float *__restrict__ Cws, float *__restrict__ Aws) {
    __shared__ float sR_[4][kOneJ * 9], sJ_[4][kOneJ * 3], sL_[4][kOneJ * 12], sG_[4][kOneJ * 12], sC_[4][4 * kMfKS];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;   // (wave-local LDS exchange only: no workgroup barrier below)
    float *R_ = sR_[w], *J_ = sJ_[w], *L_ = sL_[w], *G_ = sG_[w], *C_ = sC_[w];
    const int NC = NB + (J - 1) * 9;
    for (int i = l; i < 4 * kMfKS; i += 64) C_[i] = 0.f;
    wave_sync_lds();
    if (l < J) {
        float R[9];
        if (pose2rot) {
            const float *pp = pose + ((size_t)b * J + l) * 3;
            rodrigues(pp[0], pp[1], pp[2], R);
        } else {
            const float *pp = pose + ((size_t)b * J + l) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = pp[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) R_[l * 9 + k] = R[k];
        const float *be = betas + (size_t)b * betas_bstride;
#pragma unroll
        for (int r = 0; r < 3; ++r) {   // J_regressor (v_template + shapedirs beta) = Jt + Js beta  (lbs.py:209 through two model constants)
            float acc = Jt[l * 3 + r];
#pragma unroll 10
            for (int k = 0; k < NB; ++k) acc = fmaf(Js[(l * 3 + r) * NB + k], be[k], acc);
            J_[l * 3 + r] = acc;
        }
        if (l > 0) {                    // lbs.py:217 / :222  (R[1:] - I)
#pragma unroll
            for (int k = 0; k < 9; ++k) C_[NB + (l - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
        }
    }
    for (int i = l; i < NB; i += 64) C_[i] = betas[(size_t)b * betas_bstride + i];   // (NB may exceed the 64 lanes: few joints, many betas)
    wave_sync_lds();
    if (l < J) {  // local transform [R | J - J_parent]  (lbs.py:390-396)
        const int p = parents[l];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            L_[l * 12 + r * 4 + 0] = R_[l * 9 + r * 3 + 0];
            L_[l * 12 + r * 4 + 1] = R_[l * 9 + r * 3 + 1];
            L_[l * 12 + r * 4 + 2] = R_[l * 9 + r * 3 + 2];
            L_[l * 12 + r * 4 + 3] = (l > 0) ? (J_[l * 3 + r] - J_[p * 3 + r]) : J_[l * 3 + r];
        }
    }
    wave_sync_lds();
    if (l < 12) G_[l] = L_[l];
    wave_sync_lds();
    for (int i = 1; i < J; ++i) {  // G_i = G_parent(i) . L_i   (lbs.py:399-405)
        if (l < 12) {
            const int p = parents[i], r = l >> 2, c = l & 3;
            float acc = G_[p * 12 + r * 4 + 0] * L_[i * 12 + 0 * 4 + c];
            acc = fmaf(G_[p * 12 + r * 4 + 1], L_[i * 12 + 1 * 4 + c], acc);
            acc = fmaf(G_[p * 12 + r * 4 + 2], L_[i * 12 + 2 * 4 + c], acc);
            if (c == 3) acc += G_[p * 12 + r * 4 + 3];
            G_[i * 12 + l] = acc;
        }
        wave_sync_lds();
    }
    if (l < J) {  // A = G with the rest-pose joint removed (lbs.py:414-417)
        const float jx = J_[l * 3 + 0], jy = J_[l * 3 + 1], jz = J_[l * 3 + 2];
        float *Ag = A_out + ((size_t)b * J + l) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float g0 = G_[l * 12 + r * 4 + 0], g1 = G_[l * 12 + r * 4 + 1], g2 = G_[l * 12 + r * 4 + 2], g3 = G_[l * 12 + r * 4 + 3];
            const float tr = g3 - fmaf(g2, jz, fmaf(g1, jy, g0 * jx));
            Ag[r * 4 + 0] = g0; Ag[r * 4 + 1] = g1; Ag[r * 4 + 2] = g2; Ag[r * 4 + 3] = tr;
            if (posed_joints) posed_joints[((size_t)b * J + l) * 3 + r] = g3;  // lbs.py:410
            L_[l * 12 + r * 4 + 0] = g0; L_[l * 12 + r * 4 + 1] = g1; L_[l * 12 + r * 4 + 2] = g2; L_[l * 12 + r * 4 + 3] = tr;
        }
        Ag[12] = 0.f; Ag[13] = 0.f; Ag[14] = 0.f; Ag[15] = 1.f;
    }
    wave_sync_lds();
    // the GEMMs' B operands in fragment order PER 16-FRAME TILE: load g of the main kernel's lane (frame fi, quarter fq) is the 16 bytes at
    // [tile][g][fq * 16 + fi][0..3] -- one 1 KB run per wave and load.  (Per-frame rows, 16 bytes out of a different cache line for every
    // lane, kept the CU's vector-memory path busier than its matrix pipe: 55 us per 240 frames instead of 35.)
    const int tile = b >> 4, bi = b & 15;
    float *cw = Cws + (size_t)tile * (kMfKS / 4) * 256;
    for (int i = l; i < 4 * kMfKS; i += 64) {      // i = (g, q, s): coefficient k = 4 (4 g + s) + q
        const int g = i >> 4, q = (i >> 2) & 3, s4 = i & 3, k = 4 * (4 * g + s4) + q;
        cw[(g * 64 + q * 16 + bi) * 4 + s4] = k < NC ? C_[k] : 0.f;
    }
    float *aw = Aws + (size_t)tile * 3 * JS * 256;
    for (int i = l; i < 4 * 12 * JS; i += 64) {    // i = (q, flat = e JS + js): transform entry e of joint j = 4 js + q, flat / 4 = load, flat % 4 = element
        const int q = i / (12 * JS), flat = i - q * 12 * JS, e = flat / JS, js = flat - e * JS, j = 4 * js + q;
        aw[((flat >> 2) * 64 + q * 16 + bi) * 4 + (flat & 3)] = j < J ? L_[j * 12 + e] : 0.f;
    }
}

template <int JS>
__global__ void __launch_bounds__(64 * kMfWaves) lbs_mfma_kernel(int B, int V, int J, int NC, const float *__restrict__ v_template,
                                                                const float *__restrict__ blend_dirs, const float *__restrict__ weights,
                                                                const float *__restrict__ Cws, const float *__restrict__ Aws, float *__restrict__ verts) {
    extern __shared__ __attribute__((aligned(16))) float mf_smem[];   // [2 halves][3 coords][14 groups][4 q][16 vertices][4 k-steps]
    constexpr int KG = kMfKS / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fq = lane >> 4;
    const int v0 = blockIdx.x * kMfVT;
    const size_t E = (size_t)V * 3;
    // ---- the tile's blend rows, HBM -> LDS once.  Row k, column e = 3 (v - v0) + c  ->  [half][c][k / 16][k % 4][v % 16][(k / 4) % 4]
    // (the first frame tile's coefficients do not depend on the tile: requested first, they arrive while the rows are staged)
    const int nft = (B + 15) / 16;
    // work units: (frame tile ft, half st), u = 2 ft + st.  Many frame tiles: a wave takes both halves of tiles wave, wave + 8, ... one after
    // the other (the second half's B operands are the first's again: L1 hits); few (small batches): unit u goes to wave u % 8, so that two
    // waves share even a single tile.
    const bool split = nft < kMfWaves;
    const int nunit = 2 * nft;
    const int u_first = split ? wave : 2 * wave;
    auto next_unit = [&](int u) { return split ? u + kMfWaves : ((u & 1) ? u + 2 * kMfWaves - 1 : u + 1); };
    mf_f4 cf4[KG];
    auto load_cf = [&](int u) {                                // (always executed: a load under a condition makes the compiler drain every load at the join)
        const int tile = min(u, nunit - 1) >> 1;                // units past the end: clamped, never used.  Frames past B in the last tile: columns of
        const mf_f4 *cp = reinterpret_cast<const mf_f4 *>(Cws + (size_t)tile * KG * 256) + lane;   // their own in every MFMA, never stored (the workspace covers whole tiles)
#pragma unroll
        for (int g = 0; g < KG; ++g) cf4[g] = cp[g * 64];
    };
    load_cf(u_first);
    {
        const int ncol = min(kMfVT * 3, (int)(E - (size_t)v0 * 3));          // (the last tile is ragged)
        constexpr int NIT = 4 * kMfKS * (kMfVT * 3 / 2) / (64 * kMfWaves);   // 21 float2 per thread, all requested before the first is stored
        static_assert(NIT * 64 * kMfWaves == 4 * kMfKS * (kMfVT * 3 / 2), "row staging: whole passes");
        mf_f2 d[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 64 * kMfWaves;
            const int k = i / (kMfVT * 3 / 2), e = (i - k * (kMfVT * 3 / 2)) * 2;
            d[it] = (mf_f2){0.f, 0.f};
            if (k < NC && e < ncol) {   // 3 v0 + e is even: a row's pairs are 8-byte aligned when k E is (E even: every row; E odd: every other one)
                const float *src = blend_dirs + (size_t)k * E + (size_t)v0 * 3 + e;
                if (((E & 1) == 0 || (k & 1) == 0) && e + 1 < ncol) d[it] = *reinterpret_cast<const mf_f2 *>(src);
                else { d[it].x = src[0]; d[it].y = e + 1 < ncol ? src[1] : 0.f; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 64 * kMfWaves;
            const int k = i / (kMfVT * 3 / 2), e = (i - k * (kMfVT * 3 / 2)) * 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ee = e + h, vv = ee / 3, c = ee - vv * 3;
                mf_smem[(((((vv >> 4) * 3 + c) * KG + (k >> 4)) * 4 + (k & 3)) * 16 + (vv & 15)) * 4 + ((k >> 2) & 3)] = h ? d[it].y : d[it].x;
            }
        }
    }
    // skinning weights of the two halves as A-operand fragments (lane (vertex fi, quarter fq) holds W[v][4 js + fq]) and the template
    // coordinates of the tile, both in LDS behind the blend rows: what a lane keeps in registers across the frame tiles is B operands only
    float *s_w = mf_smem + 2 * 3 * KG * 256;          // [2][JS][64]
    float *s_t = s_w + 2 * JS * 64;                   // [32][3] + pad
    for (int i = tid; i < 2 * JS * 64; i += 64 * kMfWaves) {
        const int st = i / (JS * 64), js = (i / 64) % JS, ln = i & 63;
        const int v = min(v0 + st * 16 + (ln & 15), V - 1), j = 4 * js + (ln >> 4);
        s_w[i] = j < J ? weights[(size_t)v * J + j] : 0.f;
    }
    for (int i = tid; i < kMfVT * 3; i += 64 * kMfWaves) s_t[i] = v_template[min((size_t)v0 * 3 + i, E - 1)];
    __syncthreads();

    // B operands: the coefficients of the NEXT unit are requested when this unit's blend has consumed them (they fly during the transform
    // blend), the transforms of THIS unit at its start (first used after 168 MFMAs).
    for (int u = u_first; u < nunit; u = next_unit(u)) {
        const int ft = u >> 1, st = u & 1;
        mf_f4 am4[3 * JS];                                      // am4[flat / 4][flat % 4], flat = e JS + js: A_f[4 js + fq][e]
        {
            const mf_f4 *ap = reinterpret_cast<const mf_f4 *>(Aws + (size_t)ft * 3 * JS * 256) + lane;
#pragma unroll
            for (int i4 = 0; i4 < 3 * JS; ++i4) am4[i4] = ap[i4 * 64];
        }
        {
            mf_f4 vp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) vp[c] = (mf_f4){0.f, 0.f, 0.f, 0.f};
            // pose + shape blend: rows = the half's 16 vertices, columns = the tile's 16 frames, k ascending
            const float *dl = mf_smem + (size_t)st * 3 * KG * 256 + (fq * 16 + fi) * 4;
            {   // the three coordinates' accumulators take turns (consecutive MFMAs are independent; each accumulator still sees k ascending); the
                // fragments of k-group g + 1 are requested before group g's MFMAs; fenced, or the scheduler requests all 42 up front (168 registers)
                mf_f4 ring[2][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) ring[0][c] = *reinterpret_cast<const mf_f4 *>(dl + (c * KG) * 256);
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    if (g + 1 < KG) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) ring[(g + 1) & 1][c] = *reinterpret_cast<const mf_f4 *>(dl + (c * KG + g + 1) * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int c = 0; c < 3; ++c) vp[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[g & 1][c][s4], cf4[g][s4], vp[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_cf(next_unit(u));
            __builtin_amdgcn_sched_barrier(0);
            // v_posed = v_template + blend  (lbs.py:205, :223-229: the template is added to the finished sums, as the reference does)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) vp[c][r] += s_t[(st * 16 + 4 * fq + r) * 3 + c];
            float wfr[JS];
#pragma unroll
            for (int js = 0; js < JS; ++js) wfr[js] = s_w[(st * JS + js) * 64 + lane];
            const bool frame_ok = ft * 16 + fi < B;
            float *o = verts + (size_t)(ft * 16 + fi) * E + (size_t)(v0 + st * 16 + 4 * fq) * 3;
            float res[12];                                      // the lane's four vertices x three coordinates: 48 contiguous bytes of the frame
            // T = W . A  (lbs.py:238), one tile per entry of the 3x4 transform, a row of the transform (four entries) at a time; then
            // verts = T . [v_posed; 1]  (lbs.py:244): lane (frame fi, vertices 4 fq + r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mf_f4 T[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T[e] = (mf_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int js = 0; js < JS; ++js) {
                        const int flat = (4 * c + e) * JS + js;
                        T[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[js], am4[flat / 4][flat % 4], T[e], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) res[r * 3 + c] = fmaf(T[2][r], vp[2][r], fmaf(T[1][r], vp[1][r], T[0][r] * vp[0][r])) + T[3][r];
            }
            if (frame_ok) {
                if (v0 + st * 16 + 4 * fq + 3 < V) {   // three 16-byte stores (dword-aligned addresses: a frame's rows start at multiples of 4 E bytes)
                    typedef float mf_f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                    for (int h = 0; h < 3; ++h) *reinterpret_cast<mf_f4u *>(o + 4 * h) = (mf_f4u){res[4 * h], res[4 * h + 1], res[4 * h + 2], res[4 * h + 3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (v0 + st * 16 + 4 * fq + r < V) { o[r * 3 + 0] = res[r * 3 + 0]; o[r * 3 + 1] = res[r * 3 + 1]; o[r * 3 + 2] = res[r * 3 + 2]; }
                }
            }
        }
    }
}

}  // namespace g4d

using namespace g4d;
#define G4D_S(s) reinterpret_cast<hipStream_t>(s)

extern "C" int g4d_lbs_shape_f32(int b, int v, int nb, const float *betas, int betas_bstride, const float *v_template,
                                 const float *shapedirs, float *v_shaped, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && v >= 0 && nb >= 0, "g4d_lbs_shape_f32: negative size");
    if ((long long)b * v == 0) return G4D_OK;
    G4D_REQUIRE(betas && v_template && shapedirs && v_shaped, "g4d_lbs_shape_f32: null pointer");
    // frames in slabs so the betas fit LDS comfortably
    const int slab = 512;
    for (int b0 = 0; b0 < b; b0 += slab) {
        const int nbf = (b - b0) < slab ? (b - b0) : slab;
        hipLaunchKernelGGL(lbs_shape_kernel, dim3((v * 3 + 255) / 256), dim3(256), sizeof(float) * nbf * (nb > 0 ? nb : 1), G4D_S(stream),
                           nbf, v * 3, nb, betas_bstride, betas + (size_t)b0 * betas_bstride, v_template, shapedirs,
                           v_shaped + (size_t)b0 * v * 3);
    }
    return check_launch("g4d_lbs_shape_f32");
}

extern "C" int g4d_joint_regress_f32(int b, int j, int v, const float *jreg, int jreg_batched, const float *verts,
                                     float *joints, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && j >= 0 && v >= 0 && b <= 65535, "g4d_joint_regress_f32: bad size");
    if ((long long)b * j == 0) return G4D_OK;
    G4D_REQUIRE(jreg && verts && joints, "g4d_joint_regress_f32: null pointer");
    G4D_REQUIRE(jreg_batched >= 0, "g4d_joint_regress_f32: jreg_batched must be >= 0");
    hipLaunchKernelGGL(joint_regress_kernel, dim3(j, b), dim3(256), 0, G4D_S(stream), j, v,
                       jreg_batched ? (long long)j * v : 0ll, jreg_batched ? jreg_batched : 1, jreg, verts, joints);
    return check_launch("g4d_joint_regress_f32");
}

extern "C" int g4d_rodrigues_f32(int n, const float *rot_vecs, float *rot_mats, g4d_stream_t stream) {
    G4D_REQUIRE(n >= 0, "g4d_rodrigues_f32: negative size");
    if (n == 0) return G4D_OK;
    G4D_REQUIRE(rot_vecs && rot_mats, "g4d_rodrigues_f32: null pointer");
    hipLaunchKernelGGL(rodrigues_kernel, dim3((n + 255) / 256), dim3(256), 0, G4D_S(stream), n, rot_vecs, rot_mats);
    return check_launch("g4d_rodrigues_f32");
}

extern "C" int g4d_rigid_transform_f32(int b, int j, int pose2rot, const float *pose, const float *joints, const int *parents,
                                       float *rot_out, float *posed_joints, float *rel_transforms, float *pose_feature,
                                       g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && j > 0 && j <= 64, "g4d_rigid_transform_f32: need 1 <= J <= 64");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(pose && joints && parents && rel_transforms, "g4d_rigid_transform_f32: null pointer");
    hipLaunchKernelGGL(lbs_rigid_kernel, dim3(b), dim3(64), 0, G4D_S(stream), j, pose2rot, pose, joints, parents, rot_out,
                       posed_joints, rel_transforms, pose_feature, (j - 1) * 9, 0, (const float *)nullptr, (const float *)nullptr,
                       (const float *)nullptr, 0, 0);
    return check_launch("g4d_rigid_transform_f32");
}

extern "C" int g4d_lbs_pose_skin_f32(int b, int v, int j, int pf, const float *v_in, const float *pose_feature,
                                     const float *posedirs, const float *weights, int weights_batched, const float *A,
                                     float *v_posed_scratch, float *verts, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && v >= 0 && j > 0 && pf >= 0 && b <= 65535 && weights_batched >= 0, "g4d_lbs_pose_skin_f32: bad size");
    if ((long long)b * v == 0) return G4D_OK;
    G4D_REQUIRE(v_in && weights && A && verts, "g4d_lbs_pose_skin_f32: null pointer");
    const float *skin_in = v_in;
    if (pf > 0) {
        G4D_REQUIRE(pose_feature && posedirs && v_posed_scratch, "g4d_lbs_pose_skin_f32: pose_feature/posedirs/scratch missing");
        const size_t lds = sizeof(float) * ((size_t)kFB * pf + 3 * 64 * kFB);
        G4D_REQUIRE(lds <= 64 * 1024, "g4d_lbs_pose_skin_f32: pose feature too long for LDS staging");
        dim3 grid((v * 3 + 63) / 64, (b + kFB - 1) / kFB);
        hipLaunchKernelGGL(lbs_pose_blend_kernel, grid, dim3(256), lds, G4D_S(stream), b, v * 3, pf, v_in, (long long)v * 3, pose_feature,
                           posedirs, v_posed_scratch);
        skin_in = v_posed_scratch;
    }
    G4D_REQUIRE((size_t)j * 12 * sizeof(float) <= 64 * 1024, "g4d_lbs_pose_skin_f32: too many joints");
    hipLaunchKernelGGL(lbs_skin_kernel, dim3((v + 255) / 256, b), dim3(256), sizeof(float) * j * 12, G4D_S(stream), v, j, skin_in,
                       weights, weights_batched ? (long long)v * j : 0ll, weights_batched > 0 ? weights_batched : 1, A, verts);
    return check_launch("g4d_lbs_pose_skin_f32");
}

// lbs() in three launches: [rigid transform with the joints from betas, writes the blend coefficients [betas | R - I]] ->
// [v_posed = v_template + [betas | R - I] . [shapedirs ; posedirs]] -> [skin].  The shape blend, the V-long joint regression and
// their (B,V,3) intermediate are gone; see garment4d_amd/lbs.py for the constants.
extern "C" int g4d_lbs_fused_f32(int b, int v, int j, int nb, int pose2rot, const float *betas, int betas_bstride, const float *pose,
                                 const float *v_template, const float *blend_dirs, const float *J_template, const float *J_shapedirs,
                                 const int *parents, const float *lbs_weights, float *coeff_scratch, float *A_out, float *posed_joints,
                                 float *v_posed_scratch, float *verts, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && b <= 65535 && v >= 0 && j > 0 && j <= 64 && nb >= 0 && nb <= 64, "g4d_lbs_fused_f32: bad sizes (J <= 64, NB <= 64)");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(betas && pose && v_template && blend_dirs && J_template && J_shapedirs && parents && lbs_weights && coeff_scratch && A_out &&
                    v_posed_scratch && verts, "g4d_lbs_fused_f32: null pointer");
    const int pf = (j - 1) * 9, nc = nb + pf;
    hipLaunchKernelGGL(lbs_rigid_kernel, dim3(b), dim3(64), 0, G4D_S(stream), j, pose2rot, pose, (const float *)nullptr, parents,
                       (float *)nullptr, posed_joints, A_out, coeff_scratch, nc, nb, J_template, J_shapedirs, betas, nb, betas_bstride);
    if (v > 0) {
        const size_t lds = sizeof(float) * ((size_t)kFB * nc + 3 * 64 * kFB);
        G4D_REQUIRE(lds <= 64 * 1024, "g4d_lbs_fused_f32: too many blend coefficients for LDS staging");
        dim3 grid((v * 3 + 63) / 64, (b + kFB - 1) / kFB);
        hipLaunchKernelGGL(lbs_pose_blend_kernel, grid, dim3(256), lds, G4D_S(stream), b, v * 3, nc, v_template, 0ll, coeff_scratch, blend_dirs,
                           v_posed_scratch);
        hipLaunchKernelGGL(lbs_skin_kernel, dim3((v + 255) / 256, b), dim3(256), sizeof(float) * j * 12, G4D_S(stream), v, j, v_posed_scratch,
                           lbs_weights, 0ll, 1, A_out, verts);
    }
    return check_launch("g4d_lbs_fused_f32");
}

// lbs() in one launch (lbs_one_kernel): same constants as g4d_lbs_fused_f32, no scratch.
extern "C" int g4d_lbs_one_supported(int j, int nb) { return j > 0 && j <= kOneJ && nb >= 0 && nb + (j - 1) * 9 <= kOneKS * kOneRows; }
extern "C" int g4d_lbs_one_f32(int b, int v, int j, int nb, int pose2rot, const float *betas, int betas_bstride, const float *pose,
                               const float *v_template, const float *blend_dirs, const float *J_template, const float *J_shapedirs,
                               const int *parents, const float *lbs_weights, float *A_out, float *posed_joints, float *verts,
                               g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && v > 0 && g4d_lbs_one_supported(j, nb), "g4d_lbs_one_f32: need V > 0, J <= 32 and NB + 9 (J - 1) <= 224 (use g4d_lbs_fused_f32)");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(betas && pose && v_template && blend_dirs && J_template && J_shapedirs && parents && lbs_weights && A_out && verts,
                "g4d_lbs_one_f32: null pointer");
    const int lds = (int)sizeof(float) * (kOneKS * kOneRows * kFB + kFB * kOneJ * (9 + 3 + 12 + 12) + kOneKS * kFB * 3 * 64);
    static unsigned long long attr = 0;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(lbs_one_kernel), lds, attr, "g4d_lbs_one_f32");
    if (rc) return rc;
    // frame groups per vertex tile: all of them side by side while that still leaves under ~3 workgroups per CU, else a share each
    const int vt = (v + 63) / 64, ngroups = (b + kFB - 1) / kFB;
    static const int gy_env = getenv("G4D_LBS_ONE_GROUPS") ? atoi(getenv("G4D_LBS_ONE_GROUPS")) : 0;   // tuning hook
    int gy = ngroups;
    if (gy_env > 0) gy = gy_env < ngroups ? gy_env : ngroups;
    else if ((long long)vt * ngroups > 768) { gy = (768 + vt - 1) / vt; if (gy > ngroups) gy = ngroups; if (gy < 1) gy = 1; }
    hipLaunchKernelGGL(lbs_one_kernel, dim3(vt, gy), dim3(64 * kOneKS), lds, G4D_S(stream), b, v, j, nb, pose2rot,
                       betas, betas_bstride, pose, v_template, blend_dirs, J_template, J_shapedirs, parents, lbs_weights, A_out, posed_joints, verts);
    return check_launch("g4d_lbs_one_f32");
}

// lbs() on the matrix pipe (lbs_frame_kernel + lbs_mfma_kernel, above).  ws: device workspace of g4d_lbs_mfma_ws_bytes(b, j) bytes
// (16-byte aligned), owned by the caller (the B operands of the two GEMMs in fragment order).
extern "C" int g4d_lbs_mfma_supported(int j, int nb) { return j > 0 && j <= kOneJ && nb >= 0 && nb + (j - 1) * 9 <= 4 * kMfKS; }
extern "C" long long g4d_lbs_mfma_ws_bytes(int b, int j) {
    const int js = j <= 24 ? 6 : 8;
    return (long long)((b > 0 ? b : 0) + 15) / 16 * 16 * 4 * (kMfKS + 12 * js) * (long long)sizeof(float);   // whole 16-frame tiles
}
extern "C" int g4d_lbs_mfma_f32(int b, int v, int j, int nb, int pose2rot, const float *betas, int betas_bstride, const float *pose,
                                const float *v_template, const float *blend_dirs, const float *J_template, const float *J_shapedirs,
                                const int *parents, const float *lbs_weights, float *A_out, float *posed_joints, float *verts, void *ws,
                                long long ws_bytes, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && v > 0 && g4d_lbs_mfma_supported(j, nb), "g4d_lbs_mfma_f32: need V > 0, J <= 32 and NB + 9 (J - 1) <= 224 (use g4d_lbs_fused_f32)");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(betas && pose && v_template && blend_dirs && J_template && J_shapedirs && parents && lbs_weights && A_out && verts && ws,
                "g4d_lbs_mfma_f32: null pointer");
    G4D_REQUIRE(ws_bytes >= g4d_lbs_mfma_ws_bytes(b, j) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "g4d_lbs_mfma_f32: workspace too small or not 16-byte aligned");
    const int js = j <= 24 ? 6 : 8;
    float *Cws = reinterpret_cast<float *>(ws), *Aws = Cws + (size_t)((b + 15) / 16) * 16 * 4 * kMfKS;
    hipLaunchKernelGGL(lbs_frame_kernel, dim3((b + 3) / 4), dim3(256), 0, G4D_S(stream), b, j, nb, js, pose2rot, betas, betas_bstride, pose, J_template,
                       J_shapedirs, parents, A_out, posed_joints, Cws, Aws);
    if (const int rc = check_launch("g4d_lbs_mfma_f32(frames)")) return rc;
    const int lds = (int)sizeof(float) * (2 * 3 * (kMfKS / 4) * 256 + 2 * js * 64 + kMfVT * 3 + 32);   // 86 KB of blend rows + weights + template
    const int nc = nb + (j - 1) * 9;
    if (js == 6) {
        static unsigned long long attr = 0;
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(lbs_mfma_kernel<6>), lds, attr, "g4d_lbs_mfma_f32")) return rc;
        hipLaunchKernelGGL(lbs_mfma_kernel<6>, dim3((v + kMfVT - 1) / kMfVT), dim3(64 * kMfWaves), lds, G4D_S(stream), b, v, j, nc, v_template, blend_dirs,
                           lbs_weights, Cws, Aws, verts);
    } else {
        static unsigned long long attr = 0;
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(lbs_mfma_kernel<8>), lds, attr, "g4d_lbs_mfma_f32")) return rc;
        hipLaunchKernelGGL(lbs_mfma_kernel<8>, dim3((v + kMfVT - 1) / kMfVT), dim3(64 * kMfWaves), lds, G4D_S(stream), b, v, j, nc, v_template, blend_dirs,
                           lbs_weights, Cws, Aws, verts);
    }
    return check_launch("g4d_lbs_mfma_f32");
}
