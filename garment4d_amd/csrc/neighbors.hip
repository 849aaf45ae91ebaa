// Bucketed neighbour search on the spatial index the FPS kernel builds anyway.
//
// fps_bucket.hip sorts a level's cloud along a Morton curve and cuts it into 64-point blocks with bounding boxes; it needs
// them for its own pruning.  g4d_fps_indexed_f32 writes them out (sorted (x, y, z, original index) + block bounds), and the
// neighbour searches of the same level use them instead of scanning every point:
//
//   subset_index      the sampled points (level l+1 = cloud[fps idx]) in the SAME Morton order, with their own block bounds:
//                     one small workgroup per frame (inverse permutation in LDS, bitonic sort of <= 2048 keys);
//   three_nn_indexed  a wave takes 64 CONSECUTIVE points of the sorted cloud -- a compact patch of space -- and visits only
//                     the known blocks some lane can still improve on: the box distance is evaluated with the same un-fused
//                     fp32 expression as the point distance and every operation in it is monotone, so a skipped block
//                     cannot hold a point that beats (distance, index) lexicographically -- results are bit-identical to the
//                     index-order scan of the reference (interpolate_gpu.cu:9-52: strict < in ascending index order ==
//                     the three smallest (distance, index) pairs).
//
// Results are written at ORIGINAL indices with ORIGINAL neighbour indices: the permutation never leaves this file.
#include "g4d_common.h"

namespace g4d {

int fps_bucket_indexed(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float4 *sorted, float *boxes,
                       int *npad_out, hipStream_t s);  // fps_bucket.hip
int ref_block_size_pub(int n);                          // fps.hip

// ---- subset index: samples (fps idx) in Morton order + block bounds ------------------------------------------------------
// one workgroup (256 threads) per frame; npad <= 8192, mpad = m rounded up to 64 <= 2048
__global__ void __launch_bounds__(256) subset_index_kernel(int npad, int m, int mpad, const float4 *__restrict__ sorted_all,
                                                          const int *__restrict__ sample_idx_all, float4 *__restrict__ sub_all,
                                                          float *__restrict__ sub_boxes_all) {
    extern __shared__ unsigned smem_u[];
    unsigned short *inv = reinterpret_cast<unsigned short *>(smem_u);  // [npad] sorted position of each original index
    unsigned *keys = smem_u + npad / 2;                                // [mpad] (position << 11) | sample number
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float4 *sorted = sorted_all + (size_t)blockIdx.x * npad;
    const int *sample_idx = sample_idx_all + (size_t)blockIdx.x * m;
    float4 *sub = sub_all + (size_t)blockIdx.x * mpad;
    float *sub_boxes = sub_boxes_all + (size_t)blockIdx.x * (mpad / 64) * 6;
    for (int q = t; q < npad; q += 256) {
        const int orig = __float_as_int(sorted[q].w);
        if (orig >= 0) inv[orig] = (unsigned short)q;
    }
    __syncthreads();
    for (int j = t; j < mpad; j += 256) keys[j] = j < m ? (((unsigned)inv[sample_idx[j]] << 11) | (unsigned)j) : 0xffffffffu;
    __syncthreads();
    for (int k = 2; k <= mpad; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = t; i < mpad / 2; i += 256) {
                const int a = ((i & ~(jj - 1)) << 1) | (i & (jj - 1));
                const int b2 = a | jj;
                const unsigned ka = keys[a], kb = keys[b2];
                const bool asc = (a & k) == 0;
                if ((ka > kb) == asc) { keys[a] = kb; keys[b2] = ka; }
            }
            __syncthreads();
        }
    const float INF = __builtin_inff();
    for (int blk = wave; blk < mpad / 64; blk += 4) {
        const unsigned key = keys[blk * 64 + lane];
        const bool ok = key != 0xffffffffu;
        float4 p = make_float4(INF, INF, INF, __int_as_float(-1));
        if (ok) {
            const float4 s = sorted[key >> 11];
            p = make_float4(s.x, s.y, s.z, __int_as_float((int)(key & 2047u)));
        }
        sub[blk * 64 + lane] = p;
        float lo[3] = {ok ? p.x : INF, ok ? p.y : INF, ok ? p.z : INF}, hi[3] = {ok ? p.x : -INF, ok ? p.y : -INF, ok ? p.z : -INF};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
            }
        if (lane == 0) {
            float *o = sub_boxes + blk * 6;
            o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
        }
    }
}

// ---- three nearest known points of every cloud point ---------------------------------------------------------------------
__device__ __forceinline__ void nn3_insert_lex(float d, int k, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3) {
    const bool lt1 = d < b1 || (d == b1 && k < i1), lt2 = d < b2 || (d == b2 && k < i2), lt3 = d < b3 || (d == b3 && k < i3);
    const float nb3 = lt2 ? b2 : (lt3 ? d : b3);
    const int ni3 = lt2 ? i2 : (lt3 ? k : i3);
    const float nb2 = lt1 ? b1 : (lt2 ? d : b2);
    const int ni2 = lt1 ? i1 : (lt2 ? k : i2);
    b1 = lt1 ? d : b1; i1 = lt1 ? k : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}

constexpr int kNNMaxKnown = 2048;

// grid (npad / 256, B), 4 waves, each wave = 64 consecutive points of the sorted cloud
__global__ void __launch_bounds__(256) three_nn_indexed_kernel(int n, int npad, int mpad, const float4 *__restrict__ sorted_all,
                                                              const float4 *__restrict__ known_all, const float *__restrict__ kboxes_all,
                                                              float *__restrict__ dist2_all, int *__restrict__ idx_all) {
    __shared__ float4 sk[kNNMaxKnown];
    __shared__ float sbox[(kNNMaxKnown / 64) * 6];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.y;
    const float4 *known = known_all + (size_t)b * mpad;
    const int nkb = mpad >> 6;
    for (int j = t; j < mpad; j += 256) sk[j] = known[j];
    for (int j = t; j < nkb * 6; j += 256) sbox[j] = kboxes_all[(size_t)b * nkb * 6 + j];
    __syncthreads();
    const int pos = (blockIdx.x * 4 + wave) * 64 + lane;
    const float4 u = sorted_all[(size_t)b * npad + min(pos, npad - 1)];
    const int orig = pos < npad ? __float_as_int(u.w) : -1;
    const bool live = orig >= 0;   // padding slots carry +inf coordinates: every box test fails for them
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
        // Every block is scanned at most once per wave (a second scan would insert the same pairs again).  Pass 0 takes the
        // blocks that CONTAIN some lane's point (box distance 0): they seed tight bounds.  Pass 1 takes every other block that
        // some lane can still improve on (<=: an equal distance with a lower index also replaces).  A lane that did not ask
        // for a block still runs the inserts -- more candidates never hurt; a block nobody asks for when its turn comes can
        // never be needed later, the bounds only shrink.
        unsigned done = 0u;  // nkb <= 32
        for (int pass = 0; pass < 2; ++pass) {
            for (int kb = 0; kb < nkb; ++kb) {
                if ((done >> kb) & 1u) continue;
                const float *bx = sbox + kb * 6;
                const float ex = fmaxf(fmaxf(bx[0] - u.x, u.x - bx[3]), 0.f), ey = fmaxf(fmaxf(bx[1] - u.y, u.y - bx[4]), 0.f),
                            ez = fmaxf(fmaxf(bx[2] - u.z, u.z - bx[5]), 0.f);
                const float dbox = ex * ex + ey * ey + ez * ez;
                const bool want = live && (pass == 0 ? dbox == 0.f : dbox <= b3);
                if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
                done |= 1u << kb;
                const float4 *blk = sk + kb * 64;
#pragma unroll 4
                for (int j = 0; j < 64; ++j) {
                    const float4 kq = blk[j];
                    const float dx = u.x - kq.x, dy = u.y - kq.y, dz = u.z - kq.z;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const int k = __float_as_int(kq.w);   // padding: +inf coordinates -> d = +inf, never <= a live bound
                    if (__builtin_amdgcn_ballot_w64(live && d <= b3) != 0ull)  // wave-uniform skip: after the seed blocks most points lose
                        if (k >= 0) nn3_insert_lex(d, k, b1, b2, b3, i1, i2, i3);
                }
            }
        }
    }
    if (live && orig < n) {
        float *d2 = dist2_all + ((size_t)b * n + orig) * 3;
        int *ix = idx_all + ((size_t)b * n + orig) * 3;
        d2[0] = b1; d2[1] = b2; d2[2] = b3;
        ix[0] = i1; ix[1] = i2; ix[2] = i3;
    }
}

}  // namespace g4d

using namespace g4d;

extern "C" int g4d_fps_indexed_f32(int b, int n, int m, const float *xyz, float *temp, int *idx, float *sorted_pts, float *boxes,
                                   g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && m >= 0 && n > 2048 && n <= 8192, "g4d_fps_indexed_f32: needs 2048 < n <= 8192 (the bucketed FPS kernel)");
    if (b == 0 || m == 0) return G4D_OK;
    G4D_REQUIRE(xyz && idx && sorted_pts && boxes, "g4d_fps_indexed_f32: null pointer");
    const int bs = ref_block_size_pub(n);
    int log2bs = 0;
    while ((1 << log2bs) < bs) ++log2bs;
    int npad = 0;
    const int rc = fps_bucket_indexed(b, n, m, bs, log2bs, xyz, temp, idx, reinterpret_cast<float4 *>(sorted_pts), boxes, &npad,
                                      reinterpret_cast<hipStream_t>(stream));
    G4D_REQUIRE(rc >= 0, "g4d_fps_indexed_f32: shape not covered");
    return rc;
}

extern "C" int g4d_subset_index_f32(int b, int npad, int m, const float *sorted_pts, const int *sample_idx, float *sub_sorted,
                                    float *sub_boxes, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && (npad == 4096 || npad == 8192) && m >= 1 && m <= 2048, "g4d_subset_index_f32: npad in {4096, 8192}, 1 <= m <= 2048");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(sorted_pts && sample_idx && sub_sorted && sub_boxes, "g4d_subset_index_f32: null pointer");
    int mpad = 64;
    while (mpad < m) mpad <<= 1;   // bitonic sort: power of two
    const size_t lds = (size_t)npad * 2 + (size_t)mpad * 4;
    hipLaunchKernelGGL(subset_index_kernel, dim3(b), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), npad, m, mpad,
                       reinterpret_cast<const float4 *>(sorted_pts), sample_idx, reinterpret_cast<float4 *>(sub_sorted), sub_boxes);
    return check_launch("g4d_subset_index_f32");
}

extern "C" int g4d_three_nn_indexed_f32(int b, int n, int npad, int m, const float *sorted_pts, const float *known_sorted,
                                        const float *known_boxes, float *dist2, int *idx, g4d_stream_t stream) {
    G4D_REQUIRE(b >= 0 && b <= 65535 && n >= 1 && (npad == 4096 || npad == 8192) && n <= npad && m >= 1 && m <= kNNMaxKnown,
                "g4d_three_nn_indexed_f32: bad sizes (npad in {4096, 8192}, m <= %d)", kNNMaxKnown);
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(sorted_pts && known_sorted && known_boxes && dist2 && idx, "g4d_three_nn_indexed_f32: null pointer");
    int mpad = 64;
    while (mpad < m) mpad <<= 1;
    hipLaunchKernelGGL(three_nn_indexed_kernel, dim3(npad / 256, b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, npad, mpad,
                       reinterpret_cast<const float4 *>(sorted_pts), reinterpret_cast<const float4 *>(known_sorted), known_boxes, dist2, idx);
    return check_launch("g4d_three_nn_indexed_f32");
}
