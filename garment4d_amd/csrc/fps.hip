// Furthest point sampling for gfx950 -- replaces furthest_point_sampling_kernel(_launcher)
// (/root/reference/modules/pointnet2/pointnet2/src/sampling_gpu.cu:93-253).
//
// Semantics kept bit-for-bit: idx[0]=0; round j: mind[k] = min(d(k,old), mind[k])
// with d = dist2<FM>(dx, dy, dz) -- the squared distance under the process-wide contraction contract (g4d.h
// G4D_CONTRACT_*: nvcc's fused shape by default, or every operation rounded; g4d_common.h); next = arg-max mind[k], ties resolved like the reference's strided scan +
// shared-memory tree: let bs = min(1024, 2^floor(log2 N)); smallest bit-reversed (k mod bs) wins, then
// smallest k.  That total order lets the work be laid out for the hardware instead of copying the
// reference's thread/tree shape:
//
//   * one workgroup per cloud (the rounds are strictly serial; clouds are the parallel axis);
//   * the cloud and its running min-distances live in VGPRs for the whole kernel (N <= 16384): a thread
//     owns U residue classes x Q points, visited in ascending tie-rank so a strict `>` scan keeps the
//     right candidate with 3 VALU ops per point (cmp + 2 cndmask) on top of the 9 distance/min ops;
//   * per round ONE barrier: wave arg-max by DPP (row_shr / row_bcast, no LDS), each wave publishes a
//     16-byte candidate into a parity-double-buffered LDS slot, every thread folds the W candidates;
//   * the winner's coordinates come from an LDS SoA copy of the cloud (broadcast ds_read), never HBM.
//
// FPS is bound by the serial dependency chain (rounds x [VALU sweep + wave reduce + barrier + LDS
// round trip]), not by HBM or MFMA: algorithmic traffic is 12*N + 8*N + 4*M bytes per cloud.
#include "g4d_common.h"

namespace g4d {

template <int U>
__device__ __host__ constexpr int brev_small(int v) {
    // bit reversal on log2(U) bits, U in {1,2,4,8,16}
    int r = 0;
    for (int bit = 1, rb = U >> 1; bit < U; bit <<= 1, rb >>= 1)
        if (v & bit) r |= rb;
    return r;
}

// fminf() makes hipcc canonicalise the loop-carried operand (an extra v_max x,x per point); the bare
// instruction has the semantics wanted here (IEEE mode: a NaN operand yields the other operand).
__device__ __forceinline__ float min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct FpsCand {  // one wave's candidate for this round
    float val;
    unsigned rank;  // (bitrev(k mod bs) << 16) | (k / bs): smaller wins among equal val
    int k;
    int pad;
};

__device__ __forceinline__ unsigned fps_rank(int k, int bs, int log2bs) {
    const unsigned c = (unsigned)k & (unsigned)(bs - 1);
    const unsigned q = (unsigned)k >> log2bs;
    const unsigned br = log2bs ? (__builtin_bitreverse32(c) >> (32 - log2bs)) : 0u;
    return (br << 16) | q;
}

// max of W (4/8/16) 64-bit keys held by lanes 0..W-1 of each 16-lane row (Hillis-Steele with row_shr; lanes
// without a source read 0, which never wins).  Result valid in lane W-1.
template <int W>
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long key) {
#define G4D_STEP(CTRL)                                                                                       \
    {                                                                                                        \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)key, CTRL, 0xf, 0xf, true);        \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(key >> 32), CTRL, 0xf, 0xf, true); \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                    \
        key = o > key ? o : key;                                                                             \
    }
    G4D_STEP(0x111)
    G4D_STEP(0x112)
    if constexpr (W >= 8) G4D_STEP(0x114)
    if constexpr (W >= 16) G4D_STEP(0x118)
#undef G4D_STEP
    return key;
}

// Wave arg-max under (value desc, rank asc).  `best` per lane, `k` per lane.  Returns uniform results.
__device__ __forceinline__ void wave_argmax(float best, int k, int bs, int log2bs, float &wval, int &wk, unsigned &wrank) {
    wval = wave_max_f32(best);
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(best == wval);
    int lane;
    if (__builtin_popcountll(hit) == 1) {  // uniform branch; the common case
        lane = __builtin_ctzll(hit);
        wk = __builtin_amdgcn_readlane(k, lane);
        wrank = fps_rank(wk, bs, log2bs);
    } else {
        const unsigned r = (best == wval) ? fps_rank(k, bs, log2bs) : 0xffffffffu;
        wrank = wave_min_u32(r);
        lane = __builtin_ctzll(__builtin_amdgcn_ballot_w64(r == wrank));
        wk = __builtin_amdgcn_readlane(k, lane);
    }
}

// Register-resident FPS: T = 64*W threads; thread t owns classes t + u*T (u < U = bs/T) and, per class,
// points c + q*bs (q < Q).  Slot i = v*Q + q visits u = bitrev(v) so that tie-rank ascends with i.
template <int W, int U, int Q, int FM>
__global__ void __launch_bounds__(64 * W) fps_reg_kernel(int n, int m, int bs, int log2bs, const float *__restrict__ xyz_all,
                                                        float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ nx_all) {
    constexpr int T = 64 * W, P = U * Q;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sx = reinterpret_cast<float *>(smem_raw + 2 * 16 * 16);   // SoA copy of the cloud
    float *sy = sx + n;
    float *sz = sy + n;
    int *spick = reinterpret_cast<int *>(sz + n);                    // [m] the samples: written out once, after the loop

    const int t = threadIdx.x;
    const float *xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float *temp = temp_all + (size_t)blockIdx.x * n;
    int *idx = idx_all + (size_t)blockIdx.x * m;
    float *nx = nx_all ? nx_all + (size_t)blockIdx.x * m * 3 : nullptr;   // optional: the selected points themselves (gather fused)

    float px[P], py[P], pz[P], md[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + brev_small<U>(i / Q) * T + (i % Q) * bs;
        const bool ok = k < n;
        px[i] = ok ? xyz[k * 3 + 0] : 0.f;
        py[i] = ok ? xyz[k * 3 + 1] : 0.f;
        pz[i] = ok ? xyz[k * 3 + 2] : 0.f;
        md[i] = ok ? (temp_all ? temp[k] : 1e10f) : -2.f;  // -2 never beats the scan's initial best (-1)
        if (ok) { sx[k] = px[i]; sy[k] = py[i]; sz[k] = pz[i]; }
    }
    if (t == 0) spick[0] = 0;
    __syncthreads();

    float x1 = sx[0], y1 = sy[0], z1 = sz[0];
    const int wave = t >> 6;
    for (int j = 1; j < m; ++j) {
        float best = -1.f;
        int bslot = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
            const float d = dist2<FM>(dx, dy, dz);
            const float d2 = min_f32(d, md[i]);
            md[i] = d2;
            const bool gt = d2 > best;
            bslot = gt ? i : bslot;
            best = gt ? d2 : best;
        }
        int k;
        if constexpr (P == 1) k = t;
        else k = t + brev_small<U>(bslot / Q) * T + (bslot % Q) * bs;  // U,Q powers of two: shifts/masks
        float wval; int wk; unsigned wrank;
        wave_argmax(best, k, bs, log2bs, wval, wk, wrank);
        int old;
        if constexpr (W == 1) {
            old = wk;
        } else {
            // candidate = 64-bit key (value bits : ~rank); every wave publishes one, after the barrier lane l
            // picks up candidate l mod W and the W keys are folded by a DPP row scan (no serial compare chain)
            unsigned long long *buf = reinterpret_cast<unsigned long long *>(smem_raw) + (j & 1) * 16;
            if ((t & 63) == 0)
                buf[wave] = ((unsigned long long)__float_as_uint(fmaxf(wval, 0.f)) << 32) | (unsigned)(~wrank);
            __syncthreads();
            const unsigned long long key = row_max_u64<W>(buf[t & (W - 1)]);
            const unsigned rank = ~(unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, W - 1);
            const unsigned c = log2bs ? (__builtin_bitreverse32(rank >> 16) >> (32 - log2bs)) : 0u;
            old = (int)(((rank & 0xffffu) << log2bs) | c);
        }
        x1 = sx[old]; y1 = sy[old]; z1 = sz[old];
        if (t == 0) spick[j] = old;   // to LDS: a global store here would be drained (vmcnt(0)) by the next round's __syncthreads()
    }
    if constexpr (W > 1) __syncthreads();
    for (int j = t; j < m; j += T) {
        const int k = spick[j];
        idx[j] = k;
        if (nx) { nx[j * 3 + 0] = sx[k]; nx[j * 3 + 1] = sy[k]; nx[j * 3 + 2] = sz[k]; }
    }
    if (temp_all) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int k = t + brev_small<U>(i / Q) * T + (i % Q) * bs;
            if (k < n) temp[k] = md[i];
        }
    }
}

// Two consecutive FPS levels in ONE launch (the encoder's 1024 -> 256 -> 64): level B samples the centroids level A has just picked, which
// sit in this workgroup's LDS, so the second launch (3-5 us of the 16-batch mix, ~10 us of a single batch's latency) is not needed.
// Stage A is fps_reg_kernel<4, UA, 1> (n = 256 UA = the reference's block size, no scratch), stage B fps_reg_kernel<1, UB, 1> run by wave 0
// on the m1 = 64 UB picked points in pick order; same arithmetic, same tie-breaks, same outputs as the two launches.
template <int UA, int UB, int FM>
__global__ void __launch_bounds__(256) fps_reg_pair_kernel(int m2, const float *__restrict__ xyz_all, int *__restrict__ idx1_all,
                                                          float *__restrict__ nx1_all, int *__restrict__ idx2_all, float *__restrict__ nx2_all) {
    constexpr int T = 256, W = 4, nA = T * UA, bsA = nA, m1 = 64 * UB, bsB = m1;
    constexpr int log2bsA = UA == 1 ? 8 : (UA == 2 ? 9 : 10), log2bsB = UB == 1 ? 6 : (UB == 2 ? 7 : (UB == 4 ? 8 : 9));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sx = reinterpret_cast<float *>(smem_raw + 2 * 16 * 16);   // SoA copy of cloud A
    float *sy = sx + nA;
    float *sz = sy + nA;
    int *spick = reinterpret_cast<int *>(sz + nA);                   // [m1]
    float *bx = reinterpret_cast<float *>(spick + m1);               // cloud B = the picks of A, in pick order
    float *by = bx + m1;
    float *bz = by + m1;
    int *spick2 = reinterpret_cast<int *>(bz + m1);                  // [m2]
    const int t = threadIdx.x;
    const float *xyz = xyz_all + (size_t)blockIdx.x * nA * 3;

    // ---- stage A (fps_reg_kernel<4, UA, 1>)
    {
        float px[UA], py[UA], pz[UA], md[UA];
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int k = t + brev_small<UA>(i) * T;
            px[i] = xyz[k * 3 + 0]; py[i] = xyz[k * 3 + 1]; pz[i] = xyz[k * 3 + 2];
            md[i] = 1e10f;
            sx[k] = px[i]; sy[k] = py[i]; sz[k] = pz[i];
        }
        if (t == 0) spick[0] = 0;
        __syncthreads();
        float x1 = sx[0], y1 = sy[0], z1 = sz[0];
        const int wave = t >> 6;
        for (int j = 1; j < m1; ++j) {
            float best = -1.f;
            int bslot = 0;
#pragma unroll
            for (int i = 0; i < UA; ++i) {
                const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
                const float d = dist2<FM>(dx, dy, dz);
                const float d2 = min_f32(d, md[i]);
                md[i] = d2;
                const bool gt = d2 > best;
                bslot = gt ? i : bslot;
                best = gt ? d2 : best;
            }
            const int k = UA == 1 ? t : t + brev_small<UA>(bslot) * T;
            float wval; int wk; unsigned wrank;
            wave_argmax(best, k, bsA, log2bsA, wval, wk, wrank);
            unsigned long long *buf = reinterpret_cast<unsigned long long *>(smem_raw) + (j & 1) * 16;
            if ((t & 63) == 0) buf[wave] = ((unsigned long long)__float_as_uint(fmaxf(wval, 0.f)) << 32) | (unsigned)(~wrank);
            __syncthreads();
            const unsigned long long key = row_max_u64<W>(buf[t & (W - 1)]);
            const unsigned rank = ~(unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, W - 1);
            const unsigned c = __builtin_bitreverse32(rank >> 16) >> (32 - log2bsA);
            const int old = (int)(((rank & 0xffffu) << log2bsA) | c);
            x1 = sx[old]; y1 = sy[old]; z1 = sz[old];
            if (t == 0) spick[j] = old;
        }
        __syncthreads();
        int *idx = idx1_all + (size_t)blockIdx.x * m1;
        float *nx = nx1_all + (size_t)blockIdx.x * m1 * 3;
        for (int j = t; j < m1; j += T) {
            const int k = spick[j];
            idx[j] = k;
            const float x = sx[k], y = sy[k], z = sz[k];
            nx[j * 3 + 0] = x; nx[j * 3 + 1] = y; nx[j * 3 + 2] = z;
            bx[j] = x; by[j] = y; bz[j] = z;
        }
        __syncthreads();
    }
    if (t >= 64) return;
    // ---- stage B (fps_reg_kernel<1, UB, 1>): one wave, no barrier
    {
        float px[UB], py[UB], pz[UB], md[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int k = t + brev_small<UB>(i) * 64;
            px[i] = bx[k]; py[i] = by[k]; pz[i] = bz[k];
            md[i] = 1e10f;
        }
        if (t == 0) spick2[0] = 0;
        float x1 = bx[0], y1 = by[0], z1 = bz[0];
        for (int j = 1; j < m2; ++j) {
            float best = -1.f;
            int bslot = 0;
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
                const float d = dist2<FM>(dx, dy, dz);
                const float d2 = min_f32(d, md[i]);
                md[i] = d2;
                const bool gt = d2 > best;
                bslot = gt ? i : bslot;
                best = gt ? d2 : best;
            }
            const int k = UB == 1 ? t : t + brev_small<UB>(bslot) * 64;
            float wval; int wk; unsigned wrank;
            wave_argmax(best, k, bsB, log2bsB, wval, wk, wrank);
            const int old = wk;
            x1 = bx[old]; y1 = by[old]; z1 = bz[old];
            if (t == 0) spick2[j] = old;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int *idx = idx2_all + (size_t)blockIdx.x * m2;
        float *nx = nx2_all + (size_t)blockIdx.x * m2 * 3;
        for (int j = t; j < m2; j += 64) {
            const int k = spick2[j];
            idx[j] = k;
            nx[j * 3 + 0] = bx[k]; nx[j * 3 + 1] = by[k]; nx[j * 3 + 2] = bz[k];
        }
    }
}

// Any-N fallback: the reference's thread shape (class c = tid, strided scan from L2-resident global
// memory), the same DPP + one-barrier reduction.  1024 threads; threads >= bs idle.
template <int FM>
__global__ void __launch_bounds__(1024) fps_generic_kernel(int n, int m, int bs, int log2bs, const float *__restrict__ xyz_all,
                                                          float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ nx_all) {
    constexpr int W = 16;
    __shared__ FpsCand slots[2][W];
    const int t = threadIdx.x;
    const float *xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float *temp = temp_all + (size_t)blockIdx.x * n;
    int *idx = idx_all + (size_t)blockIdx.x * m;
    float *nx = nx_all ? nx_all + (size_t)blockIdx.x * m * 3 : nullptr;
    if (t == 0) idx[0] = 0;
    int old = 0;
    const int wave = t >> 6;
    for (int j = 1; j <= m; ++j) {   // the last pass only stores the coordinates of sample m - 1
        const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        if (t == 0 && nx) { nx[(j - 1) * 3 + 0] = x1; nx[(j - 1) * 3 + 1] = y1; nx[(j - 1) * 3 + 2] = z1; }
        if (j == m) break;
        float best = -1.f;
        int besti = 0;
        if (t < bs) {
            for (int k = t; k < n; k += bs) {
                const float dx = xyz[k * 3 + 0] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
                const float d = dist2<FM>(dx, dy, dz);
                const float d2 = fminf(d, temp[k]);
                temp[k] = d2;
                const bool gt = d2 > best;
                besti = gt ? k : besti;
                best = gt ? d2 : best;
            }
        } else {
            best = -3.f;  // below every real thread's -1: an idle thread never wins
        }
        // a thread whose points all failed `>` reports (best=-1, besti=0) like the reference; its rank
        // must be that of ITS class, not of point 0, so rank from the class id
        float wval; int wk; unsigned wrank;
        {
            wval = wave_max_f32(best);
            const unsigned r = (best == wval) ? ((t < bs ? ((log2bs ? (__builtin_bitreverse32((unsigned)t) >> (32 - log2bs)) : 0u) << 16) : 0xffff0000u) |
                                                 (unsigned)(besti >> log2bs))
                                              : 0xffffffffu;
            wrank = wave_min_u32(r);
            const int lane = __builtin_ctzll(__builtin_amdgcn_ballot_w64(r == wrank));
            wk = __builtin_amdgcn_readlane(besti, lane);
        }
        FpsCand *buf = slots[j & 1];
        if ((t & 63) == 0) { FpsCand c; c.val = wval; c.rank = wrank; c.k = wk; c.pad = 0; buf[wave] = c; }
        __syncthreads();
        FpsCand bc = buf[0];
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const FpsCand c = buf[w];
            const bool take = (c.val > bc.val) || (c.val == bc.val && c.rank < bc.rank);
            bc.val = take ? c.val : bc.val;
            bc.rank = take ? c.rank : bc.rank;
            bc.k = take ? c.k : bc.k;
        }
        old = __builtin_amdgcn_readfirstlane(bc.k);
        if (t == 0) idx[j] = old;
    }
}

// cuda_utils.h:10-14 opt_n_threads(), through double log() exactly as the reference.
static int ref_block_size(int work_size) {
    const int pow_2 = (int)(std::log((double)work_size) / std::log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

template <int W, int U, int Q, int FM>
static int launch_reg_fm(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    const size_t lds = 2 * 16 * 16 + (size_t)n * 12 + (size_t)m * 4;
    auto kern = fps_reg_kernel<W, U, Q, FM>;
    static unsigned long long attr_done = 0;  // one bit per device
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024 - 1024, attr_done, "g4d_fps_f32")) return rc;
    hipLaunchKernelGGL(kern, dim3(b), dim3(64 * W), lds, s, n, m, bs, log2bs, xyz, temp, idx, nx);
    return check_launch("g4d_fps_f32");
}

template <int W, int U, int Q>
static int launch_reg(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s) {
    G4D_WITH_FM(distance_contraction(), return (launch_reg_fm<W, U, Q, FM>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s)))
    return G4D_OK;
}

int fps_bucket_dispatch(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s);  // fps_bucket.hip
int fps_big_dispatch(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, float *nx, hipStream_t s);     // fps_big.hip
int fps_bucket_grid_launch(int b, int n, int m, int bs, int log2bs, const float *xyz, int *idx, float *nx, float rmax, void *grid_ws, hipStream_t s);


static int g_fps_force_w = -1;  // tuning hook: G4D_FPS_W=1|4|8|16|0(generic); unset = automatic (bucketed kernel for 2048 < n <= 8192)

}  // namespace g4d

static int fps_impl(int b, int n, int m, const float *xyz, float *temp, int *idx, float *nx, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "g4d_fps_f32: negative size (b=%d n=%d m=%d)", b, n, m);
    if (b == 0 || m == 0) return G4D_OK;  // sampling_gpu.cu:100  if (m <= 0) return;
    G4D_REQUIRE(n > 0, "g4d_fps_f32: n must be > 0 when m > 0");
    G4D_REQUIRE(xyz && idx, "g4d_fps_f32: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int bs = ref_block_size(n);
    int log2bs = 0;
    while ((1 << log2bs) < bs) ++log2bs;
    const int q = (n + bs - 1) / bs;
    if (g_fps_force_w < 0) {
        const char *e = getenv("G4D_FPS_W");
        g_fps_force_w = e ? atoi(e) : 99;
    }
    const int force = g_fps_force_w;
    // 4096 < n <= 8192: bucketed kernel with exact box pruning (fps_bucket.hip); G4D_FPS_BUCKET=0 turns it off
    static const int use_bucket = getenv("G4D_FPS_BUCKET") ? atoi(getenv("G4D_FPS_BUCKET")) : 1;  // 0 = off, 2 = also 2048 < n <= 4096
    if (use_bucket && force > 16 && n > (use_bucket >= 2 ? 2048 : 4096) && n <= 8192) {
        const int rc = fps_bucket_dispatch(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
        if (rc >= 0) return rc;
    }
    // 8192 < n <= 32768: the large-cloud bucketed kernel (fps_big.hip: min-distances in registers, coordinates from L2).  The
    // register-resident kernel below still takes 8192 < n <= 12800 when cloud + pick list fit LDS (G4D_FPS_BIG=2 sends those here too).
    static const int use_big = getenv("G4D_FPS_BIG") ? atoi(getenv("G4D_FPS_BIG")) : 1;
    const bool lds_ok = (size_t)n * 12 + (size_t)m * 4 + 512 <= 158 * 1024;   // SoA cloud + the pick list
    if (use_big && force > 16 && n > 8192 && n <= 32768 && (use_big >= 2 || !(lds_ok && n <= 12800))) {
        const int rc = fps_big_dispatch(b, n, m, bs, log2bs, xyz, temp, idx, nx, s);
        if (rc >= 0) return rc;
    }
#define G4D_FPS_CASE(W, U, Q) return launch_reg<W, U, Q>(b, n, m, bs, log2bs, xyz, temp, idx, nx, s)
    if (lds_ok && force != 0 && bs >= 64) {
        const int qp = q <= 1 ? 1 : q <= 2 ? 2 : q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : 0;
        // single wave: no barrier at all; preferred for small clouds
        const bool want_w1 = (force == 1) || (force > 16 && n <= 512);
        if (want_w1 && qp && (bs / 64) * qp <= 32) {
            const int u = bs / 64;
            if (u == 1 && qp == 1) G4D_FPS_CASE(1, 1, 1);
            if (u == 1 && qp == 2) G4D_FPS_CASE(1, 1, 2);
            if (u == 2 && qp == 1) G4D_FPS_CASE(1, 2, 1);
            if (u == 2 && qp == 2) G4D_FPS_CASE(1, 2, 2);
            if (u == 4 && qp == 1) G4D_FPS_CASE(1, 4, 1);
            if (u == 4 && qp == 2) G4D_FPS_CASE(1, 4, 2);
            if (u == 8 && qp == 1) G4D_FPS_CASE(1, 8, 1);
            if (u == 8 && qp == 2) G4D_FPS_CASE(1, 8, 2);
            if (u == 16 && qp == 1) G4D_FPS_CASE(1, 16, 1);
            if (u == 16 && qp == 2) G4D_FPS_CASE(1, 16, 2);
        }
        const int wsel = (force == 4 || force == 8 || force == 16) ? force : (n > 2048 ? 16 : 4);
        if (wsel == 16 && bs == 1024 && qp && qp <= 8) {
            if (qp == 1) G4D_FPS_CASE(16, 1, 1);
            if (qp == 2) G4D_FPS_CASE(16, 1, 2);
            if (qp == 4) G4D_FPS_CASE(16, 1, 4);
            if (qp == 8) G4D_FPS_CASE(16, 1, 8);
        }
        if (wsel >= 8 && bs >= 512 && qp && qp <= 8) {
            const int u = bs / 512;
            if (u == 1 && qp == 1) G4D_FPS_CASE(8, 1, 1);
            if (u == 1 && qp == 2) G4D_FPS_CASE(8, 1, 2);
            if (u == 2 && qp == 1) G4D_FPS_CASE(8, 2, 1);
            if (u == 2 && qp == 2) G4D_FPS_CASE(8, 2, 2);
            if (u == 2 && qp == 4) G4D_FPS_CASE(8, 2, 4);
            if (u == 2 && qp == 8) G4D_FPS_CASE(8, 2, 8);
        }
        if (bs >= 256 && qp) {
            const int u = bs / 256;
            if (u == 1 && qp == 1) G4D_FPS_CASE(4, 1, 1);
            if (u == 1 && qp == 2) G4D_FPS_CASE(4, 1, 2);
            if (u == 2 && qp == 1) G4D_FPS_CASE(4, 2, 1);
            if (u == 2 && qp == 2) G4D_FPS_CASE(4, 2, 2);
            if (u == 4 && qp == 1) G4D_FPS_CASE(4, 4, 1);
            if (u == 4 && qp == 2) G4D_FPS_CASE(4, 4, 2);
            if (u == 4 && qp == 4) G4D_FPS_CASE(4, 4, 4);
            if (u == 4 && qp == 8) G4D_FPS_CASE(4, 4, 8);
            if (u == 4 && qp == 16) G4D_FPS_CASE(4, 4, 16);
        }
        if (bs >= 64 && bs < 256 && qp && (bs / 64) * qp <= 32) {  // 64 <= n < 256: single wave
            const int u = bs / 64;
            if (u == 1 && qp == 1) G4D_FPS_CASE(1, 1, 1);
            if (u == 1 && qp == 2) G4D_FPS_CASE(1, 1, 2);
            if (u == 2 && qp == 1) G4D_FPS_CASE(1, 2, 1);
            if (u == 2 && qp == 2) G4D_FPS_CASE(1, 2, 2);
        }
    }
#undef G4D_FPS_CASE
    G4D_REQUIRE(temp, "g4d_fps_f32: temp scratch (B,N) is required for this N (the scratch-free kernels cover 64 <= N <= 32768)");
    G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL(fps_generic_kernel<FM>, dim3(b), dim3(1024), 0, s, n, m, bs, log2bs, xyz, temp, idx, nx))
    return check_launch("g4d_fps_f32(generic)");
}

extern "C" int g4d_fps_f32(int b, int n, int m, const float *xyz, float *temp, int *idx, g4d_stream_t stream) {
    return fps_impl(b, n, m, xyz, temp, idx, nullptr, stream);
}

extern "C" int g4d_fps_gather_f32(int b, int n, int m, const float *xyz, float *temp, int *idx, float *new_xyz, g4d_stream_t stream) {
    G4D_REQUIRE(new_xyz || b == 0 || m == 0, "g4d_fps_gather_f32: new_xyz is NULL");
    return fps_impl(b, n, m, xyz, temp, idx, new_xyz, stream);
}

// Two consecutive FPS levels (n -> m1 -> m2, level B samples the points level A picked) in one launch; idx1 / new_xyz1 / idx2 / new_xyz2
// are what g4d_fps_gather_f32(n -> m1) followed by g4d_fps_gather_f32(m1 -> m2) on new_xyz1 give.  Supported shapes:
// g4d_fps_gather_pair_supported (n = 1024 and m1 = 256, the inner levels of the Pointnet2MSGSEG encoder).
extern "C" int g4d_fps_gather_pair_supported(int n, int m1, int m2) { return n == 1024 && m1 == 256 && m2 >= 1 && m2 <= m1; }
extern "C" int g4d_fps_gather_pair_f32(int b, int n, int m1, int m2, const float *xyz, int *idx1, float *new_xyz1, int *idx2, float *new_xyz2,
                                       g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && g4d_fps_gather_pair_supported(n, m1, m2), "g4d_fps_gather_pair_f32: unsupported shape (n=%d m1=%d m2=%d)", n, m1, m2);
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(xyz && idx1 && new_xyz1 && idx2 && new_xyz2, "g4d_fps_gather_pair_f32: null pointer");
    const size_t lds = 2 * 16 * 16 + (size_t)n * 12 + (size_t)m1 * 16 + (size_t)m2 * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    G4D_WITH_FM(distance_contraction(), hipLaunchKernelGGL((fps_reg_pair_kernel<4, 4, FM>), dim3(b), dim3(256), lds, s, m2, xyz, idx1, new_xyz1, idx2, new_xyz2))
    return check_launch("g4d_fps_gather_pair_f32");
}

extern "C" int g4d_ball_grid_build_f32(int b, int n, float rmax, const float *xyz, void *grid, g4d_stream_t stream);
// g4d_fps_gather_f32 (no scratch) and g4d_ball_grid_build_f32 of the same clouds in one launch when the sampling takes the bucketed kernel
// (4096 < n <= 8192: workgroups [b, 2 b) of the sampling launch build the grids), two launches otherwise.  Outputs identical either way.
extern "C" int g4d_fps_gather_grid_f32(int b, int n, int m, const float *xyz, int *idx, float *new_xyz, float rmax, void *grid, g4d_stream_t stream) {
    using namespace g4d;
    G4D_REQUIRE(b >= 0 && n > 0 && m > 0 && rmax > 0.f, "g4d_fps_gather_grid_f32: bad sizes");
    if (b == 0) return G4D_OK;
    G4D_REQUIRE(xyz && idx && new_xyz && grid, "g4d_fps_gather_grid_f32: null pointer");
    static const int merged = getenv("G4D_FPS_GRID_MERGED") ? atoi(getenv("G4D_FPS_GRID_MERGED")) : 1;
    static const int use_bucket = getenv("G4D_FPS_BUCKET") ? atoi(getenv("G4D_FPS_BUCKET")) : 1;
    if (merged && use_bucket && !getenv("G4D_FPS_W") && !getenv("G4D_FPS_BUCKET_W") && !getenv("G4D_FPS_DEAL")) {
        const int bs = ref_block_size(n);
        int log2bs = 0;
        while ((1 << log2bs) < bs) ++log2bs;
        const int rc = fps_bucket_grid_launch(b, n, m, bs, log2bs, xyz, idx, new_xyz, rmax, grid, reinterpret_cast<hipStream_t>(stream));
        if (rc >= 0) return rc;
    }
    if (const int rc = g4d_fps_gather_f32(b, n, m, xyz, nullptr, idx, new_xyz, stream)) return rc;
    return g4d_ball_grid_build_f32(b, n, rmax, xyz, grid, stream);
}
