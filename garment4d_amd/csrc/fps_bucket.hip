// Bucketed furthest point sampling: identical output to fps.hip / the reference.  EXPERIMENTAL (opt-in with
// G4D_FPS_BUCKET=1): the pruning works -- 12 of 128 buckets swept per round at N=8192, M=1024 -- but on gfx950 the
// round is dominated by its fixed dependent chain (DPP step 19 cycles, LDS round trip 64, 16-wave barrier 60,
// measured by scripts/micro/clock.hip), and the per-bucket arg-max added here costs what the skipped sweeps save:
// 1.09 us/round either way.  Next step: prune against the global max (free) and keep one arg-max per wave.
//
// In round j only points closer to the new sample than their current min-distance change.  The cloud is sorted
// along a Morton curve (bitonic sort in LDS, once per launch) and cut into buckets of 64 consecutive points --
// one VGPR "slot" of one wave, buckets dealt round-robin to the 16 waves so that the buckets near a sample sit in
// different waves.  Each wave keeps, lane-distributed, the bounding box of each of its buckets and the bucket's
// current (max min-distance, tie rank, index).  A round is then
//
//   1. lane i tests box i against the new sample: d_box = dx*dx + dy*dy + dz*dz with dx = the gap between the
//      sample and the box along x, ... evaluated with the SAME fp32 operation order as the point distance.  Every
//      fp32 operation involved is monotone, so d(p) >= d_box holds for the rounded values of every point p in the box,
//      exactly, without any epsilon: if d_box >= bucket_max the sweep could not change a single min-distance and the
//      bucket is skipped -- pruning is bit-exact.
//   2. only the active buckets (typically 1-2 of 8 per wave, uniform branches over a ballot mask) are swept and their
//      (max, rank, index) refreshed by a DPP wave arg-max;
//   3. the wave's candidate = best of its lane-held bucket candidates (DPP row scan over 8 lanes); the workgroup
//      exchange (one barrier, parity-buffered 8-byte keys) and the LDS lookup of the winner are those of fps.hip.
//
// Tie-break = the reference's (smallest bit-reversed (k mod bs), then smallest k) through the same rank key, computed
// from ORIGINAL indices, so the permutation is invisible in the output.  Still bound by the serial round chain, but
// see the note at the top for where the time goes.
#include "g4d_common.h"

namespace g4d {

// --- pieces shared with fps.hip (kept local: both files are self-contained translation units)
__device__ __forceinline__ unsigned fpsb_rank(int k, int bs, int log2bs) {
    const unsigned c = (unsigned)k & (unsigned)(bs - 1);
    const unsigned q = (unsigned)k >> log2bs;
    const unsigned br = log2bs ? (__builtin_bitreverse32(c) >> (32 - log2bs)) : 0u;
    return (br << 16) | q;
}

__device__ __forceinline__ float fpsb_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int W>
__device__ __forceinline__ unsigned long long fpsb_row_max_u64(unsigned long long key) {
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

__device__ __forceinline__ float wave_min_f32(float v) {
    int out;
    asm volatile(
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %1, %0, 63\n\t"
        "s_nop 3"
        : "+v"(v), "=s"(out));
    return __int_as_float(out);
}

__device__ __forceinline__ unsigned part1by2(unsigned v) {  // spread the low 10 bits: b9..b0 -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// W waves, P buckets (slots) per wave; handles N <= 64*W*P points.  LDS: max(8*Npad sort keys, 12*N SoA) + 512.
template <int W, int P>
__global__ void __launch_bounds__(64 * W) fps_bucket_kernel(int n, int m, int bs, int log2bs, const float *__restrict__ xyz_all,
                                                           float *__restrict__ temp_all, int *__restrict__ idx_all) {
    constexpr int T = 64 * W, NPAD = T * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_raw);         // [2][16] candidate keys
    float *red = reinterpret_cast<float *>(smem_raw + 256);                                // [6][16] bbox partials
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem_raw + 1024);   // [NPAD] during the sort
    float *sx = reinterpret_cast<float *>(smem_raw + 1024);                                // SoA cloud afterwards
    float *sy = sx + n;
    float *sz = sy + n;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float *temp = temp_all ? temp_all + (size_t)blockIdx.x * n : nullptr;
    int *idx = idx_all + (size_t)blockIdx.x * m;
    const float INF = __builtin_inff();

    // ---- A. Morton keys of the cloud -------------------------------------------------------------------------
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    for (int k = t; k < n; k += T) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
        hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
    }
    lx = wave_min_f32(lx); ly = wave_min_f32(ly); lz = wave_min_f32(lz);
    hx = wave_max_f32(hx); hy = wave_max_f32(hy); hz = wave_max_f32(hz);
    if (lane == 0) { red[0 * 16 + wave] = lx; red[1 * 16 + wave] = ly; red[2 * 16 + wave] = lz;
                     red[3 * 16 + wave] = hx; red[4 * 16 + wave] = hy; red[5 * 16 + wave] = hz; }
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        lx = fminf(lx, red[0 * 16 + w]); ly = fminf(ly, red[1 * 16 + w]); lz = fminf(lz, red[2 * 16 + w]);
        hx = fmaxf(hx, red[3 * 16 + w]); hy = fmaxf(hy, red[4 * 16 + w]); hz = fmaxf(hz, red[5 * 16 + w]);
    }
    // the sort only has to be spatially coherent, not exact: any finite scale works (NaN/inf coordinates fall into cell 0)
    const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
    const float scale = 1023.0f / ext;
    for (int q = t; q < NPAD; q += T) {
        unsigned long long key = ~0ull;  // padding sorts to the end
        if (q < n) {
            const float x = xyz[q * 3 + 0], y = xyz[q * 3 + 1], z = xyz[q * 3 + 2];
            const unsigned cx = (unsigned)fminf(fmaxf((x - lx) * scale, 0.f), 1023.f);
            const unsigned cy = (unsigned)fminf(fmaxf((y - ly) * scale, 0.f), 1023.f);
            const unsigned cz = (unsigned)fminf(fmaxf((z - lz) * scale, 0.f), 1023.f);
            const unsigned code = part1by2(cx) | (part1by2(cy) << 1) | (part1by2(cz) << 2);
            key = ((unsigned long long)code << 32) | (unsigned)q;
        }
        keys[q] = key;
    }
    __syncthreads();
    // ---- B. bitonic sort of NPAD 64-bit keys in LDS ----------------------------------------------------------------
    for (int k = 2; k <= NPAD; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < NPAD / 2; i += T) {
                // i-th compare-exchange pair of this stage: lower index a with bit j clear
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b2 = a | j;
                const unsigned long long ka = keys[a], kb = keys[b2];
                const bool asc = (a & k) == 0;
                if ((ka > kb) == asc) { keys[a] = kb; keys[b2] = ka; }
            }
            __syncthreads();
        }
    }
    // ---- C. deal buckets of 64 sorted points to the waves, load the points into registers ----------------------------
    int pk[P];                     // original index of the slot's point (or -1)
    float px[P], py[P], pz[P], md[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int q = (i * W + wave) * 64 + lane;
        const unsigned long long key = keys[q];
        pk[i] = (key == ~0ull) ? -1 : (int)(unsigned)key;
    }
    __syncthreads();  // everybody has read its keys: the region becomes the SoA cloud
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool ok = pk[i] >= 0;
        const int k = ok ? pk[i] : 0;
        px[i] = ok ? xyz[k * 3 + 0] : 0.f;
        py[i] = ok ? xyz[k * 3 + 1] : 0.f;
        pz[i] = ok ? xyz[k * 3 + 2] : 0.f;
        md[i] = ok ? (temp ? temp[k] : 1e10f) : -2.f;  // -2: below every real min-distance, never a candidate
        if (ok) { sx[k] = px[i]; sy[k] = py[i]; sz[k] = pz[i]; }
    }
    // bucket boxes + initial bucket candidates, lane i holds bucket i
    float blx = INF, bly = INF, blz = INF, bhx = -INF, bhy = -INF, bhz = -INF;
    float smax = -2.f;
    unsigned srank = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool ok = pk[i] >= 0;
        const float a0 = wave_min_f32(ok ? px[i] : INF), a1 = wave_min_f32(ok ? py[i] : INF), a2 = wave_min_f32(ok ? pz[i] : INF);
        const float a3 = wave_max_f32(ok ? px[i] : -INF), a4 = wave_max_f32(ok ? py[i] : -INF), a5 = wave_max_f32(ok ? pz[i] : -INF);
        // initial candidate of the bucket (arg-max of md under the rank order)
        const float v = wave_max_f32(md[i]);
        const unsigned r = (md[i] == v && ok) ? fpsb_rank(pk[i], bs, log2bs) : 0xffffffffu;
        const unsigned rmin = wave_min_u32(r);
        if (lane == i) { blx = a0; bly = a1; blz = a2; bhx = a3; bhy = a4; bhz = a5; smax = v; srank = rmin; }
    }
    if (t == 0) idx[0] = 0;
    __syncthreads();

    float x1 = sx[0], y1 = sy[0], z1 = sz[0];
#ifdef G4D_FPS_DEBUG
    unsigned dbg_active = 0, dbg_maxw = 0;
#endif
    for (int j = 1; j < m; ++j) {
        // 1. which buckets can change?  gap between the sample and the box, per axis, same op order as the point distance
        const float gx = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f);
        const float gy = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f);
        const float gz = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
        const float dbox = gx * gx + gy * gy + gz * gz;
        const unsigned active = (unsigned)__builtin_amdgcn_ballot_w64(lane < P && dbox < smax);
#ifdef G4D_FPS_DEBUG
        dbg_active += __builtin_popcount(active);
#endif
        // 2. sweep the active buckets
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if ((active >> i) & 1u) {  // wave-uniform
                const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
                const float d = dx * dx + dy * dy + dz * dz;
                const float d2 = fpsb_min(d, md[i]);
                md[i] = d2;
                const float v = wave_max_f32(d2);
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(d2 == v);
                unsigned r;
                if (__builtin_popcountll(hit) == 1) {
                    const int kk = __builtin_amdgcn_readlane(pk[i], __builtin_ctzll(hit));
                    r = fpsb_rank(kk, bs, log2bs);
                } else {
                    const unsigned rr = (d2 == v && pk[i] >= 0) ? fpsb_rank(pk[i], bs, log2bs) : 0xffffffffu;
                    r = wave_min_u32(rr);
                }
                smax = (lane == i) ? v : smax;
                srank = (lane == i) ? r : srank;
            }
        }
        // 3. wave candidate = best bucket candidate (lanes 0..P-1), then the workgroup exchange
        unsigned long long key = (lane < P) ? (((unsigned long long)__float_as_uint(fmaxf(smax, 0.f)) << 32) | (unsigned)(~srank)) : 0ull;
        key = fpsb_row_max_u64<P>(key);
        const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), P - 1);
        const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, P - 1);
        unsigned long long *buf = slots + (j & 1) * 16;
        if (lane == 0) buf[wave] = ((unsigned long long)khi << 32) | klo;
        __syncthreads();
        const unsigned long long best = fpsb_row_max_u64<W>(buf[t & (W - 1)]);
        const unsigned rank = ~(unsigned)__builtin_amdgcn_readlane((int)(unsigned)best, W - 1);
        const unsigned c = log2bs ? (__builtin_bitreverse32(rank >> 16) >> (32 - log2bs)) : 0u;
        const int old = (int)(((rank & 0xffffu) << log2bs) | c);
        x1 = sx[old]; y1 = sy[old]; z1 = sz[old];
        if (t == 0) idx[j] = old;
    }
    if (temp) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (pk[i] >= 0) temp[pk[i]] = md[i];
    }
#ifdef G4D_FPS_DEBUG
    __syncthreads();
    if (lane == 0 && temp) atomicAdd(&temp[0], (float)dbg_active);  // debug only: total active (wave, bucket) sweeps
    (void)dbg_maxw;
#endif
}

template <int W, int P>
static int launch_bucket(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, hipStream_t s) {
    const size_t npad = (size_t)64 * W * P;
    const size_t body = npad * 8 > (size_t)n * 12 ? npad * 8 : (size_t)n * 12;
    const size_t lds = 1024 + body;
    auto kern = fps_bucket_kernel<W, P>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(64 * W), lds, s, n, m, bs, log2bs, xyz, temp, idx);
    return check_launch("g4d_fps_f32(bucketed)");
}

// Called by g4d_fps_f32 (fps.hip) for 2048 < n <= 8192.  Returns -1 when the shape is not covered.
int fps_bucket_dispatch(int b, int n, int m, int bs, int log2bs, const float *xyz, float *temp, int *idx, hipStream_t s) {
    if (n > 4096 && n <= 8192) return launch_bucket<16, 8>(b, n, m, bs, log2bs, xyz, temp, idx, s);
    if (n > 2048 && n <= 4096) return launch_bucket<16, 4>(b, n, m, bs, log2bs, xyz, temp, idx, s);
    return -1;
}

}  // namespace g4d
