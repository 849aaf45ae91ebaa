// Library-level entry points of libg4d_hip: version + thread-local error text.
#include <stdarg.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "g4d_common.h"

namespace g4d {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Distance contraction mode (include/g4d.h).  -1 = not initialised: first use reads G4D_DIST_CONTRACT.
static int g_contract = -1;

static int parse_contract(const char *e) {
    if (!e || !*e) return G4D_CONTRACT_NVCC;
    if (!strcmp(e, "0") || !strcmp(e, "off") || !strcmp(e, "none")) return G4D_CONTRACT_OFF;
    if (!strcmp(e, "2") || !strcmp(e, "chain") || !strcmp(e, "inner")) return G4D_CONTRACT_CHAIN;
    return G4D_CONTRACT_NVCC;
}

// per host thread override (-1: none): what a launcher called on this thread uses; the process-wide mode is only the default
static thread_local int t_contract = -1;

int distance_contraction() {
    if (t_contract >= 0) return t_contract;
    int m = __atomic_load_n(&g_contract, __ATOMIC_RELAXED);
    if (m < 0) {
        m = parse_contract(getenv("G4D_DIST_CONTRACT"));
        __atomic_store_n(&g_contract, m, __ATOMIC_RELAXED);
    }
    return m;
}
}  // namespace g4d

// ---- run-time tuning switches of the large-launch kernels (A/B experiments and tests that must drive a small shape through a kernel that would
// otherwise only see large ones).  A value set here wins over the environment variable of the same name in upper case with a G4D_ prefix
// (G4D_SA_TABLE_MIN_ROWS, ...), which wins over the built-in default.  Process-wide, like the distance contraction mode.
namespace g4d {
namespace {
struct Tune { const char *key; long long value; int state; };   // state: 0 unset, 1 from the environment / default (cached), 2 set by g4d_tuning_set
Tune g_tune[] = {{"sa_table_persistent", 0, 0}, {"sa_table_min_rows", 0, 0}, {"sa_table_128", 0, 0}, {"sa_table_oversub", 0, 0}, {"sa_table_dedup", 0, 0}, {"fp_table_persistent", 0, 0},
                 {"fp_table_min_rows", 0, 0}, {"gemm_tile", 0, 0}, {"gemm_tile_min_rows", 0, 0}, {"gemm_tile_min_cout", 0, 0}, {"gemm_tile_min_kpad", 0, 0}, {"fp_init_persistent", 0, 0}, {"fp_init_min_rows", 0, 0},
                 {"fp_head_bf16_persistent", 0, 0}, {"fp_head_bf16_min_rows", 0, 0}, {"sa_group_bf16_persistent", 0, 0}, {"sa_group_bf16_min_rows", 0, 0}};
}
// per-host-thread overrides (g4d_tuning_set_thread): an executor that holds its own tuning applies it around its launches without touching
// the process-wide table -- two executors driven from two threads do not see each other's settings (kernel selection happens on the host, at
// launch time, on the launching thread)
struct ThreadTune { long long value; bool set; };
thread_local ThreadTune t_tune[sizeof(g_tune) / sizeof(g_tune[0])] = {};

long long tuning(const char *key, long long dflt) {
    int i = 0;
    for (Tune &t : g_tune) {
        const int slot = i++;
        if (strcmp(t.key, key)) continue;
        if (t_tune[slot].set) return t_tune[slot].value;
        int st = __atomic_load_n(&t.state, __ATOMIC_ACQUIRE);
        if (st == 0) {
            char env[64] = "G4D_";
            size_t n = 4;
            for (const char *c = key; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
            env[n] = 0;
            const char *e = getenv(env);
            const long long v = e && *e ? atoll(e) : dflt;
            // state 0 -> 3 (being initialised) by ONE thread; a concurrent g4d_tuning_set (state 2) is never overwritten, a concurrent reader
            // that loses the race computes the same value from the same environment and returns it without storing
            int expect = 0;
            if (__atomic_compare_exchange_n(&t.state, &expect, 3, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
                __atomic_store_n(&t.value, v, __ATOMIC_RELAXED);
                expect = 3;
                __atomic_compare_exchange_n(&t.state, &expect, 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
                return v;
            }
            st = expect;
            if (st == 3) return v;
        }
        return __atomic_load_n(&t.value, __ATOMIC_RELAXED);
    }
    return dflt;
}
}  // namespace g4d

extern "C" int g4d_tuning_set(const char *key, long long value) {
    if (key)
        for (g4d::Tune &t : g4d::g_tune)
            if (!strcmp(t.key, key)) {
                __atomic_store_n(&t.value, value, __ATOMIC_RELAXED);
                __atomic_store_n(&t.state, 2, __ATOMIC_RELEASE);
                return G4D_OK;
            }
    g4d::set_error("g4d_tuning_set: unknown key '%s'", key ? key : "(null)");
    return G4D_EINVAL;
}

// set != 0: `value` overrides the key for launches made by the CALLING host thread; set == 0: the override is dropped (value ignored).
extern "C" int g4d_tuning_set_thread(const char *key, long long value, int set) {
    if (key) {
        int i = 0;
        for (g4d::Tune &t : g4d::g_tune) {
            const int slot = i++;
            if (!strcmp(t.key, key)) {
                g4d::t_tune[slot].value = value;
                g4d::t_tune[slot].set = set != 0;
                return G4D_OK;
            }
        }
    }
    g4d::set_error("g4d_tuning_set_thread: unknown key '%s'", key ? key : "(null)");
    return G4D_EINVAL;
}

extern "C" int g4d_version(void) { return 260; /* round 6: g4d_mlp_run / g4d_mlp_args, g4d_mlp_chain_group_table_ws_f32 + g4d_sa_table_ws_bytes / _supported (include/g4d.h); 206: round 2 */ }
extern "C" const char *g4d_last_error(void) { return g4d::g_err; }

extern "C" int g4d_get_distance_contraction(void) { return g4d::distance_contraction(); }
extern "C" int g4d_set_distance_contraction_thread(int mode) {
    const int prev = g4d::t_contract;
    if (mode < -1 || mode > G4D_CONTRACT_CHAIN) {
        g4d::set_error("g4d_set_distance_contraction_thread: mode %d is not -1 (no override) or one of G4D_CONTRACT_OFF/NVCC/CHAIN", mode);
        return -2;
    }
    g4d::t_contract = mode;
    return prev;
}

extern "C" int g4d_set_distance_contraction(int mode) {
    const int saved = g4d::t_contract;
    g4d::t_contract = -1;
    const int prev = g4d::distance_contraction();   // the process-wide mode, not this thread's override
    g4d::t_contract = saved;
    if (mode < G4D_CONTRACT_OFF || mode > G4D_CONTRACT_CHAIN) {
        g4d::set_error("g4d_set_distance_contraction: mode %d is not one of G4D_CONTRACT_OFF/NVCC/CHAIN", mode);
        return -1;
    }
    __atomic_store_n(&g4d::g_contract, mode, __ATOMIC_RELAXED);
    return prev;
}

// ---- g4d_copy_segments_f32: up to 4 device-to-device copies in ONE launch (the executor's per-step input hand-over: cloud, betas, pose --
// three runtime copy kernels cost a coalesced call of 30 steps 90 launches, ~0.5 ms of its 6)
namespace g4d {
struct CopySegs { float *dst[4]; const float *src[4]; long long n[4]; long long first_block[5]; };
__global__ void __launch_bounds__(256) copy_segments_kernel(const CopySegs c, int nseg) {
    int s = 0;
    while (s + 1 < nseg && (long long)blockIdx.x >= c.first_block[s + 1]) ++s;
    const long long i0 = ((long long)blockIdx.x - c.first_block[s]) * 1024 + threadIdx.x * 4;
    float *d = c.dst[s];
    const float *p = c.src[s];
    const long long n = c.n[s];
    if (i0 + 3 < n && ((reinterpret_cast<size_t>(d) | reinterpret_cast<size_t>(p)) & 15) == 0) {
        *reinterpret_cast<float4 *>(d + i0) = *reinterpret_cast<const float4 *>(p + i0);
    } else {
        for (int e = 0; e < 4; ++e)
            if (i0 + e < n) d[i0 + e] = p[i0 + e];
    }
}
}  // namespace g4d

extern "C" int g4d_copy_segments_f32(int nseg, float *const *dst, const float *const *src, const long long *nfloats, g4d_stream_t stream) {
    G4D_REQUIRE(nseg >= 0 && nseg <= 4 && (nseg == 0 || (dst && src && nfloats)), "g4d_copy_segments_f32: 0..4 segments");
    g4d::CopySegs c = {};
    long long blocks = 0;
    int k = 0;
    for (int i = 0; i < nseg; ++i) {
        G4D_REQUIRE(nfloats[i] >= 0 && (nfloats[i] == 0 || (dst[i] && src[i])), "g4d_copy_segments_f32: bad segment %d", i);
        if (nfloats[i] == 0) continue;
        c.dst[k] = dst[i]; c.src[k] = src[i]; c.n[k] = nfloats[i]; c.first_block[k] = blocks;
        blocks += (nfloats[i] + 1023) / 1024;
        ++k;
    }
    if (k == 0) return G4D_OK;
    c.first_block[k] = blocks;
    G4D_REQUIRE(blocks < (1ll << 31), "g4d_copy_segments_f32: too large");
    hipLaunchKernelGGL(g4d::copy_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c, k);
    return g4d::check_launch("g4d_copy_segments_f32");
}


// ---- g4d_mlp_run: the whole-stack launchers behind ONE argument block (include/g4d.h) ----------------------------------------------------
extern "C" unsigned g4d_mlp_args_size(void) { return (unsigned)sizeof(g4d_mlp_args); }

extern "C" int g4d_mlp_run(int family, const g4d_mlp_args *args, g4d_stream_t stream) {
    G4D_REQUIRE(args, "g4d_mlp_run: null argument block");
    G4D_REQUIRE(args->version == G4D_MLP_ARGS_VERSION, "g4d_mlp_run: argument block version %u, this library speaks %d", args->version, G4D_MLP_ARGS_VERSION);
    G4D_REQUIRE(args->size >= 16 && args->size <= sizeof(g4d_mlp_args), "g4d_mlp_run: argument block of %u bytes, this library's is %u (a newer caller?)",
                args->size, (unsigned)sizeof(g4d_mlp_args));
    g4d_mlp_args a;
    memset(&a, 0, sizeof(a));
    memcpy(&a, args, args->size);                 // an older caller's shorter block: the appended fields read as zero
    if (args->size < sizeof(g4d_mlp_args)) a.tap_layer = args->size > offsetof(g4d_mlp_args, tap_layer) ? a.tap_layer : -1;
    typedef const float *const *FPP;
    typedef const unsigned short *const *HPP;
    switch (family) {
        case G4D_MLP_STACK_F32:
            return g4d_mlp_stack_f32(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                     a.known_feats, a.skip, a.dist2, a.nn_idx, a.Vg, a.rowptr, a.colidx, a.vals, a.nlayers, reinterpret_cast<FPP>(a.W), a.scale,
                                     a.shift, a.Kpad, a.Cout, a.relu, a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, stream);
        case G4D_MLP_STACK_BF16:
            return g4d_mlp_stack_bf16(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                      a.known_feats, a.skip, a.dist2, a.nn_idx, a.Vg, a.rowptr, a.colidx, a.vals, a.nlayers, reinterpret_cast<HPP>(a.W), a.scale,
                                      a.shift, a.Kpad, a.Cout, a.relu, a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, stream);
        case G4D_MLP_WAVE_F32:
            G4D_REQUIRE(!a.tap_out, "g4d_mlp_run(G4D_MLP_WAVE_F32): the wave-autonomous kernel has no tap");
            return g4d_mlp_wave_f32(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                    a.known_feats, a.skip, a.dist2, a.nn_idx, a.Vg, a.rowptr, a.colidx, a.vals, a.nlayers, reinterpret_cast<FPP>(a.W), a.scale,
                                    a.shift, a.Kpad, a.Cout, a.relu, a.pool, a.out, a.ldo, a.col0, stream);
        case G4D_MLP_CHAIN_F32:
            return g4d_mlp_chain_f32(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                     a.known_feats, a.skip, a.dist2, a.nn_idx, a.nlayers, reinterpret_cast<FPP>(a.W), a.scale, a.shift, a.Kpad, a.Cout, a.relu,
                                     a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, stream);
        case G4D_MLP_CHAIN_BF16:
            if (a.unknown_grid)
                return g4d_mlp_chain_cells_bf16(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2,
                                                a.C1, a.known_feats, a.skip, a.dist2, a.nn_idx, a.nlayers, reinterpret_cast<HPP>(a.W), a.scale, a.shift, a.Kpad,
                                                a.Cout, a.relu, a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, a.unknown_grid, stream);
            return g4d_mlp_chain_bf16(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                      a.known_feats, a.skip, a.dist2, a.nn_idx, a.nlayers, reinterpret_cast<HPP>(a.W), a.scale, a.shift, a.Kpad, a.Cout, a.relu,
                                      a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, stream);
        case G4D_MLP_CHAIN_BF16X3:
            return g4d_mlp_chain_bf16x3(a.mode, a.rows, a.K0, a.X, a.ldx, a.N, a.P, a.S, a.C, a.use_xyz, a.xyz, a.new_xyz, a.feats, a.idx, a.n, a.m, a.C2, a.C1,
                                        a.known_feats, a.skip, a.dist2, a.nn_idx, a.nlayers, reinterpret_cast<HPP>(a.W), a.scale, a.shift, a.Kpad, a.Cout, a.relu,
                                        a.pool, a.out, a.ldo, a.col0, a.tap_layer, a.tap_out, a.tap_ld, stream);
    }
    g4d::set_error("g4d_mlp_run: unknown kernel family %d", family);
    return G4D_EINVAL;
}
