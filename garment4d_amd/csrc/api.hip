// Library-level entry points of libg4d_hip: version + thread-local error text.
#include <stdarg.h>
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

int distance_contraction() {
    int m = __atomic_load_n(&g_contract, __ATOMIC_RELAXED);
    if (m < 0) {
        m = parse_contract(getenv("G4D_DIST_CONTRACT"));
        __atomic_store_n(&g_contract, m, __ATOMIC_RELAXED);
    }
    return m;
}
}  // namespace g4d

extern "C" int g4d_version(void) { return 200; /* 0.2.0: round 2 (g4d_ball_query_boxes_f32 takes 16-point sub-block bounds; new entry points, see include/g4d.h) */ }
extern "C" const char *g4d_last_error(void) { return g4d::g_err; }

extern "C" int g4d_get_distance_contraction(void) { return g4d::distance_contraction(); }
extern "C" int g4d_set_distance_contraction(int mode) {
    const int prev = g4d::distance_contraction();
    if (mode < G4D_CONTRACT_OFF || mode > G4D_CONTRACT_CHAIN) {
        g4d::set_error("g4d_set_distance_contraction: mode %d is not one of G4D_CONTRACT_OFF/NVCC/CHAIN", mode);
        return -1;
    }
    __atomic_store_n(&g4d::g_contract, mode, __ATOMIC_RELAXED);
    return prev;
}
