// Library-level entry points of libmedp_hip: version / arch probe and the thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "medp_hip.h"

static thread_local char g_err[512] = "";

void medp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* medp_last_error(void) { return g_err; }
extern "C" int medp_version(void) { return 2; }      // 2: MedpVitLayer carries the LayerNorm-fold operands (round 3)
extern "C" const char* medp_arch(void) { return "gfx950"; }

// ---- RNG epoch (see common.h) and a one-word device counter ------------------------------------------------------------
#include <stdint.h>
static const uint32_t* g_rng_epoch = nullptr;
const uint32_t* medp_rng_epoch_ptr() { return g_rng_epoch; }
extern "C" int medp_rng_set_epoch_ptr(const unsigned* dev_ptr) {
    g_rng_epoch = dev_ptr;
    return 0;
}
