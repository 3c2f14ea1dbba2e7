// csm_core.hip -- error state + version of libcsm355.so
#include "csm_common.h"
#include <cstring>

namespace csm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
}  // namespace csm

extern "C" const char *csm_last_error(void) { return csm::g_err; }
extern "C" int csm_version(void) { return 1000; }
extern "C" const char *csm_build_info(void) { return "libcsm355 gfx950 hipcc " __VERSION__ " -ffp-contract=off"; }
