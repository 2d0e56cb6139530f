// Error reporting + ABI version for libsmot_emm.so.
#include "smot_common.h"

namespace smot {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace smot

extern "C" int smot_abi_version(void) { return SMOT_ABI_VERSION; }

extern "C" const char* smot_last_error(void) { return smot::g_err; }
