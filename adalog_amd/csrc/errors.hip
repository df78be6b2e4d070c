// Error reporting for the C ABI: every entry point returns 0 on success, a hipError_t value when a HIP call or a
// launch failed, or -1 for rejected arguments; adalog_last_error() then holds a description (thread-local).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void adalog_set_error(const char* where, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
extern "C" void adalog_set_error_msg(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
extern "C" const char* adalog_last_error(void) { return g_err; }
extern "C" int adalog_abi_version(void) { return 1; }
