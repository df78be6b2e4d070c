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

// Name of the scoring kernel the last adalog_gemm_score / adalog_score_act_fused call on this thread launched (measurement
// only: bench.py attributes its per-launch event times to kernels with it).
static thread_local const char* g_last_kernel = "";
extern "C" void adalog_note_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* adalog_last_kernel(void) { return g_last_kernel; }
