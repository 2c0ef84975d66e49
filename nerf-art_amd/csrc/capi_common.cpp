// capi_common.cpp - error reporting shared by every entry point of libnerfart_hip.so.
#include "nerfart_common.h"
#include <string>

namespace nerfart {
static thread_local std::string g_last_error;
void set_last_error(const char* s) { g_last_error = s ? s : ""; }
int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
}  // namespace nerfart

extern "C" {
const char* nerfart_last_error(void) { return nerfart::g_last_error.c_str(); }
int nerfart_abi_version(void) { return 1; }
}
