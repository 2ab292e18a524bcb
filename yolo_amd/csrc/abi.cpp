// Error string and version entry points of libyolo_amd.so.
#include <stdarg.h>

#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace yolo_amd

extern "C" const char* yolo_amd_last_error(void) { return yolo_amd::g_error; }
extern "C" int yolo_amd_abi_version(void) { return YOLO_AMD_ABI_VERSION; }
