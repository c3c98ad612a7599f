// Error channel of the C-ABI: functions return non-zero and leave a message here; the Python
// binding turns that into RuntimeError (the reference's TORCH_CHECK convention, curope.cpp:54-59).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/spann3r_hip.h"

static thread_local char g_err[512] = "";

void sp3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sp3_last_error(void) { return g_err; }
extern "C" int sp3_version(void) { return 1; }
