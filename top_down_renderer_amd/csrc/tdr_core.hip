// tdr_core.hip — error plumbing and the small host-side entry points of include/tdr.h.
#include "tdr_common.h"

static thread_local char g_err[512] = "";
int tdr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* tdr_last_error(void) { return g_err; }
extern "C" int tdr_set_error(int code, const char* msg) { return fail(code, "%s", msg ? msg : ""); }
extern "C" int tdr_version(void) { return 100; }
extern "C" int tdr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int tdr_rec_floats(int ncls) { return 4 * ((ncls + 1 + 3) / 4); }
