// Error plumbing shared by every entry point of libvqhip (thread-local last-error string).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vqhip.h"

static thread_local char g_vq_err[512] = "";

void vq_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_vq_err, sizeof(g_vq_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vq_last_error(void) { return g_vq_err; }
extern "C" int vq_abi_version(void) { return 10; }   // 10: vq_moments out[0] = centred second moment of |x| (was sum x).  9: range events gain counter [2] (headroom); GroupNorm partial rows are (mean, M2) pairs.  8: VqDtype gains VQ_F16X2
