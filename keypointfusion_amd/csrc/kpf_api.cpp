// Error channel and version of the C ABI (include/kpf.h).
#include "kpf_common.h"

static thread_local char g_err[512] = "";

void kpf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* kpf_last_error(void) { return g_err; }
extern "C" int kpf_abi_version(void) { return KPF_ABI_VERSION; }
