// Library-level entry points: error string, ABI version, version probe.
#include "common.h"

namespace psa {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace psa

extern "C" {

const char* psa_last_error(void) { return psa::g_last_error.c_str(); }

int psa_abi_version(void) { return PSA_ABI_VERSION; }

// reference csrc/version.cpp:14-22 returns CUDA_VERSION or -1; a HIP build
// must return -1 so paddle_sparse/__init__.py:18-32 skips the CUDA check.
int64_t psa_sparse_cuda_version(void) { return -1; }

}  // extern "C"
