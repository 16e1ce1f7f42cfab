// Library-level entry points: error string, ABI version, version probe.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
zero_kernel(uint32_t* __restrict__ p, size_t words) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  const size_t tid = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  const size_t quads = (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? words / 4 : 0;
  uint4* q = reinterpret_cast<uint4*>(p);
  for (size_t i = tid; i < quads; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = quads * 4 + tid; i < words; i += stride) p[i] = 0u;
}

}  // namespace

namespace psa {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

int zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return PSA_OK;
  if (p == nullptr || !aligned(p, 4) || (bytes & 3) != 0) {
    set_error("zero_async: pointer/size must be 4-byte aligned");
    return PSA_ERR_INVALID_ARG;
  }
  const size_t words = bytes / 4;
  size_t blocks = (words / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(zero_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                     static_cast<uint32_t*>(p), words);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("zero_async: launch failed: ") + hipGetErrorString(e));
    return PSA_ERR_HIP;
  }
  return PSA_OK;
}
}  // namespace psa

extern "C" {

const char* psa_last_error(void) { return psa::g_last_error.c_str(); }

int psa_abi_version(void) { return PSA_ABI_VERSION; }

// reference csrc/version.cpp:14-22 returns CUDA_VERSION or -1; a HIP build
// must return -1 so paddle_sparse/__init__.py:18-32 skips the CUDA check.
int64_t psa_sparse_cuda_version(void) { return -1; }

}  // extern "C"
