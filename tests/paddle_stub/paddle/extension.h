// TEST-ONLY syntax stub.  This is NOT Paddle and builds nothing that runs: it
// declares just enough of the custom-op C++ API (the names and shapes that
// /root/reference/csrc/*.cpp and integration/paddle_shim/*.cc use — Tensor,
// optional, empty/full/zeros, PD_BUILD_OP ...) for `g++ -fsyntax-only` to type-
// check the shim's calls into include/paddle_sparse_hip.h: pointer types,
// argument order and count (tests/test_abi.py::test_shim_type_checks).  Paddle
// itself is not installable in this image; the real header replaces this one
// when a maintainer builds the shim (integration/paddle_shim/setup_ops_hip.py).
#pragma once

#include <cstdint>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <vector>

namespace paddle {

enum class DataType { UINT8, INT16, INT32, INT64, FLOAT16, BFLOAT16, FLOAT32, FLOAT64 };
inline size_t SizeOf(DataType) { return 0; }

struct Place {};
struct CPUPlace : Place {};

class Tensor {
 public:
  bool is_gpu() const { return true; }
  bool is_cpu() const { return false; }
  DataType dtype() const { return DataType::INT64; }
  Place place() const { return Place(); }
  int64_t numel() const { return 0; }
  std::vector<int64_t> shape() const { return {}; }
  template <typename T>
  T* data() { return nullptr; }
  template <typename T>
  const T* data() const { return nullptr; }
  void* data() { return nullptr; }
  const void* data() const { return nullptr; }
  void* stream() const { return nullptr; }
  Tensor copy_to(const Place&, bool) const { return Tensor(); }
};

template <typename T>
class optional {
 public:
  explicit operator bool() const { return false; }
  const T& get() const { return value_; }

 private:
  T value_;
};

inline std::string Optional(const std::string& name) { return name + "@OPTIONAL"; }

inline Tensor empty(const std::vector<int64_t>&, DataType, const Place&) { return Tensor(); }
inline Tensor zeros(const std::vector<int64_t>&, DataType, const Place&) { return Tensor(); }
inline Tensor full(const std::vector<int64_t>&, int64_t, DataType, const Place&) { return Tensor(); }

namespace experimental {
inline Tensor slice(const Tensor&, const std::vector<int64_t>&, const std::vector<int64_t>&,
                    const std::vector<int64_t>&, const std::vector<int64_t>&, const std::vector<int64_t>&) {
  return Tensor();
}
inline Tensor randint(int64_t, int64_t, const std::vector<int64_t>&, DataType, const Place&) { return Tensor(); }
inline Tensor max(const Tensor&, const std::vector<int64_t>&, bool) { return Tensor(); }
}  // namespace experimental

struct OpBuilderStub {
  explicit OpBuilderStub(const char*) {}
  OpBuilderStub& Inputs(std::vector<std::string>) { return *this; }
  OpBuilderStub& Outputs(std::vector<std::string>) { return *this; }
  OpBuilderStub& Attrs(std::vector<std::string>) { return *this; }
  OpBuilderStub& SetKernelFn(const void*) { return *this; }
  OpBuilderStub& SetInferShapeFn(const void*) { return *this; }
  OpBuilderStub& SetInferDtypeFn(const void*) { return *this; }
};

}  // namespace paddle

#define PD_CHECK(cond, ...)                                   \
  do {                                                        \
    if (!(cond)) throw std::runtime_error("PD_CHECK failed"); \
  } while (0)
#define PD_THROW(...) throw std::runtime_error("PD_THROW")
#define PD_KERNEL(fn) reinterpret_cast<const void*>(&fn)
#define PD_INFER_SHAPE(fn) reinterpret_cast<const void*>(&fn)
#define PD_INFER_DTYPE(fn) reinterpret_cast<const void*>(&fn)
#define PD_BUILD_OP(name) static ::paddle::OpBuilderStub __pd_op_##name = ::paddle::OpBuilderStub(#name)
