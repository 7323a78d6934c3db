// cudf/utilities/error.hpp -- exception types and check macros of the cudf API
// (reference: cpp/include/cudf/utilities/error.hpp:35,63-86,182-235,280).
#pragma once
#include <stdexcept>
#include <string>

namespace cudf {

struct logic_error : public std::logic_error {
  using std::logic_error::logic_error;
};
struct data_type_error : public std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};
struct cuda_error : public std::runtime_error {
  cuda_error(std::string const& message, int error) : std::runtime_error(message), _err{error} {}
  [[nodiscard]] int error_code() const { return _err; }

 protected:
  int _err;
};
struct fatal_cuda_error : public cuda_error {
  using cuda_error::cuda_error;
};

}  // namespace cudf

#define CUDF_STRINGIFY_DETAIL(x) #x
#define CUDF_STRINGIFY(x) CUDF_STRINGIFY_DETAIL(x)

// CUDF_EXPECTS(cond, reason [, exception_type]) -- default exception cudf::logic_error
#define CUDF_EXPECTS_3(_cond, _reason, _exc) \
  do {                                        \
    if (!(_cond)) { throw _exc{_reason}; }    \
  } while (0)
#define CUDF_EXPECTS_2(_cond, _reason) CUDF_EXPECTS_3(_cond, _reason, cudf::logic_error)
#define CUDF_GET_MACRO(_1, _2, _3, NAME, ...) NAME
#define CUDF_EXPECTS(...) CUDF_GET_MACRO(__VA_ARGS__, CUDF_EXPECTS_3, CUDF_EXPECTS_2, 1)(__VA_ARGS__)
#define CUDF_FAIL(_reason) throw cudf::logic_error { _reason }

// runtime / kernel-layer status -> exception (CUDF_CUDA_TRY of the reference)
#define CUDF_CUDA_TRY(_call)                                                                       \
  do {                                                                                             \
    int const _status = static_cast<int>(_call);                                                   \
    if (_status != 0) {                                                                            \
      throw cudf::cuda_error{std::string{"HIP/gx error at " __FILE__ ":" CUDF_STRINGIFY(__LINE__)  \
                                         ": code "} + std::to_string(_status), _status};           \
    }                                                                                              \
  } while (0)
