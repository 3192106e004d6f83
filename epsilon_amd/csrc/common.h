// Error handling shared by the host layer.
//
// The reference aborts a solve through glog CHECK -> HandleFailure -> longjmp -> python
// `_solve.error("CHECK failed")` (reference python/epopt/solvemodule.cc:158,185,245-248).
// Here every failed check throws eps::Error; the C-ABI entry points catch it, store the
// message for eps_last_error() and return a non-zero code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>

namespace eps {

struct Error : public std::runtime_error {
  explicit Error(const std::string& msg) : std::runtime_error(msg) {}
};

[[noreturn]] inline void Fail(const char* file, int line, const std::string& msg) {
  std::ostringstream os;
  os << file << ":" << line << ": " << msg;
  throw Error(os.str());
}

#define EPS_CHECK(cond)                                                     \
  do {                                                                      \
    if (!(cond)) ::eps::Fail(__FILE__, __LINE__, "CHECK failed: " #cond);   \
  } while (0)

#define EPS_CHECK_MSG(cond, msg)                                            \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::ostringstream _os;                                               \
      _os << "CHECK failed: " #cond ": " << msg;                            \
      ::eps::Fail(__FILE__, __LINE__, _os.str());                           \
    }                                                                       \
  } while (0)

#define EPS_FATAL(msg)                                                      \
  do {                                                                      \
    std::ostringstream _os;                                                 \
    _os << msg;                                                             \
    ::eps::Fail(__FILE__, __LINE__, _os.str());                             \
  } while (0)

#define EPS_HIP(expr)                                                       \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      std::ostringstream _os;                                               \
      _os << "HIP error " << hipGetErrorString(_e) << " in " #expr;         \
      ::eps::Fail(__FILE__, __LINE__, _os.str());                           \
    }                                                                       \
  } while (0)

enum DType : int { F32 = 0, F64 = 1 };

inline size_t DTypeSize(DType dt) { return dt == F32 ? 4 : 8; }

}  // namespace eps
