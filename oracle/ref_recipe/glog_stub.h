// glog_stub.h — the few glog macros the extracted reference code uses (CHECK_*, LOG, VLOG).
// TEST INFRASTRUCTURE: lets oracle/_ref/libref.so compile the reference's own functions without glog.
// A failed CHECK does not abort(): it throws, and the extern "C" wrappers in ref_shim.cpp turn that
// into an error code (the reference process would have died here).
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>

namespace refstub {
struct CheckFailure : std::runtime_error { using std::runtime_error::runtime_error; };
struct Voidify { void operator&(std::ostream&) {} };
struct Thrower {
  std::ostringstream os;
  std::ostream& stream() { return os; }
  ~Thrower() noexcept(false) { throw CheckFailure(os.str()); }
};
struct NullStream : std::ostream { NullStream() : std::ostream(nullptr) {} };
inline std::ostream& null_stream() { static NullStream s; return s; }
}  // namespace refstub

#define REFSTUB_CHECK(cond, text) (cond) ? (void)0 : refstub::Voidify() & refstub::Thrower().stream() << "Check failed: " text " "
#define CHECK(c) REFSTUB_CHECK((c), #c)
#define CHECK_EQ(a, b) REFSTUB_CHECK((a) == (b), #a " == " #b)
#define CHECK_NE(a, b) REFSTUB_CHECK((a) != (b), #a " != " #b)
#define CHECK_GE(a, b) REFSTUB_CHECK((a) >= (b), #a " >= " #b)
#define CHECK_GT(a, b) REFSTUB_CHECK((a) > (b), #a " > " #b)
#define CHECK_LE(a, b) REFSTUB_CHECK((a) <= (b), #a " <= " #b)
#define CHECK_LT(a, b) REFSTUB_CHECK((a) < (b), #a " < " #b)
#define LOG(sev) refstub::null_stream()
#define VLOG(n) refstub::null_stream()
