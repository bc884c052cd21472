// Stand-in (see tests/stubs/README.md): glog's stream macros as no-ops that still type-check their operands.
#pragma once
#include <ostream>
namespace google_stub {
struct NullStream {
  template <typename T>
  auto operator<<(const T&) -> NullStream& { return *this; }
};
struct Voidify {
  auto operator&(const NullStream&) -> void {}
};
}  // namespace google_stub
namespace google {
inline void InitGoogleLogging(const char*) {}
}  // namespace google
#define HS_STUB_STREAM(condition) (condition) ? (void)0 : google_stub::Voidify{} & google_stub::NullStream{}
#define LOG(severity) HS_STUB_STREAM(false)
#define LOG_IF(severity, condition) HS_STUB_STREAM(!(condition))
#define DLOG(severity) HS_STUB_STREAM(false)
#define DLOG_IF(severity, condition) HS_STUB_STREAM(!(condition))
#define CHECK(condition) HS_STUB_STREAM(condition)
#define CHECK_EQ(a, b) HS_STUB_STREAM((a) == (b))
#define CHECK_NE(a, b) HS_STUB_STREAM((a) != (b))
#define CHECK_LE(a, b) HS_STUB_STREAM((a) <= (b))
#define CHECK_LT(a, b) HS_STUB_STREAM((a) < (b))
#define CHECK_GE(a, b) HS_STUB_STREAM((a) >= (b))
#define CHECK_GT(a, b) HS_STUB_STREAM((a) > (b))
#define DCHECK(condition) CHECK(condition)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_NE(a, b) CHECK_NE(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
