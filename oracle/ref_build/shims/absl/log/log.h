// TEST INFRASTRUCTURE (oracle/ref_build): LOG / VLOG / CHECK on iostream; FATAL and failed CHECKs throw
// (the C wrapper turns that into an error code, where the reference would abort the process).
#ifndef DVREF_ABSL_LOG_H_
#define DVREF_ABSL_LOG_H_
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
namespace dvref_log {
struct Fatal : std::runtime_error {
  using std::runtime_error::runtime_error;
};
class Message {
 public:
  Message(const char* file, int line, int severity) : severity_(severity) { s_ << file << ":" << line << "] "; }
  ~Message() noexcept(false) {
    if (severity_ >= 3) throw Fatal(s_.str());
    if (severity_ >= 1 && verbose()) std::cerr << s_.str() << std::endl;
  }
  template <class T>
  Message& operator<<(const T& v) {
    s_ << v;
    return *this;
  }
  Message& operator<<(std::ostream& (*f)(std::ostream&)) {
    s_ << f;
    return *this;
  }
  static bool verbose() {
    static const bool v = std::getenv("DVREF_VERBOSE") != nullptr;
    return v;
  }
 private:
  std::ostringstream s_;
  int severity_;
};
struct Voidify {
  void operator&(Message&) {}
  void operator&(const Message&) {}
};
}  // namespace dvref_log
#define DVREF_SEV_INFO 0
#define DVREF_SEV_WARNING 1
#define DVREF_SEV_ERROR 2
#define DVREF_SEV_FATAL 3
#define DVREF_SEV_DFATAL 3
#define DVREF_SEV_QFATAL 3
#define LOG(sev) ::dvref_log::Message(__FILE__, __LINE__, DVREF_SEV_##sev)
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::dvref_log::Voidify() & LOG(sev)
#define VLOG(n) if (true) {} else LOG(INFO)
#define LOG_FIRST_N(sev, n) LOG(sev)
#define LOG_EVERY_N(sev, n) LOG(sev)
#endif
