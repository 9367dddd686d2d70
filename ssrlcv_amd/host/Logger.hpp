// ssrlcv_amd/host/Logger.hpp -- minimal stand-in for the reference's global `logger` (include/Logger.hpp:151-348,
// src/Logger.cpp:4): the same logger.info / logger.warn / logger.err surface (operator<< and printf) the hot-path
// host code uses.  The reference's CSV file, state marks and Jetson power sampling are observability and out of scope
// (SURVEY.md section 2 row 14); messages go to stderr when SSRLCV_LOG is set, errors always.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sstream>
#include <string>

namespace ssrlcv {
class Logger {
 public:
  struct Channel {
    const char* tag;
    bool always;
    std::mutex* mtx;
    void emit(const std::string& s) const {
      static const bool verbose = std::getenv("SSRLCV_LOG") != nullptr;
      if (!always && !verbose) return;
      std::lock_guard<std::mutex> g(*mtx);
      std::fprintf(stderr, "[%s] %s\n", tag, s.c_str());
    }
    // every `logger.info << a << b;` statement becomes one line, like the reference's Encapsulator
    struct Line {
      const Channel* ch;
      std::ostringstream os;
      explicit Line(const Channel* c) : ch(c) {}
      Line(Line&& o) : ch(o.ch), os(std::move(o.os)) { o.ch = nullptr; }
      ~Line() { if (ch) ch->emit(os.str()); }
      template <typename T> Line& operator<<(const T& v) { os << v; return *this; }
    };
    template <typename T> Line operator<<(const T& v) const { Line l(this); l.os << v; return l; }
    void printf(const char* fmt, ...) const {
      char buf[2048];
      va_list ap;
      va_start(ap, fmt);
      std::vsnprintf(buf, sizeof buf, fmt, ap);
      va_end(ap);
      emit(buf);
    }
  };
  std::mutex mtx;
  Channel info{"info", false, &mtx};
  Channel warn{"warn", false, &mtx};
  Channel err{"error", true, &mtx};
  void logState(const std::string& state) { info << "state: " << state; }
  void logState(const char* state) { info << "state: " << state; }
};
inline Logger& global_logger() { static Logger l; return l; }
}  // namespace ssrlcv
#define logger (::ssrlcv::global_logger())
