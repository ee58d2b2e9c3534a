// PALACE_TRACE=1: wall-clock laps of an executable's stages on stderr (the reference keeps such timers commented out,
// extract_ref.cpp:259, 746-749, 1292-1293).  Off by default: stderr stays clean for the driver's logs.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace palace_host {
struct Trace {
    const char *who;
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    explicit Trace(const char *w) : who(w), on(std::getenv("PALACE_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), last(t0) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[%s] %-28s %8.1f ms  (at %8.1f ms)\n", who, what,
                     std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};
// PALACE_TRACE=1 with PALACE_TRACE_T0=<ns since the epoch, taken by the caller just before it started the program> (tools/e2e_trace.sh):
// where the caller's wait goes that the laps do not see -- loading the program, the forked start, the way out.
inline void trace_since_launch(const char *who, const char *what)
{
    const char *t0 = std::getenv("PALACE_TRACE_T0");
    if (!t0 || !std::getenv("PALACE_TRACE")) return;
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[%s] %-28s           (+%8.1f ms after the caller's launch)\n", who, what, static_cast<double>(now - std::atoll(t0)) / 1e6);
}
}  // namespace palace_host
