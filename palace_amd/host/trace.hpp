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
}  // namespace palace_host
