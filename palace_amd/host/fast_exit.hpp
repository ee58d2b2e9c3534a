// The caller of an executable waits until the process is gone -- and a process that mapped and touched gigabytes (the FASTQ /
// BAM files, the inflated stream, the HIP runtime) takes the kernel 0.1-0.2 s to tear down AFTER its last output byte is
// written (measured at the 1M-contig sample: generateGraph ~0.18 s, eref ~0.08 s).  These executables therefore do their work
// in a CHILD: main() forks first thing (no thread, no GPU state yet); the child is the program; the original process waits
// for ONE status byte, sent when every output is complete, flushed and closed, and exits with it at once, while the child's
// address space is torn down behind the caller's back.  A child that ends any other way (an error return, a signal) reports
// nothing: the original process then waits for it and passes its exit status on, so every failure path behaves as before.
// PALACE_NO_FORK=1 keeps everything in one process (debuggers, sanitizers).
// A process in which the GPU is ALREADY initialised when main() starts must not fork and go on using HIP in the child (ROCm
// does not support that: errors, or a hung GPU): that is the case under rocprofv3 and other tools whose preloaded library
// opens the device before main().  gpu_touched_before_main() looks for the signs -- a tool variable in the environment, a
// preload that names a GPU tool or the runtime, or /dev/kfd among the open descriptors -- and the program then stays one
// process, as with PALACE_NO_FORK=1.
#pragma once
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csignal>
#include <dirent.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include "trace.hpp"

extern char** environ;

namespace palace_host {

struct FastExit {
    int fd = -1;                      // >= 0: this is the worker; the status byte goes here
    // every output is written, flushed and closed: tell the caller's process, then leave (no destructors, no unmapping in user space)
    [[noreturn]] void done(int status = 0)
    {
        trace_since_launch("exit", "outputs complete");
        std::fflush(nullptr);
        if (fd >= 0) {
            ::close(1);                // a caller reading our stdout through a pipe must see its end now, not after the teardown
            ::close(2);
            const unsigned char st = static_cast<unsigned char>(status);
            ssize_t n;
            do n = ::write(fd, &st, 1); while (n < 0 && errno == EINTR);
        }
        ::_exit(status);
    }
};

inline bool gpu_touched_before_main()
{
    static const char* const exact[] = {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"};
    // A preloaded library as such says nothing (the GPU boxes of this project preload a guard into EVERY process: with a bare
    // "LD_PRELOAD is set" rule, round 5's first form, no run there ever took the forked start): it counts when it names a GPU tool
    // or the runtime itself; whatever else a preload does before main() shows in the descriptor scan below.
    static const char* const preload_marks[] = {"rocprof", "roctracer", "roctx", "rocsys", "omnitrace", "omniperf", "libamdhip", "libhsa-runtime", "librccl"};
    for (char** e = ::environ; e && *e; ++e) {
        const char* eq = std::strchr(*e, '=');
        if (!eq || !eq[1]) continue;                                    // unset or empty: not in force
        const size_t n = static_cast<size_t>(eq - *e);
        if (n == 10 && std::strncmp(*e, "LD_PRELOAD", 10) == 0) {
            for (const char* m : preload_marks)
                if (std::strstr(eq + 1, m)) return true;
            continue;
        }
        for (const char* k : exact)
            if (std::strlen(k) == n && std::strncmp(*e, k, n) == 0) return true;
        if (std::strncmp(*e, "ROCPROF", 7) == 0 || std::strncmp(*e, "ROCTRACER", 9) == 0) return true;     // ROCPROFILER_*, ROCPROFV3_*, ...
    }
    if (DIR* d = ::opendir("/proc/self/fd")) {                          // the kernel driver's node already open: a runtime is up in this process
        bool kfd = false;
        char link[64], path[300];
        while (const dirent* de = ::readdir(d)) {
            if (de->d_name[0] == '.') continue;
            std::snprintf(path, sizeof path, "/proc/self/fd/%s", de->d_name);
            const ssize_t k = ::readlink(path, link, sizeof link - 1);
            if (k <= 0) continue;
            link[k] = 0;
            if (std::strcmp(link, "/dev/kfd") == 0 || std::strncmp(link, "/dev/dri/renderD", 16) == 0) { kfd = true; break; }
        }
        ::closedir(d);
        if (kfd) return true;
    }
    return false;
}

inline FastExit fast_exit_begin()
{
    trace_since_launch("start", "main reached");
    if (std::getenv("PALACE_NO_FORK") || gpu_touched_before_main()) return {};
    int p[2];
    if (::pipe(p) != 0) return {};
    std::fflush(nullptr);
    const pid_t child = ::fork();
    if (child < 0) { ::close(p[0]); ::close(p[1]); return {}; }
    if (child == 0) {                  // the worker: the program proper
        ::close(p[0]);
        ::prctl(PR_SET_PDEATHSIG, SIGKILL);      // gone with the caller's process (which, after a status byte, only ends a teardown)
        return FastExit{p[1]};
    }
    ::close(p[1]);
    unsigned char st = 0;
    ssize_t n;
    do n = ::read(p[0], &st, 1); while (n < 0 && errno == EINTR);
    if (n == 1) { trace_since_launch("exit", "status byte read"); ::_exit(st); }           // outputs complete: the worker's teardown is nobody's wait
    int ws = 0;
    pid_t w;
    do w = ::waitpid(child, &ws, 0); while (w < 0 && errno == EINTR);
    ::_exit(w == child && WIFEXITED(ws) ? WEXITSTATUS(ws) : (w == child && WIFSIGNALED(ws) ? 128 + WTERMSIG(ws) : 1));
}

}  // namespace palace_host
