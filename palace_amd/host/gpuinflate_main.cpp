// gpuinflate <file.bam> [threads] -- every BGZF member of the file inflated on the device (palace_bgzf_inflate: one wavefront
// per member) and on the host (the loader's own decoder, zlib behind it), compared byte for byte; prints the times.
// A measuring and checking tool for N4 (host ingest); generateGraph's loader is the CPU path.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "../../include/palace_hip.h"
#include "bam.hpp"
#include "device_pick.hpp"

using namespace palace_host;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
#define CK(x) do { if ((x) != 0) { std::cerr << "gpuinflate: " << palace_last_error() << "\n"; return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { std::cerr << "usage: gpuinflate <file.bam> [threads]\n"; return 2; }
    const int threads = argc > 2 ? std::max(1, std::atoi(argv[2])) : static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    const int fd = ::open(argv[1], O_RDONLY);
    struct stat st{};
    if (fd < 0 || ::fstat(fd, &st) != 0) { std::cerr << "gpuinflate: cannot open " << argv[1] << "\n"; return 1; }
    const size_t size = static_cast<size_t>(st.st_size);
    const uint8_t *file = static_cast<const uint8_t *>(::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
    if (file == MAP_FAILED) { std::cerr << "gpuinflate: cannot map " << argv[1] << "\n"; return 1; }
    size_t total = 0;
    std::vector<BgzfMember> mem;
    try { mem = bgzf_members(file, size, &total); } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
    const int64_t n = static_cast<int64_t>(mem.size());
    std::vector<uint8_t> host(total + 8), dev(total + 8, 0xEE);
    // ---- host: every member, round-robin on the threads (what load_bam does) ----
    std::atomic<int64_t> failed{0};
    const auto h0 = Clock::now();
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                for (int64_t i = t; i < n; i += threads)
                    if (!inflate_member(file, size, mem[static_cast<size_t>(i)], host.data() + mem[static_cast<size_t>(i)].out_off)) failed++;
            });
        for (auto &th : pool) th.join();
    }
    const auto h1 = Clock::now();
    // ---- device ----
    palace_ctx *ctx = nullptr;
    CK(palace_ctx_create(pick_device(), &ctx));
    std::vector<int64_t> in_off(static_cast<size_t>(n)), out_off(static_cast<size_t>(n));
    std::vector<int32_t> in_len(static_cast<size_t>(n)), out_len(static_cast<size_t>(n)), status(static_cast<size_t>(n), -1);
    for (int64_t i = 0; i < n; i++) {
        const BgzfMember &m = mem[static_cast<size_t>(i)];
        in_off[static_cast<size_t>(i)] = static_cast<int64_t>(m.in_off); in_len[static_cast<size_t>(i)] = static_cast<int32_t>(m.in_len);
        out_off[static_cast<size_t>(i)] = static_cast<int64_t>(m.out_off); out_len[static_cast<size_t>(i)] = static_cast<int32_t>(m.out_len);
    }
    void *d_in, *d_out, *d_io, *d_il, *d_oo, *d_ol, *d_st;
    CK(palace_malloc(ctx, size + 8, &d_in)); CK(palace_malloc(ctx, total + 8, &d_out));
    CK(palace_malloc(ctx, static_cast<size_t>(n) * 8 + 8, &d_io)); CK(palace_malloc(ctx, static_cast<size_t>(n) * 8 + 8, &d_oo));
    CK(palace_malloc(ctx, static_cast<size_t>(n) * 4 + 8, &d_il)); CK(palace_malloc(ctx, static_cast<size_t>(n) * 4 + 8, &d_ol)); CK(palace_malloc(ctx, static_cast<size_t>(n) * 4 + 8, &d_st));
    CK(palace_memset(ctx, d_out, 0xEE, total + 8));
    const auto u0 = Clock::now();
    CK(palace_h2d(ctx, d_in, file, size));
    CK(palace_h2d(ctx, d_io, in_off.data(), static_cast<size_t>(n) * 8)); CK(palace_h2d(ctx, d_il, in_len.data(), static_cast<size_t>(n) * 4));
    CK(palace_h2d(ctx, d_oo, out_off.data(), static_cast<size_t>(n) * 8)); CK(palace_h2d(ctx, d_ol, out_len.data(), static_cast<size_t>(n) * 4));
    const auto u1 = Clock::now();
    float k_ms = 0, best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(palace_timer_begin(ctx));
        CK(palace_bgzf_inflate(ctx, static_cast<const uint8_t *>(d_in), n, static_cast<const int64_t *>(d_io), static_cast<const int32_t *>(d_il),
                               static_cast<const int64_t *>(d_oo), static_cast<const int32_t *>(d_ol), static_cast<uint8_t *>(d_out), static_cast<int32_t *>(d_st)));
        CK(palace_timer_end(ctx, &k_ms));
        best = std::min(best, k_ms);
    }
    const auto d0 = Clock::now();
    CK(palace_d2h(ctx, dev.data(), d_out, total));
    CK(palace_d2h(ctx, status.data(), d_st, static_cast<size_t>(n) * 4));
    const auto d1 = Clock::now();
    int64_t refused = 0, differ = 0;
    for (int64_t i = 0; i < n; i++) {
        const BgzfMember &m = mem[static_cast<size_t>(i)];
        if (status[static_cast<size_t>(i)] != 0) { refused++; continue; }
        if (std::memcmp(host.data() + m.out_off, dev.data() + m.out_off, m.out_len) != 0) differ++;
    }
    std::printf("%lld members, %.1f MB -> %.1f MB | host %d threads %.1f ms (%.2f GB/s out) | device: h2d %.1f ms, kernel %.2f ms (%.1f GB/s out), d2h %.1f ms | "
                "refused %lld, differing %lld, host failures %lld\n",
                static_cast<long long>(n), size / 1e6, total / 1e6, threads, ms(h0, h1), total / 1e6 / ms(h0, h1), ms(u0, u1), best, total / 1e6 / best, ms(d0, d1),
                static_cast<long long>(refused), static_cast<long long>(differ), static_cast<long long>(failed.load()));
    palace_ctx_destroy(ctx);
    return differ == 0 && failed.load() == 0 ? 0 : 1;
}
