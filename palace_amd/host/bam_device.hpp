// BGZF members inflated on the device beside the loader's threads (palace_bgzf_inflate: one wavefront per
// member).  The device decodes a whole BAM 2.4 x as fast as sixteen host threads (21 against 8.7 GB/s of
// output) and is idle while generateGraph reads its BAM: a helper thread takes batches of members off the BACK of the file (bam.hpp,
// BackMembers), sends their compressed bytes up, and copies the inflated bytes straight into the loader's stream
// (PALACE_BAM_DEVICE=0: the host alone; see device_inflate_helpers below for what it measured).  A member the
// device decoder refuses goes to the loader's own decoder (zlib behind it), as a member the CPU decoder refuses does.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../include/palace_hip.h"
#include "bam.hpp"

namespace palace_host {

// members per batch: the device holds ~7 000 of the decoder's wavefronts at a time (28 per CU), and a batch takes the time of its
// slowest member (~20 ms for 64 KiB) however small it is
constexpr size_t kDeviceInflateBatch = 8192;
constexpr double kDeviceBatchDeadlineS = 20.0;          // a batch takes ~0.1 s; behind this the device is taken to be hung

inline MemberHelper device_inflate_helper(int device)
{
    return [device](BackMembers &bm) {
        using Clock = std::chrono::steady_clock;
        const auto t_start = Clock::now();
        palace_ctx *ctx = nullptr;
        if (palace_ctx_create(device, &ctx)) return;                           // no device: everything stays with the loader's threads
        size_t B = kDeviceInflateBatch;
        if (const char *e = std::getenv("PALACE_BAM_DEVICE_BATCH")) B = static_cast<size_t>(std::max(64, std::min(16384, std::atoi(e))));   // (tuning runs)
        void *d_in = nullptr, *d_out = nullptr, *d_meta = nullptr;
        size_t in_cap = 0, out_cap = 0;                                        // sized by the batches claimed, not by the largest batch there could be
        // per member: in_off, out_off (int64), in_len, out_len, status (int32) -- one array each, one upload
        const size_t meta_bytes = B * (8 + 8 + 4 + 4 + 4);
        bool up = !palace_malloc(ctx, meta_bytes, &d_meta);
        auto room = [&](void *&p, size_t &cap, size_t need) {
            if (need + 64 <= cap) return true;
            if (p) palace_free(ctx, p);
            p = nullptr; cap = 0;
            if (palace_malloc(ctx, need + need / 8 + 64, &p)) return false;
            cap = need + need / 8 + 64;
            return true;
        };
        std::vector<uint8_t> meta(meta_bytes);
        int64_t *in_off = reinterpret_cast<int64_t *>(meta.data()), *out_off = in_off + B;
        int32_t *in_len = reinterpret_cast<int32_t *>(out_off + B), *out_len = in_len + B, *status = out_len + B;
        uint8_t *dm = static_cast<uint8_t *>(d_meta);
        size_t first = 0, n = 0;
        const bool trace = std::getenv("PALACE_TRACE") != nullptr;          // (laps need a sync per phase: traced runs only)
        auto ms_since = [](Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); };
        double t_up = 0, t_kernel = 0, t_down = 0, t_ready = ms_since(t_start);
        size_t batches = 0, members = 0;
        (void)t_ready;
        while (up && bm.claim(B, &first, &n)) {
            const BgzfMember &a = bm.member(first), &z = bm.member(first + n - 1);
            const uint64_t in0 = a.in_off, in1 = z.in_off + z.in_len, out0 = a.out_off, out1 = z.out_off + z.out_len;
            bool ok = room(d_in, in_cap, static_cast<size_t>(in1 - in0)) && room(d_out, out_cap, static_cast<size_t>(out1 - out0));
            for (size_t k = 0; k < n; k++) {
                const BgzfMember &m = bm.member(first + k);
                in_off[k] = static_cast<int64_t>(m.in_off - in0); in_len[k] = static_cast<int32_t>(m.in_len);
                out_off[k] = static_cast<int64_t>(m.out_off - out0); out_len[k] = static_cast<int32_t>(m.out_len);
                status[k] = -1;
            }
            auto t0 = Clock::now();
            ok = ok && !palace_h2d_async(ctx, d_in, bm.file_data + in0, static_cast<size_t>(in1 - in0)) && !palace_h2d_async(ctx, d_meta, meta.data(), meta_bytes);
            if (trace) { palace_sync(ctx); t_up += ms_since(t0); t0 = Clock::now(); }
            ok = ok && !palace_bgzf_inflate(ctx, static_cast<const uint8_t *>(d_in), static_cast<int64_t>(n), reinterpret_cast<const int64_t *>(dm),
                                      reinterpret_cast<const int32_t *>(dm + 16 * B), reinterpret_cast<const int64_t *>(dm + 8 * B),
                                      reinterpret_cast<const int32_t *>(dm + 20 * B), static_cast<uint8_t *>(d_out), reinterpret_cast<int32_t *>(dm + 24 * B));
            if (trace) { palace_sync(ctx); t_kernel += ms_since(t0); t0 = Clock::now(); }
            ok = ok && !palace_d2h_async(ctx, status, dm + 24 * B, 4 * n) && !palace_d2h_async(ctx, bm.out + out0, d_out, static_cast<size_t>(out1 - out0)) &&
                 !palace_mark(ctx, 0);
            if (ok && palace_mark_wait_for(ctx, 0, kDeviceBatchDeadlineS)) {
                // the batch never came back (a kernel or a copy that does not return): its members' bytes may still be written behind
                // our back, so nothing of this process can be trusted to finish -- say so and leave, instead of letting the loader's
                // walker and decode threads wait for done[] forever
                std::fprintf(stderr, "generateGraph: the device did not return a batch of BGZF members within %.0f s (%s); "
                                     "run with PALACE_BAM_DEVICE=0 to inflate on the host alone\n", kDeviceBatchDeadlineS, palace_last_error());
                std::fflush(stderr);
                std::_Exit(1);
            }
            if (trace) { t_down += ms_since(t0); batches++; members += n; }
            for (size_t k = 0; k < n; k++) bm.finished(first + k, ok && status[k] == 0);
            if (!ok) break;                                                    // the device is out of the game; what is left goes to the threads
        }
        if (trace)
            std::fprintf(stderr, "[bam/device] ready at %.1f ms; %zu batches, %zu members: up %.1f ms, kernel %.1f ms, down %.1f ms; left at %.1f ms\n",
                         t_ready, batches, members, t_up, t_kernel, t_down, ms_since(t_start));
        palace_free(ctx, d_in); palace_free(ctx, d_out); palace_free(ctx, d_meta);
        palace_ctx_destroy(ctx);
    };
}

// the helpers generateGraph starts: PALACE_BAM_DEVICE=<n> helper threads (default 2: one's copies overlap the other's kernel; 0 = the
// host alone).  Measured on the 1M-contig sample's BAM (30 590 members; tools/archive/r04ze.sh, settings alternated on one box): a helper is
// ready 80-100 ms into the run and takes one batch of 3 500-8 000 members (up 25-50 ms, kernel 40-65, down 35-70: the copies through
// pageable memory are the larger part), 11 000 members (38 %) are the device's by the time the sixteen threads have met them from the
// front; generateGraph 0.73 -> 0.68 s, with stage 04 in the process 0.86 -> 0.81 s.  (With the first version of the kernel -- window in
// LDS, four waves per CU, 350 ms for the file -- the helpers got 20 % of the members and the wall time did not move.)
inline std::vector<MemberHelper> device_inflate_helpers(int device)
{
    int n = 2;
    if (const char *e = std::getenv("PALACE_BAM_DEVICE")) n = std::max(0, std::min(4, std::atoi(e)));
    return std::vector<MemberHelper>(static_cast<size_t>(n), device_inflate_helper(device));
}

}  // namespace palace_host
