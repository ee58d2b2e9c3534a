// Depth stage on the host side: the mean the driver gets from
//     samtools depth <bam> | awk '{sum+=$3} END { print sum/NR }'          (palace:538-552)
// from the match segments the BAM loader collected, through palace_depth_sum_covered on the GPU.
#pragma once
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/palace_hip.h"
#include "bam.hpp"

namespace palace_host {

// what awk's `print sum/NR` writes: integral values as integers, everything else with OFMT = "%.6g"
inline std::string awk_number(double v)
{
    char buf[64];
    if (v == std::floor(v) && std::fabs(v) < 1e15) std::snprintf(buf, sizeof buf, "%lld", static_cast<long long>(v));
    else std::snprintf(buf, sizeof buf, "%.6g", v);
    return buf;
}

// returns 0 and the text in `out`; 1 when no position is covered (awk would stop with a division by zero); < 0 = library error
inline int first_depth(palace_ctx *ctx, const BamColumns &c, std::string &out, uint64_t *sum_out = nullptr, uint64_t *nr_out = nullptr,
                       std::vector<uint64_t> *contig_sum = nullptr, std::vector<uint64_t> *contig_covered = nullptr)
{
    const int32_t nt = static_cast<int32_t>(c.target_len.size());
    std::vector<int64_t> base(static_cast<size_t>(nt) + 1, 0);
    for (int32_t t = 0; t < nt; t++) base[static_cast<size_t>(t) + 1] = base[static_cast<size_t>(t)] + std::max(0, c.target_len[static_cast<size_t>(t)]);
    const int64_t n = static_cast<int64_t>(c.mseg_tid.size());
    void *d_tid = nullptr, *d_pos = nullptr, *d_len = nullptr, *d_tlen = nullptr, *d_base = nullptr;
    auto up = [&](const void *h, size_t bytes, void **d) {
        int rc = palace_malloc(ctx, bytes ? bytes : 1, d);
        return rc ? rc : palace_h2d(ctx, *d, h, bytes);
    };
    int rc = 0;
    uint64_t sum = 0, nr = 0;
    if ((rc = up(c.mseg_tid.data(), static_cast<size_t>(n) * 4, &d_tid)) == 0 && (rc = up(c.mseg_pos.data(), static_cast<size_t>(n) * 4, &d_pos)) == 0 &&
        (rc = up(c.mseg_len.data(), static_cast<size_t>(n) * 4, &d_len)) == 0 && (rc = up(c.target_len.data(), static_cast<size_t>(nt) * 4, &d_tlen)) == 0 &&
        (rc = up(base.data(), static_cast<size_t>(nt) * 8, &d_base)) == 0) {
        if (contig_sum && contig_covered) {
            void *d_cs = nullptr, *d_cc = nullptr;
            if ((rc = palace_malloc(ctx, static_cast<size_t>(nt) * 8 + 8, &d_cs)) == 0 && (rc = palace_malloc(ctx, static_cast<size_t>(nt) * 8 + 8, &d_cc)) == 0 &&
                (rc = palace_depth_per_contig(ctx, n, static_cast<int32_t *>(d_tid), static_cast<int32_t *>(d_pos), static_cast<int32_t *>(d_len), nt,
                                              static_cast<int32_t *>(d_tlen), static_cast<int64_t *>(d_base), base[static_cast<size_t>(nt)], &sum, &nr,
                                              static_cast<uint64_t *>(d_cs), static_cast<uint64_t *>(d_cc))) == 0) {
                contig_sum->resize(static_cast<size_t>(nt)); contig_covered->resize(static_cast<size_t>(nt));
                rc = palace_d2h(ctx, contig_sum->data(), d_cs, static_cast<size_t>(nt) * 8);
                if (!rc) rc = palace_d2h(ctx, contig_covered->data(), d_cc, static_cast<size_t>(nt) * 8);
            }
            palace_free(ctx, d_cs); palace_free(ctx, d_cc);
        } else {
            rc = palace_depth_sum_covered(ctx, n, static_cast<int32_t *>(d_tid), static_cast<int32_t *>(d_pos), static_cast<int32_t *>(d_len), nt,
                                          static_cast<int32_t *>(d_tlen), static_cast<int64_t *>(d_base), base[static_cast<size_t>(nt)], &sum, &nr);
        }
    }
    for (void *p : {d_tid, d_pos, d_len, d_tlen, d_base}) palace_free(ctx, p);
    if (rc) return rc;
    if (sum_out) *sum_out = sum;
    if (nr_out) *nr_out = nr;
    if (nr == 0) return 1;
    out = awk_number(static_cast<double>(sum) / static_cast<double>(nr));
    return 0;
}

}  // namespace palace_host
