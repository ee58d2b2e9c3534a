// Depth stage on gfx950: what the driver computes with
//     samtools depth -@ T <bam> > <bam>.depth ;  first_depth=$(awk '{sum+=$3} END { print sum/NR }' <bam>.depth)
// (palace:538-552) reduced to the two numbers the mean needs.  `samtools depth` (1.13 or later, default options) lists every
// reference position whose depth is > 0; a read counts at the positions of its M / = / X CIGAR operations -- deletions and
// reference skips do not count without -J -- unless one of its UNMAP, SECONDARY, QCFAIL, DUP flags is set.  So
//     sum = sum of the lengths of all such match segments,      NR = number of distinct positions they cover,
// and per-base depths are not needed for the mean: the segments mark a bit per reference position (all contigs laid end
// to end) and the covered positions are a population count.  Parity: unpinned (samtools is not in this image; the
// restatement in oracle/graph_oracle.cpp follows the samtools documentation).
#include "common.hpp"

namespace palace {

// one thread per match segment: sets bits [g, g + len) of the coverage bitmap, g = base[tid] + pos; adds len to *sum
__global__ __launch_bounds__(256) void depth_mark_kernel(const int32_t *__restrict__ seg_tid, const int32_t *__restrict__ seg_pos,
                                                         const int32_t *__restrict__ seg_len, int64_t n, int32_t n_targets,
                                                         const int32_t *__restrict__ tlen, const int64_t *__restrict__ base,
                                                         unsigned long long *__restrict__ bits, unsigned long long *__restrict__ sum,
                                                         unsigned long long *__restrict__ contig_sum)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    long long mine = 0;
    if (i < n) {
        const int32_t t = seg_tid[i];
        int64_t a = seg_pos[i], len = seg_len[i];
        if (t >= 0 && t < n_targets && a >= 0 && len > 0) {
            const int64_t L = tlen[t];
            const int64_t b = min(a + len, L);                     // (a record that runs past its contig is cut there)
            if (b > a) {
                mine = b - a;
                if (contig_sum) atomicAdd(&contig_sum[t], static_cast<unsigned long long>(mine));
                const int64_t g0 = base[t] + a, g1 = base[t] + b;   // [g0, g1)
                for (int64_t w = g0 >> 6; w <= (g1 - 1) >> 6; w++) {
                    const int64_t lo = max(g0, w << 6), hi = min(g1, (w + 1) << 6);
                    const unsigned long long m = ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1)) << (lo & 63);
                    atomicOr(&bits[w], m);
                }
            }
        }
    }
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(sum, static_cast<unsigned long long>(mine));
}

__global__ __launch_bounds__(256) void depth_popcount_kernel(const unsigned long long *__restrict__ bits, int64_t n_words,
                                                             unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_words; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        acc += __popcll(bits[i]);
    for (int d = 32; d; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// covered positions per contig: population count of its bit range (one thread per contig)
__global__ __launch_bounds__(256) void depth_contig_cover_kernel(const unsigned long long *__restrict__ bits, int32_t n_targets,
                                                                 const int32_t *__restrict__ tlen, const int64_t *__restrict__ base,
                                                                 unsigned long long *__restrict__ covered)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_targets) return;
    const int64_t g0 = base[t], g1 = g0 + max(0, tlen[t]);
    unsigned long long acc = 0;
    for (int64_t w = g0 >> 6; g1 > g0 && w <= (g1 - 1) >> 6; w++) {
        const int64_t lo = max(g0, w << 6), hi = min(g1, (w + 1) << 6);
        const unsigned long long m = ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1)) << (lo & 63);
        acc += __popcll(bits[w] & m);
    }
    covered[t] = acc;
}

}  // namespace palace

using namespace palace;

static int depth_impl(palace_ctx *ctx, int64_t n_segs, const int32_t *d_seg_tid, const int32_t *d_seg_pos,
                      const int32_t *d_seg_len, int32_t n_targets, const int32_t *d_tlen, const int64_t *d_tbase, int64_t total_len,
                      uint64_t *sum_out, uint64_t *covered_out, uint64_t *d_contig_sum, uint64_t *d_contig_covered)
{
    PALACE_REQUIRE(ctx && sum_out && covered_out && n_segs >= 0 && n_targets >= 0 && total_len >= 0, "bad argument");
    *sum_out = 0; *covered_out = 0;
    if (n_targets == 0) return PALACE_OK;
    if (n_segs == 0) {
        PALACE_HIP_TRY(hipSetDevice(ctx->device));
        if (d_contig_sum) PALACE_HIP_TRY(hipMemsetAsync(d_contig_sum, 0, static_cast<size_t>(n_targets) * 8, ctx->stream));
        if (d_contig_covered) PALACE_HIP_TRY(hipMemsetAsync(d_contig_covered, 0, static_cast<size_t>(n_targets) * 8, ctx->stream));
        return PALACE_OK;
    }
    PALACE_REQUIRE(d_seg_tid && d_seg_pos && d_seg_len && d_tlen && d_tbase, "null device pointer");
    PALACE_REQUIRE((n_segs + 255) / 256 < (1ll << 31), "too many segments for one launch");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t n_words = (total_len + 63) / 64 + 1;
    int rc = ensure_workspace(ctx, static_cast<size_t>(n_words) * 8);
    if (rc) return rc;
    unsigned long long *bits = static_cast<unsigned long long *>(ctx->ws.ptr);
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(ctx->d_small);
    PALACE_HIP_TRY(hipMemsetAsync(bits, 0, static_cast<size_t>(n_words) * 8, ctx->stream));
    PALACE_HIP_TRY(hipMemsetAsync(acc, 0, 16, ctx->stream));
    if (d_contig_sum) PALACE_HIP_TRY(hipMemsetAsync(d_contig_sum, 0, static_cast<size_t>(n_targets) * 8, ctx->stream));
    hipLaunchKernelGGL(depth_mark_kernel, dim3(static_cast<unsigned>((n_segs + 255) / 256)), dim3(256), 0, ctx->stream, d_seg_tid,
                       d_seg_pos, d_seg_len, n_segs, n_targets, d_tlen, d_tbase, bits, acc,
                       reinterpret_cast<unsigned long long *>(d_contig_sum));
    hipLaunchKernelGGL(depth_popcount_kernel, dim3(kCUs * 8), dim3(256), 0, ctx->stream, bits, n_words, acc + 1);
    if (d_contig_covered)
        hipLaunchKernelGGL(depth_contig_cover_kernel, dim3((n_targets + 255) / 256), dim3(256), 0, ctx->stream, bits, n_targets, d_tlen,
                           d_tbase, reinterpret_cast<unsigned long long *>(d_contig_covered));
    PALACE_HIP_TRY(hipGetLastError());
    uint64_t h[2];
    PALACE_HIP_TRY(hipMemcpyAsync(h, acc, 16, hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *sum_out = h[0]; *covered_out = h[1];
    return PALACE_OK;
}

extern "C" int palace_depth_sum_covered(palace_ctx *ctx, int64_t n_segs, const int32_t *d_seg_tid, const int32_t *d_seg_pos,
                                        const int32_t *d_seg_len, int32_t n_targets, const int32_t *d_tlen,
                                        const int64_t *d_tbase, int64_t total_len, uint64_t *sum_out, uint64_t *covered_out)
{
    return depth_impl(ctx, n_segs, d_seg_tid, d_seg_pos, d_seg_len, n_targets, d_tlen, d_tbase, total_len, sum_out, covered_out, nullptr, nullptr);
}

extern "C" int palace_depth_per_contig(palace_ctx *ctx, int64_t n_segs, const int32_t *d_seg_tid, const int32_t *d_seg_pos,
                                       const int32_t *d_seg_len, int32_t n_targets, const int32_t *d_tlen,
                                       const int64_t *d_tbase, int64_t total_len, uint64_t *sum_out, uint64_t *covered_out,
                                       uint64_t *d_contig_sum, uint64_t *d_contig_covered)
{
    PALACE_REQUIRE(n_targets == 0 || (d_contig_sum && d_contig_covered), "null device pointer");
    return depth_impl(ctx, n_segs, d_seg_tid, d_seg_pos, d_seg_len, n_targets, d_tlen, d_tbase, total_len, sum_out, covered_out, d_contig_sum,
                      d_contig_covered);
}
