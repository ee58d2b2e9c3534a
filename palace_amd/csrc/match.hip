// matching on gfx950: greedy maximum-weight matching of the conjugate graph by locally dominant
// arcs.  The reference implementation of `matching` is not available; this is the
// repository's own algorithm (DESIGN.md "matching"), pinned to oracle/match_oracle.cpp.
//
// A vertex owns two slots: "out" (its successor) and "in" (its predecessor).  Arc ids are ranks
// (0 = best).  Per round: every free out-slot proposes the lowest-id arc whose head's in-slot is
// free, every free in-slot proposes the lowest-id arc whose tail's out-slot is free; an arc
// proposed from both sides is taken.  The globally best remaining arc is always taken, and any
// arc taken this way is one the sequential greedy pass would also take, so the fixed point is the
// greedy matching -- independent of scheduling.  The graph here is the filtered conjugate graph
// (10^4..10^6 arcs): bandwidth-trivial, latency-bound; rounds are separate small launches.
#include "common.hpp"

namespace palace {

constexpr int32_t kNone = -1;

struct MatchArgs {
    int32_t n_vertices;
    int64_t n_arcs;
    const int32_t *src, *dst;
    const int64_t *out_off, *in_off;
    const int32_t *out_arcs, *in_arcs;
    const uint8_t *alive;
    int32_t *next, *prev, *next_arc, *want_out, *want_in;
    unsigned int *changed;
};

__global__ void match_propose_kernel(MatchArgs a)
{
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.n_vertices) return;
    int32_t wo = kNone, wi = kNone;
    if (a.alive[v]) {
        if (a.next[v] == kNone)
            for (int64_t k = a.out_off[v]; k < a.out_off[v + 1]; k++) {     // lists are ascending in arc id
                int32_t e = a.out_arcs[k], h = a.dst[e];
                if (a.alive[h] && a.prev[h] == kNone) { wo = e; break; }
            }
        if (a.prev[v] == kNone)
            for (int64_t k = a.in_off[v]; k < a.in_off[v + 1]; k++) {
                int32_t e = a.in_arcs[k], t = a.src[e];
                if (a.alive[t] && a.next[t] == kNone) { wi = e; break; }
            }
    }
    a.want_out[v] = wo;
    a.want_in[v] = wi;
}

__global__ void match_commit_kernel(MatchArgs a)
{
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.n_vertices) return;
    int32_t e = a.want_out[v];
    if (e == kNone) return;
    int32_t h = a.dst[e];
    if (a.want_in[h] != e) return;
    a.next[v] = h;                 // slot owners are unique: v owns out(v), and only arc e claims in(h)
    a.prev[h] = v;
    a.next_arc[v] = e;
    *a.changed = 1u;
}

__global__ void match_init_kernel(MatchArgs a)
{
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.n_vertices) return;
    a.next[v] = kNone; a.prev[v] = kNone; a.next_arc[v] = kNone;
}

}  // namespace palace

using namespace palace;

extern "C" int palace_match_greedy(palace_ctx *ctx, int32_t n_vertices, int64_t n_arcs, const int32_t *d_src,
                                   const int32_t *d_dst, const int64_t *d_out_off, const int32_t *d_out_arcs,
                                   const int64_t *d_in_off, const int32_t *d_in_arcs, const uint8_t *d_alive,
                                   int32_t *d_next, int32_t *d_prev, int32_t *d_next_arc, int32_t *rounds_out)
{
    PALACE_REQUIRE(ctx && n_vertices >= 0 && n_arcs >= 0, "bad argument");
    if (rounds_out) *rounds_out = 0;
    if (n_vertices == 0) return PALACE_OK;
    PALACE_REQUIRE(d_out_off && d_in_off && d_alive && d_next && d_prev && d_next_arc, "null device pointer");
    PALACE_REQUIRE(n_arcs == 0 || (d_src && d_dst && d_out_arcs && d_in_arcs), "null arc arrays");
    PALACE_REQUIRE(n_arcs < (1ll << 31), "arc ids must fit in int32");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_workspace(ctx, static_cast<size_t>(n_vertices) * 8 + 256);
    if (rc) return rc;
    MatchArgs a{};
    a.n_vertices = n_vertices; a.n_arcs = n_arcs; a.src = d_src; a.dst = d_dst;
    a.out_off = d_out_off; a.in_off = d_in_off; a.out_arcs = d_out_arcs; a.in_arcs = d_in_arcs;
    a.alive = d_alive; a.next = d_next; a.prev = d_prev; a.next_arc = d_next_arc;
    a.want_out = static_cast<int32_t *>(ctx->ws.ptr);
    a.want_in = a.want_out + n_vertices;
    a.changed = reinterpret_cast<unsigned int *>(ctx->d_small);
    const unsigned blocks = static_cast<unsigned>((n_vertices + 255) / 256);
    hipLaunchKernelGGL(match_init_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    int rounds = 0;
    for (;;) {
        PALACE_HIP_TRY(hipMemsetAsync(ctx->d_small, 0, 4, ctx->stream));
        hipLaunchKernelGGL(match_propose_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(match_commit_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
        PALACE_HIP_TRY(hipGetLastError());
        unsigned int changed = 0;
        PALACE_HIP_TRY(hipMemcpyAsync(&changed, ctx->d_small, 4, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
        rounds++;
        if (!changed) break;
        if (rounds > n_vertices + 2) { set_error("palace_match_greedy: no fixed point"); return PALACE_ESTATE; }
    }
    if (rounds_out) *rounds_out = rounds;
    return PALACE_OK;
}
