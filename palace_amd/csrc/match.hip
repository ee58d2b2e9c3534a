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
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

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

// A round that changes nothing is a no-op, so rounds are enqueued in batches between host checks of the `changed`
// word (one stream round trip per batch, not per round; later outer rounds settle within one batch).
constexpr int kRoundsPerCheck = 4;

static int enqueue_rounds(palace_ctx *ctx, const MatchArgs &a, int n)
{
    const unsigned blocks = static_cast<unsigned>((a.n_vertices + 255) / 256);
    PALACE_HIP_TRY(hipMemsetAsync(a.changed, 0, 4, ctx->stream));
    for (int k = 0; k < n; k++) {
        hipLaunchKernelGGL(match_propose_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(match_commit_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    }
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

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
        unsigned int changed = 0;
        rc = enqueue_rounds(ctx, a, kRoundsPerCheck);
        if (rc) return rc;
        PALACE_HIP_TRY(hipMemcpyAsync(&changed, ctx->d_small, 4, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
        rounds += kRoundsPerCheck;
        if (!changed) break;
        if (rounds > n_vertices + 2 + kRoundsPerCheck) { set_error("palace_match_greedy: no fixed point"); return PALACE_ESTATE; }
    }
    if (rounds_out) *rounds_out = rounds;
    return PALACE_OK;
}

// ---- whole decomposition: GPU matching per round + host read-off ------------------------------
struct SubResult {                      // components of the arc-bearing sub-graph (small)
    std::vector<int64_t> off{0};
    std::vector<int32_t> verts, iter, open_at;
    std::vector<uint8_t> kind;
};

namespace palace {

// Result arrays hold one entry per component incl. ~n_segs bare segments (tens of MB).  Fresh allocations of that
// size are mmap'ed and cost a page fault per 4 KiB on first touch -- more than writing them -- so released blocks are
// kept (a handful, library-wide) and handed out again.
class BlockPool {
public:
    void *take(size_t bytes)
    {
        bytes = std::max<size_t>(bytes, 64);
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second >= bytes && free_[i].second <= 2 * bytes + (1 << 20)) {
                    void *p = free_[i].first;
                    sizes_[p] = free_[i].second;
                    free_.erase(free_.begin() + i);
                    return p;
                }
        }
        void *p = std::malloc(bytes);
        if (p) { std::lock_guard<std::mutex> g(mu_); sizes_[p] = bytes; }
        return p;
    }
    void give(void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu_);
        const size_t bytes = sizes_[p];
        sizes_.erase(p);
        if (free_.size() < 16) free_.emplace_back(p, bytes);
        else std::free(p);
    }

private:
    std::mutex mu_;
    std::vector<std::pair<void *, size_t>> free_;
    std::map<void *, size_t> sizes_;
};
static BlockPool g_result_pool;

struct Head { int32_t first; int64_t begin, end; uint8_t cycle; int32_t open; };

// Host temporaries of palace_match_decompose, kept in the context between calls for the same reason.
struct MatchScratch {
    std::vector<int32_t> new_id, old_id, ssrc, sdst, owner, pool, live;
    std::vector<int64_t> sub_copies, po, pi, left;
    std::vector<uint8_t> seen;
    std::vector<uint64_t> has_arc;
    std::vector<Head> heads, ordered;
    SubResult sub;
};
void free_match_scratch(MatchScratch *m) { delete m; }

// A scratch vector is worked on as a LOCAL object and handed back on scope exit: loops that also store bytes
// (char-typed stores may alias anything) would otherwise reload the vector's pointers from the heap-resident
// scratch struct after every such store.
template <class T>
struct Borrowed {
    T &home;
    T v;
    explicit Borrowed(T &h) : home(h), v(std::move(h)) {}
    ~Borrowed() { home = std::move(v); }
    Borrowed(const Borrowed &) = delete;
    Borrowed &operator=(const Borrowed &) = delete;
};

}  // namespace palace

struct palace_match_result {            // final result: arrays from the block pool, written exactly once
    int64_t n = 0;
    int64_t *off = nullptr;
    int32_t *verts = nullptr, *iter = nullptr, *open_at = nullptr;
    uint8_t *kind = nullptr;
    uint64_t *bare = nullptr;            // compact results: bit s set = segment s has no arc (bare path of round 0 [+ the aggressive round])
    int64_t n_bare = 0;
    ~palace_match_result()
    {
        for (void *p : {(void *)off, (void *)verts, (void *)iter, (void *)open_at, (void *)kind, (void *)bare}) palace::g_result_pool.give(p);
    }
};

namespace palace {
// Device arrays of one decomposition live in the context's grow-only workspace, behind the two
// want arrays palace_match_greedy keeps at its start: hipMalloc/hipFree per call would wait for every
// stream of the device (the eref stream runs beside this one).
struct Arena {
    char *base;
    size_t used;
    template <class T>
    T *take(size_t n)
    {
        T *p = reinterpret_cast<T *>(base + used);
        used += (std::max<size_t>(1, n) * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};
template <class T>
static int dev_copy(palace_ctx *ctx, Arena &ar, const T *h, size_t n, T **d)
{
    *d = ar.take<T>(n);
    if (n) PALACE_HIP_TRY(hipMemcpyAsync(*d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PALACE_OK;
}
}  // namespace palace

using palace::dev_copy;

static int decompose_core(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, int64_t n_arcs,
                          const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive,
                          palace::MatchScratch &ms, SubResult *res)
{
    const int32_t V = 2 * n_segs;
    const int64_t E = n_arcs;
    // Host staging in pinned memory (grow-only, owned by the context): the link arrays that come back every round,
    // laid out like the three device arrays so that one copy fetches them all, the liveness bytes, the `changed`
    // word, and the CSR of the arcs, so that every upload is a true asynchronous copy.
    const size_t v4 = (static_cast<size_t>(V) * 4 + 255) / 256 * 256, v1 = (static_cast<size_t>(V) + 255) / 256 * 256;
    const size_t o8 = (static_cast<size_t>(V + 1) * 8 + 255) / 256 * 256, e4 = (static_cast<size_t>(E) * 4 + 255) / 256 * 256;
    {
        int rc = palace::ensure_pinned(ctx, 3 * v4 + v1 + 256 + 2 * o8 + 4 * e4);
        if (rc) return rc;
    }
    char *pin = static_cast<char *>(ctx->pin.ptr);
    int32_t *next = reinterpret_cast<int32_t *>(pin), *prev = next + v4 / 4, *narc = prev + v4 / 4;
    uint8_t *alive = reinterpret_cast<uint8_t *>(pin + 3 * v4);
    unsigned int *changed = reinterpret_cast<unsigned int *>(pin + 3 * v4 + v1);
    int64_t *out_off = reinterpret_cast<int64_t *>(pin + 3 * v4 + v1 + 256), *in_off = out_off + o8 / 8;
    int32_t *out_arcs = reinterpret_cast<int32_t *>(in_off + o8 / 8), *in_arcs = out_arcs + e4 / 4;
    int32_t *p_src = in_arcs + e4 / 4, *p_dst = p_src + e4 / 4;
    // CSR by tail and by head; arc ids ascend inside every list because arcs arrive in rank order
    palace::Borrowed<std::vector<int64_t>> b_po(ms.po), b_pi(ms.pi), b_left(ms.left);
    palace::Borrowed<std::vector<int32_t>> b_owner(ms.owner), b_pool(ms.pool);
    palace::Borrowed<std::vector<uint8_t>> b_seen(ms.seen);
    palace::Borrowed<std::vector<palace::Head>> b_heads(ms.heads), b_ordered(ms.ordered);
    auto &po = b_po.v, &pi = b_pi.v;
    std::fill(out_off, out_off + V + 1, 0); std::fill(in_off, in_off + V + 1, 0);
    for (int64_t e = 0; e < E; e++) { out_off[src[e] + 1]++; in_off[dst[e] + 1]++; }
    for (int32_t v = 0; v < V; v++) { out_off[v + 1] += out_off[v]; in_off[v + 1] += in_off[v]; }
    po.assign(out_off, out_off + V); pi.assign(in_off, in_off + V);
    for (int64_t e = 0; e < E; e++) { out_arcs[po[src[e]]++] = static_cast<int32_t>(e); in_arcs[pi[dst[e]]++] = static_cast<int32_t>(e); }
    std::copy(src, src + E, p_src); std::copy(dst, dst + E, p_dst);
    int32_t *d_src = nullptr, *d_dst = nullptr, *d_oa = nullptr, *d_ia = nullptr, *d_next = nullptr, *d_prev = nullptr, *d_narc = nullptr;
    int64_t *d_oo = nullptr, *d_io = nullptr;
    uint8_t *d_alive = nullptr;
    const size_t greedy_bytes = (static_cast<size_t>(V) * 8 + 256 + 255) / 256 * 256;
    const size_t arena_bytes = greedy_bytes + 4 * (static_cast<size_t>(E) * 4 + 256) + 2 * (static_cast<size_t>(V + 1) * 8 + 256) +
                               3 * (static_cast<size_t>(V) * 4 + 256) + static_cast<size_t>(V) + 256;
    {
        int rc = palace::ensure_workspace(ctx, arena_bytes);
        if (rc) return rc;
    }
    palace::Arena ar{static_cast<char *>(ctx->ws.ptr), greedy_bytes};
    auto cleanup = [&] { (void)hipStreamSynchronize(ctx->stream); };   // host vectors must outlive the copies
#define TRY_OR_CLEAN(expr) do { int rc__ = (expr); if (rc__) { cleanup(); return rc__; } } while (0)
    TRY_OR_CLEAN(dev_copy(ctx, ar, p_src, E, &d_src)); TRY_OR_CLEAN(dev_copy(ctx, ar, p_dst, E, &d_dst));
    TRY_OR_CLEAN(dev_copy(ctx, ar, out_arcs, E, &d_oa)); TRY_OR_CLEAN(dev_copy(ctx, ar, in_arcs, E, &d_ia));
    TRY_OR_CLEAN(dev_copy(ctx, ar, out_off, V + 1, &d_oo)); TRY_OR_CLEAN(dev_copy(ctx, ar, in_off, V + 1, &d_io));
    d_next = ar.take<int32_t>(V); d_prev = ar.take<int32_t>(V); d_narc = ar.take<int32_t>(V);   // written by match_init_kernel
    if (reinterpret_cast<char *>(d_prev) - reinterpret_cast<char *>(d_next) != static_cast<ptrdiff_t>(v4) ||
        reinterpret_cast<char *>(d_narc) - reinterpret_cast<char *>(d_prev) != static_cast<ptrdiff_t>(v4)) {
        palace::set_error("decompose: arena layout"); cleanup(); return PALACE_ESTATE;
    }
    d_alive = ar.take<uint8_t>(V);
    if (ar.used > arena_bytes) { palace::set_error("decompose: arena accounting"); cleanup(); return PALACE_ESTATE; }

    using palace::Head;
    auto &left = b_left.v;
    auto &seen = b_seen.v;
    auto &owner = b_owner.v, &pool = b_pool.v;
    auto &heads = b_heads.v, &ordered = b_ordered.v;
    left.assign(copies, copies + n_segs);
    for (auto &c : left) c = std::max<int64_t>(1, c);
    seen.assign(V, 0); owner.assign(V, -1);
    heads.clear(); pool.clear();
    palace::Borrowed<std::vector<int32_t>> b_live(ms.live);
    auto &live = b_live.v;
    res->off.assign(1, 0); res->verts.clear(); res->iter.clear(); res->open_at.clear(); res->kind.clear();
    const int rounds = iterations + (aggressive ? 1 : 0);
    for (int t = 0; t < rounds; t++) {
        if (aggressive && t == rounds - 1) std::fill(left.begin(), left.end(), 1);
        live.clear();
        for (int32_t s = 0; s < n_segs; s++) {
            const bool on = left[s] > 0;
            alive[2 * s] = alive[2 * s + 1] = on;
            if (on) { live.push_back(2 * s); live.push_back(2 * s + 1); }
        }
        if (live.empty()) continue;                           // nothing left this round (an `aggressive` round may follow)
        // Late rounds often have no arc left between two live segments: then there is nothing to match (every live
        // vertex is a path of its own) and no reason to visit the GPU.
        bool arcs_alive = t == 0;
        for (int64_t e = 0; e < E && !arcs_alive; e++) arcs_alive = alive[src[e]] && alive[dst[e]];
        if (!arcs_alive) {
            for (const int32_t v : live) { next[v] = -1; prev[v] = -1; narc[v] = -1; }
        } else {
        // One stream round trip per outer round: liveness up (pinned, stream-ordered), a batch of matching rounds,
        // the `changed` word and the three link arrays (side by side) down; more batches only if it had not settled.
            MatchArgs a{};
            a.n_vertices = V; a.n_arcs = E; a.src = d_src; a.dst = d_dst;
            a.out_off = d_oo; a.in_off = d_io; a.out_arcs = d_oa; a.in_arcs = d_ia;
            a.alive = d_alive; a.next = d_next; a.prev = d_prev; a.next_arc = d_narc;
            a.want_out = static_cast<int32_t *>(ctx->ws.ptr);
            a.want_in = a.want_out + V;
            a.changed = reinterpret_cast<unsigned int *>(ctx->d_small);
            hipError_t e = hipMemcpyAsync(d_alive, alive, static_cast<size_t>(V), hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { palace::set_error("decompose: %s", hipGetErrorString(e)); cleanup(); return PALACE_EHIP; }
            hipLaunchKernelGGL(match_init_kernel, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, ctx->stream, a);
            for (int done = 0;; done += kRoundsPerCheck) {
                int rc = enqueue_rounds(ctx, a, kRoundsPerCheck);
                if (rc) { cleanup(); return rc; }
                e = hipMemcpyAsync(changed, ctx->d_small, 4, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(next, d_next, 3 * v4, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) { palace::set_error("decompose: %s", hipGetErrorString(e)); cleanup(); return PALACE_EHIP; }
                if (!*changed) break;
                if (done > V + 2) { palace::set_error("decompose: no fixed point"); cleanup(); return PALACE_ESTATE; }
            }
        }
        // Host read-off over the vertices that are alive this round only (after round 0 that is a small part of the
        // graph); `seen` and `owner` entries are reset for exactly those vertices afterwards.
        heads.clear();
        pool.clear();
        for (const int32_t v : live) {                        // open paths, one representative per conjugate pair
            if (seen[v] || prev[v] >= 0) continue;
            const int64_t b = static_cast<int64_t>(pool.size());
            for (int32_t x = v; x >= 0; x = next[x]) { pool.push_back(x); seen[x] = 1; }
            const int64_t e = static_cast<int64_t>(pool.size());
            for (int64_t k = b; k < e; k++) seen[pool[k] ^ 1] = 1;
            if ((pool[e - 1] ^ 1) < pool[b]) {
                std::reverse(pool.begin() + b, pool.begin() + e);
                for (int64_t k = b; k < e; k++) pool[k] ^= 1;
            }
            heads.push_back({pool[b], b, e, 0, 0});
        }
        for (const int32_t v : live) {                        // closed walks
            if (seen[v]) continue;
            const int64_t b = static_cast<int64_t>(pool.size());
            for (int32_t x = v; !seen[x]; x = next[x]) { pool.push_back(x); seen[x] = 1; }
            const int64_t e = static_cast<int64_t>(pool.size());
            int32_t lo = pool[b], lo_conj = pool[b] ^ 1;
            for (int64_t k = b; k < e; k++) { seen[pool[k] ^ 1] = 1; lo = std::min(lo, pool[k]); lo_conj = std::min(lo_conj, pool[k] ^ 1); }
            if (lo_conj < lo) {
                std::reverse(pool.begin() + b, pool.begin() + e);
                for (int64_t k = b; k < e; k++) pool[k] ^= 1;
            }
            std::rotate(pool.begin() + b, std::min_element(pool.begin() + b, pool.begin() + e), pool.begin() + e);
            int64_t worst = b;
            for (int64_t k = b + 1; k < e; k++)
                if (narc[pool[k]] > narc[pool[worst]]) worst = k;
            heads.push_back({pool[b], b, e, 1, static_cast<int32_t>((worst + 1 - b) % (e - b))});
        }
        {   // emission order = ascending first vertex; first vertices are distinct, so place instead of sorting
            // (`live` ascends, and a first vertex is a live vertex)
            for (size_t c = 0; c < heads.size(); c++) owner[heads[c].first] = static_cast<int32_t>(c);
            ordered.clear();
            ordered.reserve(heads.size());
            for (const int32_t v : live)
                if (owner[v] >= 0) { ordered.push_back(heads[owner[v]]); owner[v] = -1; }
            heads.swap(ordered);
        }
        for (size_t c = 0; c < heads.size(); c++)
            for (int64_t k = heads[c].begin; k < heads[c].end; k++) owner[pool[k]] = static_cast<int32_t>(c);
        const size_t v_at = res->verts.size(), c_at = res->kind.size();
        res->verts.resize(v_at + pool.size());
        res->off.resize(c_at + 1 + heads.size());
        res->kind.resize(c_at + heads.size()); res->iter.resize(c_at + heads.size()); res->open_at.resize(c_at + heads.size());
        int32_t *ev = res->verts.data() + v_at;
        for (size_t c = 0; c < heads.size(); c++) {
            const Head &h = heads[c];
            // copies paid = min over segments of floor(left / uses); a segment is used twice when both
            // of its orientations lie on this component
            const int32_t me = static_cast<int32_t>(c);
            int64_t pay = -1;
            for (int64_t k = h.begin; k < h.end; k++) {
                const int64_t uses = 1 + (owner[pool[k] ^ 1] == me);
                const int64_t q = left[pool[k] >> 1] / uses;
                pay = pay < 0 ? q : std::min(pay, q);
            }
            pay = std::max<int64_t>(1, pay);
            for (int64_t k = h.begin; k < h.end; k++) { int64_t &l = left[pool[k] >> 1]; l = std::max<int64_t>(0, l - pay); }
            ev = std::copy(pool.begin() + h.begin, pool.begin() + h.end, ev);
            res->off[c_at + 1 + c] = static_cast<int64_t>(ev - res->verts.data());
            res->kind[c_at + c] = h.cycle;
            res->iter[c_at + c] = t;
            res->open_at[c_at + c] = h.open;
        }
        for (const int32_t v : pool) { owner[v] = -1; }       // leave the scratch clean for the next round
        for (const int32_t v : live) seen[v] = 0;
    }
    cleanup();
#undef TRY_OR_CLEAN
    return PALACE_OK;
}

namespace {

// stable LSD radix sort of `perm` by 16-bit digits of key[perm[i]]; digits that are equal everywhere are skipped
void radix_by(const std::vector<uint64_t> &key, std::vector<uint32_t> &perm, std::vector<uint32_t> &tmp)
{
    const size_t n = perm.size();
    uint64_t all_or = 0, all_and = ~0ull;
    for (size_t i = 0; i < n; i++) { all_or |= key[i]; all_and &= key[i]; }
    const uint64_t varying = all_or ^ all_and;
    std::vector<uint32_t> hist(65536);
    for (int shift = 0; shift < 64; shift += 16) {
        if (((varying >> shift) & 0xffffu) == 0) continue;
        std::fill(hist.begin(), hist.end(), 0u);
        for (size_t i = 0; i < n; i++) hist[(key[perm[i]] >> shift) & 0xffffu]++;
        uint32_t run = 0;
        for (auto &h : hist) { const uint32_t c = h; h = run; run += c; }
        for (size_t i = 0; i < n; i++) tmp[hist[(key[perm[i]] >> shift) & 0xffffu]++] = perm[i];
        perm.swap(tmp);
    }
}

}  // namespace

extern "C" {

int palace_match_arcs_from_edges(const int32_t *cn, int32_t n_segs, const palace_graph_edge *edges, int64_t n_edges,
                                 int32_t min_count, int64_t *copies, int32_t *src, int32_t *dst, int64_t *weight,
                                 int64_t *n_arcs_out)
{
    PALACE_REQUIRE(n_segs >= 0 && n_edges >= 0 && n_arcs_out, "bad argument");
    PALACE_REQUIRE(n_segs == 0 || (cn && copies), "null segment arrays");
    PALACE_REQUIRE(n_edges == 0 || (edges && src && dst && weight), "null edge arrays");
    PALACE_REQUIRE(n_segs < (1 << 30) && n_edges < (1ll << 30), "graph too large for int32 ids");
    for (int32_t s = 0; s < n_segs; s++) copies[s] = std::max(1, cn[s]);
    const uint64_t V = 2ull * static_cast<uint64_t>(n_segs);
    // arcs and conjugates of the junctions that pass the filter (generateGraph.cpp:1056-1061); the temporaries are
    // kept per thread between calls (fresh megabyte-sized vectors cost a page fault per 4 KiB)
    static thread_local std::vector<uint64_t> pair, m_pair, m_cls, m_wkey;     // pair = u * V + v
    static thread_local std::vector<int64_t> w, m_w;
    static thread_local std::vector<uint32_t> perm, tmp;
    pair.clear(); w.clear(); m_pair.clear(); m_w.clear();
    pair.reserve(2 * n_edges); w.reserve(2 * n_edges);
    for (int64_t e = 0; e < n_edges; e++) {
        const palace_graph_edge &x = edges[e];
        const int64_t tot = static_cast<int64_t>(x.counts[0]) + x.counts[1] + x.counts[2] + x.counts[3];
        if (tot < min_count) continue;
        PALACE_REQUIRE(x.left >= 0 && x.left < n_segs && x.right >= 0 && x.right < n_segs, "edge endpoint out of range");
        const uint64_t u = 2ull * x.left + (x.oL & 1), v = 2ull * x.right + (x.oR & 1);
        pair.push_back(u * V + v); w.push_back(tot);
        if ((v ^ 1) != u) { pair.push_back((v ^ 1) * V + (u ^ 1)); w.push_back(tot); }
    }
    const size_t n = pair.size();
    perm.resize(n); tmp.resize(n);
    std::iota(perm.begin(), perm.end(), 0u);
    radix_by(pair, perm, tmp);                            // by (u, v): equal arcs become adjacent
    // merge equal arcs (weights add up), as the matching executable does when it reads JUNC lines
    m_pair.reserve(n); m_w.reserve(n);
    for (size_t i = 0; i < n; i++) {
        if (!m_pair.empty() && m_pair.back() == pair[perm[i]]) m_w.back() += w[perm[i]];
        else { m_pair.push_back(pair[perm[i]]); m_w.push_back(w[perm[i]]); }
    }
    const size_t m = m_pair.size();
    int64_t w_max = 0;
    for (int64_t x : m_w) w_max = std::max(w_max, x);
    m_cls.resize(m); m_wkey.resize(m);
    for (size_t i = 0; i < m; i++) {
        const uint64_t u = m_pair[i] / V, v = m_pair[i] % V;
        m_cls[i] = std::min(m_pair[i], (v ^ 1) * V + (u ^ 1));
        m_wkey[i] = static_cast<uint64_t>(w_max - m_w[i]);           // weight descending
    }
    // rank order = (weight desc, class asc, (u, v) asc): least significant criterion first, every pass stable
    perm.resize(m); tmp.resize(m);
    std::iota(perm.begin(), perm.end(), 0u);                          // already in (u, v) order
    radix_by(m_cls, perm, tmp);
    radix_by(m_wkey, perm, tmp);
    for (size_t i = 0; i < m; i++) {
        const uint64_t p = m_pair[perm[i]];
        src[i] = static_cast<int32_t>(p / V); dst[i] = static_cast<int32_t>(p % V); weight[i] = m_w[perm[i]];
    }
    *n_arcs_out = static_cast<int64_t>(m);
    return PALACE_OK;
}

int palace_match_decompose(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, int64_t n_arcs,
                           const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive,
                           palace_match_result **out)
{
    return palace_match_decompose_ex(ctx, n_segs, copies, n_arcs, src, dst, iterations, aggressive, 0, out);
}

int palace_match_decompose_ex(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, int64_t n_arcs,
                              const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive,
                              int32_t compact, palace_match_result **out)
{
    PALACE_REQUIRE(ctx && out && n_segs >= 0 && n_arcs >= 0 && iterations >= 1, "bad argument");
    PALACE_REQUIRE(n_segs == 0 || copies, "null copies");
    PALACE_REQUIRE(n_arcs == 0 || (src && dst), "null arc arrays");
    PALACE_REQUIRE(n_segs < (1 << 30) && n_arcs < (1ll << 31), "graph too large for int32 ids");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const int32_t V = 2 * n_segs;
    for (int64_t e = 0; e < n_arcs; e++)
        PALACE_REQUIRE(src[e] >= 0 && src[e] < V && dst[e] >= 0 && dst[e] < V, "arc endpoint out of range");
    // Segments without any arc can only ever be bare one-vertex paths of round 0 (and, when
    // `aggressive`, of the extra round).  The matching runs on the sub-graph of segments that have
    // arcs, relabelled monotonically so every "smaller vertex id" decision is unchanged; the bare
    // segments are merged back in by first-vertex order.
    if (!ctx->match_scratch) ctx->match_scratch = new palace::MatchScratch();
    palace::MatchScratch &ms = *ctx->match_scratch;
    palace::Borrowed<std::vector<int32_t>> b_new(ms.new_id), b_old(ms.old_id), b_ssrc(ms.ssrc), b_sdst(ms.sdst);
    palace::Borrowed<std::vector<int64_t>> b_sc(ms.sub_copies);
    auto &new_id = b_new.v, &old_id = b_old.v, &ssrc = b_ssrc.v, &sdst = b_sdst.v;
    auto &sub_copies = b_sc.v;
    SubResult &sub = ms.sub;
    // arc-bearing segments as a bit set with a rank per 64-bit word: new id of s = rank of its bit (one bit per
    // segment instead of an int -- the set of ~1e5 among ~1e6 segments is walked by its set bits, not by all ids)
    const size_t n_words = (static_cast<size_t>(n_segs) + 63) / 64;
    palace::Borrowed<std::vector<uint64_t>> b_bits(ms.has_arc);
    auto &has_arc = b_bits.v;
    has_arc.assign(n_words, 0);
    for (int64_t e = 0; e < n_arcs; e++) {
        const int32_t a = src[e] >> 1, b = dst[e] >> 1;
        has_arc[a >> 6] |= 1ull << (a & 63);
        has_arc[b >> 6] |= 1ull << (b & 63);
    }
    new_id.resize(n_words + 1);                            // new_id[w] = number of arc-bearing segments below 64 * w
    old_id.clear();
    for (size_t w = 0; w < n_words; w++) {
        new_id[w] = static_cast<int32_t>(old_id.size());
        for (uint64_t x = has_arc[w]; x; x &= x - 1) old_id.push_back(static_cast<int32_t>(w * 64 + __builtin_ctzll(x)));
    }
    new_id[n_words] = static_cast<int32_t>(old_id.size());
    auto is_sub = [&](int32_t sg) { return (has_arc[sg >> 6] >> (sg & 63)) & 1ull; };
    auto sub_rank = [&](int32_t sg) {                      // arc-bearing segments with id < sg (= the new id of sg if it is one)
        if (sg >= n_segs) return new_id[n_words];
        return new_id[sg >> 6] + __builtin_popcountll(has_arc[sg >> 6] & ((1ull << (sg & 63)) - 1));
    };
    const int32_t n_sub = static_cast<int32_t>(old_id.size());
    sub_copies.resize(n_sub);
    for (int32_t k = 0; k < n_sub; k++) sub_copies[k] = copies[old_id[k]];
    ssrc.resize(n_arcs); sdst.resize(n_arcs);
    for (int64_t e = 0; e < n_arcs; e++) {
        ssrc[e] = 2 * sub_rank(src[e] >> 1) + (src[e] & 1);
        sdst[e] = 2 * sub_rank(dst[e] >> 1) + (dst[e] & 1);
    }
    sub.off.assign(1, 0); sub.verts.clear(); sub.iter.clear(); sub.open_at.clear(); sub.kind.clear();
    int rc = n_sub ? decompose_core(ctx, n_sub, sub_copies.data(), n_arcs, ssrc.data(), sdst.data(), iterations, aggressive, ms, &sub)
                   : PALACE_OK;
    if (rc) return rc;
    for (int32_t &v : sub.verts) v = 2 * old_id[v >> 1] + (v & 1);
    const int64_t n_sub_comp = static_cast<int64_t>(sub.kind.size());
    const int last_round = iterations + (aggressive ? 1 : 0) - 1;
    const int64_t n_bare = static_cast<int64_t>(n_segs) - n_sub;
    if (compact) {
        // only the components that hold an arc-bearing segment are listed; the bare segments -- each a one-vertex path of
        // round 0 (and of the extra round when `aggressive`), in first-vertex order between the listed ones -- are given
        // as a bit per segment
        palace_match_result *res = new palace_match_result();
        const int64_t nv = static_cast<int64_t>(sub.verts.size());
        res->off = static_cast<int64_t *>(palace::g_result_pool.take((n_sub_comp + 1) * 8));
        res->kind = static_cast<uint8_t *>(palace::g_result_pool.take(n_sub_comp + 1));
        res->iter = static_cast<int32_t *>(palace::g_result_pool.take((n_sub_comp + 1) * 4));
        res->open_at = static_cast<int32_t *>(palace::g_result_pool.take((n_sub_comp + 1) * 4));
        res->verts = static_cast<int32_t *>(palace::g_result_pool.take((nv + 1) * 4));
        res->bare = static_cast<uint64_t *>(palace::g_result_pool.take((n_words + 1) * 8));
        if (!res->off || !res->kind || !res->iter || !res->open_at || !res->verts || !res->bare) {
            delete res;
            palace::set_error("palace_match_decompose: out of host memory");
            return PALACE_ENOMEM;
        }
        std::copy(sub.off.begin(), sub.off.end(), res->off);
        std::copy(sub.kind.begin(), sub.kind.end(), res->kind);
        std::copy(sub.iter.begin(), sub.iter.end(), res->iter);
        std::copy(sub.open_at.begin(), sub.open_at.end(), res->open_at);
        std::copy(sub.verts.begin(), sub.verts.end(), res->verts);
        for (size_t w = 0; w < n_words; w++) res->bare[w] = ~has_arc[w];
        if (n_segs & 63) res->bare[n_words - 1] &= (1ull << (n_segs & 63)) - 1;
        res->n = n_sub_comp;
        res->n_bare = n_bare;
        *out = res;
        return PALACE_OK;
    }
    const int64_t n_out = n_sub_comp + n_bare * (aggressive && last_round > 0 ? 2 : 1);
    const int64_t nv_out = static_cast<int64_t>(sub.verts.size()) + (n_out - n_sub_comp);
    palace_match_result *res = new palace_match_result();
    res->off = static_cast<int64_t *>(palace::g_result_pool.take((n_out + 1) * 8));
    res->kind = static_cast<uint8_t *>(palace::g_result_pool.take(n_out));
    res->iter = static_cast<int32_t *>(palace::g_result_pool.take(n_out * 4));
    res->open_at = static_cast<int32_t *>(palace::g_result_pool.take(n_out * 4));
    res->verts = static_cast<int32_t *>(palace::g_result_pool.take(nv_out * 4));
    if (!res->off || !res->kind || !res->iter || !res->open_at || !res->verts) {
        delete res;
        palace::set_error("palace_match_decompose: out of host memory");
        return PALACE_ENOMEM;
    }
    int64_t *r_off = res->off;
    int32_t *r_verts = res->verts, *r_iter = res->iter, *r_open = res->open_at;
    uint8_t *r_kind = res->kind;
    r_off[0] = 0;
    // Components of a round come out in first-vertex order: sub-graph components (already ordered) interleaved with
    // the bare segments s (first vertex 2s); a component goes after every bare s with 2s < its first vertex, i.e.
    // s < split(c).  The ~n_segs outputs of such a round are written by a few threads, each owning a range of
    // segment ids: its bare segments, and the components whose split point falls into the range.
    auto split = [&](int64_t c) { return (sub.verts[sub.off[c]] + 1) >> 1; };
    auto merge_range = [&](int32_t a, int32_t b, int64_t c_lo, int64_t c_hi, int64_t oc, int64_t ov, int round) {
        int32_t s = a;
        auto bare_until = [&](int32_t until) {
            for (; s < until; s++) {
                if (is_sub(s)) continue;
                r_verts[ov++] = 2 * s;
                r_kind[oc] = 0; r_iter[oc] = round; r_open[oc] = 0;
                r_off[++oc] = ov;
            }
        };
        for (int64_t c = c_lo; c < c_hi; c++) {
            bare_until(std::min<int32_t>(b, split(c)));
            const int64_t len = sub.off[c + 1] - sub.off[c];
            std::copy(sub.verts.begin() + sub.off[c], sub.verts.begin() + sub.off[c + 1], r_verts + ov);
            ov += len;
            r_kind[oc] = sub.kind[c]; r_iter[oc] = sub.iter[c]; r_open[oc] = sub.open_at[c];
            r_off[++oc] = ov;
        }
        bare_until(b);
    };
    int64_t oc = 0, ov = 0, c = 0;                        // next component / vertex slot, next sub-graph component
    for (int round = 0; round <= last_round; round++) {
        int64_t c_end = c;
        while (c_end < n_sub_comp && sub.iter[c_end] == round) c_end++;
        const bool bare_round = round == 0 || (aggressive && round == last_round);
        if (!bare_round) {                                // sub-graph components only
            merge_range(0, 0, c, c_end, oc, ov, round);
            oc += c_end - c; ov += sub.off[c_end] - sub.off[c];
            c = c_end;
            continue;
        }
        const int n_thr = n_segs >= (1 << 17) ? 4 : 1;
        std::vector<std::thread> pool;
        int64_t c_lo = c;
        for (int k = 0; k < n_thr; k++) {
            const int32_t a = static_cast<int32_t>(static_cast<int64_t>(n_segs) * k / n_thr);
            const int32_t b = static_cast<int32_t>(static_cast<int64_t>(n_segs) * (k + 1) / n_thr);
            int64_t c_hi = c_end;                         // components with split(c) < b (all that are left, in the last range)
            if (k + 1 < n_thr) {
                int64_t lo = c_lo, hi = c_end;
                while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (split(mid) < b) lo = mid + 1; else hi = mid; }
                c_hi = lo;
            }
            const int64_t n_sub_in = sub_rank(b) - sub_rank(a);
            const int64_t n_bare_in = (b - a) - n_sub_in;
            if (n_thr == 1) merge_range(a, b, c_lo, c_hi, oc, ov, round);
            else pool.emplace_back(merge_range, a, b, c_lo, c_hi, oc, ov, round);
            oc += n_bare_in + (c_hi - c_lo);
            ov += n_bare_in + (sub.off[c_hi] - sub.off[c_lo]);
            c_lo = c_hi;
        }
        for (auto &t : pool) t.join();
        c = c_end;
    }
    res->n = oc;
    *out = res;
    return PALACE_OK;
}

int64_t palace_match_result_count(const palace_match_result *r) { return r ? r->n : 0; }
const int64_t *palace_match_result_offsets(const palace_match_result *r) { return r ? r->off : nullptr; }
const int32_t *palace_match_result_verts(const palace_match_result *r) { return r ? r->verts : nullptr; }
const uint8_t *palace_match_result_kind(const palace_match_result *r) { return r ? r->kind : nullptr; }
const int32_t *palace_match_result_iter(const palace_match_result *r) { return r ? r->iter : nullptr; }
const int32_t *palace_match_result_open_at(const palace_match_result *r) { return r ? r->open_at : nullptr; }
const uint64_t *palace_match_result_bare(const palace_match_result *r) { return r ? r->bare : nullptr; }
int64_t palace_match_result_bare_count(const palace_match_result *r) { return r ? r->n_bare : 0; }
void palace_match_result_free(palace_match_result *r) { delete r; }

}  // extern "C"
