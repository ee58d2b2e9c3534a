// matching on gfx950: greedy maximum-weight matching of the conjugate graph by locally dominant
// arcs.  The reference implementation of `matching` is not available; this is the
// repository's own algorithm (DESIGN.md "matching"), pinned to oracle/match_oracle.cpp.
//
// This file: palace_match_greedy (one matching, CSR lists, arc ids = ranks), the host glue that ranks arcs
// (palace_match_arcs_from_edges) and palace_match_decompose[_ex], which hands the arc-bearing sub-graph to the
// device-resident decomposition of decomp.hip and expands its result.
//
// palace_match_greedy: a vertex owns two slots, "out" (its successor) and "in" (its predecessor).  Per round every
// free out-slot proposes the lowest-id arc whose head's in-slot is free, every free in-slot the lowest-id arc whose
// tail's out-slot is free; an arc proposed from both sides is taken.  The globally best remaining arc is always
// taken, and any arc taken this way is one the sequential greedy pass would also take, so the fixed point is the
// greedy matching -- independent of scheduling.
#include "common.hpp"
#include "decomp.hpp"
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

// ---- whole decomposition: the arc-bearing sub-graph on the device, bare segments merged in on the host ----
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

// Host temporaries of palace_match_decompose, kept in the context between calls (fresh megabyte-sized vectors cost a page
// fault per 4 KiB).
struct MatchScratch {
    std::vector<int32_t> new_id, old_id, ssrc, sdst;
    std::vector<int64_t> sub_copies;
    std::vector<uint64_t> has_arc;
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

__global__ void iota_keys_kernel(uint64_t *__restrict__ khi, uint64_t *__restrict__ klo, int64_t n)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        khi[i] = static_cast<uint64_t>(i);                       // arcs arrive in rank order: the rank is the key
        klo[i] = 0;
    }
}

}  // namespace palace

palace_match_result::~palace_match_result()
{
    if (borrowed) return;
    for (void *p : {(void *)off, (void *)verts, (void *)iter, (void *)open_at, (void *)kind, (void *)bare}) palace::g_result_pool.give(p);
}

// The decomposition of the arc-bearing sub-graph, all of it on the device (decomp.hip): arcs, copy numbers and the segment
// ids of the caller's graph go up, the component arrays come back -- one synchronisation in the middle of nothing: the host
// enqueues everything, then waits once for the counters and once for the arrays.
static int decompose_core(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, const int32_t *orig, int64_t n_arcs,
                          const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive, SubResult *res)
{
    using namespace palace;
    const int64_t S = n_segs, E = n_arcs;
    const int rounds = iterations + (aggressive ? 1 : 0);
    PALACE_REQUIRE(rounds <= kMaxRounds, "too many iterations");
    // output room: a segment is alive in at most min(iterations, copies) rounds (it pays at least one copy per round) plus the
    // aggressive one; a round lists a live segment once, twice on a component that is its own conjugate
    int64_t live_rounds = 0;
    for (int64_t s = 0; s < S; s++) live_rounds += std::min<int64_t>(iterations, std::max<int64_t>(1, copies[s]));
    if (aggressive) live_rounds += S;
    const int64_t comp_cap = live_rounds, vert_cap = 2 * live_rounds;
    const size_t dev_bytes = decomp_bytes(S, E, comp_cap, vert_cap, rounds, kMaxIters);
    int rc = ensure_workspace(ctx, dev_bytes);
    if (rc) return rc;
    DecompBufs b;
    decomp_carve(b, static_cast<char *>(ctx->ws.ptr), S, E, comp_cap, vert_cap, rounds, kMaxIters);
    // staging in pinned memory so that every upload is a true asynchronous copy
    const size_t e4 = (static_cast<size_t>(E) * 4 + 255) / 256 * 256, s8 = (static_cast<size_t>(S) * 8 + 255) / 256 * 256,
                 s4 = (static_cast<size_t>(S) * 4 + 255) / 256 * 256;
    rc = ensure_pinned(ctx, 2 * e4 + s8 + s4 + 512);
    if (rc) return rc;
    char *pin = static_cast<char *>(ctx->pin.ptr);
    int32_t *p_src = reinterpret_cast<int32_t *>(pin), *p_dst = reinterpret_cast<int32_t *>(pin + e4);
    int64_t *p_left = reinterpret_cast<int64_t *>(pin + 2 * e4);
    int32_t *p_orig = reinterpret_cast<int32_t *>(pin + 2 * e4 + s8);
    DecompState *p_st = reinterpret_cast<DecompState *>(pin + 2 * e4 + s8 + s4);
    DecompState *p_back = p_st + 1;
    std::copy(src, src + E, p_src); std::copy(dst, dst + E, p_dst);
    for (int64_t s = 0; s < S; s++) p_left[s] = std::max<int64_t>(1, copies[s]);
    std::copy(orig, orig + S, p_orig);
    std::memset(p_st, 0, sizeof *p_st);
    p_st->S = n_segs; p_st->V = 2 * n_segs; p_st->E = E;
    hipStream_t st = ctx->stream;
    auto upload = [&]() -> int {
        PALACE_HIP_TRY(hipMemcpyAsync(b.st, p_st, sizeof *p_st, hipMemcpyHostToDevice, st));
        if (E) {
            PALACE_HIP_TRY(hipMemcpyAsync(b.src, p_src, static_cast<size_t>(E) * 4, hipMemcpyHostToDevice, st));
            PALACE_HIP_TRY(hipMemcpyAsync(b.dst, p_dst, static_cast<size_t>(E) * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(iota_keys_kernel, dim3(kDecompGrid), dim3(kDecompBlock), 0, st, b.khi, b.klo, E);
        }
        PALACE_HIP_TRY(hipMemcpyAsync(b.left, p_left, static_cast<size_t>(S) * 8, hipMemcpyHostToDevice, st));
        PALACE_HIP_TRY(hipMemcpyAsync(b.orig, p_orig, static_cast<size_t>(S) * 4, hipMemcpyHostToDevice, st));
        return PALACE_OK;
    };
    auto fail = [&](int code) { (void)hipStreamSynchronize(st); return code; };      // the pinned staging must outlive the copies
    hipError_t e = hipSuccess;
    if ((rc = upload())) return fail(rc);
    if ((rc = decomp_begin(ctx, b, rounds, comp_cap, vert_cap))) return fail(rc);
    DecompRun run;
    if ((rc = decomp_group(ctx, b, run, rounds, aggressive, true))) return fail(rc);
    auto reset_left = [&]() -> int {                          // (the checked run starts over: copy numbers as uploaded)
        PALACE_HIP_TRY(hipMemcpyAsync(b.left, p_left, static_cast<size_t>(S) * 8, hipMemcpyHostToDevice, st));
        return PALACE_OK;
    };
    if ((rc = decomp_finish(ctx, b, run, rounds, aggressive, true, comp_cap, vert_cap, 2 * E + 64, p_back, reset_left))) return fail(rc);
    if (p_back->overflow || p_back->n_comp > comp_cap || p_back->n_vert > vert_cap) {
        set_error("decompose: component arrays too small (%lld components, %lld vertices)", (long long)p_back->n_comp, (long long)p_back->n_vert);
        return PALACE_ESTATE;
    }
    const size_t nc = static_cast<size_t>(p_back->n_comp), nv = static_cast<size_t>(p_back->n_vert);
    res->off.resize(nc + 1); res->verts.resize(nv); res->iter.resize(nc); res->open_at.resize(nc); res->kind.resize(nc);
    e = hipMemcpyAsync(res->off.data(), b.o_off, (nc + 1) * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && nv) e = hipMemcpyAsync(res->verts.data(), b.o_verts, nv * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(res->iter.data(), b.o_iter, nc * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(res->open_at.data(), b.o_open, nc * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(res->kind.data(), b.o_kind, nc, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("decompose: %s", hipGetErrorString(e)); return fail(PALACE_EHIP); }
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

int palace_match_set_option(palace_ctx *ctx, const char *name, int64_t value)
{
    PALACE_REQUIRE(ctx && name, "null argument");
    if (!std::strcmp(name, "iters_per_round")) {
        PALACE_REQUIRE(value >= 0 && value <= palace::kMaxIters, "iters_per_round out of range");
        ctx->match_iters = static_cast<int>(value);
    } else if (!std::strcmp(name, "one_word_keys")) {
        PALACE_REQUIRE(value == 0 || value == 1, "one_word_keys must be 0 or 1");
        ctx->match_two_word_keys = value == 0;
    } else if (!std::strcmp(name, "decomp_grid")) {
        PALACE_REQUIRE(value >= 0 && value <= 65536, "decomp_grid out of range");
        ctx->match_grid = static_cast<int>(value);
    } else {
        palace::set_error("palace_match_set_option: unknown option '%s'", name);
        return PALACE_EINVAL;
    }
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
    int rc = n_sub ? decompose_core(ctx, n_sub, sub_copies.data(), old_id.data(), n_arcs, ssrc.data(), sdst.data(), iterations, aggressive, &sub)
                   : PALACE_OK;                            // (vertices come back as ids of the caller's graph)
    if (rc) return rc;
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
void palace_match_result_free(palace_match_result *r) { if (r && !r->borrowed) delete r; }

}  // extern "C"
