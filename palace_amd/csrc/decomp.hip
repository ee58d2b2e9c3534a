// matching on gfx950, whole decomposition on the device: rounds of {greedy matching of the conjugate graph, read the paths
// and cycles off the successor links, charge copy numbers, drop exhausted segments}.  The reference's `matching` binary is
// absent; the algorithm is this repository's own, pinned to oracle/match_oracle.cpp.
//
// Everything here is latency-bound indexing work on a small graph (10^4 .. 10^6 arcs): what counts is that the host never
// waits.  So every kernel is grid-stride with a fixed grid and reads its sizes (S, V, E, output cursors) from device memory,
// a fixed number of matching iterations is enqueued per round and the ones behind the fixed point return at once (a flag per
// iteration says whether it still took an arc), and the host reads two counters back at the very end.
//
//   matching     an arc is taken when it is the best remaining arc of both its tail's out-slot and its head's in-slot
//                (locally dominant arcs: the fixed point is the sequential greedy matching in rank order, whatever the
//                scheduling).  Rank = 128-bit key (khi, klo), lower is better: per iteration a pass of 64-bit atomicMin on
//                the stamped khi, a pass of atomicMin on klo among the arcs that tie on khi, and the commit pass.  The
//                stamp (iteration number, descending) in the top bits of khi lets newer proposals displace older ones, so
//                the slot arrays are never cleared inside a round.  The slot word IS the slot's state (round 6): 0 = closed
//                (the vertex is dead, or the slot has been given to an arc) -- below every stamped key, so no proposal
//                displaces it and no key ever equals it; an iteration therefore reads one word per end of an arc instead
//                of alive / next / prev beside it.
//   read-off     a vertex without predecessor walks its path; what no walk reaches lies on cycles, and the smallest vertex
//                of a cycle walks it.  Of a component and its conjugate twin exactly one walker -- the one with the smaller
//                first vertex -- reports (make_final_fa.py:20-34 for the conjugate rule).  Reported lengths are scanned over
//                the vertex ids, which puts the components of a round in ascending first-vertex order, and a second walk
//                writes the vertices out and charges the copy numbers.
#include "decomp.hpp"

#include <algorithm>

namespace palace {

namespace {

constexpr uint64_t kNoKey = ~0ull;
constexpr uint64_t kClosed = 0;                        // slot word of a closed slot: stamps are >= 1 in every key that is proposed (decomp_begin,
                                                       // decomp_run_checked bound the iteration count), so a stamped key is never 0
constexpr int kStampShift = 42;                        // khi < 2^42; the iteration stamp lives above
constexpr int kPackedShift = 52, kPackedStamps = 1 << (64 - kPackedShift);      // one-word keys < 2^52: 12 bits of stamp
constexpr uint64_t kLenMask = (1ull << 40) - 1;


// ---- exclusive scan over u64, count on the device -------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}

constexpr int kMaxWaves = 16;                          // workgroups of up to 1024 threads

// the three steps of an exclusive scan over `n` values by `nblk` workgroups (this one is number `blk`): chunk sums, their
// prefix (one workgroup, nblk <= its size), the chunks
__device__ void scan_partials_body(const uint64_t *__restrict__ in, int64_t n, uint64_t *__restrict__ partials, int blk, int nblk)
{
    __shared__ uint64_t part[kMaxWaves];
    const int64_t chunk = (n + nblk - 1) / nblk;
    const int64_t a = min(n, static_cast<int64_t>(blk) * chunk), e = min(n, a + chunk);
    uint64_t s = 0;
    for (int64_t i = a + threadIdx.x; i < e; i += blockDim.x) s += in[i];
    const int lane = threadIdx.x & 63, waves = static_cast<int>(blockDim.x >> 6);
    s = wave_incl_scan(s, lane);
    if (lane == 63) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < waves; w++) t += part[w];
        partials[blk] = t;
    }
    __syncthreads();
}

__device__ void scan_prefix_body(uint64_t *__restrict__ partials, int n_parts, uint64_t *__restrict__ total)
{
    __shared__ uint64_t part[kMaxWaves];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t x = static_cast<int>(threadIdx.x) < n_parts ? partials[threadIdx.x] : 0;
    const uint64_t incl = wave_incl_scan(x, lane);
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    uint64_t before = 0;
    for (int w = 0; w < wv; w++) before += part[w];
    if (static_cast<int>(threadIdx.x) < n_parts) partials[threadIdx.x] = before + incl - x;
    if (threadIdx.x == blockDim.x - 1) *total = before + incl;
    __syncthreads();
}

__device__ void scan_apply_body(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int64_t n, const uint64_t *__restrict__ partials,
                                int blk, int nblk)
{
    __shared__ uint64_t part[kMaxWaves];
    const int64_t chunk = (n + nblk - 1) / nblk;
    const int64_t a = min(n, static_cast<int64_t>(blk) * chunk), e = min(n, a + chunk);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = static_cast<int>(blockDim.x >> 6);
    uint64_t carry = partials[blk];
    for (int64_t base = a; base < e; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const uint64_t x = i < e ? in[i] : 0;
        const uint64_t incl = wave_incl_scan(x, lane);
        if (lane == 63) part[wv] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
        for (int w = 0; w < waves; w++) { if (w < wv) before += part[w]; all += part[w]; }
        if (i < e) out[i] = carry + before + incl - x;
        carry += all;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kDecompBlock) void scan_partials_kernel(const uint64_t *__restrict__ in, const int32_t *__restrict__ n_dev,
                                                                     uint64_t *__restrict__ partials)
{
    scan_partials_body(in, *n_dev, partials, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}
__global__ __launch_bounds__(kDecompBlock) void scan_prefix_kernel(uint64_t *__restrict__ partials, int n_parts, uint64_t *__restrict__ total)
{
    scan_prefix_body(partials, n_parts, total);
}
__global__ __launch_bounds__(kDecompBlock) void scan_apply_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                                  const int32_t *__restrict__ n_dev, const uint64_t *__restrict__ partials)
{
    scan_apply_body(in, out, *n_dev, partials, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}

// ---- decomposition: the phases as device functions over a span of threads (T.tid of T.n), launched as
// one kernel per phase ----
struct Span { int tid, n; };
__device__ __forceinline__ Span whole_grid() { return Span{static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x), static_cast<int>(gridDim.x * blockDim.x)}; }

__device__ void ph_init(const DecompBufs &b, Span T, int64_t comp_cap, int64_t vert_cap, int n_flags, int allow_packed)
{
    DecompState *st = b.st;
    const int V = st->V;
    for (int i = T.tid; i < V; i += T.n) { b.bo_hi[i] = kNoKey; b.bi_hi[i] = kNoKey; b.on_path[i] = 0; }
    for (int i = T.tid; i < n_flags; i += T.n) b.changed[i] = 0;
    if (T.tid == 0) {
        st->n_comp = 0; st->n_vert = 0; st->comp_cap = comp_cap; st->vert_cap = vert_cap;
        st->unsettled = 0; st->overflow = 0; st->bad = 0; st->dead = 0; st->scan_total = 0; st->alive_after = 0;
        st->packed = (allow_packed && b.kc && b.pack_bad && *b.pack_bad == 0) ? 1u : 0u;
        if (comp_cap >= 0) b.o_off[0] = 0;
    }
}

__device__ void ph_round_begin(const DecompBufs &b, Span T, int reset_left)
{
    const int V = b.st->V;
    if (!reset_left && b.st->dead) return;                     // nobody kept a copy: this round is empty (the aggressive round hands out new ones)
    if (T.tid == 0) { b.st->alive_after = 0; if (reset_left) b.st->dead = 0; }
    const int64_t E = b.st->E;
    for (int64_t e = T.tid; e < E; e += T.n) b.done[e] = 0;    // every arc is looked at afresh in the round's first iteration
    const bool packed = b.st->packed != 0;                     // (the klo slot arrays are not used then)
    for (int i = T.tid; i < V; i += T.n) {
        const int s = i >> 1;
        const int64_t l = reset_left ? 1 : b.left[s];          // the aggressive round: every segment gets one more copy
        if (reset_left && !(i & 1)) b.left[s] = 1;
        b.alive[i] = l > 0;
        b.next[i] = -1; b.prev[i] = -1;
        b.bo_hi[i] = b.bi_hi[i] = l > 0 ? kNoKey : kClosed;      // both slots of a live vertex are open
        if (!packed) { b.bo_lo[i] = kNoKey; b.bo_lo[V + i] = kNoKey; b.bi_lo[i] = kNoKey; b.bi_lo[V + i] = kNoKey; }
        b.len_a[i] = 0;
    }
}

struct IterArgs {
    uint64_t stamp;           // (descending iteration stamp) << kStampShift
    uint64_t stamp_p;         // the same for one-word keys: << kPackedShift
    int parity;               // which half of the klo slot arrays this iteration uses
    int prev_flag, flag;      // index of the previous iteration's `changed` word in this round (-1: first) / of this one's
    int unique_hi;
};

// An arc is open while both its slots are: the out slot of its tail and the in slot of its head.  A closed slot stays closed until
// the round ends (vertices do not come back to life and slots are not given up inside a round), so whoever sees an arc closed says so
// in `done`: from then on an iteration spends one sequential byte on it.  After the first iteration of a round most arcs are closed.
// Per open arc and iteration the three passes do five random accesses to the two slot words (a load and two atomicMin, one of them
// returning, in the proposal pass; two loads in the commit pass) -- until round 6 they looked up alive[u], alive[v], next[u], prev[v] in
// every pass on top of the slot words (twelve), and the counting kernels beside them pay for every one.

// pass 1: stamped khi into both slots; the klo halves of the OTHER parity (written one iteration ago, needed again in the
// next one) are reset here, where nothing writes next / prev and "open" is the same for every thread that looks
__device__ void ph_propose_hi(const DecompBufs &b, Span T, const IterArgs &a)
{
    if (b.st->dead) return;
    if (a.prev_flag >= 0 && !b.changed[a.prev_flag]) return;           // the round reached its fixed point
    const int64_t E = b.st->E;
    const int V = b.st->V;
    const int other = (a.parity ^ 1) * V;
    const bool packed = b.st->packed != 0;
    for (int64_t e = T.tid; e < E; e += T.n) {
        if (b.done[e]) continue;
        const int u = b.src[e], v = b.dst[e];
        // the head's slot is looked at BEFORE the tail's takes the proposal: an arc that is closed must leave its key nowhere (a key
        // left in the tail's slot by an arc whose head is closed could keep the slot's best open arc from being taken in this
        // iteration, and an iteration that takes no arc ends the round).  Nothing closes a slot during this pass.
        if (b.bi_hi[v] == kClosed) { b.done[e] = 1; continue; }
        const uint64_t k = packed ? (a.stamp_p | b.kc[e]) : (a.stamp | b.khi[e]);
        if (atomicMin(reinterpret_cast<unsigned long long *>(&b.bo_hi[u]), static_cast<unsigned long long>(k)) == kClosed) { b.done[e] = 1; continue; }
        atomicMin(reinterpret_cast<unsigned long long *>(&b.bi_hi[v]), static_cast<unsigned long long>(k));
        if (!a.unique_hi && !packed) { b.bo_lo[other + u] = kNoKey; b.bi_lo[other + v] = kNoKey; }
    }
}

// pass 2: among the arcs that tie on khi in a slot, the smallest klo (a closed arc's key stands in no slot)
__device__ void ph_propose_lo(const DecompBufs &b, Span T, const IterArgs &a)
{
    if (b.st->dead) return;
    if (a.prev_flag >= 0 && !b.changed[a.prev_flag]) return;
    if (b.st->packed) return;                                          // one-word keys: the first pass has decided every slot
    const int64_t E = b.st->E;
    const int mine = a.parity * b.st->V;
    for (int64_t e = T.tid; e < E; e += T.n) {
        if (b.done[e]) continue;
        const int u = b.src[e], v = b.dst[e];
        const uint64_t k = a.stamp | b.khi[e], lo = b.klo[e];
        if (b.bo_hi[u] == k) atomicMin(reinterpret_cast<unsigned long long *>(&b.bo_lo[mine + u]), static_cast<unsigned long long>(lo));
        if (b.bi_hi[v] == k) atomicMin(reinterpret_cast<unsigned long long *>(&b.bi_lo[mine + v]), static_cast<unsigned long long>(lo));
    }
}

// pass 3: an arc that is the best of both its slots is taken (slot owners are unique: keys are distinct) and closes them
__device__ void ph_commit_flag(const DecompBufs &b, Span T, const IterArgs &a, volatile unsigned int *took);
__device__ void ph_commit(const DecompBufs &b, Span T, const IterArgs &a)
{
    if (b.st->dead) return;
    if (a.prev_flag >= 0 && !b.changed[a.prev_flag]) return;
    ph_commit_flag(b, T, a, b.changed + a.flag);
}
__device__ void ph_commit_flag(const DecompBufs &b, Span T, const IterArgs &a, volatile unsigned int *took)
{
    bool any = false;
    const int64_t E = b.st->E;
    const int mine = a.parity * b.st->V;
    const bool packed = b.st->packed != 0;
    for (int64_t e = T.tid; e < E; e += T.n) {
        if (b.done[e]) continue;
        const int u = b.src[e], v = b.dst[e];
        const uint64_t k = packed ? (a.stamp_p | b.kc[e]) : (a.stamp | b.khi[e]);
        if (b.bo_hi[u] != k || b.bi_hi[v] != k) continue;      // (a slot closed a moment ago by another arc of this pass held that arc's key, not this one)
        const uint64_t lo = b.klo[e];
        if (!a.unique_hi && !packed && (b.bo_lo[mine + u] != lo || b.bi_lo[mine + v] != lo)) continue;
        b.next[u] = v; b.prev[v] = u;
        b.nhi[u] = b.khi[e]; b.nlo[u] = lo;
        b.bo_hi[u] = kClosed; b.bi_hi[v] = kClosed;
        b.done[e] = 1;                               // (taken: both its slots are closed now)
        any = true;
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) *took = 1u;         // one store per workgroup, not one per arc, to the one word
}

// open walks: every live vertex without predecessor walks to the end of its path; the walker whose first vertex is the smaller
// one of {P, conj P} reports
__device__ void ph_heads(const DecompBufs &b, Span T, int round, int last_flag)
{
    DecompState *st = b.st;
    if (st->dead) return;
    if (T.tid == 0 && last_flag >= 0 && b.changed[last_flag]) st->unsettled = 1;
    const int V = st->V;
    for (int v = T.tid; v < V; v += T.n) {
        if (!b.alive[v] || b.prev[v] >= 0) continue;
        int len = 0, last = v;
        int64_t least = b.left[v >> 1];
        for (int x = v; x >= 0; x = b.next[x]) {
            b.on_path[x] = round + 1;
            last = x;
            len++;
            least = min(least, b.left[x >> 1]);
        }
        const int twin_first = last ^ 1;                          // first vertex of the conjugate path
        if (v > twin_first) continue;                             // the twin's walker reports
        const int64_t uses = twin_first == v ? 2 : 1;             // the path is its own conjugate: every segment lies on it twice
        b.len_a[v] = (1ull << 40) | static_cast<uint64_t>(len);
        b.pay[v] = max(static_cast<int64_t>(1), least / uses);
        b.kind[v] = 0;
        b.open_at[v] = 0;
    }
}

// closed walks: a live vertex no open walk reached lies on a cycle; it walks until it meets a smaller vertex (then it is not
// the cycle's first vertex) or itself
__device__ void ph_cycles(const DecompBufs &b, Span T, int round)
{
    const int V = b.st->V;
    if (b.st->dead) return;
    for (int v = T.tid; v < V; v += T.n) {
        if (!b.alive[v] || b.on_path[v] == round + 1) continue;
        int len = 1, twin_least = v ^ 1, worst = 0;
        uint64_t whi = b.nhi[v], wlo = b.nlo[v];
        int64_t least = b.left[v >> 1];
        bool first = true;
        for (int x = b.next[v]; x != v; x = b.next[x]) {
            if (x < v) { first = false; break; }           // (also ends the walk should x ever be -1: every vertex here has a successor)
            twin_least = min(twin_least, x ^ 1);
            const uint64_t hi = b.nhi[x], lo = b.nlo[x];
            if (hi > whi || (hi == whi && lo > wlo)) { whi = hi; wlo = lo; worst = len; }
            least = min(least, b.left[x >> 1]);
            len++;
        }
        if (!first || v > twin_least) continue;                   // the conjugate cycle starts lower: its first vertex reports
        const int64_t uses = twin_least == v ? 2 : 1;
        b.len_a[v] = (1ull << 40) | static_cast<uint64_t>(len);
        b.pay[v] = max(static_cast<int64_t>(1), least / uses);
        b.kind[v] = 1;
        b.open_at[v] = (worst + 1) % len;                         // position behind the weakest arc (where -b opens the cycle)
    }
}

// second walk of the reporting vertices: the component into the output arrays, copies charged
__device__ void ph_emit(const DecompBufs &b, Span T, int round, int64_t c0, int64_t v0)
{
    DecompState *st = b.st;
    const int V = st->V;
    if (st->dead) return;
    int64_t alive = 0;
    for (int v = T.tid; v < V; v += T.n) {
        const uint64_t la = b.len_a[v];
        if (!la) continue;
        const int64_t len = static_cast<int64_t>(la & kLenMask);
        const uint64_t p = b.pos[v];
        const int64_t c = c0 + static_cast<int64_t>(p >> 40), at = v0 + static_cast<int64_t>(p & kLenMask);
        const int64_t pay = b.pay[v];
        const bool room = c < st->comp_cap && at + len <= st->vert_cap;
        if (!room) st->overflow = 1;
        else { b.o_off[c] = at; b.o_kind[c] = b.kind[v]; b.o_iter[c] = round; b.o_open[c] = b.open_at[v]; }
        int x = v;
        int64_t still = 0;                                         // vertices of this component whose segment keeps copies
        for (int64_t k = 0; k < len; k++) {
            if (room) b.o_verts[at + k] = 2 * b.orig[x >> 1] + (x & 1);
            const int64_t l = b.left[x >> 1];
            b.left[x >> 1] = max(static_cast<int64_t>(0), l - pay);
            still += l - pay > 0;
            x = b.next[x];
        }
        alive += still;
    }
    // one add per workgroup
    __shared__ unsigned long long wg_alive;
    if (threadIdx.x == 0) wg_alive = 0;
    __syncthreads();
    if (alive) atomicAdd(&wg_alive, static_cast<unsigned long long>(alive));
    __syncthreads();
    if (threadIdx.x == 0 && wg_alive) atomicAdd(reinterpret_cast<unsigned long long *>(&st->alive_after), wg_alive);
}


// one kernel per phase
__global__ void dec_init_kernel(DecompBufs b, int64_t comp_cap, int64_t vert_cap, int n_flags, int allow_packed) { ph_init(b, whole_grid(), comp_cap, vert_cap, n_flags, allow_packed); }
__global__ void dec_round_begin_kernel(DecompBufs b, int reset_left) { ph_round_begin(b, whole_grid(), reset_left); }
__global__ void dec_propose_hi_kernel(DecompBufs b, IterArgs a) { ph_propose_hi(b, whole_grid(), a); }
__global__ void dec_propose_lo_kernel(DecompBufs b, IterArgs a) { ph_propose_lo(b, whole_grid(), a); }
__global__ void dec_commit_kernel(DecompBufs b, IterArgs a) { ph_commit(b, whole_grid(), a); }
__global__ void dec_heads_kernel(DecompBufs b, int round, int last_flag) { ph_heads(b, whole_grid(), round, last_flag); }
__global__ void dec_cycles_kernel(DecompBufs b, int round) { ph_cycles(b, whole_grid(), round); }
__global__ void dec_emit_kernel(DecompBufs b, int round) { ph_emit(b, whole_grid(), round, b.st->n_comp, b.st->n_vert); }

__global__ void dec_round_end_kernel(DecompBufs b)
{
    DecompState *st = b.st;
    if (blockIdx.x || threadIdx.x || st->dead) return;
    st->dead = st->alive_after == 0;
    st->n_comp += static_cast<int64_t>(st->scan_total >> 40);
    st->n_vert += static_cast<int64_t>(st->scan_total & kLenMask);
    if (st->n_comp <= st->comp_cap) b.o_off[st->n_comp] = st->n_vert;       // (o_off has comp_cap + 1 entries)
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Carver {
    char *base;
    size_t used = 0;
    template <class T>
    T *take(size_t n)
    {
        T *p = base ? reinterpret_cast<T *>(base + used) : nullptr;
        used += up256(std::max<size_t>(1, n) * sizeof(T));
        return p;
    }
};

size_t carve_all(DecompBufs &b, char *base, int64_t s_cap, int64_t e_cap, int64_t comp_cap, int64_t vert_cap, int rounds, int iters)
{
    Carver c{base};
    const size_t S = static_cast<size_t>(std::max<int64_t>(1, s_cap)), V = 2 * S, E = static_cast<size_t>(std::max<int64_t>(1, e_cap));
    b.st = c.take<DecompState>(1);
    b.src = c.take<int32_t>(E); b.dst = c.take<int32_t>(E);
    b.khi = c.take<uint64_t>(E); b.klo = c.take<uint64_t>(E);
    b.done = c.take<uint8_t>(E);
    b.kc = c.take<uint64_t>(E);
    b.left = c.take<int64_t>(S); b.orig = c.take<int32_t>(S);
    b.next = c.take<int32_t>(V); b.prev = c.take<int32_t>(V); b.on_path = c.take<int32_t>(V); b.open_at = c.take<int32_t>(V);
    b.nhi = c.take<uint64_t>(V); b.nlo = c.take<uint64_t>(V);
    b.bo_hi = c.take<uint64_t>(V); b.bi_hi = c.take<uint64_t>(V);
    b.bo_lo = c.take<uint64_t>(2 * V); b.bi_lo = c.take<uint64_t>(2 * V);
    b.len_a = c.take<uint64_t>(V); b.pos = c.take<uint64_t>(V);
    b.pay = c.take<int64_t>(V);
    b.alive = c.take<uint8_t>(V); b.kind = c.take<uint8_t>(V);
    b.partials = c.take<uint64_t>(kDecompGrid + 1);
    b.changed = c.take<uint32_t>(static_cast<size_t>(rounds) * static_cast<size_t>(iters));
    b.o_off = c.take<int64_t>(static_cast<size_t>(comp_cap) + 1);
    b.o_verts = c.take<int32_t>(static_cast<size_t>(vert_cap));
    b.o_iter = c.take<int32_t>(static_cast<size_t>(comp_cap));
    b.o_open = c.take<int32_t>(static_cast<size_t>(comp_cap));
    b.o_kind = c.take<uint8_t>(static_cast<size_t>(comp_cap));
    return c.used;
}

}  // namespace

size_t decomp_bytes(int64_t s_cap, int64_t e_cap, int64_t comp_cap, int64_t vert_cap, int rounds, int iters)
{
    DecompBufs tmp;
    return carve_all(tmp, nullptr, s_cap, e_cap, comp_cap, vert_cap, rounds, iters);
}

void decomp_carve(DecompBufs &b, char *base, int64_t s_cap, int64_t e_cap, int64_t comp_cap, int64_t vert_cap, int rounds, int iters)
{
    carve_all(b, base, s_cap, e_cap, comp_cap, vert_cap, rounds, iters);
}

int scan_u64(palace_ctx *ctx, const uint64_t *in, uint64_t *out, const int32_t *n_dev, uint64_t *partials, uint64_t *total_dev)
{
    hipLaunchKernelGGL(scan_partials_kernel, dim3(kDecompGrid), dim3(kDecompBlock), 0, ctx->stream, in, n_dev, partials);
    hipLaunchKernelGGL(scan_prefix_kernel, dim3(1), dim3(kDecompBlock), 0, ctx->stream, partials, kDecompGrid, total_dev);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(kDecompGrid), dim3(kDecompBlock), 0, ctx->stream, in, out, n_dev, partials);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

namespace {

const dim3 kBlock(kDecompBlock);
// Grid of the arc- and vertex-sized phases.  They are chains of dependent random look-ups (done -> src / dst -> alive / next /
// prev -> the slot words): what bounds them is how many of those chains are in flight, not bytes.  With the scans' 256 workgroups
// (one wave per SIMD) the first iteration of a round -- every arc open, half a million arcs at the 1M-contig sample -- took
// 0.35 + 0.43 + 0.60 ms for its three passes, as much as the 22 iterations behind it together; with one thread per arc it is a few
// waves of latency.  Still a fixed grid (the launch sequence does not depend on a number only the device knows): workgroups
// beyond the arcs read one count and leave.  Option "decomp_grid" (palace_match_set_option).
constexpr int kWideGridDefault = 2048;
inline dim3 wide_grid(const palace_ctx *ctx) { return dim3(ctx->match_grid > 0 ? ctx->match_grid : kWideGridDefault); }

// `n` matching iterations of one round; `first` = the number (inside the round) of the first of them, `count` = iterations
// enqueued since the state was initialised (the stamp), flags[flag0 ..] their `changed` words.  The first iteration of a batch
// runs unconditionally (the caller knows the previous batch, if any, still took an arc).
void enqueue_iterations(palace_ctx *ctx, const DecompBufs &b, int first, int n, uint64_t *count, int flag0, bool unique_hi)
{
    for (int k = 0; k < n; k++) {
        IterArgs a{((1ull << 21) - 1 - *count) << kStampShift, static_cast<uint64_t>((kPackedStamps - 1 - *count) & (kPackedStamps - 1)) << kPackedShift,
                   (first + k) & 1, k ? flag0 + k - 1 : -1, flag0 + k, unique_hi ? 1 : 0};
        (*count)++;
        hipLaunchKernelGGL(dec_propose_hi_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, a);
        if (!unique_hi) hipLaunchKernelGGL(dec_propose_lo_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, a);
        hipLaunchKernelGGL(dec_commit_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, a);
    }
}

int enqueue_read_off(palace_ctx *ctx, const DecompBufs &b, int round, int last_flag)
{
    hipLaunchKernelGGL(dec_heads_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, round, last_flag);
    hipLaunchKernelGGL(dec_cycles_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, round);
    int rc = scan_u64(ctx, b.len_a, b.pos, &b.st->V, b.partials, &b.st->scan_total);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_emit_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, round);
    hipLaunchKernelGGL(dec_round_end_kernel, dim3(1), dim3(64), 0, ctx->stream, b);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

}  // namespace

int decomp_begin(palace_ctx *ctx, const DecompBufs &b, int rounds, int64_t comp_cap, int64_t vert_cap)
{
    PALACE_REQUIRE(rounds >= 1 && rounds <= kMaxRounds, "round count out of range");
    static_assert(kDecompGrid <= kDecompBlock, "scan_prefix_kernel scans the block sums with one workgroup");
    // one-word keys carry 12 bits of stamp: only when every iteration this decomposition can enqueue has a stamp of its own
    const int allow_packed = !ctx->match_two_word_keys && static_cast<int64_t>(rounds) * kMaxIters < kPackedStamps ? 1 : 0;
    hipLaunchKernelGGL(dec_init_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, comp_cap, vert_cap, rounds * kMaxIters, allow_packed);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int decomp_rounds(palace_ctx *ctx, const DecompBufs &b, int t0, int t1, int rounds, int aggressive, int iters, bool unique_hi,
                  uint64_t *count)
{
    PALACE_REQUIRE(0 <= t0 && t0 <= t1 && t1 <= rounds && rounds <= kMaxRounds && iters >= 1 && iters <= kMaxIters, "round / iteration count out of range");
    for (int t = t0; t < t1; t++) {
        hipLaunchKernelGGL(dec_round_begin_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, (aggressive && t == rounds - 1) ? 1 : 0);
        enqueue_iterations(ctx, b, 0, iters, count, t * kMaxIters, unique_hi);       // stamps descend over the whole decomposition
        int rc = enqueue_read_off(ctx, b, t, t * kMaxIters + iters - 1);
        if (rc) return rc;
    }
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

// The same decomposition with the host in the loop: batches of iterations until the round's fixed point, however many it
// takes (a chain of ascending weights needs as many iterations as it has arcs).  Only used when decomp_enqueue's fixed
// number of iterations did not suffice (`unsettled`); needs rounds * 2^21 / ... never more than 2^21 iterations in all.
int decomp_run_checked(palace_ctx *ctx, const DecompBufs &b, int rounds, int aggressive, bool unique_hi, int64_t comp_cap,
                       int64_t vert_cap, int64_t max_iterations)
{
    PALACE_REQUIRE(rounds >= 1 && rounds <= kMaxRounds, "round count out of range");
    constexpr int kBatch = 16;                                              // even: the klo parity carries over
    hipLaunchKernelGGL(dec_init_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, comp_cap, vert_cap, kBatch, 0);      // (any number of iterations: two-word keys)
    uint64_t count = 0;
    for (int t = 0; t < rounds; t++) {
        hipLaunchKernelGGL(dec_round_begin_kernel, wide_grid(ctx), kBlock, 0, ctx->stream, b, (aggressive && t == rounds - 1) ? 1 : 0);
        for (int first = 0;; first += kBatch) {
            PALACE_HIP_TRY(hipMemsetAsync(b.changed, 0, kBatch * sizeof(uint32_t), ctx->stream));
            enqueue_iterations(ctx, b, first, kBatch, &count, 0, unique_hi);
            uint32_t last = 0;
            PALACE_HIP_TRY(hipMemcpyAsync(&last, b.changed + kBatch - 1, sizeof last, hipMemcpyDeviceToHost, ctx->stream));
            PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (!last) break;
            if (static_cast<int64_t>(count) > max_iterations || count + kBatch >= (1ull << 21)) {
                set_error("matching: no fixed point after %llu iterations", static_cast<unsigned long long>(count));
                return PALACE_ESTATE;
            }
        }
        PALACE_HIP_TRY(hipMemsetAsync(b.changed, 0, kBatch * sizeof(uint32_t), ctx->stream));    // settled: heads must not flag the round
        int rc = enqueue_read_off(ctx, b, t, kBatch - 1);
        if (rc) return rc;
    }
    return PALACE_OK;
}

int decomp_group(palace_ctx *ctx, const DecompBufs &b, DecompRun &run, int rounds, int aggressive, bool unique_hi)
{
    const int first = kFirstGroupRounds;
    const int t0 = run.next_round, t1 = std::min(rounds, t0 + (t0 == 0 ? first : kRoundsPerGroup));
    for (int t = t0; t < t1; t++) {
        const int iters = ctx->match_iters > 0 ? std::min(ctx->match_iters, kMaxIters) : (t == 0 ? kFirstRoundIters : kLaterRoundIters);
        int rc = decomp_rounds(ctx, b, t, t + 1, rounds, aggressive, iters, unique_hi, &run.count);
        if (rc) return rc;
    }
    run.next_round = t1;
    return PALACE_OK;
}

int decomp_finish(palace_ctx *ctx, const DecompBufs &b, DecompRun &run, int rounds, int aggressive, bool unique_hi, int64_t comp_cap,
                  int64_t vert_cap, int64_t max_iterations, DecompState *h_state, const std::function<int()> &reset_left)
{
    for (;;) {
        PALACE_HIP_TRY(hipMemcpyAsync(h_state, b.st, sizeof *h_state, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (h_state->unsettled) {
            // a round was still taking arcs in its last enqueued iteration (long chains of ascending weights): the whole
            // decomposition once more, with the host watching every round's fixed point
            int rc = reset_left();
            if (rc) return rc;
            rc = decomp_run_checked(ctx, b, rounds, aggressive, unique_hi, comp_cap, vert_cap, max_iterations);
            if (rc) return rc;
            PALACE_HIP_TRY(hipMemcpyAsync(h_state, b.st, sizeof *h_state, hipMemcpyDeviceToHost, ctx->stream));
            PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
            return PALACE_OK;
        }
        if (run.next_round >= rounds) return PALACE_OK;
        if (h_state->alive_after == 0) {
            // nobody keeps a copy: every further round is empty -- except an aggressive last round, which hands every segment one
            if (!aggressive) return PALACE_OK;
            run.next_round = rounds - 1;
        }
        int rc = decomp_group(ctx, b, run, rounds, aggressive, unique_hi);
        if (rc) return rc;
    }
}

}  // namespace palace
