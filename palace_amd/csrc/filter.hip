// Stage 04 resident in HBM: what the pipeline does between generateGraph's numbers and `all_result` (palace:566-600) without
// the text files in between -- filter_graph.py's selection on the edges generateGraph aggregated, then `matching` on exactly
// the graph that selection leaves, both on the device, the host only formatting.
//
//   B2  share/palace/scripts/filter_graph.py:201-264 on (per-contig seed bits, edges): a JUNC line exists for an edge whose
//       counters sum to >= MIN_COUNT (generate_graph.cpp:1056-1061); pass 2 keeps junctions that touch a core seed or are
//       self loops and marks their ends (1 hop); pass 3 keeps junctions that touch a seed or a marked end; SEG lines:
//       seeds, ends of kept junctions, and -- from contigs.paths (:126-151) -- every member of a path half (or 2000 bases)
//       of whose length lies on supported contigs.
//   M1  the filtered graph's segments numbered as the file lists them (selected SEG lines in `_graph.txt` order, then the
//       path-rescued ones in that order), arcs = kept junctions + conjugates + path-backed arcs (matching -l), merged in a
//       hash table, ranked by a 128-bit key instead of a sort, decomposed by decomp.hip.
// Every kernel is grid-stride with a fixed grid and reads its counts from device memory: the call sequence is the same for
// every sample, nothing waits for the host, and the host reads results back once.
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <vector>

#include "common.hpp"
#include "decomp.hpp"

namespace palace {
namespace {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint64_t kWeightTop = 1ull << 40;           // arc weights are below this (sums of 32-bit counters)

struct FilterState {                                  // device scalars
    int32_t n_segs, S_f, n_sel, n_resc;
    int64_t n_edges, n_junc, n_kept2, n_kept3, n_arcs;
    int64_t n_static, n_dyn;                          // arcs backed by contigs.paths / junction arcs beside them
    int64_t n_path_arcs;                              // (create) distinct arcs of contigs.paths
    uint32_t bad, pack_bad;                           // pack_bad: arcs whose rank key does not fit the decomposition's one-word form (decomp.hpp)
    uint64_t scan_total;
};
enum : uint32_t { kBadPathToken = 1, kBadNoSeg = 2, kBadEdgeBound = 4, kBadArcTable = 8 };

struct ArcTable {
    uint64_t *key;
    unsigned long long *w;
    uint32_t *backed;
    uint64_t mask;
    int64_t *list, list_cap;                          // slots in insertion order
};

// contigs.paths as matching -l uses it, once per sample: the distinct arcs (and conjugates) its consecutive tokens back, as
// oriented CONTIG vertices (2 * contig + minus), in a hash table for look-up plus a list; which of them exist in a filtered
// graph is a matter of which contigs that graph keeps
struct PathArcs {
    uint64_t *key;                                    // hash: (u << 32 | v), kEmptyKey = free
    int32_t *idx;                                     //       -> position in the list
    uint64_t mask;
    int32_t *u, *v;                                   // the list
    int32_t *arc_of;                                  // per step: arc number the entry got in the filtered graph, -1 = not in it
    int64_t n, cap;
};

struct F {                                            // what the kernels see
    FilterState *fs;
    int32_t n_segs, min_count;
    const uint8_t *seed;
    const int32_t *tlen, *rank, *name_len;
    int32_t *by_rank;
    int64_t n_paths;
    const int64_t *path_off;
    const int32_t *path_tok;
    uint8_t *core, *sel, *near, *resc, *seg_flags, *edge_flags, *has_arc;
    int32_t *fid, *contig_of;
    uint64_t *scan_in, *scan_out;
    int64_t edge_bound;
    ArcTable t;                                       // junction arcs that no path backs (per step)
    PathArcs pa;
    int32_t *arc_u, *arc_v;                           // filtered-graph vertex ids of the merged arcs
    unsigned long long *arc_w;                        // their weights
    uint8_t *arc_backed;
    int64_t arc_cap;
    int use_paths;
    uint64_t *bare;
};

__device__ __forceinline__ int gtid() { return static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); }
__device__ __forceinline__ int gsize() { return static_cast<int>(gridDim.x * blockDim.x); }

__global__ void st4_by_rank_kernel(F f)
{
    for (int s = gtid(); s < f.n_segs; s += gsize()) f.by_rank[f.rank[s]] = s;
}

__global__ void st4_begin_kernel(F f, const int64_t *__restrict__ d_n_edges, int64_t call_bound)
{
    for (int s = gtid(); s < f.n_segs; s += gsize()) {
        const uint8_t c = f.seed[s] != 0 && f.tlen[s] > 0;          // a seed with a SEG line (filter_graph.py:210-218)
        f.core[s] = c; f.sel[s] = c; f.near[s] = 0; f.resc[s] = 0;
    }
    if (gtid() == 0) {
        FilterState *fs = f.fs;
        const int64_t n = *d_n_edges;
        fs->n_segs = f.n_segs; fs->S_f = 0; fs->n_sel = 0; fs->n_resc = 0;
        const int64_t bound = call_bound < f.edge_bound ? call_bound : f.edge_bound;     // this call's edge array, and the flags' room
        fs->n_edges = n < 0 ? 0 : (n > bound ? bound : n);
        fs->n_junc = 0; fs->n_kept2 = 0; fs->n_kept3 = 0; fs->n_arcs = 0; fs->n_static = 0; fs->n_dyn = 0;
        fs->bad = n > bound ? kBadEdgeBound : 0u;
        fs->pack_bad = 0;
        fs->scan_total = 0;
    }
}

__device__ __forceinline__ int64_t edge_total(const palace_graph_edge &e)
{
    return static_cast<int64_t>(e.counts[0]) + e.counts[1] + e.counts[2] + e.counts[3];
}

// pass 2 (filter_graph.py:223-233): junctions touching a core seed, and self loops; their ends are one hop away
__global__ void st4_pass2_kernel(F f, const palace_graph_edge *__restrict__ edges)
{
    const int64_t n = f.fs->n_edges;
    int64_t juncs = 0, kept = 0;
    for (int64_t i = gtid(); i < n; i += gsize()) {
        const palace_graph_edge e = edges[i];
        uint8_t fl = 0;
        if (edge_total(e) >= f.min_count) {               // the JUNC line exists (generate_graph.cpp:1061)
            fl = 1; juncs++;
            if (e.left == e.right || f.core[e.left] || f.core[e.right]) {
                fl |= 2; kept++;
                f.near[e.left] = 1; f.near[e.right] = 1;
                f.sel[e.left] = 1; f.sel[e.right] = 1;
            }
        }
        f.edge_flags[i] = fl;
    }
    if (juncs) atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_junc), static_cast<unsigned long long>(juncs));
    if (kept) atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_kept2), static_cast<unsigned long long>(kept));
}

// pass 3 (:237-245): junctions touching a seed or a one-hop end
__global__ void st4_pass3_kernel(F f, const palace_graph_edge *__restrict__ edges)
{
    const int64_t n = f.fs->n_edges;
    int64_t kept = 0;
    for (int64_t i = gtid(); i < n; i += gsize()) {
        const uint8_t fl = f.edge_flags[i];
        if (!(fl & 1)) continue;
        const palace_graph_edge e = edges[i];
        if (f.core[e.left] || f.near[e.left] || f.core[e.right] || f.near[e.right]) {
            f.edge_flags[i] = fl | 4;
            if (!(fl & 2)) kept++;
            f.sel[e.left] = 1; f.sel[e.right] = 1;
        }
    }
    if (kept) atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_kept3), static_cast<unsigned long long>(kept));
}

// contigs.paths rescue (:126-151): one thread per path line
__global__ void st4_paths_kernel(F f)
{
    for (int64_t p = gtid(); p < f.n_paths; p += gsize()) {
        const int64_t a = f.path_off[p], b = f.path_off[p + 1];
        int64_t total = 0, backed = 0;
        bool known = true;
        for (int64_t k = a; k < b; k++) {
            const int32_t t = f.path_tok[k];
            if (t < 0) { known = false; break; }
            const int32_t m = t >> 1;
            total += f.name_len[m];
            if (f.seed[m]) backed += f.name_len[m];            // support = blast | gene | score (:247), with or without SEG line
        }
        if (!known) { atomicOr(&f.fs->bad, kBadPathToken); continue; }     // the reference dies on num_to_full_name[...]
        if (backed > 0 && (static_cast<double>(backed) / static_cast<double>(total) >= 0.5 || backed > 2000))
            for (int64_t k = a; k < b; k++) f.resc[f.path_tok[k] >> 1] = 1;
    }
}

// the filtered graph's SEG order: selected lines in `_graph.txt` order (= name rank), then the rescued ones in that order
__global__ void st4_order_kernel(F f)
{
    for (int r = gtid(); r < f.n_segs; r += gsize()) {
        const int s = f.by_rank[r];
        const bool sel = f.sel[s], resc = f.resc[s] && !sel;
        if (resc && f.tlen[s] <= 0) atomicOr(&f.fs->bad, kBadNoSeg);   // all_segs[seg] of a contig without SEG line (:257)
        f.scan_in[r] = (sel ? (1ull << 32) : 0ull) | (resc ? 1ull : 0ull);
    }
}

__global__ void st4_ids_kernel(F f)
{
    FilterState *fs = f.fs;
    const int32_t n_sel = static_cast<int32_t>(fs->scan_total >> 32), n_resc = static_cast<int32_t>(fs->scan_total & 0xffffffffu);
    for (int r = gtid(); r < f.n_segs; r += gsize()) {
        const int s = f.by_rank[r];
        const uint64_t in = f.scan_in[r], before = f.scan_out[r];
        int32_t id = -1;
        if (in >> 32) id = static_cast<int32_t>(before >> 32);
        else if (in & 1) id = n_sel + static_cast<int32_t>(before & 0xffffffffu);
        f.fid[s] = id;
        if (id >= 0) { f.contig_of[id] = s; f.has_arc[id] = 0; }
        f.seg_flags[s] = static_cast<uint8_t>((in >> 32 ? 1 : 0) | (in & 1 ? 2 : 0) | (f.core[s] ? 4 : 0));
    }
    if (gtid() == 0) { fs->n_sel = n_sel; fs->n_resc = n_resc; fs->S_f = n_sel + n_resc; }
}

// ---- arcs of the filtered graph -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// create(): the distinct arcs of contigs.paths (consecutive tokens of a line; an unknown id breaks the chain) and their conjugates
__device__ void path_arc_insert(const F &f, int32_t u, int32_t v)
{
    const PathArcs &pa = f.pa;
    const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
    uint64_t s = mix(key) & pa.mask;
    for (uint64_t probes = 0; probes <= pa.mask; probes++) {
        const uint64_t cur = pa.key[s];
        if (cur == key) return;
        if (cur == kEmptyKey) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&pa.key[s]), kEmptyKey, key);
            if (old == kEmptyKey) {
                const int64_t at = static_cast<int64_t>(atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_path_arcs), 1ull));
                if (at < pa.cap) { pa.u[at] = u; pa.v[at] = v; pa.idx[s] = static_cast<int32_t>(at); }
                return;
            }
            if (old == key) return;
        }
        s = (s + 1) & pa.mask;
    }
}
__global__ void st4_path_arcs_kernel(F f)
{
    for (int64_t p = gtid(); p < f.n_paths; p += gsize()) {
        int32_t before = -1;
        for (int64_t k = f.path_off[p]; k < f.path_off[p + 1]; k++) {
            const int32_t here = f.path_tok[k];
            if (before >= 0 && here >= 0) {
                path_arc_insert(f, before, here);
                if ((here ^ 1) != before) path_arc_insert(f, here ^ 1, before ^ 1);       // the conjugate (make_final_fa.py:20-34)
            }
            before = here;
        }
    }
}
__device__ __forceinline__ int32_t path_arc_find(const PathArcs &pa, int32_t u, int32_t v)
{
    const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
    uint64_t s = mix(key) & pa.mask;
    for (;;) {
        const uint64_t cur = pa.key[s];
        if (cur == key) return pa.idx[s];
        if (cur == kEmptyKey) return -1;
        s = (s + 1) & pa.mask;
    }
}

// matching -l contigs.paths on the filtered graph: the path arcs both of whose contigs it keeps (weight 0, path-backed)
__global__ __launch_bounds__(kDecompBlock) void st4_arcs_static_kernel(F f)
{
    __shared__ uint32_t wave_n[kDecompBlock / 64];
    __shared__ long long block_base;
    const PathArcs &pa = f.pa;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t p0 = static_cast<int64_t>(blockIdx.x) * blockDim.x; p0 < pa.n; p0 += static_cast<int64_t>(gridDim.x) * blockDim.x) {   // (uniform per workgroup)
        const int64_t p = p0 + threadIdx.x;
        int32_t fu = -1, fv = -1;
        if (p < pa.n) {                                            // the byte per contig first (1 MB, cache resident), the ids for the few that pass
            const int32_t cu = pa.u[p] >> 1, cv = pa.v[p] >> 1;
            if ((f.seg_flags[cu] & 3) && (f.seg_flags[cv] & 3)) { fu = f.fid[cu]; fv = f.fid[cv]; }
        }
        const bool in = fu >= 0 && fv >= 0;
        // one returning add on the arc counter per workgroup and round (adds on one address are a serial resource of the device)
        const unsigned long long m = __ballot(in);
        if (lane == 0) wave_n[wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kDecompBlock / 64; w++) { const uint32_t c = wave_n[w]; wave_n[w] = total; total += c; }
            block_base = total ? static_cast<long long>(atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_static), static_cast<unsigned long long>(total))) : 0ll;
        }
        __syncthreads();
        if (p < pa.n) {
            int32_t at = -1;
            if (in) {
                const int64_t slot = block_base + wave_n[wave] + __popcll(m & ((1ull << lane) - 1));
                if (slot < f.arc_cap) {
                    at = static_cast<int32_t>(slot);
                    f.arc_u[at] = 2 * fu + (pa.u[p] & 1); f.arc_v[at] = 2 * fv + (pa.v[p] & 1);
                    f.arc_w[at] = 0; f.arc_backed[at] = 1;
                } else atomicOr(&f.fs->bad, kBadArcTable);
            }
            pa.arc_of[p] = at;
        }
        __syncthreads();                                           // (wave_n / block_base are rewritten in the next round)
    }
}

// a junction arc: its weight goes to the path arc between the same two oriented contigs if the graph has one, else it is an arc
// of its own (merged with equal ones in the table of this step)
__device__ void bump_arc(const F &f, int32_t cu, int32_t cv, int32_t u, int32_t v, int64_t w)
{
    if (f.use_paths) {
        const int32_t p = path_arc_find(f.pa, cu, cv);
        if (p >= 0) {
            const int32_t at = f.pa.arc_of[p];                   // (both contigs are kept, so the path arc is in the graph)
            if (at >= 0) { atomicAdd(&f.arc_w[at], static_cast<unsigned long long>(w)); return; }
        }
    }
    const ArcTable &t = f.t;
    const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
    uint64_t s = mix(key) & t.mask;
    for (uint64_t probes = 0;; probes++) {
        if (probes > t.mask) { atomicOr(&f.fs->bad, kBadArcTable); return; }
        const uint64_t cur = t.key[s];
        if (cur == key) break;
        if (cur == kEmptyKey) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&t.key[s]), kEmptyKey, key);
            if (old == kEmptyKey) {                          // this thread created the arc: it also lists it
                const int64_t at = static_cast<int64_t>(atomicAdd(reinterpret_cast<unsigned long long *>(&f.fs->n_dyn), 1ull));
                if (at < t.list_cap) t.list[at] = static_cast<int64_t>(s);
                else atomicOr(&f.fs->bad, kBadArcTable);
                break;
            }
            if (old == key) break;
        }
        s = (s + 1) & t.mask;
    }
    if (w) atomicAdd(&t.w[s], static_cast<unsigned long long>(w));
}

__global__ void st4_arcs_juncs_kernel(F f, const palace_graph_edge *__restrict__ edges)
{
    const int64_t n = f.fs->n_edges;
    for (int64_t i = gtid(); i < n; i += gsize()) {
        if (!(f.edge_flags[i] & 6)) continue;
        const palace_graph_edge e = edges[i];
        const int64_t w = min(edge_total(e), static_cast<int64_t>(kWeightTop - 1));
        const int32_t cu = 2 * e.left + (e.oL & 1), cv = 2 * e.right + (e.oR & 1);
        const int32_t u = 2 * f.fid[e.left] + (e.oL & 1), v = 2 * f.fid[e.right] + (e.oR & 1);
        bump_arc(f, cu, cv, u, v, w);
        if ((v ^ 1) != u) bump_arc(f, cv ^ 1, cu ^ 1, v ^ 1, u ^ 1, w);          // the conjugate; an arc that is its own conjugate once
    }
}

// the junction arcs of this step's table join the list behind the path arcs; the table is left clean for the next sample
__global__ void st4_arcs_dyn_kernel(F f)
{
    FilterState *fs = f.fs;
    const int64_t n_static = min(fs->n_static, f.arc_cap), n = min(fs->n_dyn, f.t.list_cap);
    for (int64_t a = gtid(); a < n; a += gsize()) {
        const int64_t s = f.t.list[a];
        const uint64_t key = f.t.key[s];
        const int64_t at = n_static + a;
        if (at < f.arc_cap) {
            f.arc_u[at] = static_cast<int32_t>(key >> 32); f.arc_v[at] = static_cast<int32_t>(key & 0xffffffffu);
            f.arc_w[at] = f.t.w[s]; f.arc_backed[at] = 0;
        } else atomicOr(&fs->bad, kBadArcTable);
        f.t.key[s] = kEmptyKey; f.t.w[s] = 0;
    }
    if (gtid() == 0) fs->n_arcs = min(n_static + n, f.arc_cap);
}

// rank keys (lower = better): weight descending, path-backed first, class {arc, conjugate} ascending, the arc that equals
// its class key first -- the order palace_amd/host/matching_main.cpp sorts by
__global__ void st4_arc_keys_kernel(F f, DecompBufs b)
{
    const int64_t n = min(f.fs->n_static, f.arc_cap) + min(f.fs->n_dyn, f.t.list_cap);
    for (int64_t a = gtid(); a < n && a < f.arc_cap; a += gsize()) {
        const int32_t u = f.arc_u[a], v = f.arc_v[a];
        const uint64_t w = min(static_cast<uint64_t>(f.arc_w[a]), kWeightTop - 1);
        b.khi[a] = ((kWeightTop - w) << 1) | (f.arc_backed[a] ? 0u : 1u);
        const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
        const uint64_t twin = (static_cast<uint64_t>(static_cast<uint32_t>(v ^ 1)) << 32) | static_cast<uint32_t>(u ^ 1);
        const uint64_t cls = min(key, twin);
        b.klo[a] = (cls << 1) | (key != cls ? 1u : 0u);
        f.has_arc[u >> 1] = 1; f.has_arc[v >> 1] = 1;
    }
}

__global__ void st4_sub_in_kernel(F f)
{
    const int n = f.fs->S_f;
    for (int i = gtid(); i < n; i += gsize()) f.scan_in[i] = f.has_arc[i];
}

// arc-bearing segments renumbered (monotone in the filtered ids, so every "smaller vertex" decision is unchanged)
__global__ void st4_sub_kernel(F f, DecompBufs b, const int32_t *__restrict__ cn)
{
    FilterState *fs = f.fs;
    const int n = fs->S_f;
    const int32_t n_sub = static_cast<int32_t>(fs->scan_total);
    for (int i = gtid(); i < n; i += gsize()) {
        if (!f.has_arc[i]) continue;
        const int32_t sub = static_cast<int32_t>(f.scan_out[i]);
        b.orig[sub] = i;
        b.left[sub] = max(1, cn[f.contig_of[i]]);
    }
    const int64_t n_arcs = min(fs->n_arcs, f.arc_cap);
    // The rank key of an arc in one word, where it fits (decomp.hpp, DecompBufs::kc): [weight term | not backed | class of (u, v)] with
    // the sub-graph's vertex ids -- the renumbering is monotone, so classes compare as they do in st4_arc_keys_kernel's two words, and
    // the bit that tells an arc from its conjugate there is not needed to rank the arcs of ONE slot (the two never share one).
    const int vbits = 32 - __clz(max(1, 2 * n_sub - 1)), wbits = min(40, 51 - 2 * vbits);
    const uint64_t wmax = wbits > 0 ? (1ull << wbits) - 1 : 0;
    if (gtid() == 0 && wbits <= 0) fs->pack_bad = 1;
    for (int64_t a = gtid(); a < n_arcs; a += gsize()) {
        const int32_t u = f.arc_u[a], v = f.arc_v[a];
        const int32_t su = 2 * static_cast<int32_t>(f.scan_out[u >> 1]) + (u & 1), sv = 2 * static_cast<int32_t>(f.scan_out[v >> 1]) + (v & 1);
        b.src[a] = su;
        b.dst[a] = sv;
        if (wbits > 0) {
            const uint64_t w = f.arc_w[a];
            if (w > wmax) atomicOr(&fs->pack_bad, 1u);
            const uint64_t key = (static_cast<uint64_t>(su) << vbits) | static_cast<uint32_t>(sv);
            const uint64_t twin = (static_cast<uint64_t>(sv ^ 1) << vbits) | static_cast<uint32_t>(su ^ 1);
            b.kc[a] = ((wmax - min(w, wmax)) << (2 * vbits + 1)) | (static_cast<uint64_t>(f.arc_backed[a] ? 0u : 1u) << (2 * vbits)) | min(key, twin);
        }
    }
    if (gtid() == 0) { b.st->S = n_sub; b.st->V = 2 * n_sub; b.st->E = n_arcs; }
}

__global__ void st4_left_kernel(F f, DecompBufs b, const int32_t *__restrict__ cn)       // (before a second run of the decomposition)
{
    const int n = f.fs->S_f;
    for (int i = gtid(); i < n; i += gsize())
        if (f.has_arc[i]) b.left[static_cast<int32_t>(f.scan_out[i])] = max(1, cn[f.contig_of[i]]);
}

__global__ void st4_bare_kernel(F f)
{
    const int n = f.fs->S_f, words = (n + 63) / 64;
    for (int w = gtid(); w < words; w += gsize()) {
        uint64_t bits = 0;
        for (int k = 0; k < 64 && w * 64 + k < n; k++)
            if (!f.has_arc[w * 64 + k]) bits |= 1ull << k;
        f.bare[w] = bits;
    }
}

size_t up(size_t v) { return (v + 255) / 256 * 256; }
uint64_t pow2_at_least(uint64_t v) { uint64_t p = 1024; while (p < v) p <<= 1; return p; }

template <class T>
T *carve(char *&p, size_t n)
{
    T *r = reinterpret_cast<T *>(p);
    p += up(std::max<size_t>(1, n) * sizeof(T));
    return r;
}

}  // namespace
}  // namespace palace

struct palace_stage04 {
    palace::F f{};
    palace::DecompBufs b{};
    char *fixed = nullptr;            // device block sized by the sample (create)
    char *grown = nullptr;            // device block sized by the edge bound (grow-only)
    size_t grown_bytes = 0;
    int64_t n_tok = 0, edge_bound = 0, s_cap = 0, e_cap = 0, comp_cap = 0, vert_cap = 0;
    int rounds = 0, aggressive = 0, last_rounds = 0;
    palace::DecompRun run;
    bool filtered = false, matched = false;
    const int32_t *d_cn = nullptr;
    // pinned host
    char *pin = nullptr;
    size_t pin_bytes = 0;
    palace::FilterState *h_fs = nullptr;      // the two state blocks come back into their own small pinned block
    palace::DecompState *h_ds = nullptr;
    palace_match_result res;
    int32_t *h_contig_of = nullptr;
};

using namespace palace;

namespace {

const dim3 kG(1024), kB(kDecompBlock);           // fixed grid of the grid-stride kernels over contigs / edges / path lines

// (The launch sequences of filter / match as hipGraphs, captured once and replayed -- option launch_graphs of rounds 3-5 -- saved the host
// 0.3 ms of enqueueing per step and never shortened the step: the kernels run back to back either way.  Removed in round 6.)
int grow_pinned(palace_ctx *ctx, palace_stage04 *s, size_t bytes)
{
    if (s->pin_bytes >= bytes) return PALACE_OK;
    if (s->pin) { PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream)); PALACE_HIP_TRY(hipHostFree(s->pin)); s->pin = nullptr; s->pin_bytes = 0; }
    const size_t want = bytes + bytes / 4 + (1 << 16);
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&s->pin), want, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("stage04: hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return PALACE_ENOMEM; }
    s->pin_bytes = want;
    return PALACE_OK;
}

// the device block that depends on the edge bound: edge flags, the arc table and list, the decomposition's arrays
int grow_device(palace_ctx *ctx, palace_stage04 *s, int64_t edge_bound, int rounds)
{
    const int64_t e_dyn = 2 * edge_bound + 16;                                 // junction arcs incl. conjugates
    const int64_t e_cap = e_dyn + s->f.pa.n + 16;                              // + the arcs contigs.paths backs
    const int64_t s_cap = std::min<int64_t>(s->f.n_segs, e_cap);               // arc-bearing segments: an arc has two ends, ends are shared
    const int64_t comp_cap = static_cast<int64_t>(rounds) * s_cap + 16, vert_cap = 2 * comp_cap;
    const uint64_t slots = pow2_at_least(2 * static_cast<uint64_t>(e_dyn));
    const size_t dec = decomp_bytes(s_cap, e_cap, comp_cap, vert_cap, rounds, kMaxIters);
    const size_t bytes = up(static_cast<size_t>(edge_bound) + 1) + up(slots * 8) * 2 + up(static_cast<size_t>(e_dyn) * 8) +
                         2 * up(static_cast<size_t>(e_cap) * 4) + up(static_cast<size_t>(e_cap) * 8) + up(static_cast<size_t>(e_cap)) + dec + 8192;
    if (s->grown_bytes < bytes || s->edge_bound < edge_bound || s->rounds < rounds) {
        if (s->grown) { PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream)); PALACE_HIP_TRY(hipFree(s->grown)); s->grown = nullptr; s->grown_bytes = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->grown), bytes);
        if (e != hipSuccess) { set_error("stage04: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return PALACE_ENOMEM; }
        s->grown_bytes = bytes;
        char *p = s->grown;
        s->f.edge_flags = carve<uint8_t>(p, static_cast<size_t>(edge_bound) + 1);
        s->f.t.key = carve<uint64_t>(p, slots);
        s->f.t.w = carve<unsigned long long>(p, slots);
        s->f.t.backed = nullptr;
        s->f.t.mask = slots - 1;
        s->f.t.list = carve<int64_t>(p, static_cast<size_t>(e_dyn));
        s->f.t.list_cap = e_dyn;
        s->f.arc_u = carve<int32_t>(p, static_cast<size_t>(e_cap));
        s->f.arc_v = carve<int32_t>(p, static_cast<size_t>(e_cap));
        s->f.arc_w = carve<unsigned long long>(p, static_cast<size_t>(e_cap));
        s->f.arc_backed = carve<uint8_t>(p, static_cast<size_t>(e_cap));
        s->f.arc_cap = e_cap;
        decomp_carve(s->b, p, s_cap, e_cap, comp_cap, vert_cap, rounds, kMaxIters);
        s->b.pack_bad = &s->f.fs->pack_bad;
        PALACE_HIP_TRY(hipMemsetAsync(s->f.t.key, 0xff, slots * 8, ctx->stream));       // empty; every use leaves it empty again
        PALACE_HIP_TRY(hipMemsetAsync(s->f.t.w, 0, slots * 8, ctx->stream));
        s->edge_bound = edge_bound; s->s_cap = s_cap; s->e_cap = e_cap; s->comp_cap = comp_cap; s->vert_cap = vert_cap; s->rounds = rounds;
    }
    s->f.edge_bound = s->edge_bound;
    return PALACE_OK;
}

}  // namespace

extern "C" {

int palace_stage04_create(palace_ctx *ctx, const palace_stage04_inputs *in, palace_stage04 **out)
{
    PALACE_REQUIRE(ctx && in && out, "null argument");
    PALACE_REQUIRE(in->n_segs >= 0 && in->n_segs < (1 << 30) && in->n_paths >= 0 && in->min_count >= 0, "bad size");
    PALACE_REQUIRE(in->n_segs == 0 || (in->seed && in->tlen && in->rank && in->name_len), "null per-contig array");
    PALACE_REQUIRE(in->n_paths == 0 || (in->path_off && in->path_tok), "null path arrays");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = static_cast<size_t>(in->n_segs);
    const int64_t n_tok = in->n_paths ? in->path_off[in->n_paths] : 0;
    PALACE_REQUIRE(n_tok >= 0 && (in->n_paths == 0 || in->path_off[0] == 0), "path offsets must start at 0 and ascend");
    int64_t n_pairs = 0;                                     // consecutive token pairs over all path lines (a line may be empty)
    for (int64_t k = 0; k < in->n_paths; k++) {
        const int64_t len = in->path_off[k + 1] - in->path_off[k];
        PALACE_REQUIRE(len >= 0 && in->path_off[k + 1] <= n_tok, "path offsets must start at 0 and ascend");
        n_pairs += len > 1 ? len - 1 : 0;
    }
    for (int32_t i = 0; i < in->n_segs; i++) PALACE_REQUIRE(in->rank[i] >= 0 && in->rank[i] < in->n_segs, "rank out of range");
    for (int64_t k = 0; k < n_tok; k++) PALACE_REQUIRE(in->path_tok[k] < 2 * static_cast<int64_t>(in->n_segs), "path token out of range");
    std::unique_ptr<palace_stage04> s(new palace_stage04());
    const int64_t pa_cap = 2 * n_pairs + 16;                                             // consecutive pairs and their conjugates
    const uint64_t pa_slots = pow2_at_least(2 * static_cast<uint64_t>(pa_cap));
    const size_t bytes = up(sizeof(FilterState)) + up(n) * 8 + up(n * 4) * 6 + up(n * 8) * 2 + up((static_cast<size_t>(in->n_paths) + 1) * 8) +
                         up(static_cast<size_t>(n_tok) * 4 + 4) + up((n + 63) / 64 * 8 + 8) + up(pa_slots * 8) + up(pa_slots * 4) +
                         3 * up(static_cast<size_t>(pa_cap) * 4) + 16384;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->fixed), bytes);
    if (e != hipSuccess) { set_error("stage04: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return PALACE_ENOMEM; }
    char *p = s->fixed;
    F &f = s->f;
    f.fs = carve<FilterState>(p, 1);
    f.n_segs = in->n_segs; f.min_count = in->min_count; f.n_paths = in->n_paths;
    uint8_t *seed = carve<uint8_t>(p, n);
    f.core = carve<uint8_t>(p, n); f.sel = carve<uint8_t>(p, n); f.near = carve<uint8_t>(p, n); f.resc = carve<uint8_t>(p, n);
    f.seg_flags = carve<uint8_t>(p, n); f.has_arc = carve<uint8_t>(p, n);
    int32_t *tlen = carve<int32_t>(p, n), *rank = carve<int32_t>(p, n), *name_len = carve<int32_t>(p, n);
    f.by_rank = carve<int32_t>(p, n); f.fid = carve<int32_t>(p, n); f.contig_of = carve<int32_t>(p, n);
    f.scan_in = carve<uint64_t>(p, n); f.scan_out = carve<uint64_t>(p, n);
    int64_t *path_off = carve<int64_t>(p, static_cast<size_t>(in->n_paths) + 1);
    int32_t *path_tok = carve<int32_t>(p, static_cast<size_t>(n_tok) + 1);
    f.bare = carve<uint64_t>(p, (n + 63) / 64 + 1);
    f.pa.key = carve<uint64_t>(p, pa_slots); f.pa.idx = carve<int32_t>(p, pa_slots); f.pa.mask = pa_slots - 1;
    f.pa.u = carve<int32_t>(p, static_cast<size_t>(pa_cap)); f.pa.v = carve<int32_t>(p, static_cast<size_t>(pa_cap));
    f.pa.arc_of = carve<int32_t>(p, static_cast<size_t>(pa_cap));
    f.pa.cap = pa_cap; f.pa.n = 0;
    f.seed = seed; f.tlen = tlen; f.rank = rank; f.name_len = name_len; f.path_off = path_off; f.path_tok = path_tok;
    s->n_tok = n_tok;
    auto fail = [&](hipError_t err) {
        set_error("stage04: upload failed: %s", hipGetErrorString(err));
        (void)hipFree(s->fixed);
        return PALACE_EHIP;
    };
    hipStream_t st = ctx->stream;
    if (n) {
        if ((e = hipMemcpyAsync(seed, in->seed, n, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(tlen, in->tlen, n * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(rank, in->rank, n * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(name_len, in->name_len, n * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    }
    if (in->n_paths) {
        if ((e = hipMemcpyAsync(path_off, in->path_off, (static_cast<size_t>(in->n_paths) + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        if (n_tok && (e = hipMemcpyAsync(path_tok, in->path_tok, static_cast<size_t>(n_tok) * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    } else if ((e = hipMemsetAsync(path_off, 0, 8, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(st4_by_rank_kernel, kG, kB, 0, st, f);
    // the arcs contigs.paths backs, once per sample (what `matching -l` derives from the file every time it runs)
    if ((e = hipMemsetAsync(f.fs, 0, sizeof(FilterState), st)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(f.pa.key, 0xff, pa_slots * 8, st)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(st4_path_arcs_kernel, kG, kB, 0, st, f);
    FilterState h_fs{};
    if ((e = hipMemcpyAsync(&h_fs, f.fs, sizeof h_fs, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);          // the caller's arrays are free again
    if (h_fs.n_path_arcs > pa_cap) { (void)hipFree(s->fixed); set_error("stage04: path arc table too small"); return PALACE_ESTATE; }
    f.pa.n = h_fs.n_path_arcs;
    // the small pinned block the two state structs come back into
    e = hipHostMalloc(reinterpret_cast<void **>(&s->h_fs), up(sizeof(FilterState)) + up(sizeof(DecompState)), hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(s->fixed); set_error("stage04: hipHostMalloc failed: %s", hipGetErrorString(e)); return PALACE_ENOMEM; }
    s->h_ds = reinterpret_cast<DecompState *>(reinterpret_cast<char *>(s->h_fs) + up(sizeof(FilterState)));
    s->res.borrowed = true;
    *out = s.release();
    return PALACE_OK;
}

int palace_stage04_destroy(palace_ctx *ctx, palace_stage04 *s)
{
    if (!s) return PALACE_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    if (s->fixed) (void)hipFree(s->fixed);
    if (s->grown) (void)hipFree(s->grown);
    if (s->pin) (void)hipHostFree(s->pin);
    if (s->h_fs) (void)hipHostFree(s->h_fs);
    delete s;
    return PALACE_OK;
}

int palace_stage04_reserve(palace_ctx *ctx, palace_stage04 *s, int64_t edge_bound)
{
    PALACE_REQUIRE(ctx && s && edge_bound >= 0, "bad argument");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    return grow_device(ctx, s, std::max<int64_t>(edge_bound, s->edge_bound), std::max(s->rounds, 11));
}

int palace_stage04_filter(palace_ctx *ctx, palace_stage04 *s, const palace_graph_edge *d_edges, const int64_t *d_n_edges,
                          int64_t edge_bound)
{
    PALACE_REQUIRE(ctx && s && d_n_edges && edge_bound >= 0, "bad argument");
    PALACE_REQUIRE(edge_bound == 0 || d_edges, "null edge array");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = grow_device(ctx, s, std::max<int64_t>(edge_bound, s->edge_bound), std::max(s->rounds, 11));
    if (rc) return rc;
    rc = [&]() -> int {
        const F &f = s->f;
        hipStream_t st = ctx->stream;
        hipLaunchKernelGGL(st4_begin_kernel, kG, kB, 0, st, f, d_n_edges, edge_bound);
        hipLaunchKernelGGL(st4_pass2_kernel, kG, kB, 0, st, f, d_edges);
        hipLaunchKernelGGL(st4_pass3_kernel, kG, kB, 0, st, f, d_edges);
        hipLaunchKernelGGL(st4_paths_kernel, kG, kB, 0, st, f);
        hipLaunchKernelGGL(st4_order_kernel, kG, kB, 0, st, f);
        PALACE_HIP_TRY(hipGetLastError());
        int rc2 = scan_u64(ctx, f.scan_in, f.scan_out, &f.fs->n_segs, s->b.partials, &f.fs->scan_total);
        if (rc2) return rc2;
        hipLaunchKernelGGL(st4_ids_kernel, kG, kB, 0, st, f);
        PALACE_HIP_TRY(hipGetLastError());
        return PALACE_OK;
    }();
    if (rc) return rc;
    s->filtered = true;
    s->matched = false;
    return PALACE_OK;
}

static int enqueue_match(palace_ctx *ctx, palace_stage04 *s, const palace_graph_edge *d_edges, bool use_paths)
{
    s->f.use_paths = use_paths ? 1 : 0;
    const F &f = s->f;
    hipStream_t st = ctx->stream;
    if (use_paths) hipLaunchKernelGGL(st4_arcs_static_kernel, kG, kB, 0, st, f);
    hipLaunchKernelGGL(st4_arcs_juncs_kernel, kG, kB, 0, st, f, d_edges);
    hipLaunchKernelGGL(st4_arcs_dyn_kernel, kG, kB, 0, st, f);
    hipLaunchKernelGGL(st4_arc_keys_kernel, kG, kB, 0, st, f, s->b);
    hipLaunchKernelGGL(st4_sub_in_kernel, kG, kB, 0, st, f);
    PALACE_HIP_TRY(hipGetLastError());
    int rc = scan_u64(ctx, f.scan_in, f.scan_out, &f.fs->S_f, s->b.partials, &f.fs->scan_total);
    if (rc) return rc;
    hipLaunchKernelGGL(st4_sub_kernel, kG, kB, 0, st, f, s->b, s->d_cn);
    hipLaunchKernelGGL(st4_bare_kernel, kG, kB, 0, st, f);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

// (palace_stage04_match_after -- the rounds held back until a mark of another context's stream, so that they would not run beside its
// counting kernels -- put them on the critical path wherever it was tried (rounds 3-5).  Removed in round 6.)
int palace_stage04_match(palace_ctx *ctx, palace_stage04 *s, const palace_graph_edge *d_edges, const int32_t *d_cn,
                         int32_t iterations, int32_t aggressive, int32_t use_paths)
{
    PALACE_REQUIRE(ctx && s && d_cn && iterations >= 1, "bad argument");
    PALACE_REQUIRE(s->filtered, "palace_stage04_filter has not run");
    const int rounds = iterations + (aggressive ? 1 : 0);
    PALACE_REQUIRE(rounds <= kMaxRounds, "too many iterations");
    PALACE_REQUIRE(rounds <= s->rounds, "more rounds than the filter call reserved room for (at most 10 iterations + aggressive)");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    s->d_cn = d_cn; s->aggressive = aggressive ? 1 : 0;
    // the first group of rounds goes out now; palace_stage04_result looks at the state after it and continues while segments
    // keep copies (a typical sample is done after the first group)
    s->f.use_paths = use_paths ? 1 : 0;
    int rc = enqueue_match(ctx, s, d_edges, use_paths != 0);
    if (rc) return rc;
    s->run = DecompRun{};
    if ((rc = decomp_begin(ctx, s->b, rounds, s->comp_cap, s->vert_cap))) return rc;
    if ((rc = decomp_group(ctx, s->b, s->run, rounds, s->aggressive, false))) return rc;
    PALACE_HIP_TRY(hipMemcpyAsync(s->h_fs, s->f.fs, sizeof(FilterState), hipMemcpyDeviceToHost, ctx->stream));
    s->rounds = std::max(s->rounds, rounds);
    s->matched = true;
    s->last_rounds = rounds;
    return PALACE_OK;
}

static int check_filter_state(const FilterState &fs)
{
    if (fs.bad & kBadPathToken) { set_error("stage04: contigs.paths names an unknown contig id (filter_graph.py:138 raises)"); return PALACE_EINVAL; }
    if (fs.bad & kBadNoSeg) { set_error("stage04: contigs.paths rescues a contig without SEG line (filter_graph.py:257 raises)"); return PALACE_EINVAL; }
    if (fs.bad & kBadEdgeBound) { set_error("stage04: more edges on the device than the bound given to palace_stage04_filter"); return PALACE_EINVAL; }
    if (fs.bad & kBadArcTable) { set_error("stage04: arc table full"); return PALACE_ESTATE; }
    return PALACE_OK;
}

int palace_stage04_counts(palace_ctx *ctx, palace_stage04 *s, int64_t counts[8])
{
    PALACE_REQUIRE(ctx && s && counts && s->filtered, "bad argument");
    PALACE_HIP_TRY(hipMemcpyAsync(s->h_fs, s->f.fs, sizeof(FilterState), hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const FilterState &fs = *s->h_fs;
    int rc = check_filter_state(fs);
    if (rc) return rc;
    counts[0] = fs.n_edges; counts[1] = fs.n_junc; counts[2] = fs.n_kept2; counts[3] = fs.n_kept3;
    counts[4] = fs.n_sel; counts[5] = fs.n_resc; counts[6] = s->matched ? fs.n_arcs : -1; counts[7] = fs.S_f;
    return PALACE_OK;
}

int palace_stage04_flags(palace_ctx *ctx, palace_stage04 *s, uint8_t *h_seg_flags, uint8_t *h_edge_flags, int64_t n_edges)
{
    PALACE_REQUIRE(ctx && s && s->filtered && n_edges >= 0 && n_edges <= s->edge_bound, "bad argument");
    if (h_seg_flags && s->f.n_segs)
        PALACE_HIP_TRY(hipMemcpyAsync(h_seg_flags, s->f.seg_flags, static_cast<size_t>(s->f.n_segs), hipMemcpyDeviceToHost, ctx->stream));
    if (h_edge_flags && n_edges)
        PALACE_HIP_TRY(hipMemcpyAsync(h_edge_flags, s->f.edge_flags, static_cast<size_t>(n_edges), hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PALACE_OK;
}

int palace_stage04_result(palace_ctx *ctx, palace_stage04 *s, palace_match_result **out, const int32_t **contig_of_out,
                          int64_t *n_segs_filtered_out)
{
    PALACE_REQUIRE(ctx && s && out, "null argument");
    PALACE_REQUIRE(s->matched, "palace_stage04_match has not run");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int rounds = s->last_rounds;
    auto reset_left = [&]() -> int {
        hipLaunchKernelGGL(st4_left_kernel, kG, kB, 0, st, s->f, s->b, s->d_cn);
        PALACE_HIP_TRY(hipGetLastError());
        return PALACE_OK;
    };
    int rc = decomp_finish(ctx, s->b, s->run, rounds, s->aggressive, false, s->comp_cap, s->vert_cap, 2 * s->e_cap + 64, s->h_ds, reset_left);
    if (rc) return rc;
    rc = check_filter_state(*s->h_fs);                          // (in since the first wait of decomp_finish)
    if (rc) return rc;
    const DecompState &ds = *s->h_ds;
    if (ds.overflow || ds.n_comp > s->comp_cap || ds.n_vert > s->vert_cap) { set_error("stage04: component arrays too small"); return PALACE_ESTATE; }
    const size_t nc = static_cast<size_t>(ds.n_comp), nv = static_cast<size_t>(ds.n_vert), sf = static_cast<size_t>(s->h_fs->S_f);
    const size_t words = (sf + 63) / 64;
    rc = grow_pinned(ctx, s, up((nc + 1) * 8) + up(nv * 4 + 4) + 2 * up(nc * 4 + 4) + up(nc + 1) + up(words * 8 + 8) + up(sf * 4 + 4));
    if (rc) return rc;
    char *p = s->pin;
    palace_match_result &r = s->res;
    r.off = carve<int64_t>(p, nc + 1); r.verts = carve<int32_t>(p, nv + 1); r.iter = carve<int32_t>(p, nc + 1);
    r.open_at = carve<int32_t>(p, nc + 1); r.kind = carve<uint8_t>(p, nc + 1); r.bare = carve<uint64_t>(p, words + 1);
    s->h_contig_of = carve<int32_t>(p, sf + 1);
    PALACE_HIP_TRY(hipMemcpyAsync(r.off, s->b.o_off, (nc + 1) * 8, hipMemcpyDeviceToHost, st));
    if (nv) PALACE_HIP_TRY(hipMemcpyAsync(r.verts, s->b.o_verts, nv * 4, hipMemcpyDeviceToHost, st));
    if (nc) {
        PALACE_HIP_TRY(hipMemcpyAsync(r.iter, s->b.o_iter, nc * 4, hipMemcpyDeviceToHost, st));
        PALACE_HIP_TRY(hipMemcpyAsync(r.open_at, s->b.o_open, nc * 4, hipMemcpyDeviceToHost, st));
        PALACE_HIP_TRY(hipMemcpyAsync(r.kind, s->b.o_kind, nc, hipMemcpyDeviceToHost, st));
    }
    if (words) PALACE_HIP_TRY(hipMemcpyAsync(r.bare, s->f.bare, words * 8, hipMemcpyDeviceToHost, st));
    if (sf) PALACE_HIP_TRY(hipMemcpyAsync(s->h_contig_of, s->f.contig_of, sf * 4, hipMemcpyDeviceToHost, st));
    PALACE_HIP_TRY(hipStreamSynchronize(st));
    r.n = static_cast<int64_t>(nc);
    r.n_bare = static_cast<int64_t>(sf) - ds.S;
    *out = &s->res;
    if (contig_of_out) *contig_of_out = s->h_contig_of;
    if (n_segs_filtered_out) *n_segs_filtered_out = static_cast<int64_t>(sf);
    return PALACE_OK;
}

}  // extern "C"
