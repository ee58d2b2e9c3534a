// generateGraph on gfx950: BAM evidence -> SEG depths + JUNC edges of the conjugate graph.
// Functional spec: bin/generate_graph.cpp of the reference (rows G2-G6 of SURVEY.md section 8).
//
// Design: the reference is one sequential pass whose only order-dependent state is
// (a) hasSupplementEvidence of the current record and (b) the processedPairedReads name set
// ("first record in file order that finds a layout wins; every later eligible record of that
// name is skipped and its reference span is added to its MATE's contig", :890-893, :938).
// Everything else is a pure function of one record.  So:
//   classify kernel   one thread per record over structure-of-arrays columns (44 B/record):
//                     filters, per-contig depth sums (wave-level run combining, records are
//                     coordinate sorted so a wave mostly sees one or two contigs), layout search,
//                     canonical edge key, FASTG membership by binary search -> compact candidates.
//   resolve kernels   over the (few %) candidates only: split evidence sets a per-record bit;
//                     pairs insert min(file ordinal) per read-name key into a hash table, then
//                     each eligible pair record compares its ordinal with that minimum; accepted
//                     evidence is counted per canonical edge in a second hash table.
// Integer adds commute, so every count is independent of scheduling; the only floating point is
// the `score > 0` gate, which is decided exactly: mapq 0 -> 0, small distances -> positive, and
// the exp() underflow zone is evaluated by the host's libm with the reference's own expression.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace palace {

constexpr uint64_t kEmpty = ~0ull;

struct GraphArgs {
    palace_bam_cols c;
    const palace_sa_item *sa;
    int32_t n_targets;
    const int32_t *tlen, *trank;
    const uint64_t *fastg;
    int64_t n_fastg;
    const uint32_t *fastg_first;       // optional: fastg_first[t] = index of the first key whose left contig is >= t (n_targets + 1 entries)
    palace_graph_params p;
    double lambda, safe_dist;
    int64_t ord_base;
    unsigned long long *consumed;
    palace_graph_cand *cands;
    int64_t cap;
    unsigned long long *n_cands;
};

enum : int { kStart = 0, kEnd = 1, kMiddle = 2 };

__device__ __forceinline__ int region_of(int pos1, int len, int max_end)       // :56-62
{
    int pref = min(max_end, len / 2), suff = max(len - max_end, len / 2);
    return pos1 <= pref ? kStart : (pos1 > suff ? kEnd : kMiddle);
}
__device__ __forceinline__ int flip_region(int r) { return r == kStart ? kEnd : (r == kEnd ? kStart : kMiddle); }
__device__ __forceinline__ int near_dist(int reg, int o_minus, int pos, int len)   // :310-318
{
    int g = o_minus ? flip_region(reg) : reg;
    return g == kStart ? max(0, pos - 1) : max(0, len - pos);
}

struct Side {
    int rev, reg, pos, len, tid, mapq, nm;
    int rank;                       // dense rank of the contig's name (trank[tid]), fetched with the length
};

// membership in the sorted key array; with the per-contig offsets the search starts inside the left contig's few links (two or
// three dependent loads instead of ~22 over the whole array: this search was most of the classify kernel's latency chain)
__device__ __forceinline__ bool fastg_has(const uint64_t *__restrict__ keys, int64_t n, const uint32_t *__restrict__ first, uint64_t k)
{
    int64_t lo = 0, hi = n;
    if (first) { const uint32_t t = static_cast<uint32_t>(k >> 33); lo = first[t]; hi = first[t + 1]; }
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        uint64_t v = keys[mid];
        if (v == k) return true;
        if (v < k) lo = mid + 1; else hi = mid;
    }
    return false;
}

// Fill the layout-dependent part of a candidate (l = left side, r = right side, orientations
// as found; 1 = '-').  generate_graph.cpp:800-872 / :940-1008 reduced to what is observable.
__device__ __forceinline__ void fill_evidence(const GraphArgs &a, const Side &l, const Side &r, int oL, int oR,
                                              palace_graph_cand &c)
{
    const int rl = l.rank, rr = r.rank;
    const bool left_is_a = rl <= rr;                                    // :802/:846 via name order
    const int eL = left_is_a ? oL : oR, eR = left_is_a ? oR : oL;       // :847-848 (orientations swap!)
    c.dL = near_dist(l.reg, eL, l.pos, l.len);
    c.dR = near_dist(r.reg, eR, r.pos, r.len);
    c.mapqL = static_cast<int16_t>(l.mapq); c.nmL = l.nm;
    c.mapqR = static_cast<int16_t>(r.mapq); c.nmR = r.nm;
    if (l.mapq == 0 || r.mapq == 0) c.cls = 0;
    else c.cls = (static_cast<double>(c.dL) + static_cast<double>(c.dR) <= a.safe_dist) ? 1 : 2;
    int lt = l.tid, rt = r.tid, kL = oL, kR = oR;
    if (!a.p.both_order && rr < rl) {                                   // :856-861
        lt = r.tid; rt = l.tid; kL = !oR; kR = !oL;
    }
    c.left = lt; c.right = rt; c.oL = static_cast<uint8_t>(kL); c.oR = static_cast<uint8_t>(kR);
    uint64_t fk = (static_cast<uint64_t>(lt) << 33) | (static_cast<uint64_t>(rt) << 2) | (oL << 1) | oR;   // :863
    c.in_fastg = fastg_has(a.fastg, a.n_fastg, a.fastg_first, fk);
}

// first[t] = index of the first key whose left contig (key >> 33) is >= t, t = 0 .. n_targets (a lower bound per thread)
__global__ void fastg_offsets_kernel(const uint64_t *__restrict__ keys, int64_t n, int32_t n_targets, uint32_t *__restrict__ first)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t > n_targets) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (static_cast<int64_t>(keys[mid] >> 33) < t) lo = mid + 1; else hi = mid;
    }
    first[t] = static_cast<uint32_t>(lo);
}

// Append the candidates of the lanes that have one (`has`; the call is made by all 64 lanes of the wave): ONE returning add on the
// candidate counter per wave, not one per candidate -- ~420 000 adds on a single address were most of this kernel's time.
__device__ __forceinline__ void emit_wave(const GraphArgs &a, bool has, const palace_graph_cand &c)
{
    const unsigned long long m = __ballot(has);
    if (!m) return;                                                  // uniform
    const int lane = threadIdx.x & 63, first = __builtin_ctzll(m);
    const unsigned long long border = __ballot(has && c.found && c.cls == 2);      // exp() underflow zone: the host decides these (rare)
    unsigned long long base = 0;
    if (lane == first) {
        base = atomicAdd(a.n_cands, static_cast<unsigned long long>(__popcll(m)));
        if (border) atomicAdd(a.n_cands + 1, static_cast<unsigned long long>(__popcll(border)));
    }
    base = __shfl(base, first);
    if (has) {
        const unsigned long long i = base + __popcll(m & ((1ull << lane) - 1));
        if (static_cast<int64_t>(i) < a.cap) a.cands[i] = c;
    }
}

// Pass 1, one thread per record, streaming: the filters, the per-contig depth sums (:654-662) and the SELECTION of the records
// that can bear evidence at all -- an SA list or a mate on another contig: ~7 % of a sample -- whose indices are appended to
// `list`, per wave.  Pass 2 runs the layout search on those only, with full waves.  (As one kernel, thread per record, 93 % of
// the waves went down the whole chain of dependent loads -- contig tables, SA item, FASTG keys, two returning appends -- for
// the four or five of their lanes that had a candidate: 0.84 ms for 6.67 M records, 5 % of the HBM rate.)
constexpr int kSelectThreads = 1024;
__global__ __launch_bounds__(kSelectThreads) void graph_depth_select_kernel(GraphArgs a, uint32_t *__restrict__ list, unsigned long long *__restrict__ n_list)
{
    __shared__ uint32_t wave_n[kSelectThreads / 64];
    __shared__ unsigned long long block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a resident grid that strides over the records (two workgroups per CU): launched one workgroup per 1024 records the kernel
    // took 0.68 ms however little it did -- 5 waves per CU in flight on average, the rest of the time went into getting waves started
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kSelectThreads;
    for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * kSelectThreads; i0 < a.c.n; i0 += stride) {     // (uniform per workgroup)
        const int64_t i = i0 + threadIdx.x;
        const bool in_range = i < a.c.n;
        int flag = 0x4, tid = -1, ref_len = 0, mapq = 0, nm = 0, sa0 = 0, sa1 = 0, mtid = -1;
        if (in_range) {                                                     // every column this pass needs, at once
            flag = a.c.flag[i]; tid = a.c.tid[i]; ref_len = a.c.ref_len[i]; mapq = a.c.mapq[i]; nm = a.c.nm[i];
            sa0 = a.c.sa_off[i]; sa1 = a.c.sa_off[i + 1]; mtid = a.c.mtid[i];
        }
        const bool live = in_range && !(flag & (0x800 | 0x100 | 0x4));      // :647-649
        if (!live) { tid = -1; ref_len = 0; }
        // ---- depth (:654-662): combine runs of equal tid inside the wave, one atomic per run -------
        {
            int add = (live && tid >= 0 && tid < a.n_targets && ref_len > 0) ? ref_len : 0;
            int key = (live && tid >= 0 && tid < a.n_targets) ? tid : -1;
            int prev = __shfl_up(key, 1);
            bool sorted = __all(lane == 0 || prev <= key);
            if (sorted) {
                long long v = add;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    long long u = __shfl_up(v, d);
                    int k2 = __shfl_up(key, d);
                    if (lane >= d && k2 == key) v += u;
                }
                int next = __shfl_down(key, 1);
                if ((lane == 63 || next != key) && key >= 0 && v > 0)
                    atomicAdd(&a.consumed[key], static_cast<unsigned long long>(v));
            } else if (add > 0) {
                atomicAdd(&a.consumed[key], static_cast<unsigned long long>(add));
            }
        }
        // a record that can bear evidence: passes the flag filter, :679, names a target -- and has an SA list or a mate elsewhere
        const bool pass = live && mapq >= a.p.min_mapq && nm <= a.p.max_nm && tid >= 0 && tid < a.n_targets;
        const bool pair = pass && a.p.enable_paired && (flag & 0x1) && !(flag & 0x8) && mtid >= 0 && mtid < a.n_targets && mtid != tid;
        const bool want = pass && (pair || sa1 > sa0);
        // one returning add on the list counter per workgroup and round, not one per wave (adds on one address are taken one at a
        // time by the memory side -- and while ~100 000 of them queued there, the counting kernels' own reservations on the other
        // stream waited too: level 1 ran 4.47 ms beside the old classify kernel, 3.6 ms beside this one)
        const unsigned long long m = __ballot(want);
        if (lane == 0) wave_n[wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kSelectThreads / 64; w++) { const uint32_t c = wave_n[w]; wave_n[w] = total; total += c; }
            block_base = total ? atomicAdd(n_list, static_cast<unsigned long long>(total)) : 0ull;
        }
        __syncthreads();
        if (want) list[block_base + wave_n[wave] + __popcll(m & ((1ull << lane) - 1))] = static_cast<uint32_t>(i);
        __syncthreads();                                                     // (wave_n / block_base are rewritten in the next round)
    }
}

// Pass 2: a lane per SELECTED record (grid-stride over the list, whole waves: candidates are appended per wave, emit_wave).
__global__ __launch_bounds__(256) void graph_classify_kernel(GraphArgs a, const uint32_t *__restrict__ list, const unsigned long long *__restrict__ n_list)
{
    const int64_t n = static_cast<int64_t>(*n_list);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) & ~63ll; k0 < n; k0 += stride) {   // k0: uniform per wave
    const int64_t k = k0 + (threadIdx.x & 63);
    const bool sel = k < n;
    const int64_t i = sel ? list[k] : 0;
    int flag = 0x4, tid = 0, ref_len = 0, mapq = 0, nm = 0, sa0 = 0, sa1 = 0, mtid = -1, pos0 = 0, mpos0 = 0, read_len = 0;
    if (sel) {                                                          // batch 1: the record
        flag = a.c.flag[i]; tid = a.c.tid[i]; ref_len = a.c.ref_len[i]; mapq = a.c.mapq[i]; nm = a.c.nm[i];
        sa0 = a.c.sa_off[i]; sa1 = a.c.sa_off[i + 1]; mtid = a.c.mtid[i]; pos0 = a.c.pos[i]; mpos0 = a.c.mpos[i]; read_len = a.c.read_len[i];
    }
    if (!sel) { sa0 = sa1 = 0; }
    const int64_t ord = a.ord_base + i;
    // (pass 1 selected the record: it is live, passes :679 and names a target)
    const bool pair = sel && a.p.enable_paired && (flag & 0x1) && !(flag & 0x8) && mtid >= 0 && mtid < a.n_targets && mtid != tid;
    Side s1{(flag & 0x10) != 0, kMiddle, pos0 + 1, 0, tid, mapq, nm, 0};
    Side sm{(flag & 0x20) != 0, kMiddle, mpos0 + 1, 0, mtid, mapq, nm, 0};  // the mate; its mapq/nm := own (:950)
    unsigned long long qkey = 0;
    int clip_s = 0, clip_e = 0;
    if (sel) {                                                           // batch 2: the contig tables (and the rare columns)
        s1.len = a.tlen[tid]; s1.rank = a.trank[tid];
        if (pair) { sm.len = a.tlen[mtid]; sm.rank = a.trank[mtid]; qkey = a.c.qkey[i]; }
        if (sa1 > sa0) { clip_s = a.c.clip_s[i]; clip_e = a.c.clip_e[i]; }
        s1.reg = region_of(s1.pos, s1.len, a.p.max_end);
        if (pair) sm.reg = region_of(sm.pos, sm.len, a.p.max_end);
    }

    // ---- split reads (:684-879): item r of every lane's SA list in step (lists are short; most lanes have none) ----------
    int st1 = 0, en1 = 0;
    if (sa1 > sa0) {                                                     // :369-380 (len == read_len)
        if (clip_s < 0) { st1 = 0; en1 = 0; }                            // empty CIGAR text (:332)
        else if (s1.rev && read_len > 0) { st1 = read_len - (read_len - clip_e) + 1; en1 = read_len - clip_s; }
        else { st1 = clip_s + 1; en1 = read_len - clip_e; }
    }
    for (int r = 0; __any(sa0 + r < sa1); r++) {
        palace_graph_cand c{};
        bool has = false;
        if (sa0 + r < sa1) {
            const palace_sa_item it = a.sa[sa0 + r];
            bool ok = it.mapq2 >= a.p.min_mapq && it.nm2 <= a.p.max_nm && it.tid2 >= 0 && it.tid2 < a.n_targets;   // :724, :731-734
            if (ok) {
                Side s2{it.rev2 != 0, 0, it.pos2, a.tlen[it.tid2], it.tid2, it.mapq2, it.nm2, a.trank[it.tid2]};
                s2.reg = region_of(s2.pos, s2.len, a.p.max_end);
                ok = s1.reg != kMiddle && s2.reg != kMiddle;                         // :742
                int st2, en2;
                if (it.clip_s2 < 0) { st2 = 0; en2 = 0; }
                else if (s2.rev && read_len > 0) { st2 = read_len - (it.len2 - it.clip_e2) + 1; en2 = read_len - it.clip_s2; }
                else { st2 = it.clip_s2 + 1; en2 = it.len2 - it.clip_e2; }
                bool first1 = false, stitch = false;                                 // :401-428, gap/overlap 150
                if (en1 <= st2 && st2 - en1 - 1 <= 150) { first1 = true; stitch = true; }
                else if (en2 <= st1 && st1 - en2 - 1 <= 150) { first1 = false; stitch = true; }
                else if (st1 <= en2 && st2 <= en1) {
                    int ov = min(en1, en2) - max(st1, st2) + 1;
                    if (ov <= 150) { first1 = st1 <= st2; stitch = true; }
                }
                ok = ok && stitch;
                if (ok) {
                    const Side &l = first1 ? s1 : s2, &rr = first1 ? s2 : s1;
                    const int oL = l.rev, oR = rr.rev;                               // both read forward (:524-527)
                    if (l.reg == (oL ? kStart : kEnd) && rr.reg == (oR ? kEnd : kStart)) {   // :531-535
                        c.ord = ord; c.kind = 0; c.found = 1; c.sa_index = r;
                        fill_evidence(a, l, rr, oL, oR, c);
                        has = true;
                    }
                }
            }
        }
        emit_wave(a, has, c);
    }
    // ---- read pairs (:887-1011); the hasSupplementEvidence gate is applied in resolve -------------
    palace_graph_cand c{};
    if (pair) {
        c.ord = ord; c.kind = 1; c.qkey = qkey; c.mtid = mtid; c.ref_len = max(0, ref_len);
        const Side &s2 = sm;
        if (s1.reg != kMiddle && s2.reg != kMiddle) {                            // :910
            for (int order = 0; order < 2 && !c.found; order++) {                // :916-934
                const Side &l = order == 0 ? s1 : s2, &r = order == 0 ? s2 : s1;
                const int oL = l.rev, oR = !r.rev;                               // left forward, right reverse
                if (l.reg != (oL ? kStart : kEnd) || r.reg != (oR ? kEnd : kStart)) continue;
                int dl = l.reg == kStart ? max(0, l.pos - 1) : max(0, l.len - l.pos);
                int dr = r.reg == kStart ? max(0, r.pos - 1) : max(0, r.len - r.pos);
                double fl = l.len > 0 ? static_cast<double>(dl) / l.len : 1.0;
                double fr = r.len > 0 ? static_cast<double>(dr) / r.len : 1.0;
                if (fl > a.p.max_span_frac || fr > a.p.max_span_frac) continue;  // :503
                c.found = 1;
                fill_evidence(a, l, r, oL, oR, c);
            }
        }
    }
    emit_wave(a, pair, c);
    }
}

// ---- resolve -----------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t slot_for(uint64_t *__restrict__ keys, uint64_t mask, uint64_t key)
{
    uint64_t s = mix64(key) & mask;
    for (;;) {
        uint64_t cur = keys[s];
        if (cur == key) return s;
        if (cur == kEmpty) {
            uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[s]), kEmpty, key);
            if (old == kEmpty || old == key) return s;
        }
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ int64_t slot_find(const uint64_t *__restrict__ keys, uint64_t mask, uint64_t key)
{
    uint64_t s = mix64(key) & mask;
    for (;;) {
        uint64_t cur = keys[s];
        if (cur == key) return static_cast<int64_t>(s);
        if (cur == kEmpty) return -1;
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ uint64_t edge_key(const palace_graph_cand &c)
{
    return (static_cast<uint64_t>(static_cast<uint32_t>(c.left)) << 33) |
           (static_cast<uint64_t>(static_cast<uint32_t>(c.right)) << 2) | (c.oL << 1) | c.oR;
}

struct ResolveArgs {
    palace_graph_cand *cands;
    int64_t n;
    uint32_t *supp_bits;            // per record ordinal
    uint64_t *q_keys, *q_min, q_mask;
    uint64_t *e_keys, e_mask;
    uint32_t *e_counts;             // 4 per slot
    unsigned long long *consumed;
    palace_graph_edge *edges;
    int64_t edge_cap;
    unsigned long long *counters;   // [0] edges, [1] border
};

__device__ __forceinline__ void count_edge(const ResolveArgs &a, const palace_graph_cand &c)
{
    uint64_t s = slot_for(a.e_keys, a.e_mask, edge_key(c));
    atomicAdd(&a.e_counts[4 * s + 2 * c.kind + (c.in_fastg ? 0 : 1)], 1u);      // :866-872, :1002-1008
}

// pass 1: accepted split evidence marks its record (:874, :881-883) and is counted
__global__ void resolve_split_kernel(ResolveArgs a)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const palace_graph_cand c = a.cands[i];
    if (c.kind != 0 || c.cls != 1) return;
    atomicOr(&a.supp_bits[c.ord >> 5], 1u << (c.ord & 31));
    count_edge(a, c);
}
// pass 2: min file ordinal per read name among pair records that found a layout (:938)
__global__ void resolve_pair_insert_kernel(ResolveArgs a)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const palace_graph_cand c = a.cands[i];
    if (c.kind != 1 || !c.found) return;
    if ((a.supp_bits[c.ord >> 5] >> (c.ord & 31)) & 1) return;                   // :887 !hasSupplementEvidence
    uint64_t s = slot_for(a.q_keys, a.q_mask, c.qkey == kEmpty ? kEmpty - 1 : c.qkey);
    atomicMin(reinterpret_cast<unsigned long long *>(&a.q_min[s]), static_cast<unsigned long long>(c.ord));
}
// pass 3: later records of a processed name add their span to the MATE's contig (:890-893);
// the first one is scored and counted (:990-1008)
__global__ void resolve_pair_apply_kernel(ResolveArgs a)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const palace_graph_cand c = a.cands[i];
    if (c.kind != 1) return;
    if ((a.supp_bits[c.ord >> 5] >> (c.ord & 31)) & 1) return;
    int64_t s = slot_find(a.q_keys, a.q_mask, c.qkey == kEmpty ? kEmpty - 1 : c.qkey);
    if (s >= 0 && a.q_min[s] < static_cast<uint64_t>(c.ord)) {
        if (c.ref_len > 0) atomicAdd(&a.consumed[c.mtid], static_cast<unsigned long long>(c.ref_len));
        return;
    }
    if (c.found && c.cls == 1) count_edge(a, c);
}
constexpr int kCompactThreads = 1024;
__global__ __launch_bounds__(kCompactThreads) void compact_edges_kernel(ResolveArgs a)
{
    __shared__ uint32_t wave_n[kCompactThreads / 64];
    __shared__ unsigned long long block_base;
    uint64_t s = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool have = s <= a.e_mask && a.e_keys[s] != kEmpty;
    const unsigned long long m = __ballot(have);                  // one counter add per WORKGROUP (adds on one address are taken one at a time)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_n[wave] = static_cast<uint32_t>(__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kCompactThreads / 64; w++) { const uint32_t c = wave_n[w]; wave_n[w] = total; total += c; }
        block_base = total ? atomicAdd(&a.counters[0], static_cast<unsigned long long>(total)) : 0ull;
    }
    __syncthreads();
    if (!have) return;
    const unsigned long long i = block_base + wave_n[wave] + __popcll(m & ((1ull << lane) - 1));
    if (static_cast<int64_t>(i) >= a.edge_cap) return;
    uint64_t k = a.e_keys[s];
    palace_graph_edge e{};
    e.left = static_cast<int32_t>(k >> 33);
    e.right = static_cast<int32_t>((k >> 2) & 0x7fffffffu);
    e.oL = (k >> 1) & 1; e.oR = k & 1;
    for (int j = 0; j < 4; j++) e.counts[j] = a.e_counts[4 * s + j];
    a.edges[i] = e;
}

__global__ void copy_number_kernel(const unsigned long long *__restrict__ consumed, const int32_t *__restrict__ tlen,
                                   int32_t n, double avg, int32_t *__restrict__ cn)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double depth = static_cast<double>(consumed[i]) / static_cast<double>(max(1, tlen[i]));   // :1029
    double cnf = avg > 0.0 ? depth / avg : 0.0;                                                // :1030
    cn[i] = static_cast<int32_t>(floor(cnf + 0.5));                                            // :1031
}

static uint64_t pow2_at_least(uint64_t v) { uint64_t p = 64; while (p < v) p <<= 1; return p; }
static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// computeLayoutScore (:432-461) with the host's libm, for candidates in the exp() underflow zone
static bool host_score_positive(const palace_graph_cand &c, const palace_graph_params &p)
{
    double lambda = std::max(50.0, static_cast<double>(p.max_end) / 2.0);
    double w1 = std::exp(-static_cast<double>(c.dL) / lambda), w2 = std::exp(-static_cast<double>(c.dR) / lambda);
    double w_end = w1 * w2;
    double qL = std::min(1.0, static_cast<double>(c.mapqL) / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, c.nmL)));
    double qR = std::min(1.0, static_cast<double>(c.mapqR) / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, c.nmR)));
    volatile double score = w_end * qL * qR;
    return score > 0.0;
}

}  // namespace palace

using namespace palace;

extern "C" {

int palace_graph_classify(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                          int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                          const uint64_t *d_fastg, int64_t n_fastg, const palace_graph_params *prm,
                          int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                          int64_t cand_cap, int64_t *n_cands_out)
{
    return palace_graph_classify_ex(ctx, cols, d_sa, n_targets, d_tlen, d_trank, d_fastg, n_fastg, prm, ord_base, d_consumed,
                                    d_cands, cand_cap, n_cands_out, nullptr);
}

int palace_graph_fastg_offsets(palace_ctx *ctx, const uint64_t *d_fastg, int64_t n_fastg, int32_t n_targets, uint32_t *d_first)
{
    PALACE_REQUIRE(ctx && d_first && n_fastg >= 0 && n_fastg < (1ll << 32) && n_targets >= 0, "bad argument");
    PALACE_REQUIRE(n_fastg == 0 || d_fastg, "null FASTG key array");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const unsigned blocks = static_cast<unsigned>((static_cast<int64_t>(n_targets) + 1 + 255) / 256);
    hipLaunchKernelGGL(fastg_offsets_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_fastg, n_fastg, n_targets, d_first);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_graph_classify_ex(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                             int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                             const uint64_t *d_fastg, int64_t n_fastg, const palace_graph_params *prm,
                             int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                             int64_t cand_cap, int64_t *n_cands_out, int64_t *n_border_out)
{
    return palace_graph_classify_ix(ctx, cols, d_sa, n_targets, d_tlen, d_trank, d_fastg, n_fastg, nullptr, prm, ord_base, d_consumed,
                                    d_cands, cand_cap, n_cands_out, n_border_out);
}

int palace_graph_classify_ix(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                             int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                             const uint64_t *d_fastg, int64_t n_fastg, const uint32_t *d_fastg_first, const palace_graph_params *prm,
                             int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                             int64_t cand_cap, int64_t *n_cands_out, int64_t *n_border_out)
{
    PALACE_REQUIRE(ctx && cols && prm && n_cands_out, "null argument");
    PALACE_REQUIRE(cols->n >= 0 && n_targets >= 0 && n_fastg >= 0 && cand_cap >= 0, "negative size");
    *n_cands_out = 0;
    if (n_border_out) *n_border_out = 0;
    if (cols->n == 0) return PALACE_OK;
    PALACE_REQUIRE(cols->tid && cols->pos && cols->mtid && cols->mpos && cols->nm && cols->ref_len &&
                       cols->read_len && cols->clip_s && cols->clip_e && cols->flag && cols->mapq &&
                       cols->qkey && cols->sa_off && d_tlen && d_trank && d_consumed && d_cands,
                   "null device pointer");
    PALACE_REQUIRE(n_fastg == 0 || d_fastg, "null FASTG key array");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_pinned(ctx, 256);
    if (rc) return rc;
    GraphArgs a{};
    a.c = *cols; a.sa = d_sa; a.n_targets = n_targets; a.tlen = d_tlen; a.trank = d_trank;
    a.fastg = d_fastg; a.n_fastg = n_fastg; a.fastg_first = n_fastg > 0 ? d_fastg_first : nullptr; a.p = *prm; a.ord_base = ord_base;
    a.lambda = std::max(50.0, static_cast<double>(prm->max_end) / 2.0);
    a.safe_dist = 600.0 * a.lambda;                 // exp(-600) ~ 2.6e-261: far above underflow
    a.consumed = reinterpret_cast<unsigned long long *>(d_consumed);
    a.cands = d_cands; a.cap = cand_cap;
    a.n_cands = reinterpret_cast<unsigned long long *>(ctx->d_small);
    PALACE_HIP_TRY(hipMemsetAsync(ctx->d_small, 0, 24, ctx->stream));
    const int64_t blocks = (cols->n + 255) / 256, sel_blocks = (cols->n + kSelectThreads - 1) / kSelectThreads;
    PALACE_REQUIRE(blocks < (1ll << 31) && cols->n < (1ll << 32), "too many records for one launch");
    // pass 1 over every record (depth, selection), pass 2 over the selected ones (their indices: 4 B per record of workspace)
    rc = ensure_workspace(ctx, static_cast<size_t>(cols->n) * 4 + 256);
    if (rc) return rc;
    uint32_t *list = static_cast<uint32_t *>(ctx->ws.ptr);
    unsigned long long *n_list = reinterpret_cast<unsigned long long *>(ctx->d_small) + 2;
    hipLaunchKernelGGL(graph_depth_select_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(sel_blocks, 2 * kCUs))), dim3(kSelectThreads), 0, ctx->stream, a, list, n_list);
    hipLaunchKernelGGL(graph_classify_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(blocks, kCUs * 8))), dim3(256), 0, ctx->stream, a, list,
                       static_cast<const unsigned long long *>(n_list));
    PALACE_HIP_TRY(hipGetLastError());
    // both counters come back in one copy into pinned memory, behind one wait (no second round trip in resolve)
    unsigned long long *n = static_cast<unsigned long long *>(ctx->pin.ptr);
    PALACE_HIP_TRY(hipMemcpyAsync(n, ctx->d_small, 16, hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *n_cands_out = static_cast<int64_t>(n[0]);
    if (n_border_out) *n_border_out = static_cast<int64_t>(n[1]);
    ctx->graph_border = static_cast<int64_t>(n[1]);
    ctx->graph_border_cands = d_cands;
    ctx->graph_border_n = static_cast<int64_t>(n[0]);
    if (static_cast<int64_t>(n[0]) > cand_cap) {
        set_error("palace_graph_classify: %llu candidates exceed capacity %lld", n[0], (long long)cand_cap);
        return PALACE_EINVAL;
    }
    return PALACE_OK;
}

int palace_graph_copy_numbers(palace_ctx *ctx, const uint64_t *d_consumed, const int32_t *d_tlen,
                              int32_t n_targets, double avg_depth, int32_t *d_cn)
{
    PALACE_REQUIRE(ctx && n_targets >= 0, "bad argument");
    if (n_targets == 0) return PALACE_OK;
    PALACE_REQUIRE(d_consumed && d_tlen && d_cn, "null device pointer");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(copy_number_kernel, dim3((n_targets + 255) / 256), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long *>(d_consumed), d_tlen, n_targets, avg_depth, d_cn);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_graph_resolve(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands,
                         int64_t n_records_total, const palace_graph_params *prm, uint64_t *d_consumed,
                         palace_graph_edge *d_edges, int64_t edge_cap, int64_t *n_edges_out)
{
    PALACE_REQUIRE(n_edges_out, "null argument");
    // Exactly the candidates the last classify call left (same buffer AND same count) carry its border count, once; anything
    // else -- other ranks' candidates gathered in place behind them, a buffer classified into twice -- is looked at (-1).
    // Callers that know the count pass it to palace_graph_resolve_ex themselves.
    int64_t n_border = -1;
    if (ctx && d_cands == ctx->graph_border_cands && n_cands == ctx->graph_border_n) n_border = ctx->graph_border;
    if (ctx) { ctx->graph_border_cands = nullptr; ctx->graph_border_n = -1; }
    return palace_graph_resolve_ex(ctx, d_cands, n_cands, n_border, n_records_total, prm, d_consumed, d_edges, edge_cap, nullptr,
                                   n_edges_out);
}

int palace_graph_score_border(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands, int64_t n_border,
                              const palace_graph_params *prm)
{
    PALACE_REQUIRE(ctx && prm && n_cands >= 0, "bad argument");
    if (n_cands == 0 || n_border == 0) return PALACE_OK;
    PALACE_REQUIRE(d_cands, "null device pointer");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    std::vector<palace_graph_cand> h(static_cast<size_t>(n_cands));
    PALACE_HIP_TRY(hipMemcpyAsync(h.data(), d_cands, h.size() * sizeof(palace_graph_cand), hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    bool any = false;
    for (auto &c : h)
        if (c.found && c.cls == 2) { c.cls = host_score_positive(c, *prm) ? 1 : 0; any = true; }
    if (any) {
        PALACE_HIP_TRY(hipMemcpyAsync(d_cands, h.data(), h.size() * sizeof(palace_graph_cand), hipMemcpyHostToDevice, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));     // h leaves scope
    }
    return PALACE_OK;
}

int palace_graph_resolve_ex(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands, int64_t n_border,
                            int64_t n_records_total, const palace_graph_params *prm, uint64_t *d_consumed,
                            palace_graph_edge *d_edges, int64_t edge_cap, int64_t *d_n_edges, int64_t *n_edges_out)
{
    PALACE_REQUIRE(ctx && prm, "null argument");
    PALACE_REQUIRE(n_cands >= 0 && n_records_total >= 0 && edge_cap >= 0, "negative size");
    if (n_edges_out) *n_edges_out = 0;
    if (n_cands == 0) {
        if (d_n_edges) PALACE_HIP_TRY(hipMemsetAsync(d_n_edges, 0, 8, ctx->stream));
        return PALACE_OK;
    }
    PALACE_REQUIRE(d_cands && d_consumed && (d_edges || edge_cap == 0), "null device pointer");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t qcap = pow2_at_least(2 * static_cast<uint64_t>(n_cands));
    const size_t bits_bytes = up256((static_cast<size_t>(n_records_total) + 31) / 32 * 4 + 4);
    const size_t total = bits_bytes + up256(qcap * 8) * 3 + up256(qcap * 16);
    int rc = ensure_workspace(ctx, total);
    if (rc) return rc;
    char *ws = static_cast<char *>(ctx->ws.ptr);
    ResolveArgs a{};
    a.cands = d_cands; a.n = n_cands;
    a.supp_bits = reinterpret_cast<uint32_t *>(ws); ws += bits_bytes;
    a.q_keys = reinterpret_cast<uint64_t *>(ws); ws += up256(qcap * 8);
    a.q_min = reinterpret_cast<uint64_t *>(ws); ws += up256(qcap * 8);
    a.e_keys = reinterpret_cast<uint64_t *>(ws); ws += up256(qcap * 8);
    a.e_counts = reinterpret_cast<uint32_t *>(ws);
    a.q_mask = a.e_mask = qcap - 1;
    a.consumed = reinterpret_cast<unsigned long long *>(d_consumed);
    a.edges = d_edges; a.edge_cap = edge_cap;
    a.counters = reinterpret_cast<unsigned long long *>(ctx->d_small);
    PALACE_HIP_TRY(hipMemsetAsync(ctx->d_small, 0, 16, ctx->stream));
    PALACE_HIP_TRY(hipMemsetAsync(a.supp_bits, 0, bits_bytes, ctx->stream));
    PALACE_HIP_TRY(hipMemsetAsync(a.q_keys, 0xff, up256(qcap * 8) * 3, ctx->stream));   // keys, mins, edge keys
    PALACE_HIP_TRY(hipMemsetAsync(a.e_counts, 0, up256(qcap * 16), ctx->stream));
    const unsigned blocks = static_cast<unsigned>((n_cands + 255) / 256);
    if (n_border != 0) {    // exp() underflow zone: decide with the host's libm (:432-461).  n_border < 0: not known, look
        rc = palace_graph_score_border(ctx, d_cands, n_cands, n_border, prm);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(resolve_split_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(resolve_pair_insert_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(resolve_pair_apply_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(compact_edges_kernel, dim3(static_cast<unsigned>((qcap + kCompactThreads - 1) / kCompactThreads)), dim3(kCompactThreads), 0,
                       ctx->stream, a);
    PALACE_HIP_TRY(hipGetLastError());
    if (d_n_edges) PALACE_HIP_TRY(hipMemcpyAsync(d_n_edges, ctx->d_small, 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (!n_edges_out) return PALACE_OK;              // the count stays on the device: nothing to wait for
    rc = ensure_pinned(ctx, 256);
    if (rc) return rc;
    unsigned long long *cnt = static_cast<unsigned long long *>(ctx->pin.ptr);
    PALACE_HIP_TRY(hipMemcpyAsync(cnt, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *n_edges_out = static_cast<int64_t>(cnt[0]);
    if (static_cast<int64_t>(cnt[0]) > edge_cap) {
        set_error("palace_graph_resolve: %llu edges exceed capacity %lld", cnt[0], (long long)edge_cap);
        return PALACE_EINVAL;
    }
    return PALACE_OK;
}

}  // extern "C"

static_assert(sizeof(palace_graph_cand) == 64, "candidate layout is part of the ABI");
static_assert(sizeof(palace_graph_edge) == 32, "edge layout is part of the ABI");
static_assert(sizeof(palace_sa_item) == 32, "SA item layout is part of the ABI");
static_assert(sizeof(palace_graph_params) == 32, "params layout is part of the ABI");
