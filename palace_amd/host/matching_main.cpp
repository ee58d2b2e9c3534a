// matching -- stands in for the reference's `matching` executable (absent from the reference tree,
// SURVEY.md F1) with the command line the pipeline uses (palace:587-590, 684-688, 734-739):
//     matching -g <graph> -r <linear.txt> -c <cycle.txt> [-s] -i <iterations> [-b] [-l contigs.paths] [--aggressive]
// Input grammar: SEG name depth cn gene score blast [order] / JUNC L oL R oR n1 n2
// (filter_graph.py:197,258; create_sub_graph.py:77,89).  Output grammar, as every consumer reads it
// (filter_result.py:125-134, make_fa_from_path.py:94-137, remove_cycle_dup.py:9-13): the linear
// file holds one path per line, tokens `<seg><+|->` separated by tabs, no marker lines; the cycle
// file holds two-line records, a marker line (`iter <n>` or `self`) followed by the cycle's tokens.
// The decomposition itself is this repository's algorithm (DESIGN.md "matching"); the greedy
// matching of every iteration runs on the GPU (palace_match_greedy), the rest is bookkeeping.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <unistd.h>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/palace_hip.h"
#include "fastx.hpp"
#include "textio.hpp"
#include "trace.hpp"
#include "fast_exit.hpp"
#include "device_pick.hpp"

namespace {

using palace_host::sv;

#define CK(call)                                                                       \
    do {                                                                               \
        int rc__ = (call);                                                             \
        if (rc__ != 0) {                                                               \
            std::cerr << "matching: " #call " failed: " << palace_last_error() << "\n"; \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

struct Options {
    std::string graph, linear, cycle, paths, batch;
    int iterations = 10;
    bool self_loops = false, break_cycles = false, aggressive = false;
};

bool parse_args(int argc, char **argv, Options &o)
{
    auto need = [&](int &i) -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        const char *v = nullptr;
        auto take = [&](std::string &dst) { v = need(i); if (v) dst = v; return v != nullptr; };
        if (a == "-g" || a == "--graph") { if (!take(o.graph)) return false; }
        else if (a == "-r" || a == "--result") { if (!take(o.linear)) return false; }
        else if (a == "-c" || a == "--result_c") { if (!take(o.cycle)) return false; }
        else if (a == "-l" || a == "--local_order") { if (!take(o.paths)) return false; }
        else if (a == "-i" || a == "--iteration") { v = need(i); if (!v) return false; o.iterations = std::max(1, std::atoi(v)); }
        else if (a == "-s" || a == "--self_l") o.self_loops = true;
        else if (a == "-b" || a == "--break_c") o.break_cycles = true;
        else if (a == "--aggressive") o.aggressive = true;
        else if (a == "--batch") { if (!take(o.batch)) return false; }
        else if (a == "-h" || a == "--help") return false;
        else { std::cerr << "matching: unknown option " << a << "\n"; return false; }
    }
    return !o.batch.empty() || (!o.graph.empty() && !o.linear.empty() && !o.cycle.empty());
}

struct Arc { int32_t u, v; int64_t w; int32_t backed; uint64_t cls; };

struct ConjGraph {
    std::vector<int64_t> copies;
    palace_host::Names seg_of, seg_of_id;              // keys are views of the mapped graph text (kept by the job); seg_of.names[s] = segment s's name
    std::vector<int32_t> id_seg;                       // seg_of_id's dense id -> segment (a later SEG with the same id wins)
    // (u << 32 | v) -> index in arcs: a flat open-addressing table (a node-based map took ~0.15 s for the 1.3 M arc look-ups of a
    // 1M-contig sample's contigs.paths)
    std::vector<uint64_t> arc_key;
    std::vector<uint32_t> arc_at;                      // index + 1; 0 = empty slot
    size_t arc_mask = 0;
    void arc_grow()
    {
        const size_t cap = arc_key.empty() ? (1u << 16) : 2 * arc_key.size();
        std::vector<uint64_t> k(cap);
        std::vector<uint32_t> a(cap, 0);
        for (size_t i = 0; i < arc_key.size(); i++)
            if (arc_at[i]) {
                size_t at = static_cast<size_t>((arc_key[i] * 0x9E3779B97F4A7C15ull) >> 17) & (cap - 1);
                while (a[at]) at = (at + 1) & (cap - 1);
                k[at] = arc_key[i]; a[at] = arc_at[i];
            }
        arc_key.swap(k); arc_at.swap(a); arc_mask = cap - 1;
    }
    std::vector<Arc> arcs;

    static sv id_token(sv n)                           // EDGE_<id>_...: the token between the first two '_' ("" without a '_')
    {
        const size_t a = n.find('_');
        if (a == sv::npos) return sv();
        const size_t b = n.find('_', a + 1);
        return n.substr(a + 1, b == sv::npos ? sv::npos : b - a - 1);
    }
    int32_t seg(sv n, uint64_t h)                      // h = hash_bytes(n)
    {
        const int before = static_cast<int>(seg_of.names.size());
        const int32_t s = seg_of.intern_hashed(n, h);
        if (s < before) return s;
        copies.push_back(1);
        if (n.find('_') != sv::npos) {
            const int t = seg_of_id.intern(id_token(n));
            if (static_cast<size_t>(t) >= id_seg.size()) id_seg.resize(static_cast<size_t>(t) + 1);
            id_seg[static_cast<size_t>(t)] = s;
        }
        return s;
    }
    int32_t seg_by_id(sv id) const
    {
        const int t = seg_of_id.find(id);
        return t < 0 ? -1 : id_seg[static_cast<size_t>(t)];
    }
    void bump(int32_t u, int32_t v, int64_t w, int32_t backed)
    {
        uint64_t k = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
        if (2 * (arcs.size() + 1) > arc_key.size()) arc_grow();
        size_t at = static_cast<size_t>((k * 0x9E3779B97F4A7C15ull) >> 17) & arc_mask;
        while (arc_at[at] && arc_key[at] != k) at = (at + 1) & arc_mask;
        if (!arc_at[at]) { arc_key[at] = k; arc_at[at] = static_cast<uint32_t>(arcs.size() + 1); arcs.push_back({u, v, w, backed, 0}); }
        else { Arc &a = arcs[arc_at[at] - 1]; a.w += w; a.backed |= backed; }
    }
    void add(int32_t u, int32_t v, int64_t w, int32_t backed)          // the arc and its conjugate (make_final_fa.py:20-34)
    {
        bump(u, v, w, backed);
        if (!((v ^ 1) == u && (u ^ 1) == v)) bump(v ^ 1, u ^ 1, w, backed);
    }
};

std::vector<std::string> words(const std::string &line)
{
    std::vector<std::string> t;
    std::istringstream ss(line);
    for (std::string x; ss >> x;) t.push_back(x);
    return t;
}

// atof / atol of a token of a mapped file (no terminator there): the longest numeric prefix, 0 without one
double num_prefix(sv tok)
{
    char buf[64];
    const size_t n = std::min(tok.size(), sizeof buf - 1);
    std::memcpy(buf, tok.data(), n);
    buf[n] = 0;
    return std::atof(buf);
}
long int_prefix(sv tok)
{
    char buf[64];
    const size_t n = std::min(tok.size(), sizeof buf - 1);
    std::memcpy(buf, tok.data(), n);
    buf[n] = 0;
    return std::atol(buf);
}

// contigs.paths (SPAdes): lines of comma-separated `<contig id><+|->` tokens (NODE header lines skipped); consecutive
// tokens back the arc between them.  Read once, straight from the mapped file; a batch run applies to each graph only
// the lines that mention one of its contigs.
struct PathTok { sv id; bool minus; bool ok; };

template <class F>
void for_each_path_line(const char *data, size_t size, F f)           // f(tokens of one line)
{
    std::vector<PathTok> pl;
    palace_host::for_each_line(data, size, [&](sv line) {
        if (!line.empty() && line.back() == '\n') line.remove_suffix(1);
        if (line.substr(0, 4) == "NODE") return;
        pl.clear();
        size_t p = 0;
        while (p <= line.size()) {
            const size_t c = line.find(',', p);
            sv tok = line.substr(p, c == sv::npos ? sv::npos : c - p);
            p = c == sv::npos ? line.size() + 1 : c + 1;
            while (!tok.empty() && (tok.back() == ';' || tok.back() == '\r' || tok.back() == ' ')) tok.remove_suffix(1);
            PathTok t{sv(), false, false};
            if (tok.size() >= 2 && (tok.back() == '+' || tok.back() == '-')) { t.id = tok.substr(0, tok.size() - 1); t.minus = tok.back() == '-'; t.ok = true; }
            pl.push_back(t);
        }
        f(pl);
    });
}

// the arcs a path line backs in graph g (consecutive tokens the graph knows both of): f(tail, head); reads g only
template <class F>
void backed_by(const ConjGraph &g, const std::vector<PathTok> &line, F f)
{
    int32_t before = -1;
    for (const PathTok &t : line) {
        int32_t here = -1;
        if (t.ok) {
            const int32_t s = g.seg_by_id(t.id);
            if (s >= 0) here = 2 * s + (t.minus ? 1 : 0);
        }
        if (before >= 0 && here >= 0) f(before, here);
        before = here;
    }
}
void apply_path(ConjGraph &g, const std::vector<PathTok> &line)
{
    backed_by(g, line, [&](int32_t u, int32_t v) { g.add(u, v, 0, 1); });
}

size_t host_threads(size_t bytes)
{
    if (const char *e = std::getenv("PALACE_HOST_THREADS")) return static_cast<size_t>(std::max(1, std::atoi(e)));
    return bytes < (1u << 20) ? 1 : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
}

// The lines of the graph text are split, hashed and their numbers read in parts on threads; segments and arcs are made from
// them in file order by the caller's thread (a segment's number is the order of its first mention, SEG or JUNC).
void load_graph_text(ConjGraph &g, const palace_host::MappedText &txt)
{
    struct Rec { sv a, b; uint64_t ha, hb; int64_t w; uint8_t junc, minus_a, minus_b; };
    const std::vector<size_t> cut = palace_host::line_cuts(txt.data, txt.size, host_threads(txt.size));
    std::vector<std::vector<Rec>> recs(cut.size() - 1);
    palace_host::for_parts(cut, [&](size_t k, size_t lo, size_t hi) {
        std::vector<sv> t;
        palace_host::for_each_line(txt.data + lo, hi - lo, [&](sv line) {
            palace_host::split_ws(line, t);
            if (t.size() >= 4 && t[0] == "SEG")
                recs[k].push_back(Rec{t[1], sv(), palace_host::hash_bytes(t[1]), 0, std::max<int64_t>(1, static_cast<int64_t>(num_prefix(t[3]))), 0, 0, 0});
            else if (t.size() >= 7 && t[0] == "JUNC")
                recs[k].push_back(Rec{t[1], t[3], palace_host::hash_bytes(t[1]), palace_host::hash_bytes(t[3]), int_prefix(t[5]) + int_prefix(t[6]),
                                      1, static_cast<uint8_t>(t[2] == "-"), static_cast<uint8_t>(t[4] == "-")});
        });
    });
    size_t n = 0;
    for (const auto &part : recs) n += part.size();
    g.seg_of.reserve(n + 16);
    g.seg_of_id.reserve(n + 16);
    g.copies.reserve(n + 16);
    for (const auto &part : recs)
        for (size_t i = 0; i < part.size(); i++) {
            const Rec &r = part[i];
            if (i + 8 < part.size()) { g.seg_of.prefetch(part[i + 8].ha); if (part[i + 8].junc) g.seg_of.prefetch(part[i + 8].hb); }
            if (!r.junc) { g.copies[static_cast<size_t>(g.seg(r.a, r.ha))] = r.w; continue; }
            const int32_t a = g.seg(r.a, r.ha), b = g.seg(r.b, r.hb);
            g.add(2 * a + r.minus_a, 2 * b + r.minus_b, r.w, 0);
        }
}

}  // namespace

int main(int argc, char **argv)
{
    Options opt;
    if (!parse_args(argc, argv, opt)) {
        std::cerr << "Usage: matching -g <graph> -r <linear out> -c <cycle out> [-s] [-i <iterations>] [-b] "
                     "[-l <contigs.paths>] [--aggressive]\n"
                     "       matching --batch <list of '<graph> <linear out> <cycle out>' lines> [-s] [-i <iterations>] [-b] [-l ...] [--aggressive]\n";
        return 1;
    }
    palace_host::FastExit fast_exit = palace_host::fast_exit_begin();   // from here on this is the worker process (fast_exit.hpp)
    const int device = palace_host::pick_device();                       // PALACE_DEVICE (device_pick.hpp): before anything touches HIP
    palace_host::Trace tr("matching");
    palace_ctx *ctx = nullptr;                                // the HIP runtime comes up while the graph text is read
    int ctx_rc = 0;
    std::string ctx_err;
    std::thread hip_up([&] {
        ctx_rc = palace_ctx_create(device, &ctx);
        if (ctx_rc) ctx_err = palace_last_error();
    });
    // one job per graph: the plain command line is a batch of one
    struct Job { std::string graph, linear, cycle; std::unique_ptr<palace_host::MappedText> text; ConjGraph g; int32_t v0 = 0; std::string lin, cyc, selfs; std::unordered_set<sv> lin_seen, cyc_seen; };   // (the sets hold views of the lines' part texts)
    std::vector<Job> jobs;
    std::unique_ptr<palace_host::MappedText> paths_text;
    try {
        if (opt.batch.empty()) { jobs.emplace_back(); jobs[0].graph = opt.graph; jobs[0].linear = opt.linear; jobs[0].cycle = opt.cycle; }
        else {
            // --batch <list>: one "<graph> <linear out> <cycle out>" triple per line (white space separated); every graph is
            // decomposed with the options of this command line, all of them in ONE run on the GPU -- the step-5 loop of the
            // driver (palace:651-806) starts a process per *.second sub-graph, hundreds per sample
            std::ifstream lf(opt.batch);
            if (!lf) throw std::runtime_error("cannot open batch list " + opt.batch);
            for (std::string line; std::getline(lf, line);) {
                auto t = words(line);
                if (t.empty()) continue;
                if (t.size() != 3) throw std::runtime_error("batch list: expected '<graph> <linear> <cycle>' per line");
                jobs.emplace_back(); jobs.back().graph = t[0]; jobs.back().linear = t[1]; jobs.back().cycle = t[2];
            }
        }
        for (Job &j : jobs) {
            try { j.text = std::make_unique<palace_host::MappedText>(j.graph); }
            catch (const std::exception &) { throw std::runtime_error("cannot open graph " + j.graph); }
            load_graph_text(j.g, *j.text);
        }
        tr.lap("graph text read");
        if (!opt.paths.empty()) {
            try { paths_text = std::make_unique<palace_host::MappedText>(opt.paths); }
            catch (const std::exception &) { paths_text.reset(); }                  // (an unreadable paths file backs nothing, as before)
        }
        if (paths_text && jobs.size() == 1) {
            // contigs.paths of a 1M-contig assembly is ~100 MB of which a filtered graph knows a few per cent of the ids: parts
            // of the file on threads (tokens, look-ups in the graph: read-only), the arcs they back applied in file order here
            const std::vector<size_t> cut = palace_host::line_cuts(paths_text->data, paths_text->size, host_threads(paths_text->size));
            std::vector<std::vector<std::pair<int32_t, int32_t>>> backed(cut.size() - 1);
            const ConjGraph &g = jobs[0].g;
            palace_host::for_parts(cut, [&](size_t k, size_t a, size_t b) {
                for_each_path_line(paths_text->data + a, b - a, [&](const std::vector<PathTok> &pl) {
                    backed_by(g, pl, [&](int32_t u, int32_t v) { backed[k].emplace_back(u, v); });
                });
            });
            for (const auto &part : backed)
                for (const auto &uv : part) jobs[0].g.add(uv.first, uv.second, 0, 1);
        } else if (paths_text) {
            // which graphs know a contig id: a path line is applied, in file order, to every graph that knows one of its ids
            palace_host::Names ids;
            std::vector<std::vector<uint32_t>> jobs_of;
            for (uint32_t ji = 0; ji < jobs.size(); ji++)
                for (sv id : jobs[ji].g.seg_of_id.names) {
                    const int t = ids.intern(id);
                    if (static_cast<size_t>(t) >= jobs_of.size()) jobs_of.resize(static_cast<size_t>(t) + 1);
                    jobs_of[static_cast<size_t>(t)].push_back(ji);
                }
            std::vector<uint32_t> touched;
            for_each_path_line(paths_text->data, paths_text->size, [&](const std::vector<PathTok> &pl) {
                touched.clear();
                for (const PathTok &t : pl) {
                    if (!t.ok) continue;
                    const int k = ids.find(t.id);
                    if (k >= 0) touched.insert(touched.end(), jobs_of[static_cast<size_t>(k)].begin(), jobs_of[static_cast<size_t>(k)].end());
                }
                std::sort(touched.begin(), touched.end());
                touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
                for (uint32_t ji : touched) apply_path(jobs[ji].g, pl);
            });
        }
    } catch (const std::exception &e) { std::cerr << "matching: " << e.what() << "\n"; hip_up.join(); return 1; }
    tr.lap("contigs.paths applied");
    // The union of all graphs as one conjugate graph (vertex ids offset per graph).  Components never span two graphs and
    // come out in first-vertex order, and the arc order (weight, path-backed, class key, ends) compares two arcs of one
    // graph the same way with or without the offset, so every graph gets exactly the decomposition of a run of its own.
    int64_t S_total = 0, E = 0;
    for (Job &j : jobs) { j.v0 = static_cast<int32_t>(2 * S_total); S_total += static_cast<int64_t>(j.g.seg_of.names.size()); E += static_cast<int64_t>(j.g.arcs.size()); }
    if (2 * S_total >= (1ll << 31)) { std::cerr << "matching: too many segments\n"; hip_up.join(); return 1; }
    const int32_t S = static_cast<int32_t>(S_total), V = 2 * S;
    std::vector<Arc> arcs;
    arcs.reserve(static_cast<size_t>(E));
    std::vector<int64_t> copies;
    copies.reserve(static_cast<size_t>(S));
    for (Job &j : jobs) {
        copies.insert(copies.end(), j.g.copies.begin(), j.g.copies.end());
        for (Arc a : j.g.arcs) { a.u += j.v0; a.v += j.v0; arcs.push_back(a); }
    }
    for (Arc &a : arcs) {
        uint64_t k1 = static_cast<uint64_t>(a.u) * V + a.v, k2 = static_cast<uint64_t>(a.v ^ 1) * V + (a.u ^ 1);
        a.cls = std::min(k1, k2);
    }
    std::sort(arcs.begin(), arcs.end(), [](const Arc &x, const Arc &y) {       // rank order
        if (x.w != y.w) return x.w > y.w;
        if (x.backed != y.backed) return x.backed > y.backed;
        if (x.cls != y.cls) return x.cls < y.cls;
        return x.u != y.u ? x.u < y.u : x.v < y.v;
    });
    std::vector<int32_t> src(static_cast<size_t>(E)), dst(static_cast<size_t>(E));
    for (int64_t e = 0; e < E; e++) { src[static_cast<size_t>(e)] = arcs[static_cast<size_t>(e)].u; dst[static_cast<size_t>(e)] = arcs[static_cast<size_t>(e)].v; }
    tr.lap("arcs ranked");
    hip_up.join();
    if (ctx_rc) { std::cerr << "matching: cannot set up the GPU: " << ctx_err << "\n"; return 1; }
    tr.lap("hip runtime up (joined)");
    palace_match_result *res = nullptr;
    CK(palace_match_decompose(ctx, S, copies.data(), E, src.data(), dst.data(), opt.iterations, opt.aggressive, &res));
    tr.lap("decompose");
    palace_ctx_destroy(ctx);

    const int64_t n_comp = palace_match_result_count(res);
    const int64_t *off = palace_match_result_offsets(res);
    const int32_t *verts = palace_match_result_verts(res), *iter = palace_match_result_iter(res),
                  *open_at = palace_match_result_open_at(res);
    const uint8_t *kind = palace_match_result_kind(res);
    std::vector<int32_t> v0s;
    for (const Job &j : jobs) v0s.push_back(j.v0);
    auto job_of = [&](int64_t c) -> Job & { return jobs[static_cast<size_t>(std::upper_bound(v0s.begin(), v0s.end(), verts[off[c]]) - v0s.begin() - 1)]; };
    // the lines of the components (and, with -b, of the cycles opened at their weakest arc) are written out by parts of the
    // component range on threads; which of them are new, and where they go, is decided in component order by this thread
    struct PartText { std::string text; std::vector<uint64_t> at; };              // at[2k], at[2k+1]: start of component k's line / its opened line
    const size_t n_parts = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(n_comp), host_threads(n_comp < 4096 ? 0 : size_t{1} << 20)));   // (PALACE_HOST_THREADS forces parts: tests)
    std::vector<PartText> part(n_parts);
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < n_parts; k++)
            pool.emplace_back([&, k] {
                const int64_t c0 = n_comp * static_cast<int64_t>(k) / static_cast<int64_t>(n_parts), c1 = n_comp * static_cast<int64_t>(k + 1) / static_cast<int64_t>(n_parts);
                PartText &pt = part[k];
                pt.at.reserve(2 * static_cast<size_t>(c1 - c0) + 1);
                pt.text.reserve(static_cast<size_t>(off[c1] - off[c0]) * 40 + 64);
                for (int64_t c = c0; c < c1; c++) {
                    const int64_t n = off[c + 1] - off[c];
                    const Job &j = n ? job_of(c) : jobs[0];
                    for (int pass = 0; pass < 2; pass++) {
                        pt.at.push_back(pt.text.size());
                        if (n == 0 || (pass == 1 && !(kind[c] && opt.break_cycles))) continue;
                        const int64_t first = pass ? open_at[c] : 0;
                        for (int64_t i = 0; i < n; i++) {
                            const int32_t v = verts[off[c] + (first + i) % n] - j.v0;
                            if (i) pt.text += '\t';
                            pt.text += j.g.seg_of.names[static_cast<size_t>(v >> 1)];
                            pt.text += (v & 1) ? '-' : '+';
                        }
                        pt.text += '\n';
                    }
                }
                pt.at.push_back(pt.text.size());
            });
        for (auto &t : pool) t.join();
    }
    for (size_t k = 0; k < n_parts; k++) {
        const int64_t c0 = n_comp * static_cast<int64_t>(k) / static_cast<int64_t>(n_parts), c1 = n_comp * static_cast<int64_t>(k + 1) / static_cast<int64_t>(n_parts);
        const PartText &pt = part[k];
        for (int64_t c = c0; c < c1; c++) {
            const int64_t n = off[c + 1] - off[c];
            if (n == 0) continue;
            Job &j = job_of(c);
            const size_t a = 2 * static_cast<size_t>(c - c0);
            const sv s = sv(pt.text).substr(pt.at[a], pt.at[a + 1] - pt.at[a]), open = sv(pt.text).substr(pt.at[a + 1], pt.at[a + 2] - pt.at[a + 1]);
            if (!kind[c]) {
                if (n == 1 && iter[c] > 0) continue;               // a bare segment is reported once, in round 0
                if (j.lin_seen.insert(s).second) j.lin += s;
                continue;
            }
            if (!j.cyc_seen.insert(s).second) continue;
            if (n == 1 && opt.self_loops) { j.selfs += "self\n"; j.selfs += s; }
            else { j.cyc += "iter " + std::to_string(iter[c]) + "\n"; j.cyc += s; }
            if (opt.break_cycles && j.lin_seen.insert(open).second) j.lin += open;   // also reported opened at its weakest arc
        }
    }
    palace_match_result_free(res);
    for (Job &j : jobs) {
        j.cyc += j.selfs;
        std::ofstream fl(j.linear, std::ios::binary), fc(j.cycle, std::ios::binary);
        if (!fl || !fc) { std::cerr << "matching: cannot write outputs of " << j.graph << "\n"; return 1; }
        fl << j.lin;
        fc << j.cyc;
        fl.close(); fc.close();
        if (fl.fail() || fc.fail()) { std::cerr << "matching: failed writing the outputs of " << j.graph << "\n"; return 1; }
    }
    tr.lap("text output");
    fast_exit.done(0);          // outputs are complete and closed: the caller goes on, the teardown happens behind it
}
