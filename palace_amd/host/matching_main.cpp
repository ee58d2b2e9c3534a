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
#include <sstream>
#include <string>
#include <unistd.h>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/palace_hip.h"
#include "trace.hpp"

namespace {

#define CK(call)                                                                       \
    do {                                                                               \
        int rc__ = (call);                                                             \
        if (rc__ != 0) {                                                               \
            std::cerr << "matching: " #call " failed: " << palace_last_error() << "\n"; \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

struct Options {
    std::string graph, linear, cycle, paths;
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
        else if (a == "-h" || a == "--help") return false;
        else { std::cerr << "matching: unknown option " << a << "\n"; return false; }
    }
    return !o.graph.empty() && !o.linear.empty() && !o.cycle.empty();
}

struct Arc { int32_t u, v; int64_t w; int32_t backed; uint64_t cls; };

struct ConjGraph {
    std::vector<std::string> name;
    std::vector<int64_t> copies;
    std::unordered_map<std::string, int32_t> seg_of, seg_of_id;
    std::unordered_map<uint64_t, size_t> arc_of;       // (u << 32 | v) -> index in arcs
    std::vector<Arc> arcs;

    int32_t seg(const std::string &n)
    {
        auto it = seg_of.find(n);
        if (it != seg_of.end()) return it->second;
        int32_t s = static_cast<int32_t>(name.size());
        seg_of.emplace(n, s);
        name.push_back(n);
        copies.push_back(1);
        size_t a = n.find('_');
        if (a != std::string::npos) {
            size_t b = n.find('_', a + 1);
            seg_of_id[n.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1)] = s;   // later wins
        }
        return s;
    }
    void bump(int32_t u, int32_t v, int64_t w, int32_t backed)
    {
        uint64_t k = (static_cast<uint64_t>(static_cast<uint32_t>(u)) << 32) | static_cast<uint32_t>(v);
        auto it = arc_of.find(k);
        if (it == arc_of.end()) { arc_of.emplace(k, arcs.size()); arcs.push_back({u, v, w, backed, 0}); }
        else { arcs[it->second].w += w; arcs[it->second].backed |= backed; }
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

void load_graph(ConjGraph &g, const Options &o)
{
    std::ifstream in(o.graph);
    if (!in) throw std::runtime_error("cannot open graph " + o.graph);
    std::string line;
    while (std::getline(in, line)) {
        auto t = words(line);
        if (t.size() >= 4 && t[0] == "SEG") {
            int32_t s = g.seg(t[1]);
            g.copies[s] = std::max<int64_t>(1, static_cast<int64_t>(std::atof(t[3].c_str())));
        } else if (t.size() >= 7 && t[0] == "JUNC") {
            int32_t a = g.seg(t[1]), b = g.seg(t[3]);
            g.add(2 * a + (t[2] == "-"), 2 * b + (t[4] == "-"), std::atol(t[5].c_str()) + std::atol(t[6].c_str()), 0);
        }
    }
    if (o.paths.empty()) return;
    std::ifstream pin(o.paths);
    while (std::getline(pin, line)) {
        if (line.rfind("NODE", 0) == 0) continue;
        int32_t before = -1;
        size_t p = 0;
        while (p <= line.size()) {
            size_t c = line.find(',', p);
            std::string tok = line.substr(p, c == std::string::npos ? std::string::npos : c - p);
            p = c == std::string::npos ? line.size() + 1 : c + 1;
            while (!tok.empty() && (tok.back() == ';' || tok.back() == '\r' || tok.back() == ' ')) tok.pop_back();
            int32_t here = -1;
            if (tok.size() >= 2 && (tok.back() == '+' || tok.back() == '-')) {
                auto it = g.seg_of_id.find(tok.substr(0, tok.size() - 1));
                if (it != g.seg_of_id.end()) here = 2 * it->second + (tok.back() == '-');
            }
            if (before >= 0 && here >= 0) g.add(before, here, 0, 1);
            before = here;
        }
    }
}

}  // namespace

int main(int argc, char **argv)
{
    Options opt;
    if (!parse_args(argc, argv, opt)) {
        std::cerr << "Usage: matching -g <graph> -r <linear out> -c <cycle out> [-s] [-i <iterations>] [-b] "
                     "[-l <contigs.paths>] [--aggressive]\n";
        return 1;
    }
    palace_host::Trace tr("matching");
    palace_ctx *ctx = nullptr;                                // the HIP runtime comes up while the graph text is read
    int ctx_rc = 0;
    std::string ctx_err;
    std::thread hip_up([&] {
        ctx_rc = palace_ctx_create(0, &ctx);
        if (ctx_rc) ctx_err = palace_last_error();
    });
    ConjGraph g;
    try {
        load_graph(g, opt);
    } catch (const std::exception &e) { std::cerr << "matching: " << e.what() << "\n"; hip_up.join(); return 1; }
    tr.lap("graph + paths read");
    const int32_t S = static_cast<int32_t>(g.name.size()), V = 2 * S;
    for (Arc &a : g.arcs) {
        uint64_t k1 = static_cast<uint64_t>(a.u) * V + a.v, k2 = static_cast<uint64_t>(a.v ^ 1) * V + (a.u ^ 1);
        a.cls = std::min(k1, k2);
    }
    std::sort(g.arcs.begin(), g.arcs.end(), [](const Arc &x, const Arc &y) {       // rank order
        if (x.w != y.w) return x.w > y.w;
        if (x.backed != y.backed) return x.backed > y.backed;
        if (x.cls != y.cls) return x.cls < y.cls;
        return x.u != y.u ? x.u < y.u : x.v < y.v;
    });
    const int64_t E = static_cast<int64_t>(g.arcs.size());
    std::vector<int32_t> src(E), dst(E);
    for (int64_t e = 0; e < E; e++) { src[e] = g.arcs[e].u; dst[e] = g.arcs[e].v; }

    tr.lap("arcs ranked");
    hip_up.join();
    if (ctx_rc) { std::cerr << "matching: cannot set up the GPU: " << ctx_err << "\n"; return 1; }
    tr.lap("hip runtime up (joined)");
    palace_match_result *res = nullptr;
    CK(palace_match_decompose(ctx, S, g.copies.data(), E, src.data(), dst.data(), opt.iterations, opt.aggressive, &res));
    tr.lap("decompose");
    palace_ctx_destroy(ctx);

    const int64_t n_comp = palace_match_result_count(res);
    const int64_t *off = palace_match_result_offsets(res);
    const int32_t *verts = palace_match_result_verts(res), *iter = palace_match_result_iter(res),
                  *open_at = palace_match_result_open_at(res);
    const uint8_t *kind = palace_match_result_kind(res);
    auto line_of = [&](int64_t c, int64_t first) {
        const int64_t n = off[c + 1] - off[c];
        std::string s;
        for (int64_t i = 0; i < n; i++) {
            const int32_t v = verts[off[c] + (first + i) % n];
            if (i) s += '\t';
            s += g.name[v >> 1];
            s += (v & 1) ? '-' : '+';
        }
        s += '\n';
        return s;
    };
    std::string lin, cyc, selfs;
    std::unordered_set<std::string> lin_seen, cyc_seen;
    for (int64_t c = 0; c < n_comp; c++) {
        const int64_t n = off[c + 1] - off[c];
        if (!kind[c]) {
            if (n == 1 && iter[c] > 0) continue;               // a bare segment is reported once, in round 0
            std::string s = line_of(c, 0);
            if (lin_seen.insert(s).second) lin += s;
            continue;
        }
        std::string s = line_of(c, 0);
        if (!cyc_seen.insert(s).second) continue;
        if (n == 1 && opt.self_loops) selfs += "self\n" + s;
        else cyc += "iter " + std::to_string(iter[c]) + "\n" + s;
        if (opt.break_cycles) {                                 // also report it opened at its weakest arc
            std::string open = line_of(c, open_at[c]);
            if (lin_seen.insert(open).second) lin += open;
        }
    }
    palace_match_result_free(res);
    cyc += selfs;
    std::ofstream fl(opt.linear, std::ios::binary), fc(opt.cycle, std::ios::binary);
    if (!fl || !fc) { std::cerr << "matching: cannot write outputs\n"; return 1; }
    fl << lin;
    fc << cyc;
    fl.close(); fc.close();
    tr.lap("text output");
    std::fflush(nullptr);
    _exit(0);                   // outputs are complete and closed: skip the teardown of the host containers
}
