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
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/palace_hip.h"

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

template <class T>
int to_device(palace_ctx *ctx, const std::vector<T> &v, T **d)
{
    void *p = nullptr;
    int rc = palace_malloc(ctx, std::max<size_t>(1, v.size()) * sizeof(T), &p);
    if (rc) return rc;
    *d = static_cast<T *>(p);
    return palace_h2d(ctx, p, v.data(), v.size() * sizeof(T));
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
    ConjGraph g;
    try {
        load_graph(g, opt);
    } catch (const std::exception &e) { std::cerr << "matching: " << e.what() << "\n"; return 1; }
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
    std::vector<int32_t> src(E), dst(E), out_arcs(E), in_arcs(E);
    std::vector<int64_t> out_off(V + 1, 0), in_off(V + 1, 0);
    for (int64_t e = 0; e < E; e++) { src[e] = g.arcs[e].u; dst[e] = g.arcs[e].v; out_off[src[e] + 1]++; in_off[dst[e] + 1]++; }
    for (int32_t v = 0; v < V; v++) { out_off[v + 1] += out_off[v]; in_off[v + 1] += in_off[v]; }
    {
        std::vector<int64_t> po(out_off.begin(), out_off.end() - 1), pi(in_off.begin(), in_off.end() - 1);
        for (int64_t e = 0; e < E; e++) { out_arcs[po[src[e]]++] = static_cast<int32_t>(e); in_arcs[pi[dst[e]]++] = static_cast<int32_t>(e); }
    }

    palace_ctx *ctx = nullptr;
    CK(palace_ctx_create(0, &ctx));
    int32_t *d_src, *d_dst, *d_oa, *d_ia, *d_next, *d_prev, *d_narc; int64_t *d_oo, *d_io; uint8_t *d_alive;
    CK(to_device(ctx, src, &d_src)); CK(to_device(ctx, dst, &d_dst)); CK(to_device(ctx, out_arcs, &d_oa));
    CK(to_device(ctx, in_arcs, &d_ia)); CK(to_device(ctx, out_off, &d_oo)); CK(to_device(ctx, in_off, &d_io));
    std::vector<int32_t> next(V), prev(V), narc(V);
    std::vector<uint8_t> alive(V);
    CK(to_device(ctx, next, &d_next)); CK(to_device(ctx, prev, &d_prev)); CK(to_device(ctx, narc, &d_narc));
    CK(to_device(ctx, alive, &d_alive));

    auto tok = [&](int32_t v) { return g.name[v >> 1] + ((v & 1) ? "-" : "+"); };
    auto line_of = [&](const std::vector<int32_t> &vs, size_t first) {
        std::string s;
        for (size_t i = 0; i < vs.size(); i++) { if (i) s += '\t'; s += tok(vs[(first + i) % vs.size()]); }
        s += '\n';
        return s;
    };
    std::vector<int64_t> left(g.copies);
    std::string lin, cyc, selfs;
    std::unordered_set<std::string> lin_seen, cyc_seen;
    const int rounds = opt.iterations + (opt.aggressive ? 1 : 0);
    for (int t = 0; t < rounds; t++) {
        if (opt.aggressive && t == rounds - 1) std::fill(left.begin(), left.end(), 1);
        bool any = false;
        for (int32_t s = 0; s < S; s++) { alive[2 * s] = alive[2 * s + 1] = left[s] > 0; any |= left[s] > 0; }
        if (!any) break;
        CK(palace_h2d(ctx, d_alive, alive.data(), alive.size()));
        CK(palace_match_greedy(ctx, V, E, d_src, d_dst, d_oo, d_oa, d_io, d_ia, d_alive, d_next, d_prev, d_narc, nullptr));
        CK(palace_d2h(ctx, next.data(), d_next, next.size() * 4));
        CK(palace_d2h(ctx, prev.data(), d_prev, prev.size() * 4));
        CK(palace_d2h(ctx, narc.data(), d_narc, narc.size() * 4));

        struct Comp { std::vector<int32_t> v; bool cycle; };
        std::vector<Comp> comps;
        std::vector<uint8_t> seen(V, 0);
        for (int32_t v = 0; v < V; v++) {                     // open paths, one representative per conjugate pair
            if (!alive[v] || seen[v] || prev[v] >= 0) continue;
            std::vector<int32_t> p;
            for (int32_t x = v; x >= 0; x = next[x]) { p.push_back(x); seen[x] = 1; }
            const int32_t conj_head = p.back() ^ 1;
            for (int32_t x : p) seen[x ^ 1] = 1;
            if (conj_head < p.front()) {
                std::reverse(p.begin(), p.end());
                for (int32_t &x : p) x ^= 1;
            }
            comps.push_back({std::move(p), false});
        }
        for (int32_t v = 0; v < V; v++) {                     // closed walks
            if (!alive[v] || seen[v]) continue;
            std::vector<int32_t> c;
            for (int32_t x = v; !seen[x]; x = next[x]) { c.push_back(x); seen[x] = 1; }
            int32_t lo = *std::min_element(c.begin(), c.end()), lo_conj = c[0] ^ 1;
            for (int32_t x : c) { seen[x ^ 1] = 1; lo_conj = std::min(lo_conj, x ^ 1); }
            if (lo_conj < lo) {
                std::reverse(c.begin(), c.end());
                for (int32_t &x : c) x ^= 1;
            }
            std::rotate(c.begin(), std::min_element(c.begin(), c.end()), c.end());
            comps.push_back({std::move(c), true});
        }
        std::sort(comps.begin(), comps.end(), [](const Comp &a, const Comp &b) { return a.v.front() < b.v.front(); });
        for (const Comp &c : comps) {
            std::unordered_map<int32_t, int64_t> uses;
            for (int32_t x : c.v) uses[x >> 1]++;
            int64_t pay = -1;
            for (auto &kv : uses) { int64_t q = left[kv.first] / kv.second; pay = pay < 0 ? q : std::min(pay, q); }
            pay = std::max<int64_t>(1, pay);
            for (auto &kv : uses) left[kv.first] = std::max<int64_t>(0, left[kv.first] - pay * kv.second);
            if (!c.cycle) {
                if (c.v.size() == 1 && t > 0) continue;        // a bare segment is reported once, in round 0
                std::string s = line_of(c.v, 0);
                if (lin_seen.insert(s).second) lin += s;
                continue;
            }
            std::string s = line_of(c.v, 0);
            if (!cyc_seen.insert(s).second) continue;
            if (c.v.size() == 1 && opt.self_loops) selfs += "self\n" + s;
            else cyc += "iter " + std::to_string(t) + "\n" + s;
            if (opt.break_cycles) {                             // also report it opened at its weakest arc
                size_t worst = 0;
                for (size_t i = 1; i < c.v.size(); i++)
                    if (narc[c.v[i]] > narc[c.v[worst]]) worst = i;
                std::string open = line_of(c.v, worst + 1);
                if (lin_seen.insert(open).second) lin += open;
            }
        }
    }
    palace_ctx_destroy(ctx);
    cyc += selfs;
    std::ofstream fl(opt.linear, std::ios::binary), fc(opt.cycle, std::ios::binary);
    if (!fl || !fc) { std::cerr << "matching: cannot write outputs\n"; return 1; }
    fl << lin;
    fc << cyc;
    return 0;
}
