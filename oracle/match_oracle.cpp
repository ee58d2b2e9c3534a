// oracle/match_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU statement of THIS REPOSITORY'S OWN `matching` algorithm (DESIGN.md, section "matching").
// The reference's `matching` executable is absent from the reference tree, source and binary
// alike (bin/matching is listed in .MISSING_LARGE_BLOBS, its source directory
// seqGraph/ is only mentioned in .gitignore:1-3 and a commented-out README block), and no file of
// the tree pins its results.  Only its command line (palace:587-590, 684-688) and the grammar its
// consumers accept (filter_result.py:125-134, make_fa_from_path.py:94-96, remove_cycle_dup.py:9-13,
// make_final_fa.py:20-34 for the conjugate rule) are recoverable.
//
// Parity status: UNPINNED against the reference; this file pins the product's GPU path to a
// sequential, obviously-greedy statement of the same algorithm.
//
// Algorithm: oriented vertices v = 2*seg + (orient == '-'); every JUNC L oL R oR n1 n2 is the arc
// (L,oL)->(R,oR) of weight n1+n2 together with its conjugate (R,~oR)->(L,~oL).  Consecutive tokens
// of a SPAdes path line add (or mark) arcs as "path backed".  Arc classes {a, conj a} are ranked by
// (weight desc, path backed first, canonical key asc).  One iteration = a greedy matching in rank
// order (an arc class is taken iff its tail has no successor yet and its head no predecessor),
// followed by reading off the paths and cycles the successor links form; every segment on an
// emitted component pays copies from its copy number, exhausted segments leave the graph, and the
// next iteration re-matches what is left (at most `iterations` times).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Arc { int u, v; long w; int backed; uint64_t cls; };

struct Graph {
    std::vector<std::string> name;
    std::vector<long> cn;
    std::map<std::string, int> index;
    std::map<std::string, int> by_id;          // "123" -> segment index (EDGE_123_...)
    std::map<std::pair<int, int>, std::pair<long, int>> arcs;   // (u,v) -> (weight, backed)
    int seg(const std::string &n)
    {
        auto it = index.find(n);
        if (it != index.end()) return it->second;
        int i = (int)name.size();
        index[n] = i; name.push_back(n); cn.push_back(1);
        size_t a = n.find('_');
        if (a != std::string::npos) {
            size_t b = n.find('_', a + 1);
            by_id[n.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1)] = i;
        }
        return i;
    }
    void add(int u, int v, long w, int backed)
    {
        auto &a = arcs[{u, v}];
        a.first += w; a.second |= backed;
        if (std::make_pair(v ^ 1, u ^ 1) != std::make_pair(u, v)) {
            auto &b = arcs[{v ^ 1, u ^ 1}];
            b.first += w; b.second |= backed;
        }
    }
};

void load(Graph &g, const char *graph_path, const char *paths_path)
{
    std::ifstream in(graph_path);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::vector<std::string> t;
        for (std::string x; ss >> x;) t.push_back(x);
        if (t.size() >= 4 && t[0] == "SEG") {
            int s = g.seg(t[1]);
            g.cn[s] = std::max(1L, (long)std::atof(t[3].c_str()));
        } else if (t.size() >= 7 && t[0] == "JUNC") {
            int a = g.seg(t[1]), b = g.seg(t[3]);
            g.add(2 * a + (t[2] == "-"), 2 * b + (t[4] == "-"), std::atol(t[5].c_str()) + std::atol(t[6].c_str()), 0);
        }
    }
    if (!paths_path || !*paths_path) return;
    std::ifstream pin(paths_path);
    while (std::getline(pin, line)) {
        if (line.compare(0, 4, "NODE") == 0) continue;
        int prev = -1;
        std::string tok;
        std::istringstream ss(line);
        while (std::getline(ss, tok, ',')) {
            while (!tok.empty() && (tok.back() == ';' || tok.back() == '\r' || tok.back() == ' ')) tok.pop_back();
            int cur = -1;
            if (tok.size() >= 2 && (tok.back() == '+' || tok.back() == '-')) {
                auto it = g.by_id.find(tok.substr(0, tok.size() - 1));
                if (it != g.by_id.end()) cur = 2 * it->second + (tok.back() == '-');
            }
            if (prev >= 0 && cur >= 0) g.add(prev, cur, 0, 1);
            prev = cur;
        }
    }
}

std::string token(const Graph &g, int v) { return g.name[v >> 1] + ((v & 1) ? "-" : "+"); }

}  // namespace

extern "C" long orc_match_run(const char *graph_path, const char *paths_path, int iterations, int self_loops,
                              int break_cycles, int aggressive, char *lin_out, size_t lin_cap, char *cyc_out,
                              size_t cyc_cap, long *cyc_len)
{
    Graph g;
    load(g, graph_path, paths_path);
    const int S = (int)g.name.size(), V = 2 * S;
    std::vector<Arc> arcs;
    for (auto &kv : g.arcs) {
        int u = kv.first.first, v = kv.first.second;
        uint64_t k1 = (uint64_t)u * V + v, k2 = (uint64_t)(v ^ 1) * V + (u ^ 1);
        arcs.push_back({u, v, kv.second.first, kv.second.second, std::min(k1, k2)});
    }
    std::sort(arcs.begin(), arcs.end(), [](const Arc &a, const Arc &b) {
        if (a.w != b.w) return a.w > b.w;
        if (a.backed != b.backed) return a.backed > b.backed;
        if (a.cls != b.cls) return a.cls < b.cls;
        return std::make_pair(a.u, a.v) < std::make_pair(b.u, b.v);
    });
    std::vector<long> left(g.cn);
    std::string lin, cyc, selfs;
    std::set<std::string> seen_lin, seen_cyc;
    const int rounds = iterations + (aggressive ? 1 : 0);
    for (int t = 0; t < rounds; t++) {
        if (aggressive && t == rounds - 1) std::fill(left.begin(), left.end(), 1L);
        std::vector<char> alive(S);
        bool any = false;
        for (int s = 0; s < S; s++) { alive[s] = left[s] > 0; any |= alive[s]; }
        if (!any) continue;
        std::vector<int> nxt(V, -1), prv(V, -1);
        std::vector<long> wt(V, 0);                              // weight rank of the arc leaving v
        for (size_t r = 0; r < arcs.size(); r++) {
            const Arc &a = arcs[r];
            if (!alive[a.u >> 1] || !alive[a.v >> 1]) continue;
            if (nxt[a.u] >= 0 || prv[a.v] >= 0) continue;
            nxt[a.u] = a.v; prv[a.v] = a.u; wt[a.u] = (long)r;
        }
        std::vector<char> done(V, 0);
        struct Comp { std::vector<int> v; bool cycle; };
        std::vector<Comp> comps;
        for (int v = 0; v < V; v++) {                            // paths: start at vertices without predecessor
            if (!alive[v >> 1] || done[v] || prv[v] >= 0) continue;
            Comp c{{}, false};
            for (int x = v; x >= 0; x = nxt[x]) { c.v.push_back(x); done[x] = 1; }
            std::vector<int> rc;
            for (auto it = c.v.rbegin(); it != c.v.rend(); ++it) rc.push_back(*it ^ 1);
            for (int x : rc) done[x] = 1;
            if (rc.front() < c.v.front()) c.v = rc;              // representative of {P, conj P}
            comps.push_back(c);
        }
        for (int v = 0; v < V; v++) {                            // what is left are cycles
            if (!alive[v >> 1] || done[v]) continue;
            std::vector<int> cy;
            for (int x = v; !done[x]; x = nxt[x]) { cy.push_back(x); done[x] = 1; }
            std::vector<int> rc;
            for (auto it = cy.rbegin(); it != cy.rend(); ++it) rc.push_back(*it ^ 1);
            for (int x : rc) done[x] = 1;
            auto rot = [](std::vector<int> c) { std::rotate(c.begin(), std::min_element(c.begin(), c.end()), c.end()); return c; };
            std::vector<int> a = rot(cy), b = rot(rc);
            comps.push_back(Comp{b.front() < a.front() ? b : a, true});
        }
        std::sort(comps.begin(), comps.end(), [](const Comp &a, const Comp &b) { return a.v.front() < b.v.front(); });
        for (const Comp &c : comps) {
            std::map<int, long> occ;
            for (int x : c.v) occ[x >> 1]++;
            long m = -1;
            for (auto &kv : occ) { long q = left[kv.first] / kv.second; m = m < 0 ? q : std::min(m, q); }
            if (m < 1) m = 1;
            for (auto &kv : occ) left[kv.first] = std::max(0L, left[kv.first] - m * kv.second);
            if (c.v.size() == 1 && !c.cycle && t > 0) continue;   // bare segments are reported once (t = 0)
            std::string body;
            for (size_t i = 0; i < c.v.size(); i++) body += (i ? "\t" : "") + token(g, c.v[i]);
            body += "\n";
            if (!c.cycle) {
                if (seen_lin.insert(body).second) lin += body;
                continue;
            }
            if (!seen_cyc.insert(body).second) continue;
            if (c.v.size() == 1 && self_loops) selfs += "self\n" + body;
            else cyc += "iter " + std::to_string(t) + "\n" + body;
            if (break_cycles) {                                   // open the cycle at its weakest arc
                size_t worst = 0;
                for (size_t i = 1; i < c.v.size(); i++)
                    if (wt[c.v[i]] > wt[c.v[worst]]) worst = i;
                std::string open;
                for (size_t i = 0; i < c.v.size(); i++) open += (i ? "\t" : "") + token(g, c.v[(worst + 1 + i) % c.v.size()]);
                open += "\n";
                if (seen_lin.insert(open).second) lin += open;
            }
        }
    }
    cyc += selfs;
    if (lin.size() > lin_cap || cyc.size() > cyc_cap) return -1;
    std::memcpy(lin_out, lin.data(), lin.size());
    std::memcpy(cyc_out, cyc.data(), cyc.size());
    *cyc_len = (long)cyc.size();
    return (long)lin.size();
}
