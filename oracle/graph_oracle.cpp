// oracle/graph_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's `generateGraph` (bin/generate_graph.cpp), written from the
// behaviour of that file.  It starts from decoded BAM records (what htslib's sam_read1 hands the
// reference) because htslib is not in this image: the reference translation unit cannot be built
// here without stand-ins for its headers, so no reference-run golden exists for this stage and
// the reference ships no tests or fixtures for it.
//
// Parity status: UNPINNED (restatement checked only by reading it against the cited lines and by
// hand-computed cases in tests/test_oracle_graph.py).
//
// Used as the checker by tests/ and as bench.py's cpu_baseline leg; never linked by the product.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Opts {                      // generate_graph.cpp:20-44, set by :573-593
    int max_end = 300, min_mapq = 0, max_nm = 5;
    int enable_paired = 1, both_order = 0, min_count = 5;
    double max_span_frac = 0.80;
};

enum Region { R_START = 0, R_END = 1, R_MIDDLE = 2 };

// generate_graph.cpp:56-62
Region region_of(int pos1, int len, const Opts &o)
{
    int pref = std::min(o.max_end, len / 2), suff = std::max(len - o.max_end, len / 2);
    if (pos1 <= pref) return R_START;
    if (pos1 > suff) return R_END;
    return R_MIDDLE;
}
Region flipped(Region r) { return r == R_START ? R_END : r == R_END ? R_START : R_MIDDLE; }   // :81-85
int to_start(int pos) { return std::max(0, pos - 1); }                                         // :67-69
int to_end(int pos, int len) { return std::max(0, len - pos); }                                // :74-76

struct Interval { int start = 0, end = 0, len = 0, clip_s = 0, clip_e = 0; };

// generate_graph.cpp:330-383 on a textual CIGAR (zero-length ops are dropped, :340)
Interval read_interval(const std::string &cigar, bool rev, int read_len)
{
    Interval r;
    if (cigar.empty()) return r;
    std::vector<std::pair<int, char>> ops;
    int n = 0;
    for (char c : cigar) {
        if (std::isdigit((unsigned char)c)) n = n * 10 + (c - '0');
        else { if (n > 0) ops.push_back({n, c}); n = 0; }
    }
    if (!ops.empty() && ops.front().second == 'S') r.clip_s = ops.front().first;
    if (ops.size() > 1 && ops.back().second == 'S') r.clip_e = ops.back().first;
    for (auto &op : ops)
        if (op.second == 'M' || op.second == 'I' || op.second == 'S' || op.second == '=' || op.second == 'X')
            r.len += op.first;
    if (rev && read_len > 0) { r.start = read_len - (r.len - r.clip_e) + 1; r.end = read_len - r.clip_s; }
    else { r.start = r.clip_s + 1; r.end = r.len - r.clip_e; }
    return r;
}

// generate_graph.cpp:401-428
bool stitchable(const Interval &a, const Interval &b, int max_gap, int max_ov, bool &first1)
{
    if (a.end <= b.start && b.start - a.end - 1 <= max_gap) { first1 = true; return true; }
    if (b.end <= a.start && a.start - b.end - 1 <= max_gap) { first1 = false; return true; }
    if (a.start <= b.end && b.start <= a.end) {
        int ov = std::min(a.end, b.end) - std::max(a.start, b.start) + 1;
        if (ov <= max_ov) { first1 = a.start <= b.start; return true; }
    }
    return false;
}

struct Side { bool rev; Region reg; int pos, len; };

// generate_graph.cpp:510-538 (split) and :465-506 (paired): is (oL, oR) a valid layout with
// `l` on the left and `r` on the right?
bool split_layout_ok(const Side &l, const Side &r, char oL, char oR)
{
    bool fwdL = (oL == '-') ? l.rev : !l.rev, fwdR = (oR == '-') ? r.rev : !r.rev;
    if (!fwdL || !fwdR) return false;
    if (l.reg == R_MIDDLE || r.reg == R_MIDDLE) return false;
    if (l.reg != (oL == '+' ? R_END : R_START)) return false;
    if (r.reg != (oR == '+' ? R_START : R_END)) return false;
    return true;
}
bool paired_layout_ok(const Side &l, const Side &r, char oL, char oR, const Opts &o)
{
    bool fwdL = (oL == '-') ? l.rev : !l.rev, fwdR = (oR == '-') ? r.rev : !r.rev;
    if (!fwdL || fwdR) return false;
    if (l.reg == R_MIDDLE || r.reg == R_MIDDLE) return false;
    if (l.reg != (oL == '+' ? R_END : R_START)) return false;
    if (r.reg != (oR == '+' ? R_START : R_END)) return false;
    int dL = (l.reg == R_START) ? to_start(l.pos) : to_end(l.pos, l.len);
    int dR = (r.reg == R_START) ? to_start(r.pos) : to_end(r.pos, r.len);
    double fL = l.len > 0 ? (double)dL / l.len : 1.0, fR = r.len > 0 ? (double)dR / r.len : 1.0;
    return !(fL > o.max_span_frac || fR > o.max_span_frac);
}

// generate_graph.cpp:432-461 with :255-260 and :310-318.  `oL`/`oR` are the *_eval orientations.
// dbg: --debug's line of :454-458 (ostream's default formatting of the four doubles); score_out: the value :852 prints
bool score_positive(const Side &l, int mapqL, int nmL, char oL, const Side &r, int mapqR, int nmR, char oR,
                    const Opts &o, std::ostream *dbg = nullptr, double *score_out = nullptr)
{
    Region gl = (oL == '-') ? flipped(l.reg) : l.reg, gr = (oR == '-') ? flipped(r.reg) : r.reg;
    int dL = (gl == R_START) ? to_start(l.pos) : to_end(l.pos, l.len);
    int dR = (gr == R_START) ? to_start(r.pos) : to_end(r.pos, r.len);
    double lambda = std::max(50.0, (double)o.max_end / 2.0);
    double w1 = std::exp(-(double)dL / lambda), w2 = std::exp(-(double)dR / lambda);
    double w_end = w1 * w2;
    double qL = std::min(1.0, (double)mapqL / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, nmL)));
    double qR = std::min(1.0, (double)mapqR / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, nmR)));
    double score = w_end * qL * qR;
    if (dbg) *dbg << "Score calculation: w_end=" << w_end << " w_qualL=" << qL << " w_qualR=" << qR << " total=" << score << "\n";
    if (score_out) *score_out = score;
    return score > 0.0;
}

void trim(std::string &s)
{
    size_t i = 0, j = s.size();
    while (i < j && std::isspace((unsigned char)s[i])) ++i;
    while (j > i && std::isspace((unsigned char)s[j - 1])) --j;
    s = s.substr(i, j - i);
}

struct SaItem { std::string rname, cigar; int pos = -1, mapq = 0, nm = 0; bool rev = false, ok = false; };

// generate_graph.cpp:185-206
SaItem parse_sa_item(const std::string &item)
{
    SaItem it;
    std::stringstream ss(item);
    std::string f[6];
    for (int k = 0; k < 6; k++)
        if (!std::getline(ss, f[k], ',')) return it;
    for (auto &x : f) trim(x);
    if (f[0].empty() || f[1].empty()) return it;
    it.rname = f[0]; it.pos = std::atoi(f[1].c_str()); it.rev = (f[2] == "-"); it.cigar = f[3];
    it.mapq = std::atoi(f[4].c_str()); it.nm = std::atoi(f[5].c_str()); it.ok = true;
    return it;
}

using PairKey = std::tuple<std::string, std::string, char, char>;

// generate_graph.cpp:119-169: every link of the FASTG .fai and its reverse-complement twin
std::set<PairKey> parse_fastg_fai(const char *path)
{
    std::set<PairKey> out;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        std::string full = line.substr(0, line.find(';'));
        size_t colon = full.find(':');
        std::string name = full.substr(0, colon);
        bool rev = false;
        if (!name.empty() && name.back() == '\'') { rev = true; name.pop_back(); }
        if (colon == std::string::npos) continue;
        std::stringstream links(full.substr(colon + 1));
        std::string lk;
        while (std::getline(links, lk, ',')) {
            if (lk.empty()) continue;
            bool lrev = false;
            if (lk.back() == '\'') { lrev = true; lk.pop_back(); }
            char o1 = rev ? '-' : '+';
            char o2 = rev ? (lrev ? '+' : '-') : (lrev ? '-' : '+');
            out.insert(PairKey{name, lk, o1, o2});
            out.insert(PairKey{lk, name, o1 == '+' ? '-' : '+', o2 == '+' ? '-' : '+'});
        }
    }
    return out;
}

struct Agg { int supp = 0, span = 0, supp_nf = 0, span_nf = 0; std::vector<std::pair<std::string, int>> reads; };   // AggStats :300-306 (supportingReads: name, flag)

}  // namespace

extern "C" {

struct OrcRecords {
    int64_t n;
    const uint16_t *flag;
    const int32_t *tid, *pos, *mtid, *mpos;
    const uint8_t *mapq;
    const int32_t *nm;            // 0 when the NM tag is absent (generate_graph.cpp:665-667)
    const int64_t *cigar_off;     // n+1
    const uint32_t *cigar;        // BAM encoding: len << 4 | op
    const int64_t *qname_off;     // n+1
    const char *qname;
    const int64_t *sa_off;        // n+1; has_sa[i] says whether the tag exists at all
    const char *sa;
    const uint8_t *has_sa;
};

struct OrcGraphOpts { int max_end, min_mapq, max_nm, enable_paired, both_order, min_count; double max_span_frac; int debug; };   // debug: --debug (:44, :590)

void orc_graph_default_opts(OrcGraphOpts *o)
{
    Opts d;
    *o = OrcGraphOpts{d.max_end, d.min_mapq, d.max_nm, d.enable_paired, d.both_order, d.min_count, d.max_span_frac, 0};
}

// generate_graph.cpp:644-1076.  Writes the SEG/JUNC text into out; returns its size or -1.
// trace (may be null): what --debug writes to stderr on the way (:454-458, :607-609, :711-717, :746-750, :758-767, :789-797, :851-853)
static long graph_run(const OrcRecords *R, int n_targets, const char *tnames, const int64_t *tname_off,
                      const int32_t *tlen, const char *fastg_fai, double avg_depth, const OrcGraphOpts *oo,
                      char *out, size_t cap, std::string *trace)
{
    std::ostringstream dbg_text;
    std::ostream *dbg = trace ? &dbg_text : nullptr;
    auto region_name = [](Region r) { return r == R_START ? "START" : r == R_END ? "END" : "MIDDLE"; };
    Opts o;
    o.max_end = oo->max_end; o.min_mapq = oo->min_mapq; o.max_nm = oo->max_nm; o.enable_paired = oo->enable_paired;
    o.both_order = oo->both_order; o.min_count = oo->min_count; o.max_span_frac = oo->max_span_frac;
    std::vector<std::string> tname(n_targets);
    std::unordered_map<std::string, int> name_to_tid;
    for (int i = 0; i < n_targets; i++) {
        tname[i].assign(tnames + tname_off[i], tnames + tname_off[i + 1]);
        name_to_tid[tname[i]] = i;                              // :624-627 (last duplicate wins)
    }
    std::set<PairKey> fastg = parse_fastg_fai(fastg_fai);
    if (dbg) *dbg << "Loaded " << fastg.size() << " expected connections from FastG\n";      // :607-609
    std::unordered_map<std::string, double> consumed;           // :631
    std::map<PairKey, Agg> agg;                                 // :632 (same ordering as LayoutKey)
    std::unordered_set<std::string> seen_pairs;                 // :635
    static const char opchr[] = "MIDNSHP=XB";

    auto pass = [&](int mapq, int nm) { return mapq >= o.min_mapq && nm <= o.max_nm; };   // :246-248
    auto add_edge = [&](std::string cL, char oL, std::string cR, char oR, bool split, const std::string &qname, int flag) {   // :855-872, :991-1008
        PairKey key{cL, cR, oL, oR};
        if (!o.both_order && cR < cL) {
            std::swap(cL, cR);
            key = PairKey{cL, cR, oR == '-' ? '+' : '-', oL == '-' ? '+' : '-'};
        }
        bool in_fastg = fastg.count(PairKey{cL, cR, oL, oR}) > 0;      // unflipped orientations (:863, :999)
        Agg &a = agg[key];
        if (split) (in_fastg ? a.supp : a.supp_nf)++;
        else (in_fastg ? a.span : a.span_nf)++;
        a.reads.emplace_back(qname, flag);                             // :872, :1008
    };

    for (int64_t i = 0; i < R->n; i++) {
        const uint16_t f = R->flag[i];
        if (f & 0x800 || f & 0x100 || f & 0x4) continue;                       // :647-649
        const uint32_t *cg = R->cigar + R->cigar_off[i];
        const int ncg = (int)(R->cigar_off[i + 1] - R->cigar_off[i]);
        int ref_len = 0, read_len = 0;
        for (int k = 0; k < ncg; k++) {
            int op = cg[k] & 15, len = cg[k] >> 4;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += len;     // bam_cigar2rlen
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) read_len += len;    // :385-397
        }
        const int tid = R->tid[i];
        if (tid >= 0 && ref_len > 0) consumed[tname[tid]] += ref_len;          // :654-662
        const int mapq = R->mapq[i], nm = R->nm[i];
        if (!pass(mapq, nm)) continue;                                         // :679
        const std::string qname(R->qname + R->qname_off[i], R->qname + R->qname_off[i + 1]);
        bool has_supp = false;
        if (R->has_sa[i] && tid >= 0) {                                        // :687
            const std::string &r1 = tname[tid];
            Side s1{(f & 0x10) != 0, R_MIDDLE, R->pos[i] + 1, tlen[tid]};
            s1.reg = region_of(s1.pos, s1.len, o);
            std::string cigar1;
            for (int k = 0; k < ncg; k++) cigar1 += std::to_string(cg[k] >> 4) + opchr[std::min<int>(cg[k] & 15, 9)];
            Interval iv1 = read_interval(cigar1, s1.rev, read_len);
            if (dbg)                                                           // :711-717
                *dbg << "\n=== Split-read: " << qname << " (len=" << read_len << ") ===\n"
                     << "Primary: " << r1 << " pos=" << s1.pos << " rev=" << s1.rev << " region=" << region_name(s1.reg)
                     << " read[" << iv1.start << "-" << iv1.end << "]" << " CIGAR=" << cigar1 << "\n";
            std::stringstream ss(std::string(R->sa + R->sa_off[i], R->sa + R->sa_off[i + 1]));
            std::string item;
            while (std::getline(ss, item, ';')) {                              // :719
                if (item.empty()) continue;
                SaItem it = parse_sa_item(item);
                if (!it.ok || !pass(it.mapq, it.nm)) continue;
                if (r1 == it.rname) continue;
                auto tt = name_to_tid.find(it.rname);
                if (tt == name_to_tid.end()) continue;
                Side s2{it.rev, R_MIDDLE, it.pos, tlen[tt->second]};
                s2.reg = region_of(s2.pos, s2.len, o);
                if (s1.reg == R_MIDDLE || s2.reg == R_MIDDLE) continue;        // :742
                Interval iv2 = read_interval(it.cigar, it.rev, read_len);
                if (dbg)                                                       // :746-750
                    *dbg << "SA: " << it.rname << " pos=" << s2.pos << " rev=" << s2.rev << " region=" << region_name(s2.reg)
                         << " read[" << iv2.start << "-" << iv2.end << "]" << " CIGAR=" << it.cigar << "\n";
                bool first1 = false;
                if (!stitchable(iv1, iv2, 150, 150, first1)) {                 // :757
                    if (dbg) *dbg << "  -> Cannot stitch: intervals too far apart or too much overlap\n";
                    continue;
                }
                if (dbg) *dbg << "  -> Can stitch! " << (first1 ? "Primary first" : "SA first") << "\n";
                const Side &l = first1 ? s1 : s2, &r = first1 ? s2 : s1;
                char oL = 0, oR = 0;
                for (char a : {'+', '-'}) {
                    for (char b : {'+', '-'})
                        if (split_layout_ok(l, r, a, b)) { oL = a; oR = b; break; }
                    if (oL) break;
                }
                if (!oL) {
                    if (dbg) *dbg << "  -> No valid layout found\n";           // :789-791
                    continue;
                }
                const std::string &cL = first1 ? r1 : it.rname, &cR = first1 ? it.rname : r1;
                if (dbg) *dbg << "  -> Found valid layout: " << cL << "(" << oL << ") -> " << cR << "(" << oR << ")\n";   // :795-797
                int mqL = first1 ? mapq : it.mapq, nmL = first1 ? nm : it.nm;
                int mqR = first1 ? it.mapq : mapq, nmR = first1 ? it.nm : nm;
                bool left_is_a = cL <= cR;                                      // :802, :846
                double score = 0.0;
                if (score_positive(l, mqL, nmL, left_is_a ? oL : oR, r, mqR, nmR, left_is_a ? oR : oL, o, dbg, &score)) {
                    if (dbg) *dbg << "  -> Passed eval with score=" << score << "\n";      // :851-853
                    add_edge(cL, oL, cR, oR, true, qname, f);
                    has_supp = true;
                }
            }
        }
        const int mtid = R->mtid[i];
        if (!has_supp && o.enable_paired && (f & 0x1) && !(f & 0x8) && mtid >= 0 && mtid != tid &&
            tid >= 0) {   // :887-888 (tid < 0 on a mapped record would index target_name[-1] there: skipped here)
            if (seen_pairs.count(qname)) {                                     // :890-893
                consumed[tname[mtid]] += std::max(0, ref_len);
                continue;
            }
            Side s1{(f & 0x10) != 0, R_MIDDLE, R->pos[i] + 1, tlen[tid]};
            Side s2{(f & 0x20) != 0, R_MIDDLE, R->mpos[i] + 1, tlen[mtid]};
            s1.reg = region_of(s1.pos, s1.len, o);
            s2.reg = region_of(s2.pos, s2.len, o);
            if (s1.reg == R_MIDDLE || s2.reg == R_MIDDLE) continue;            // :910
            char oL = 0, oR = 0;
            bool first1 = true;
            for (int order = 0; order < 2 && !oL; order++) {                   // :916-934
                bool f1 = order == 0;
                const Side &l = f1 ? s1 : s2, &r = f1 ? s2 : s1;
                for (char a : {'+', '-'}) {
                    for (char b : {'+', '-'})
                        if (paired_layout_ok(l, r, a, b, o)) { oL = a; oR = b; first1 = f1; break; }
                    if (oL) break;
                }
            }
            if (!oL) continue;
            seen_pairs.insert(qname);                                          // :938
            const Side &l = first1 ? s1 : s2, &r = first1 ? s2 : s1;
            const std::string &cL = first1 ? tname[tid] : tname[mtid], &cR = first1 ? tname[mtid] : tname[tid];
            bool left_is_a = cL <= cR;
            if (score_positive(l, mapq, nm, left_is_a ? oL : oR, r, mapq, nm, left_is_a ? oR : oL, o, dbg))   // :950-951, :990 (its debug line: :454-458)
                add_edge(cL, oL, cR, oR, false, qname, f);
        }
    }

    // :1019-1076
    std::map<std::string, std::pair<double, int>> seg;
    for (int i = 0; i < n_targets; i++) {
        int L = tlen[i];
        if (L <= 0) continue;
        auto it = consumed.find(tname[i]);
        double c = it == consumed.end() ? 0.0 : it->second;
        double depth = c / std::max(1, L);
        double cnf = avg_depth > 0.0 ? depth / avg_depth : 0.0;
        seg[tname[i]] = {depth, (int)std::floor(cnf + 0.5)};
    }
    std::ostringstream os;
    for (auto &kv : seg) os << "SEG " << kv.first << " " << kv.second.first << " " << kv.second.second << "\n";
    for (auto &kv : agg) {
        const Agg &a = kv.second;
        int total = a.supp + a.span + a.supp_nf + a.span_nf;
        if (total == 0 || total < o.min_count) continue;
        os << "JUNC " << std::get<0>(kv.first) << " " << std::get<2>(kv.first) << " " << std::get<1>(kv.first) << " "
           << std::get<3>(kv.first) << " " << (a.supp + a.span + a.supp_nf) << " " << a.span_nf;
        if (oo->debug) {                                                      // :1068-1073
            os << " READS:";
            for (const auto &r : a.reads) os << " " << r.first << "(" << r.second << ")";
        }
        os << "\n";
    }
    if (trace) *trace = dbg_text.str();
    std::string s = os.str();
    if (s.size() > cap) return -1;
    std::memcpy(out, s.data(), s.size());
    return (long)s.size();
}

long orc_graph_run(const OrcRecords *R, int n_targets, const char *tnames, const int64_t *tname_off,
                   const int32_t *tlen, const char *fastg_fai, double avg_depth, const OrcGraphOpts *oo,
                   char *out, size_t cap)
{
    return graph_run(R, n_targets, tnames, tname_off, tlen, fastg_fai, avg_depth, oo, out, cap, nullptr);
}

// The graph text as orc_graph_run, and what --debug writes to stderr meanwhile into trace_out; returns the graph text's size, -1 when
// either buffer is too small.
long orc_graph_run_trace(const OrcRecords *R, int n_targets, const char *tnames, const int64_t *tname_off,
                         const int32_t *tlen, const char *fastg_fai, double avg_depth, const OrcGraphOpts *oo,
                         char *out, size_t cap, char *trace_out, size_t trace_cap, long *trace_len)
{
    std::string trace;
    long n = graph_run(R, n_targets, tnames, tname_off, tlen, fastg_fai, avg_depth, oo, out, cap, &trace);
    if (n < 0 || trace.size() > trace_cap) return -1;
    std::memcpy(trace_out, trace.data(), trace.size());
    *trace_len = (long)trace.size();
    return n;
}

}  // extern "C"

// ---- depth stage (palace:538-552): `samtools depth <bam> | awk '{sum+=$3} END { print sum/NR }'` ------------------------
// Restated from the samtools documentation (samtools is not in this image: parity UNPINNED).  samtools depth with default
// options (1.13 or later: no depth cap) prints one line per reference position whose depth is > 0; a record counts at the
// positions of its M, = and X operations (deletions and reference skips only with -J), unless one of UNMAP (0x4),
// SECONDARY (0x100), QCFAIL (0x200), DUP (0x400) is set.  Per-base counters, as the tool keeps them; the awk line then
// divides the sum of column 3 by the number of lines.  Writes the text awk prints into `out` (integral values are printed
// as integers, everything else with OFMT = "%.6g"); returns its length, or -1 when no line would exist (awk: division by zero).
extern "C" long orc_depth_mean(const OrcRecords *R, int n_targets, const int32_t *tlen, char *out, size_t cap,
                               uint64_t *sum_out, uint64_t *nr_out)
{
    std::vector<std::vector<uint32_t>> depth(static_cast<size_t>(n_targets));
    for (int64_t i = 0; i < R->n; i++) {
        if (R->flag[i] & 0x704) continue;
        const int32_t t = R->tid[i];
        if (t < 0 || t >= n_targets || R->pos[i] < 0) continue;
        auto &d = depth[static_cast<size_t>(t)];
        if (d.empty()) d.assign(static_cast<size_t>(std::max(0, tlen[t])), 0);
        int64_t p = R->pos[i];
        for (int64_t k = R->cigar_off[i]; k < R->cigar_off[i + 1]; k++) {
            const uint32_t op = R->cigar[k] & 15, len = R->cigar[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {                        // M, =, X
                for (int64_t q = p; q < p + len && q < static_cast<int64_t>(d.size()); q++) d[static_cast<size_t>(q)]++;
                p += len;
            } else if (op == 2 || op == 3) p += len;                     // D, N: reference advances, nothing counted
        }
    }
    uint64_t sum = 0, nr = 0;
    for (auto &d : depth)
        for (uint32_t v : d)
            if (v) { sum += v; nr++; }
    if (sum_out) *sum_out = sum;
    if (nr_out) *nr_out = nr;
    if (nr == 0) return -1;
    const double mean = static_cast<double>(sum) / static_cast<double>(nr);
    int n;
    if (mean == std::floor(mean) && std::fabs(mean) < 1e15) n = std::snprintf(out, cap, "%lld", static_cast<long long>(mean));
    else n = std::snprintf(out, cap, "%.6g", mean);
    return n;
}
