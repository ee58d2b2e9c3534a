// generateGraph --debug: the per-read text the reference writes to stderr while it walks the records (generate_graph.cpp:454-458 the
// score line, :607-609 the FASTG count, :711-717 / :746-750 the split-read and SA headers, :758-767 stitching, :789-797 the layout,
// :851-853 the accepted score).  A DIAGNOSTIC of the host side: the graph itself -- which evidence counts, the JUNC numbers -- is the
// device's (palace_graph_classify / _resolve) and does not pass through here; this walks the decoded records once more, on one
// thread, in file order, and prints what the reference prints for each.  It reads the records where the loader left them (the
// inflated stream: CIGAR and SA texts are not among the columns) and the columns for everything the loader has already derived.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <string>
#include <string_view>
#include <tuple>

#include "bam.hpp"
#include "textio.hpp"

namespace palace_host {

namespace dbgtrace {

enum Reg { START, END, MIDDLE };
inline const char *reg_name(Reg r) { return r == START ? "START" : r == END ? "END" : "MIDDLE"; }
inline Reg region(int pos1, int len, int max_end)                       // :56-62
{
    if (pos1 <= std::min(max_end, len / 2)) return START;
    if (pos1 > std::max(len - max_end, len / 2)) return END;
    return MIDDLE;
}
inline Reg flip(Reg r) { return r == START ? END : r == END ? START : MIDDLE; }

// the read interval a CIGAR text covers (:330-383): ops of length 0 do not exist; a clip counts at the text's first op and, when
// there is more than one op, at its last
struct Span { int start = 0, end = 0; };
inline Span span_of(std::string_view cigar, bool rev, int read_len)
{
    Span s;
    if (cigar.empty()) return s;                                          // (a text without a single op, "*" say, is NOT this case: [1, 0])
    int n = 0, n_ops = 0, first_len = 0, last_len = 0, total = 0;
    char first_op = 0, last_op = 0;
    for (char ch : cigar) {
        if (ch >= '0' && ch <= '9') { n = n * 10 + (ch - '0'); continue; }
        if (n > 0) {
            if (n_ops == 0) { first_op = ch; first_len = n; }
            last_op = ch; last_len = n;
            n_ops++;
            if (ch == 'M' || ch == 'I' || ch == 'S' || ch == '=' || ch == 'X') total += n;
        }
        n = 0;
    }
    const int clip_s = (n_ops > 0 && first_op == 'S') ? first_len : 0, clip_e = (n_ops > 1 && last_op == 'S') ? last_len : 0;
    if (rev && read_len > 0) { s.start = read_len - (total - clip_e) + 1; s.end = read_len - clip_s; }
    else { s.start = clip_s + 1; s.end = total - clip_e; }
    return s;
}

// :401-428 with both limits at 150
inline bool can_stitch(const Span &a, const Span &b, bool &first1)
{
    if (a.end <= b.start && b.start - a.end - 1 <= 150) { first1 = true; return true; }
    if (b.end <= a.start && a.start - b.end - 1 <= 150) { first1 = false; return true; }
    if (a.start <= b.end && b.start <= a.end && std::min(a.end, b.end) - std::max(a.start, b.start) + 1 <= 150) { first1 = a.start <= b.start; return true; }
    return false;
}

struct End { bool rev; Reg reg; int pos, len; };

inline void put(std::string &out, std::string_view s) { out.append(s.data(), s.size()); }
inline void put(std::string &out, long long v) { out += std::to_string(v); }
inline void put_g(std::string &out, double v)                          // ostream << double, default format
{
    char b[40];
    out.append(b, format_g6(v, b));
}

// computeLayoutScore (:432-461) with its debug line; orientations are the *_eval ones
inline double score_line(std::string &out, const End &l, int mapqL, int nmL, char oL, const End &r, int mapqR, int nmR, char oR, int max_end)
{
    const Reg gl = oL == '-' ? flip(l.reg) : l.reg, gr = oR == '-' ? flip(r.reg) : r.reg;
    const int dL = gl == START ? std::max(0, l.pos - 1) : std::max(0, l.len - l.pos);
    const int dR = gr == START ? std::max(0, r.pos - 1) : std::max(0, r.len - r.pos);
    const double lambda = std::max(50.0, static_cast<double>(max_end) / 2.0);
    const double w_end = std::exp(-static_cast<double>(dL) / lambda) * std::exp(-static_cast<double>(dR) / lambda);
    const double qL = std::min(1.0, static_cast<double>(mapqL) / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, nmL)));
    const double qR = std::min(1.0, static_cast<double>(mapqR) / 60.0) * (1.0 / (1.0 + 0.2 * std::max(0, nmR)));
    const double total = w_end * qL * qR;
    put(out, "Score calculation: w_end="); put_g(out, w_end);
    put(out, " w_qualL="); put_g(out, qL);
    put(out, " w_qualR="); put_g(out, qR);
    put(out, " total="); put_g(out, total);
    out += '\n';
    return total;
}

inline std::string_view trimmed(std::string_view s)
{
    while (!s.empty() && std::isspace(static_cast<unsigned char>(s.front()))) s.remove_prefix(1);
    while (!s.empty() && std::isspace(static_cast<unsigned char>(s.back()))) s.remove_suffix(1);
    return s;
}

// how many (name, name, orientation, orientation) entries parseFastgFile's set holds (:119-169): every link and its twin, of ANY
// names -- the set is built before the BAM header is read
inline size_t fastg_set_size(const std::string &path)
{
    std::set<std::tuple<std::string, std::string, char, char>> set;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        const std::string head = line.substr(0, line.find(';'));
        const size_t colon = head.find(':');
        std::string name = head.substr(0, colon);
        bool rev = false;
        if (!name.empty() && name.back() == '\'') { rev = true; name.pop_back(); }
        if (colon == std::string::npos) continue;
        size_t q = colon + 1;
        while (q <= head.size()) {
            const size_t comma = head.find(',', q);
            std::string lk = head.substr(q, comma == std::string::npos ? std::string::npos : comma - q);
            q = comma == std::string::npos ? head.size() + 1 : comma + 1;
            if (lk.empty()) continue;
            bool lrev = false;
            if (lk.back() == '\'') { lrev = true; lk.pop_back(); }
            const char o1 = rev ? '-' : '+', o2 = (rev != lrev) ? '-' : '+';
            set.emplace(name, lk, o1, o2);
            set.emplace(lk, name, o1 == '+' ? '-' : '+', o2 == '+' ? '-' : '+');
        }
    }
    return set.size();
}

}  // namespace dbgtrace

// The whole text, in the order the reference writes it.
inline std::string debug_trace(const BamColumns &c, const std::string &fastg_fai, const palace_graph_params &prm)
{
    using namespace dbgtrace;
    std::string out;
    put(out, "Loaded "); put(out, static_cast<long long>(fastg_set_size(fastg_fai))); put(out, " expected connections from FastG\n");
    const uint8_t *raw = c.raw.data();
    static const char opchr[] = "MIDNSHP=XB??????";
    const int32_t n_ref = static_cast<int32_t>(c.target_name.size());
    // read names that have given a pair evidence already (:635, :890-893, :938): exact on names
    std::set<std::string> paired_seen;
    auto pass = [&](int mapq, int nm) { return mapq >= prm.min_mapq && nm <= prm.max_nm; };
    for (int64_t i = 0; i < c.n(); i++) {
        const uint16_t flag = c.flag[i];
        if (flag & (0x800 | 0x100 | 0x4)) continue;                           // :647-649
        const int mapq = c.mapq[i], nm = c.nm[i];
        if (!pass(mapq, nm)) continue;                                        // :679
        const int32_t tid = c.tid[i];
        if (tid >= n_ref) continue;
        const uint8_t *name = raw + c.qname_at[i], *r = name - 32;            // the record where the loader found it
        const uint8_t *rec_end = r + (static_cast<uint32_t>(r[-4]) | static_cast<uint32_t>(r[-3]) << 8 | static_cast<uint32_t>(r[-2]) << 16 | static_cast<uint32_t>(r[-1]) << 24);
        const size_t l_name = r[8], n_cig = static_cast<size_t>(r[12]) | static_cast<size_t>(r[13]) << 8;
        const size_t l_seq = static_cast<size_t>(r[16]) | static_cast<size_t>(r[17]) << 8 | static_cast<size_t>(r[18]) << 16 | static_cast<size_t>(r[19]) << 24;
        const uint8_t *cg = name + l_name, *aux = cg + 4 * n_cig + (l_seq + 1) / 2 + l_seq;
        auto u32 = [](const uint8_t *p) { return static_cast<uint32_t>(p[0]) | static_cast<uint32_t>(p[1]) << 8 | static_cast<uint32_t>(p[2]) << 16 | static_cast<uint32_t>(p[3]) << 24; };
        // first SA:Z and (for CIGARs of more than 65535 ops, which htslib puts back in place) first CG:B,I
        const char *sa = nullptr;
        size_t sa_len = 0;
        const uint8_t *ops = cg;
        size_t n_ops = n_cig;
        const bool placeholder = n_cig > 0 && tid >= 0 && c.pos[i] >= 0 && (u32(cg) & 15) == 4 && (u32(cg) >> 4) == l_seq;
        bool have_sa = false, have_cg = false;
        for (const uint8_t *x = aux; x + 3 <= rec_end && !(have_sa && (have_cg || !placeholder));) {
            const uint8_t ty = x[2];
            const uint8_t *v = x + 3;
            size_t sz = 0;
            switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const void *z = std::memchr(v, 0, static_cast<size_t>(rec_end - v)); sz = z ? static_cast<const uint8_t *>(z) - v + 1 : 0; break; }
            case 'B': {
                if (v + 5 > rec_end) break;
                const size_t el = (v[0] == 'c' || v[0] == 'C') ? 1 : (v[0] == 's' || v[0] == 'S') ? 2 : (v[0] == 'i' || v[0] == 'I' || v[0] == 'f') ? 4 : 0;
                sz = el ? 5 + el * u32(v + 1) : 0;
                break;
            }
            default: sz = 0;
            }
            if (!sz || sz > static_cast<size_t>(rec_end - v)) break;
            if (!have_sa && x[0] == 'S' && x[1] == 'A' && ty == 'Z') { have_sa = true; sa = reinterpret_cast<const char *>(v); sa_len = sz - 1; }
            if (!have_cg && x[0] == 'C' && x[1] == 'G') {
                have_cg = true;
                if (placeholder && ty == 'B' && (v[0] == 'I' || v[0] == 'i') && u32(v + 1) >= n_cig && u32(v + 1) < (1u << 29)) { ops = v + 5; n_ops = u32(v + 1); }
            }
            x = v + sz;
        }
        const std::string qname = c.qname(i);
        const int read_len = c.read_len[i];
        bool has_split = false;
        if (have_sa && tid >= 0) {                                             // :687
            const std::string &r1 = c.target_name[static_cast<size_t>(tid)];
            End e1{(flag & 0x10) != 0, MIDDLE, c.pos[i] + 1, c.target_len[static_cast<size_t>(tid)]};
            e1.reg = region(e1.pos, e1.len, prm.max_end);
            std::string cigar1;
            for (size_t k = 0; k < n_ops; k++) { const uint32_t v = u32(ops + 4 * k); cigar1 += std::to_string(v >> 4); cigar1 += opchr[v & 15]; }
            const Span s1 = span_of(cigar1, e1.rev, read_len);
            put(out, "\n=== Split-read: "); put(out, qname); put(out, " (len="); put(out, read_len); put(out, ") ===\n");
            put(out, "Primary: "); put(out, r1); put(out, " pos="); put(out, e1.pos); put(out, " rev="); put(out, e1.rev ? 1 : 0);
            put(out, " region="); put(out, reg_name(e1.reg)); put(out, " read["); put(out, s1.start); put(out, "-"); put(out, s1.end);
            put(out, "] CIGAR="); put(out, cigar1); out += '\n';
            // the C string the reference reads the tag as ends at its first NUL: sa_len is that; items split at ';' (:719)
            for (size_t p = 0; p <= sa_len;) {
                const void *semi = p < sa_len ? std::memchr(sa + p, ';', sa_len - p) : nullptr;
                const size_t e = semi ? static_cast<size_t>(static_cast<const char *>(semi) - sa) : sa_len;
                const std::string_view item(sa + p, e - p);
                p = e + 1;
                if (item.empty()) continue;
                // parseSAItem (:185-206): six comma-separated fields, trimmed; the first two non-empty
                // (a field is what std::getline hands out: an empty one between two commas exists, one behind the item's last comma does not)
                std::string_view f[6];
                size_t q = 0;
                int got = 0;
                while (got < 6 && q < item.size()) {
                    const size_t comma = item.find(',', q);
                    f[got++] = trimmed(item.substr(q, comma == std::string_view::npos ? std::string_view::npos : comma - q));
                    if (comma == std::string_view::npos) break;
                    q = comma + 1;
                }
                if (got < 6 || f[0].empty() || f[1].empty()) continue;
                const std::string a1(f[1]), a4(f[4]), a5(f[5]);
                const int pos2 = std::atoi(a1.c_str()), mapq2 = std::atoi(a4.c_str()), nm2 = std::atoi(a5.c_str());
                const bool rev2 = f[2] == "-";
                if (!pass(mapq2, nm2)) continue;
                if (f[0] == std::string_view(r1)) continue;                    // :733
                const int32_t tid2 = c.tid_of(f[0]);
                if (tid2 < 0) continue;                                        // :735-737
                End e2{rev2, MIDDLE, pos2, c.target_len[static_cast<size_t>(tid2)]};
                e2.reg = region(e2.pos, e2.len, prm.max_end);
                if (e1.reg == MIDDLE || e2.reg == MIDDLE) continue;            // :742
                const Span s2 = span_of(f[3], rev2, read_len);
                put(out, "SA: "); put(out, f[0]); put(out, " pos="); put(out, pos2); put(out, " rev="); put(out, rev2 ? 1 : 0);
                put(out, " region="); put(out, reg_name(e2.reg)); put(out, " read["); put(out, s2.start); put(out, "-"); put(out, s2.end);
                put(out, "] CIGAR="); put(out, f[3]); out += '\n';
                bool first1 = false;
                if (!can_stitch(s1, s2, first1)) { put(out, "  -> Cannot stitch: intervals too far apart or too much overlap\n"); continue; }
                put(out, first1 ? "  -> Can stitch! Primary first\n" : "  -> Can stitch! SA first\n");
                const End &l = first1 ? e1 : e2, &rr = first1 ? e2 : e1;
                // checkSplitReadLayout (:510-538): both segments read forward in the layout, the left one at the left contig's
                // right end, the right one at the right contig's left end; first of ++, +-, -+, --
                char oL = 0, oR = 0;
                for (int k = 0; k < 4 && !oL; k++) {
                    const char a = (k & 2) ? '-' : '+', b = (k & 1) ? '-' : '+';
                    const bool fwdL = a == '-' ? l.rev : !l.rev, fwdR = b == '-' ? rr.rev : !rr.rev;
                    if (fwdL && fwdR && l.reg == (a == '+' ? END : START) && rr.reg == (b == '+' ? START : END)) { oL = a; oR = b; }
                }
                if (!oL) { put(out, "  -> No valid layout found\n"); continue; }
                const std::string_view cL = first1 ? std::string_view(r1) : f[0], cR = first1 ? f[0] : std::string_view(r1);
                put(out, "  -> Found valid layout: "); put(out, cL); out += '('; out += oL; put(out, ") -> "); put(out, cR); out += '('; out += oR; put(out, ")\n");
                const bool left_is_a = cL <= cR;                                // :802, :846-848
                const double score = score_line(out, l, first1 ? mapq : mapq2, first1 ? nm : nm2, left_is_a ? oL : oR,
                                                rr, first1 ? mapq2 : mapq, first1 ? nm2 : nm, left_is_a ? oR : oL, prm.max_end);
                if (score > 0.0) {
                    put(out, "  -> Passed eval with score="); put_g(out, score); out += '\n';
                    has_split = true;
                }
            }
        }
        // the paired branch prints nothing of its own, but every layout it scores goes through computeLayoutScore (:990 -> :454-458)
        const int32_t mtid = c.mtid[i];
        if (has_split || !prm.enable_paired || !(flag & 0x1) || (flag & 0x8) || mtid < 0 || mtid == tid || tid < 0 || mtid >= n_ref) continue;
        if (paired_seen.count(qname)) continue;                                // :890-893
        End e1{(flag & 0x10) != 0, MIDDLE, c.pos[i] + 1, c.target_len[static_cast<size_t>(tid)]};
        End e2{(flag & 0x20) != 0, MIDDLE, c.mpos[i] + 1, c.target_len[static_cast<size_t>(mtid)]};
        e1.reg = region(e1.pos, e1.len, prm.max_end);
        e2.reg = region(e2.pos, e2.len, prm.max_end);
        if (e1.reg == MIDDLE || e2.reg == MIDDLE) continue;                    // :910
        char oL = 0, oR = 0;
        bool first1 = true;
        for (int order = 0; order < 2 && !oL; order++) {                       // :916-934: this read on the left first, then its mate
            const End &l = order == 0 ? e1 : e2, &rr = order == 0 ? e2 : e1;
            for (int k = 0; k < 4 && !oL; k++) {
                const char a = (k & 2) ? '-' : '+', b = (k & 1) ? '-' : '+';
                const bool fwdL = a == '-' ? l.rev : !l.rev, fwdR = b == '-' ? rr.rev : !rr.rev;
                if (!fwdL || fwdR) continue;                                    // left read forward, right read reverse in the layout (:465-506)
                if (l.reg != (a == '+' ? END : START) || rr.reg != (b == '+' ? START : END)) continue;
                const int dL = l.reg == START ? std::max(0, l.pos - 1) : std::max(0, l.len - l.pos);
                const int dR = rr.reg == START ? std::max(0, rr.pos - 1) : std::max(0, rr.len - rr.pos);
                const double fL = l.len > 0 ? static_cast<double>(dL) / l.len : 1.0, fR = rr.len > 0 ? static_cast<double>(dR) / rr.len : 1.0;
                if (fL > prm.max_span_frac || fR > prm.max_span_frac) continue;
                oL = a; oR = b; first1 = order == 0;
            }
        }
        if (!oL) continue;
        paired_seen.insert(qname);                                             // :938
        const End &l = first1 ? e1 : e2, &rr = first1 ? e2 : e1;
        const std::string &cL = c.target_name[static_cast<size_t>(first1 ? tid : mtid)], &cR = c.target_name[static_cast<size_t>(first1 ? mtid : tid)];
        const bool left_is_a = cL <= cR;
        score_line(out, l, mapq, nm, left_is_a ? oL : oR, rr, mapq, nm, left_is_a ? oR : oL, prm.max_end);      // (the mate's mapq / NM are this read's, :950-951)
    }
    return out;
}

}  // namespace palace_host
