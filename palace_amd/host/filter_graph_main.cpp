// filter_graph -- native core of palace_amd/scripts/filter_graph.py (same 13 positional arguments, same two output files,
// byte for byte the script's output; the script hands over to this executable when it is built).
//
// Counterpart of the reference's share/palace/scripts/filter_graph.py (call site palace:568-579).  At the 1M-contig
// configuration the script is 60 % of the files -> files time of the whole path (4.9 s of pure-Python line handling);
// this is the same selection over mapped files with interned names.  No GPU work: the stage is a few hash look-ups per
// line.  Line citations below are to the reference script.
//
//   filter_graph fastg.fai graph.txt out.txt depth f_th hit_seqs.out node_scores.out contigs.blast blast_ratio
//                contigs.fasta.fai all_hit_segs.txt contigs.paths score_threshold
//
// Inputs the script would die on with a Python exception (a junction or path naming an unknown contig, a short line)
// end this program with exit code 1 and a message instead.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <algorithm>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "textio.hpp"
#include "trace.hpp"

namespace {

using palace_host::sv;
using palace_host::strip;
using palace_host::rstrip;
using palace_host::split_on;
using palace_host::split_ws;
using palace_host::Names;

[[noreturn]] void die(const std::string &what)
{
    std::fprintf(stderr, "filter_graph: %s\n", what.c_str());
    std::fflush(stderr);
    _exit(1);                                      // (may be called from a parser thread: no static destructors beside running threads)
}

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    explicit Mapped(const char *path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) die(std::string("cannot open ") + path);
        struct stat st{};
        if (fstat(fd, &st) != 0) die(std::string("cannot stat ") + path);
        n = static_cast<size_t>(st.st_size);
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) die(std::string("cannot map ") + path);
            p = static_cast<const char *>(m);
        }
        ::close(fd);
    }
    // calls f(line) for every line, the line INCLUDING its '\n' when it has one (Python's iteration over a file)
    template <class F>
    void lines(F f) const
    {
        size_t a = 0;
        while (a < n) {
            const void *e = std::memchr(p + a, '\n', n - a);
            const size_t b = e ? static_cast<size_t>(static_cast<const char *>(e) - p) + 1 : n;
            f(sv(p + a, b - a));
            a = b;
        }
    }
};

bool has_e(sv s) { return s.find('e') != sv::npos || s.find('E') != sv::npos; }

// float(text): the whole (stripped) token must be a number
bool to_double(sv tok, double &v)
{
    const std::string t(strip(tok));
    if (t.empty() || t.find('x') != std::string::npos || t.find('X') != std::string::npos) return false;
    char *end = nullptr;
    v = std::strtod(t.c_str(), &end);
    return end == t.c_str() + t.size();
}
double need_double(sv tok, const char *what)
{
    double v;
    if (!to_double(tok, v)) die(std::string("not a number in ") + what + ": '" + std::string(tok) + "'");
    return v;
}
long long need_int(sv tok, const char *what)
{
    const sv t = strip(tok);
    size_t i = 0;
    bool neg = false;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
    long long v = 0;
    const size_t first = i;
    for (; i < t.size() && t[i] >= '0' && t[i] <= '9'; i++) v = v * 10 + (t[i] - '0');
    if (i == first || i != t.size() || i - first > 18) die(std::string("not an integer in ") + what + ": '" + std::string(t) + "'");
    return neg ? -v : v;
}
std::string fixed3(double v)
{
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.3f", v);
    return buf;
}
sv field(const std::vector<sv> &cols, size_t i, const char *what)
{
    if (i >= cols.size()) die(std::string("short line in ") + what);
    return cols[i];
}
// fields written in scientific notation become plain (l.178-188)
std::string plain_number(sv tok)
{
    double v;
    if (!has_e(tok) || !to_double(tok, v)) return std::string(tok);
    char buf[400];
    if (std::isfinite(v) && v == std::floor(v)) {
        std::snprintf(buf, sizeof buf, "%.0f", v);
        if (buf[0] == '-' && buf[1] == '0' && buf[2] == 0) return "0";      // str(int(-0.0))
        return buf;
    }
    std::string s = fixed3(v);
    while (!s.empty() && s.back() == '0') s.pop_back();
    if (!s.empty() && s.back() == '.') s.pop_back();
    return s;
}

struct Facts {                                     // per name id
    long long length = -1;                         // fasta .fai
    bool blast = false, score_hit = false, has_score = false, gene = false;
    std::string score_text;                        // "%.3f" text (or "0.0")
    double score = 0;                              // float(score_text), 0 without a score
    sv raw;                                        // the name's latest SEG line of the graph
    bool has_raw = false, seed = false, near = false, in_already = false;
    size_t text_at = 0, text_len = 0;              // the name's first SEG text in the output block
    const char *text_of = nullptr;                 // the SEG line the latest text was made from
    std::vector<std::string> more_texts;           // further, different texts of the same name (duplicate SEG lines)
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 14) {
        std::fprintf(stderr, "usage: filter_graph fastg.fai graph.txt out.txt depth f_th hit_seqs.out node_scores.out contigs.blast "
                             "blast_ratio contigs.fasta.fai all_hit_segs.txt contigs.paths score_threshold\n");
        return 2;
    }
    const char *fastg_fai = argv[1], *graph_path = argv[2], *out_path = argv[3], *gene_file = argv[6], *score_file = argv[7],
               *blast_file = argv[8], *fasta_fai = argv[10], *hit_segs_path = argv[11], *paths_file = argv[12];
    (void)need_double(argv[4], "<depth>");                                        // parsed, unused (l.11)
    const double blast_ratio = need_double(argv[9], "<blast_ratio>"), score_threshold = need_double(argv[13], "<score_threshold>");

    palace_host::Trace trace("filter_graph");
    // Every large file is parsed in parts on threads (line splitting, number parsing, name hashing and look-up: read-only
    // work) and the results are applied in file order by this thread, so the outcome does not depend on the thread count.
    size_t n_threads = std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    bool forced = false;                                                          // (tests: small files in several parts)
    if (const char *e = std::getenv("PALACE_HOST_THREADS")) { n_threads = static_cast<size_t>(std::max(1, std::atoi(e))); forced = true; }
    auto cuts_of = [&](const Mapped &m) { return palace_host::line_cuts(m.p, m.n, (m.n < (1u << 20) && !forced) ? 1 : n_threads); };
    Names names;
    std::vector<Facts> facts;
    auto grow = [&](int id) { if (static_cast<size_t>(id) >= facts.size()) facts.resize(static_cast<size_t>(id) + 1); return id; };
    auto id_of = [&](sv s) { return grow(names.intern(s)); };

    // contigs.fasta.fai: lengths, and the id token -> name map contigs.paths is read with
    Mapped fai(fasta_fai);
    Names tokens;                                  // id token of a name (EDGE_<token>_...) -> name id
    std::vector<int> token_name;
    {
        struct Row { sv name, token; uint64_t h_name, h_token; long long len; };
        const std::vector<size_t> cut = cuts_of(fai);
        std::vector<std::vector<Row>> rows(cut.size() - 1);
        palace_host::for_parts(cut, [&](size_t k, size_t a, size_t b) {
            std::vector<sv> cols, parts;
            palace_host::for_each_line(fai.p + a, b - a, [&](sv line) {
                split_on(strip(line), '\t', cols);
                split_on(cols[0], '_', parts);
                const sv token = field(parts, 1, "a fasta index name");
                rows[k].push_back(Row{cols[0], token, palace_host::hash_bytes(cols[0]), palace_host::hash_bytes(token),
                                      need_int(field(cols, 1, "the fasta index"), "the fasta index")});
            });
        });
        size_t n_lines = 0;
        for (const auto &r : rows) n_lines += r.size();
        names.reserve(n_lines + 16);
        tokens.reserve(n_lines + 16);
        facts.reserve(n_lines + 16);
        for (const auto &part : rows)
            for (size_t i = 0; i < part.size(); i++) {
                const Row &r = part[i];
                if (i + 8 < part.size()) { names.prefetch(part[i + 8].h_name); tokens.prefetch(part[i + 8].h_token); }
                const int id = grow(names.intern_hashed(r.name, r.h_name));
                facts[static_cast<size_t>(id)].length = r.len;
                const int t = tokens.intern_hashed(r.token, r.h_token);
                if (static_cast<size_t>(t) >= token_name.size()) token_name.resize(static_cast<size_t>(t) + 1);
                token_name[static_cast<size_t>(t)] = id;                              // (a later line with the same token wins, as in a dict)
            }
    }
    trace.lap("fasta index");

    // contigs.blast: consecutive rows of one (query, subject) pair form a group (l.66-94)
    Mapped blast(blast_file);
    {
        std::vector<sv> cols;
        sv cur_q, cur_s;
        long long aligned = 0;
        const double cut = blast_ratio * 100;
        auto group_done = [&](sv q) {
            const int id = names.find(q);
            if (id < 0 || facts[static_cast<size_t>(id)].length < 0) die("BLAST query not in the fasta index: " + std::string(q));
            const long long len = facts[static_cast<size_t>(id)].length;
            if (len == 0) die("contig of length 0 in the fasta index: " + std::string(q));
            if (static_cast<double>(aligned) / static_cast<double>(len) > blast_ratio || aligned > 2000) facts[static_cast<size_t>(id)].blast = true;
        };
        blast.lines([&](sv line) {
            split_on(strip(line), '\t', cols);
            const sv q = field(cols, 0, "the BLAST table"), s = field(cols, 1, "the BLAST table");
            const double ident = need_double(field(cols, 2, "the BLAST table"), "the BLAST table");
            const long long alen = need_int(field(cols, 3, "the BLAST table"), "the BLAST table");
            const bool new_group = (cur_q != q && !cur_q.empty()) || (cur_s != s && !cur_s.empty());
            if (new_group) {
                group_done(cur_q);
                aligned = ident > cut ? alen : 0;
            } else if (ident > cut) {
                aligned += alen;
            }
            cur_q = q; cur_s = s;
        });
        if (!cur_q.empty()) {
            const int id = names.find(cur_q);
            if (id >= 0 && facts[static_cast<size_t>(id)].length >= 0) group_done(cur_q);
        }
    }
    trace.lap("blast table");

    // hit_seqs.out: first column, untrimmed (l.101)
    Mapped genes(gene_file);
    genes.lines([&](sv line) {
        const size_t t = line.find('\t');
        facts[static_cast<size_t>(id_of(t == sv::npos ? line : line.substr(0, t)))].gene = true;
    });

    // node_scores.out (l.104-112)
    Mapped scores(score_file);
    {
        struct Row { sv name; uint64_t h; double score; char text[24]; };      // "%.3f" of anything below 1e19 fits
        const std::vector<size_t> cut = cuts_of(scores);
        std::vector<std::vector<Row>> rows(cut.size() - 1);
        std::vector<std::vector<std::string>> long_text(cut.size() - 1);        // (texts that do not fit: row.text[0] == 0)
        palace_host::for_parts(cut, [&](size_t k, size_t a, size_t b) {
            std::vector<sv> cols;
            palace_host::for_each_line(scores.p + a, b - a, [&](sv line) {
                split_on(strip(line), '\t', cols);
                const sv val = field(cols, 1, "the score table");
                const std::string text = has_e(val) ? std::string("0.0") : fixed3(need_double(val, "the score table"));
                Row r{cols[0], palace_host::hash_bytes(cols[0]), std::strtod(text.c_str(), nullptr), {0}};
                if (text.size() < sizeof r.text) std::memcpy(r.text, text.c_str(), text.size() + 1);
                else long_text[k].push_back(text);
                rows[k].push_back(r);
            });
        });
        for (size_t k = 0; k < rows.size(); k++) {
            size_t next_long = 0;
            for (size_t i = 0; i < rows[k].size(); i++) {
                const Row &r = rows[k][i];
                if (i + 8 < rows[k].size()) names.prefetch(rows[k][i + 8].h);
                Facts &f = facts[static_cast<size_t>(grow(names.intern_hashed(r.name, r.h)))];
                f.score_text = r.text[0] ? std::string(r.text) : long_text[k][next_long++];
                f.has_score = true;
                f.score = r.score;
                f.score_hit = f.score > score_threshold;
            }
        }
    }
    trace.lap("gene hits, scores");
    { Mapped unused(fastg_fai); }                                                 // opened like the reference does (l.114-120)
    // contigs.paths rescue (l.126-151) needs the facts the tables above gave (BLAST / gene / score flags, name lengths) and nothing of
    // the graph: the parts work out which lines pass and list their members on threads of their own while the graph passes below go
    // on.  It reads snapshots (the graph passes add names and facts), and keeps its first complaint for the place the script makes it.
    std::vector<uint8_t> backs(facts.size());
    for (size_t i = 0; i < facts.size(); i++) backs[i] = facts[i].blast || facts[i].gene || facts[i].score_hit;
    const std::vector<sv> names_then(names.names.begin(), names.names.end());
    std::unique_ptr<Mapped> paths;
    std::vector<std::vector<int>> passing;                                        // members of the passing lines of a part, in order
    std::string rescue_err;
    std::thread t_rescue([&] {
        if (::access(paths_file, R_OK) != 0) { rescue_err = std::string("cannot open ") + paths_file; return; }
        paths.reset(new Mapped(paths_file));
        const std::vector<size_t> cut = cuts_of(*paths);
        passing.resize(cut.size() - 1);
        std::vector<std::string> errs(cut.size() - 1);
        palace_host::for_parts(cut, [&](size_t k, size_t a, size_t b) {
            std::vector<sv> cols;
            std::string clean;
            std::vector<int> members;
            palace_host::for_each_line(paths->p + a, b - a, [&](sv raw_line) {
                if (!errs[k].empty()) return;
                const sv s = strip(raw_line);
                clean.clear();
                for (char c : s) if (c != ';') clean += c;
                if (sv(clean).substr(0, 4) == "NODE") return;
                split_on(clean, ',', cols);
                members.clear();
                long long total = 0, backed = 0;
                for (sv tok : cols) {
                    const sv key = tok.empty() ? tok : tok.substr(0, tok.size() - 1);
                    const int t = tokens.find(key);
                    if (t < 0) { errs[k] = "contigs.paths names an unknown contig id: '" + std::string(key) + "'"; return; }
                    const int m = token_name[static_cast<size_t>(t)];
                    members.push_back(m);
                    const sv name = names_then[static_cast<size_t>(m)];
                    size_t at = 0;                                                // name_length(), complaining instead of dying
                    for (int u = 0; u < 3 && at != sv::npos; u++) { at = name.find('_', at); if (at != sv::npos) at++; }
                    if (at == sv::npos) { errs[k] = "contig name without a length token: " + std::string(name); return; }
                    const size_t e = name.find('_', at);
                    const sv digits = strip(name.substr(at, e == sv::npos ? sv::npos : e - at));
                    long long len = 0;
                    size_t i = 0;
                    bool neg = false;
                    if (i < digits.size() && (digits[i] == '+' || digits[i] == '-')) neg = digits[i++] == '-';
                    const size_t first = i;
                    for (; i < digits.size() && digits[i] >= '0' && digits[i] <= '9'; i++) len = len * 10 + (digits[i] - '0');
                    if (i == first || i != digits.size() || i - first > 18) { errs[k] = "not an integer in a contig name: '" + std::string(digits) + "'"; return; }
                    if (neg) len = -len;
                    total += len;
                    if (backs[static_cast<size_t>(m)]) backed += len;
                }
                if (backed > 0 && (static_cast<double>(backed) / static_cast<double>(total) >= 0.5 || backed > 2000))
                    passing[k].insert(passing[k].end(), members.begin(), members.end());
            });
        });
        for (const std::string &e : errs) if (!e.empty()) { rescue_err = e; break; }
    });

    Mapped graph(graph_path);
    std::string seg_block;                                                        // SEG texts as they are selected ...
    struct Piece { const char *src; size_t at, len; };                           // ... written in the order of the SEG lines they
    std::vector<Piece> pieces;                                                    // were made from (src), as the script sorts them
    auto flags_of = [&](const Facts &f) {
        std::string s;
        if (f.blast) s += "ref+";
        if (f.score > score_threshold) s += "score+";
        if (f.gene) s += "gene+";
        return s;
    };
    // the SEG text of a name whose latest SEG line is `raw` (l.190-197)
    auto render = [&](const Facts &f, sv raw, std::vector<sv> &toks, std::string &text) {
        split_ws(raw, toks);
        text.clear();
        for (size_t i = 0; i < toks.size(); i++) {
            if (i) text += ' ';
            if (i < 2) text.append(toks[i]); else text += plain_number(toks[i]);
        }
        text += ' ';
        text += f.gene ? "1" : "0";
        text += ' ';
        text += f.has_score ? f.score_text : std::string("0.000");
        text += f.blast ? " 1\n" : " 0\n";
    };
    auto emit = [&](Facts &f, sv text, const char *src) {     // select() of the script: a text is written once
        if (f.text_len) {
            if (sv(seg_block).substr(f.text_at, f.text_len) == text) return;
            for (const std::string &e : f.more_texts)
                if (e == text) return;
            f.more_texts.emplace_back(text);
        } else {
            f.text_at = seg_block.size();
            f.text_len = text.size();
        }
        pieces.push_back(Piece{src, seg_block.size(), text.size()});
        seg_block.append(text);
        // `already` of the script: the second space-separated token of every selected text
        const size_t a = text.find(' ');
        if (a != sv::npos) {
            const size_t b = text.find(' ', a + 1);
            const sv second = text.substr(a + 1, (b == sv::npos ? text.size() : b) - a - 1);
            const size_t own = static_cast<size_t>(&f - facts.data());                 // (as a rule the text's second token is the segment's own name: no look-up)
            if (own < names.names.size() && names.names[own] == second) f.in_already = true;
            else {
                const int t = names.find(second);
                if (t >= 0) facts[static_cast<size_t>(t)].in_already = true;
            }
        }
    };
    std::string text;
    std::vector<sv> toks;
    auto select = [&](int id) {
        Facts &f = facts[static_cast<size_t>(id)];
        if (!f.has_raw) die("junction names a contig without a SEG line: " + std::string(names.names[static_cast<size_t>(id)]));
        if (f.text_of == f.raw.data()) return;                // the text of this very line is out already
        f.text_of = f.raw.data();
        render(f, f.raw, toks, text);
        emit(f, text, f.raw.data());
    };

    struct End { sv line; int left, right; };
    std::vector<End> ends;
    std::vector<std::pair<int, std::string>> hit_rows;       // first position of every name that has flags
    {
        // pass 1: SEG lines, seeds; junction ends.  The parts resolve names and, for seeds, render the text (every fact a
        // text depends on is final by now); names the graph alone knows (id -1) are interned when the part is applied.
        struct Rec { sv line, a, b; int ia, ib; bool seg; uint32_t text_at, text_len; };
        const std::vector<size_t> cut = cuts_of(graph);
        std::vector<std::vector<Rec>> recs(cut.size() - 1);
        std::vector<std::string> texts(cut.size() - 1);
        const size_t known = facts.size();
        palace_host::for_parts(cut, [&](size_t k, size_t a, size_t b) {
            std::vector<sv> cols, tk;
            std::string one;
            palace_host::for_each_line(graph.p + a, b - a, [&](sv line) {
                split_on(rstrip(line), ' ', cols);
                if (cols[0] != "SEG") {
                    const sv l = field(cols, 1, "a JUNC line"), r = field(cols, 3, "a JUNC line");
                    recs[k].push_back(Rec{line, l, r, names.find(l), names.find(r), false, 0, 0});
                    return;
                }
                const sv name = field(cols, 1, "a SEG line");
                Rec rec{line, name, sv(), names.find(name), -1, true, 0, 0};
                if (rec.ia >= 0 && static_cast<size_t>(rec.ia) < known) {
                    const Facts &f = facts[static_cast<size_t>(rec.ia)];
                    if (f.blast || f.score > score_threshold || f.gene) {
                        render(f, line, tk, one);
                        rec.text_at = static_cast<uint32_t>(texts[k].size());
                        rec.text_len = static_cast<uint32_t>(one.size());
                        texts[k] += one;
                    }
                }
                recs[k].push_back(rec);
            });
        });
        std::vector<char> hit_listed;
        for (size_t k = 0; k < recs.size(); k++)
            for (size_t i = 0; i < recs[k].size(); i++) {
                const Rec &r = recs[k][i];
                if (i + 8 < recs[k].size() && recs[k][i + 8].ia >= 0) __builtin_prefetch(&facts[static_cast<size_t>(recs[k][i + 8].ia)]);
                if (!r.seg) {
                    const int l = r.ia >= 0 ? r.ia : id_of(r.a), rr = r.ib >= 0 ? r.ib : id_of(r.b);
                    ends.push_back(End{r.line, l, rr});
                    continue;
                }
                const int id = r.ia >= 0 ? r.ia : id_of(r.a);
                Facts &f = facts[static_cast<size_t>(id)];
                f.raw = r.line;
                f.has_raw = true;
                if (r.text_len) {                             // a seed
                    f.seed = true;
                    f.text_of = r.line.data();
                    emit(f, sv(texts[k]).substr(r.text_at, r.text_len), r.line.data());
                    if (hit_listed.size() < facts.size()) hit_listed.resize(facts.size(), 0);
                    if (!hit_listed[static_cast<size_t>(id)]) { hit_listed[static_cast<size_t>(id)] = 1; hit_rows.emplace_back(id, flags_of(f)); }
                }
            }
    }
    trace.lap("graph pass 1");
    // (the script collects the junction lines after pass 1; a junction in front of its SEG lines is therefore fine, and
    //  select() below sees every name's LAST SEG line, as the script's raw_seg does)
    std::vector<sv> juncs;
    for (const End &e : ends) {                               // pass 2: junctions touching a seed, and self loops
        if (e.left == e.right || facts[static_cast<size_t>(e.left)].seed || facts[static_cast<size_t>(e.right)].seed) {
            juncs.push_back(e.line);
            select(e.left);
            select(e.right);
            facts[static_cast<size_t>(e.left)].near = facts[static_cast<size_t>(e.right)].near = true;
        }
    }
    for (Facts &f : facts) f.near = f.near || f.seed;
    for (const End &e : ends) {                               // pass 3: junctions touching seeds or their neighbours
        if (facts[static_cast<size_t>(e.left)].near || facts[static_cast<size_t>(e.right)].near) {
            juncs.push_back(e.line);
            select(e.left);
            select(e.right);
        }
    }
    trace.lap("graph passes 2, 3");

    // contigs.paths rescue (l.126-151): worked out beside the graph passes (t_rescue, started in front of pass 1); its complaints are
    // made here, where the script would make them
    t_rescue.join();
    if (!rescue_err.empty()) die(rescue_err);
    std::vector<int> rescued;
    {
        std::vector<char> seen(facts.size(), 0);
        for (const auto &part : passing)
            for (int m : part)
                if (!seen[static_cast<size_t>(m)]) { seen[static_cast<size_t>(m)] = 1; rescued.push_back(m); }
    }

    trace.lap("paths rescue");
    // all_hit_segs.txt is written by a thread of its own while this one writes the graph (its complaints are made behind the graph's)
    std::string hits_err;
    std::thread t_hits([&] {
        FILE *hits = std::fopen(hit_segs_path, "wb");
        if (!hits) { hits_err = std::string("cannot write ") + hit_segs_path; return; }
        {
            std::string block;
            block.reserve(hit_rows.size() * 56);
            for (const auto &row : hit_rows) {
                block += "SAMPLE\t";
                block.append(names.names[static_cast<size_t>(row.first)]);
                block += '\t';
                block += row.second;
                block += '\n';
            }
            std::fwrite(block.data(), 1, block.size(), hits);
        }
        if (std::fclose(hits) != 0) hits_err = std::string("write failed: ") + hit_segs_path;
    });
    FILE *out = std::fopen(out_path, "wb");
    if (!out) die(std::string("cannot write ") + out_path);
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &a, const Piece &b) { return a.src < b.src; });
    {
        std::string ordered;                                  // (one write instead of one per SEG line)
        ordered.reserve(seg_block.size());
        for (const Piece &pc : pieces) ordered.append(seg_block, pc.at, pc.len);
        std::fwrite(ordered.data(), 1, ordered.size(), out);
    }
    std::vector<sv> tail;                                     // the rescued segments' SEG lines, in the order of the graph file
    for (int m : rescued) {
        const Facts &f = facts[static_cast<size_t>(m)];
        if (f.in_already) continue;
        if (!f.has_raw) die("contigs.paths names a contig without a SEG line: " + std::string(names.names[static_cast<size_t>(m)]));
        tail.push_back(f.raw);
    }
    std::sort(tail.begin(), tail.end(), [](sv a, sv b) { return a.data() < b.data(); });
    {
        std::string block;
        block.reserve(tail.size() * 64);
        for (sv line : tail) { block.append(strip(line)); block += " 0 1.0 0\n"; }
        std::fwrite(block.data(), 1, block.size(), out);
    }
    {
        std::unordered_set<sv> emitted;
        emitted.reserve(2 * juncs.size() + 16);
        for (sv j : juncs)
            if (emitted.insert(j).second) std::fwrite(j.data(), 1, j.size(), out);
    }
    if (std::fclose(out) != 0) die(std::string("write failed: ") + out_path);

    t_hits.join();                                            // all_hit_segs.txt was written meanwhile
    if (!hits_err.empty()) die(hits_err);
    trace.lap("outputs");
    std::fflush(nullptr);
    _exit(0);                   // both outputs are complete and closed: skip tearing down a million small host containers
}
