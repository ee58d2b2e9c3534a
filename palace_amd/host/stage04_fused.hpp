// Stage 04 inside generateGraph (optional): BAM -> `_graph.txt` -> `_filtered_graph_pre.txt` / `_filtered_graph.txt` ->
// `linear` / `cycle` / `cycle_nodup` / `all_result` in ONE process, every named artefact of palace:555-600 written, none of
// them read back.  The reference runs five processes here (generateGraph, filter_graph.py, uniq, matching,
// remove_cycle_dup.py + cat) coupled by text files; this replacement's separate executables stay as they are for the
// unchanged driver, and this path gives byte for byte the files that chain gives (tests/test_gpu_stage04.py) while the graph
// stays in HBM: the selection and the decomposition are the library's palace_stage04_* (csrc/filter.hip, decomp.hip).
//
// Host work here: the side files of filter_graph.py (read by threads while the BAM is inflated), and text.
// Line citations are to share/palace/scripts/filter_graph.py unless they name another file.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/palace_hip.h"
#include "bam.hpp"
#include "device_pick.hpp"
#include "fastx.hpp"
#include "textio.hpp"
#include "trace.hpp"

namespace palace_host {

struct Stage04Options {
    // inputs of filter_graph.py (argv 6-10, 12, 13; palace:568-579)
    std::string gene_file, score_file, blast_file, fasta_fai, paths_file;
    double blast_ratio = 0.7, score_threshold = 0.7;
    // outputs
    std::string pre_out, filtered_out, hit_segs_out, linear_out, cycle_out, nodup_out, result_out;
    // matching options (palace:587-590)
    int iterations = 10;
    bool self_loops = false, break_cycles = false, aggressive = false;
    bool enabled() const { return !pre_out.empty() || !filtered_out.empty() || !result_out.empty() || !linear_out.empty(); }
};

namespace s4 {

inline bool has_e(sv s) { return s.find('e') != sv::npos || s.find('E') != sv::npos; }
inline bool to_double(sv tok, double &v)                   // float(text): the whole (stripped) token must be a number
{
    const std::string t(strip(tok));
    if (t.empty() || t.find('x') != std::string::npos || t.find('X') != std::string::npos) return false;
    char *end = nullptr;
    v = std::strtod(t.c_str(), &end);
    return end == t.c_str() + t.size();
}
inline bool to_int(sv tok, long long &v)
{
    const sv t = strip(tok);
    size_t i = 0;
    bool neg = false;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
    long long x = 0;
    const size_t first = i;
    for (; i < t.size() && t[i] >= '0' && t[i] <= '9'; i++) x = x * 10 + (t[i] - '0');
    if (i == first || i != t.size() || i - first > 18) return false;
    v = neg ? -x : x;
    return true;
}
inline std::string fixed3(double v)
{
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.3f", v);
    return buf;
}
// fields written in scientific notation become plain (l.178-188)
inline std::string plain_number(sv tok)
{
    double v;
    if (!has_e(tok) || !to_double(tok, v)) return std::string(tok);
    char buf[400];
    if (std::isfinite(v) && v == std::floor(v)) {
        std::snprintf(buf, sizeof buf, "%.0f", v);
        if (buf[0] == '-' && buf[1] == '0' && buf[2] == 0) return "0";
        return buf;
    }
    std::string s = fixed3(v);
    while (!s.empty() && s.back() == '0') s.pop_back();
    if (!s.empty() && s.back() == '.') s.pop_back();
    return s;
}

}  // namespace s4

// what the side files say about every BAM target (names that are no target cannot have a SEG line and are ignored, as the
// script ignores facts about names its graph does not list)
struct Stage04Side {
    std::vector<uint8_t> seed;                     // bit 0 blast (l.66-94), bit 1 gene (l.99-102), bit 2 score > threshold (l.104-112)
    std::vector<std::string> score_text;           // "%.3f" text or "0.0"; empty = no line ("0.000" in the SEG text)
    std::vector<int32_t> name_len;                 // get_edge_len (l.50-52); -1 = the name has no such token
    std::vector<int64_t> path_off{0};
    std::vector<int32_t> path_tok;
    std::string error;                             // first thing the script would have died on
};

inline void stage04_read_side_files(const Stage04Options &o, const BamColumns &c, Stage04Side &out)
{
    Trace tr("generateGraph/side");
    const size_t nt = c.target_name.size();
    out.seed.assign(nt, 0);
    out.score_text.assign(nt, std::string());
    out.name_len.assign(nt, -1);
    std::vector<long long> fai_len(nt, -1);
    Names tokens;                                  // id token (EDGE_<token>_...) -> target, from the fasta index as the script builds it
    std::vector<int32_t> token_tid;
    std::string err_fai, err_blast, err_gene, err_score, err_paths;
    // name lengths: every target, not only those the fasta index lists
    constexpr int kSideThreads = 10;               // (beside the inflate threads; since the device takes part of the inflate this work is what the BAM phase ends with)
    pool_for(64, kSideThreads, [&](size_t part) {
        for (size_t t = nt * part / 64; t < nt * (part + 1) / 64; t++) {
            const std::string &nm = c.target_name[t];
            size_t a = 0;
            int k = 0;
            for (; k < 3; k++) { a = nm.find('_', a); if (a == std::string::npos) break; a++; }
            if (k == 3) {
                const size_t b = nm.find('_', a);
                long long v;
                if (s4::to_int(sv(nm).substr(a, b == std::string::npos ? std::string::npos : b - a), v) && v < (1ll << 31)) out.name_len[t] = static_cast<int32_t>(v);
            }
        }
    });
    std::unique_ptr<MappedText> fai, blast, genes, scores, paths;
    auto open = [](const std::string &p, std::unique_ptr<MappedText> &m, std::string &err) {
        try { m.reset(new MappedText(p)); } catch (const std::exception &e) { err = e.what(); }
    };
    open(o.fasta_fai, fai, err_fai); open(o.blast_file, blast, err_blast); open(o.gene_file, genes, err_gene);
    open(o.score_file, scores, err_score); open(o.paths_file, paths, err_paths);
    for (const std::string *e : {&err_fai, &err_blast, &err_gene, &err_score, &err_paths})
        if (!e->empty()) { out.error = *e; return; }

    // the two tables that need nothing but the target names start at once, each with bytes of its own (merged into `seed` at the end)
    std::vector<uint8_t> score_hit(nt, 0);
    std::vector<uint8_t> gene_hit(nt, 0);          // (its own bytes: the BLAST thread is writing `seed` meanwhile)
    std::thread t_gene([&] {                       // first column, untrimmed (l.101)
        for_each_line(genes->data, genes->size, [&](sv line) {
            const size_t t = line.find('\t');
            const int32_t tid = c.tid_of(t == sv::npos ? line : line.substr(0, t));
            if (tid >= 0) gene_hit[static_cast<size_t>(tid)] = 1;
        });
    });
    std::thread t_score([&] {                      // l.104-112
        const std::vector<size_t> cut = line_cuts(scores->data, scores->size, 8);
        std::vector<std::string> errs(cut.size() - 1);
        for_parts(cut, [&](size_t k, size_t a, size_t b) {
            std::vector<sv> cols;
            for_each_line(scores->data + a, b - a, [&](sv line) {
                if (!errs[k].empty()) return;
                split_on(strip(line), '\t', cols);
                double v = 0;
                if (cols.size() < 2 || (!s4::has_e(cols[1]) && !s4::to_double(cols[1], v))) { errs[k] = "malformed line in the score table"; return; }
                const int32_t tid = c.tid_of(cols[0]);
                if (tid < 0) return;
                std::string text = s4::has_e(cols[1]) ? std::string("0.0") : s4::fixed3(v);
                const double rounded = std::strtod(text.c_str(), nullptr);
                // (distinct targets per line; a repeated name: the later line wins, parts are in file order only per part --
                //  a name listed twice in the score file is not something the pipeline produces)
                out.score_text[static_cast<size_t>(tid)] = std::move(text);
                score_hit[static_cast<size_t>(tid)] = rounded > o.score_threshold;
            });
        });
        for (const auto &e : errs) if (!e.empty() && err_score.empty()) err_score = e;
    });
    // contigs.paths: every line that is no NODE header (l.126-137), in parts.  Its text work (lines, the ';' removed, tokens, their
    // hashes) needs nothing and runs beside the fasta index; the look-ups of the tokens wait for the index's id tokens.
    std::atomic<bool> t_paths_go{false};
    std::thread t_paths([&] {
        const std::vector<size_t> cut = line_cuts(paths->data, paths->size, 16);
        struct Part { std::string text; std::vector<uint32_t> at; std::vector<uint16_t> len; std::vector<uint64_t> hash; std::vector<uint8_t> kind; std::vector<int32_t> ends; };
        std::vector<Part> part(cut.size() - 1);                           // kind: 0 empty token, 1 '<id>+' (or any other last character), 2 '<id>-'
        pool_for(part.size(), kSideThreads / 2, [&](size_t k) {
            Part &pt = part[k];
            std::string clean;
            pt.text.reserve(cut[k + 1] - cut[k]);
            for_each_line(paths->data + cut[k], cut[k + 1] - cut[k], [&](sv raw) {
                const sv s = strip(raw);
                clean.clear();
                for (char ch : s) if (ch != ';') clean += ch;
                if (sv(clean).substr(0, 4) == "NODE") return;
                const size_t base = pt.text.size();
                pt.text += clean;
                size_t p = 0;
                while (p <= clean.size()) {
                    const size_t comma = clean.find(',', p);
                    const sv t = sv(clean).substr(p, comma == std::string::npos ? sv::npos : comma - p);
                    const size_t at = base + p;
                    p = comma == std::string::npos ? clean.size() + 1 : comma + 1;
                    const sv id = t.empty() ? t : t.substr(0, t.size() - 1);
                    pt.at.push_back(static_cast<uint32_t>(at));
                    pt.len.push_back(static_cast<uint16_t>(std::min<size_t>(id.size(), 65535)));
                    pt.hash.push_back(hash_bytes(id));
                    pt.kind.push_back(t.empty() ? 0 : t.back() == '-' ? 2 : 1);
                }
                pt.ends.push_back(static_cast<int32_t>(pt.at.size()));
            });
        });
        while (!t_paths_go.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(100));
        std::vector<std::vector<int32_t>> tok(part.size());
        pool_for(part.size(), kSideThreads, [&](size_t k) {
            const Part &pt = part[k];
            tok[k].resize(pt.at.size());
            for (size_t i = 0; i < pt.at.size(); i++) {
                if (i + 8 < pt.at.size()) tokens.prefetch(pt.hash[i + 8]);
                int32_t code = -1;
                if (pt.kind[i] && pt.len[i] < 65535) {
                    const int id = tokens.find_hashed(sv(pt.text).substr(pt.at[i], pt.len[i]), pt.hash[i]);
                    const int32_t tid = id < 0 ? -1 : token_tid[static_cast<size_t>(id)];
                    if (tid >= 0) code = 2 * tid + (pt.kind[i] == 2);
                }
                tok[k][i] = code;
            }
        });
        size_t n_tok = 0, n_lines = 0;
        for (size_t k = 0; k < tok.size(); k++) { n_tok += tok[k].size(); n_lines += part[k].ends.size(); }
        out.path_tok.reserve(n_tok + 1);
        out.path_off.reserve(n_lines + 1);
        for (size_t k = 0; k < tok.size(); k++) {
            const int64_t base = static_cast<int64_t>(out.path_tok.size());
            out.path_tok.insert(out.path_tok.end(), tok[k].begin(), tok[k].end());
            for (int32_t e : part[k].ends) out.path_off.push_back(base + e);
        }
    });
    // fasta index first: lengths (the BLAST rule divides by them) and the id tokens (contigs.paths speaks in them).  Parsed in
    // parts on threads (splitting, numbers, the name look-up); applied in file order by this thread
    {
        struct Row { sv token; uint64_t h_token; long long len; int32_t tid; };
        const std::vector<size_t> cut = line_cuts(fai->data, fai->size, 24);
        std::vector<std::vector<Row>> rows(cut.size() - 1);
        std::vector<std::string> errs(cut.size() - 1);
        pool_for(cut.size() - 1, kSideThreads, [&](size_t k) {
            std::vector<sv> cols, parts;
            rows[k].reserve((cut[k + 1] - cut[k]) / 40 + 16);
            for_each_line(fai->data + cut[k], cut[k + 1] - cut[k], [&](sv line) {
                if (!errs[k].empty()) return;
                split_on(strip(line), '\t', cols);
                long long len;
                if (cols.size() < 2 || !s4::to_int(cols[1], len)) { errs[k] = "short or non-numeric line in the fasta index"; return; }
                split_on(cols[0], '_', parts);
                if (parts.size() < 2) { errs[k] = "fasta index name without an id token: " + std::string(cols[0]); return; }
                rows[k].push_back(Row{parts[1], hash_bytes(parts[1]), len, c.tid_of(cols[0])});
            });
        });
        size_t n_rows = 0;
        for (size_t k = 0; k < rows.size(); k++) { n_rows += rows[k].size(); if (!errs[k].empty() && err_fai.empty()) err_fai = errs[k]; }
        tokens.reserve(n_rows + 16);
        token_tid.reserve(n_rows + 16);
        for (const auto &part : rows)
            for (size_t i = 0; i < part.size(); i++) {
                const Row &r = part[i];
                if (i + 8 < part.size()) tokens.prefetch(part[i + 8].h_token);
                if (r.tid >= 0) fai_len[static_cast<size_t>(r.tid)] = r.len;
                const int t = tokens.intern_hashed(r.token, r.h_token);
                if (static_cast<size_t>(t) >= token_tid.size()) token_tid.resize(static_cast<size_t>(t) + 1, -1);
                token_tid[static_cast<size_t>(t)] = r.tid;             // a later line with the same token wins, as in a dict (-1: no target)
            }
    }
    tr.lap("name lengths + fasta index");
    std::thread t_blast([&] {                      // consecutive rows of one (query, subject) pair form a group (l.66-94)
        std::vector<sv> cols;
        sv cur_q, cur_s;
        long long aligned = 0;
        const double cut = o.blast_ratio * 100;
        auto group_done = [&](sv q, bool last) {
            const int32_t tid = c.tid_of(q);
            if (tid < 0 || fai_len[static_cast<size_t>(tid)] < 0) {
                // (the script looks the query up in the fasta index; a query that is no BAM target cannot have a SEG line)
                if (tid >= 0 && !last) err_blast = "BLAST query not in the fasta index: " + std::string(q);
                return;
            }
            const long long len = fai_len[static_cast<size_t>(tid)];
            if (len == 0) { err_blast = "contig of length 0 in the fasta index: " + std::string(q); return; }
            if (static_cast<double>(aligned) / static_cast<double>(len) > o.blast_ratio || aligned > 2000) out.seed[static_cast<size_t>(tid)] |= 1;
        };
        for_each_line(blast->data, blast->size, [&](sv line) {
            if (!err_blast.empty()) return;
            split_on(strip(line), '\t', cols);
            double ident;
            long long alen;
            if (cols.size() < 4 || !s4::to_double(cols[2], ident) || !s4::to_int(cols[3], alen)) { err_blast = "malformed line in the BLAST table"; return; }
            const sv q = cols[0], s = cols[1];
            const bool new_group = (cur_q != q && !cur_q.empty()) || (cur_s != s && !cur_s.empty());
            if (new_group) { group_done(cur_q, false); aligned = ident > cut ? alen : 0; }
            else if (ident > cut) aligned += alen;
            cur_q = q; cur_s = s;
        });
        if (err_blast.empty() && !cur_q.empty()) group_done(cur_q, true);
    });
    tr.lap("blast started");
    t_paths_go.store(true, std::memory_order_release);           // (the id tokens are final: the path reader may look its tokens up)
    t_blast.join();
    t_gene.join();
    t_score.join();
    for (size_t t = 0; t < nt; t++) out.seed[t] = static_cast<uint8_t>((out.seed[t] & ~6u) | (gene_hit[t] ? 2u : 0u) | (score_hit[t] ? 4u : 0u));
    tr.lap("blast, gene hits, scores (joined)");
    t_paths.join();
    tr.lap("paths (joined)");
    for (const std::string *e : {&err_fai, &err_blast, &err_gene, &err_score, &err_paths})
        if (!e->empty() && out.error.empty()) out.error = *e;
}

// remove_cycle_dup.py:3-30: pairs of lines (an odd last line is paired with "\n"), first occurrence of every pair kept
inline std::string cycle_without_duplicates(const std::string &cyc)
{
    std::vector<sv> lines;
    for_each_line(cyc.data(), cyc.size(), [&](sv l) { lines.push_back(l); });
    std::string out;
    std::unordered_set<std::string> seen;
    for (size_t i = 0; i < lines.size(); i += 2) {
        std::string pair(lines[i]);
        pair += '\x01';
        pair += i + 1 < lines.size() ? std::string(lines[i + 1]) : std::string("\n");
        if (!seen.insert(pair).second) continue;
        out.append(lines[i]);
        out.append(i + 1 < lines.size() ? lines[i + 1] : sv("\n"));
    }
    return out;
}

inline bool write_file(const std::string &path, const std::string &text)
{
    if (path.empty()) return true;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
    return (std::fclose(f) == 0) && ok;
}

// `uniq`: a line equal to the one before it is dropped
inline std::string uniq_lines(const std::string &text)
{
    std::string out;
    out.reserve(text.size());
    sv prev;
    bool have = false;
    for_each_line(text.data(), text.size(), [&](sv l) {
        if (have && l == prev) return;
        out.append(l);
        prev = l; have = true;
    });
    return out;
}

// The resident object (side arrays on the device, the path-arc table, the big scratch blocks) can be set up while the BAM is
// still being inflated: on a context of its own, from the thread that read the side files.  edge_guess sizes the scratch (it
// grows later if the real bound is larger).
inline palace_stage04 *stage04_prepare(const Stage04Options &o, const BamColumns &c, const Stage04Side &side, const std::vector<int32_t> &rank,
                                       int min_count, int64_t edge_guess, std::string &err)
{
    if (!side.error.empty()) { err = side.error; return nullptr; }
    Trace tr("generateGraph/prepare");
    palace_ctx *ctx = nullptr;
    if (palace_ctx_create(pick_device(), &ctx)) { err = palace_last_error(); return nullptr; }
    tr.lap("context");
    palace_stage04_inputs in{};
    in.n_segs = static_cast<int32_t>(c.target_name.size()); in.min_count = min_count;
    in.seed = side.seed.data(); in.tlen = c.target_len.data(); in.rank = rank.data();
    std::vector<int32_t> name_len(side.name_len);
    for (int32_t &v : name_len) if (v < 0) v = 0;                         // (only read for members of contigs.paths lines, whose names have the token)
    in.name_len = name_len.data();
    in.n_paths = static_cast<int64_t>(side.path_off.size()) - 1;
    in.path_off = side.path_off.data(); in.path_tok = side.path_tok.data();
    palace_stage04 *st = nullptr;
    if (palace_stage04_create(ctx, &in, &st)) { err = palace_last_error(); st = nullptr; }
    tr.lap("create");
    if (st && (palace_stage04_reserve(ctx, st, edge_guess) || palace_sync(ctx))) { err = palace_last_error(); st = nullptr; }
    tr.lap("reserve");
    palace_ctx_destroy(ctx);                                              // (the object's memory belongs to the device, not to the context)
    (void)o;
    return st;
}

// After generateGraph's resolve: selection, decomposition, every file.  raw_seg[t] = the SEG line of target t as `_graph.txt`
// has it (empty: none); edges = the aggregated edges in DEVICE order (d_edges), sorted_index = their `_graph.txt` order.
// Returns 0, or 1 with `err` set.
inline int stage04_run(palace_ctx *ctx, palace_stage04 *st, const Stage04Options &o, const BamColumns &c, const Stage04Side &side,
                       const std::vector<int32_t> &rank, const std::vector<sv> &raw_seg, const std::vector<palace_graph_edge> &edges,
                       const std::vector<uint32_t> &sorted_index, const palace_graph_edge *d_edges, int64_t n_cands,
                       const int32_t *d_cn, int threads, Trace &tr, std::string &err)
{
    const int32_t nt = static_cast<int32_t>(c.target_name.size());
    auto fail = [&](const std::string &what) { err = what; return 1; };
    void *p = nullptr;
    if (palace_malloc(ctx, 8, &p)) return fail(palace_last_error());
    const int64_t n_edges = static_cast<int64_t>(edges.size());
    if (palace_h2d(ctx, p, &n_edges, 8)) return fail(palace_last_error());
    if (palace_stage04_filter(ctx, st, d_edges, static_cast<const int64_t *>(p), std::max<int64_t>(1, std::max(n_cands, n_edges)))) return fail(palace_last_error());
    if (palace_stage04_match(ctx, st, d_edges, d_cn, o.iterations, o.aggressive ? 1 : 0, 1)) return fail(palace_last_error());
    std::vector<uint8_t> seg_flags(static_cast<size_t>(nt)), edge_flags(static_cast<size_t>(n_edges));
    if (palace_stage04_flags(ctx, st, seg_flags.data(), edge_flags.data(), n_edges)) return fail(palace_last_error());
    int64_t counts[8];
    if (palace_stage04_counts(ctx, st, counts)) return fail(palace_last_error());   // (also what the script dies on)
    tr.lap("stage 04: selection on the device");
    // the decomposition is a few milliseconds of device work behind the selection: wait for it now, then the two text outputs
    // (filtered graph; linear / cycle / all_result) are independent of each other and are made side by side
    palace_match_result *res = nullptr;
    const int32_t *contig_of = nullptr;
    int64_t n_f = 0;
    if (palace_stage04_result(ctx, st, &res, &contig_of, &n_f)) return fail(palace_last_error());
    tr.lap("stage 04: decomposition (waited)");
    std::string err_filtered;
    std::thread filtered_text([&] {

    // ---- `_filtered_graph_pre.txt` (l.252-264): selected SEG lines and rescued ones in graph order, the kept junctions ----
    std::vector<int32_t> by_rank(static_cast<size_t>(nt));
    for (int32_t t = 0; t < nt; t++) by_rank[static_cast<size_t>(rank[static_cast<size_t>(t)])] = t;
    std::string pre, hits;
    {
        const size_t n_parts = static_cast<size_t>(std::max(1, threads)) * 2;
        std::vector<std::string> sel_part(n_parts), resc_part(n_parts), hit_part(n_parts);
        pool_for(n_parts, threads, [&](size_t part) {
            std::string &sel = sel_part[part], &resc = resc_part[part], &hit = hit_part[part];
            std::vector<sv> toks;
            const int32_t r0 = static_cast<int32_t>(static_cast<int64_t>(nt) * part / n_parts), r1 = static_cast<int32_t>(static_cast<int64_t>(nt) * (part + 1) / n_parts);
            for (int32_t r = r0; r < r1; r++) {
                const int32_t t = by_rank[static_cast<size_t>(r)];
                const uint8_t sd = side.seed[static_cast<size_t>(t)];
                if (sd && !raw_seg[static_cast<size_t>(t)].empty()) {   // all_hit_segs.txt in SEG order (l.163-171, 266-269)
                    hit += "SAMPLE\t"; hit += c.target_name[static_cast<size_t>(t)]; hit += '\t';
                    if (sd & 1) hit += "ref+";
                    if (sd & 4) hit += "score+";
                    if (sd & 2) hit += "gene+";
                    hit += '\n';
                }
                const uint8_t fl = seg_flags[static_cast<size_t>(t)];
                if (fl & 1) {
                    split_ws(raw_seg[static_cast<size_t>(t)], toks);    // l.173-197
                    for (size_t i = 0; i < toks.size(); i++) {
                        if (i) sel += ' ';
                        if (i < 2) sel.append(toks[i]); else sel += s4::plain_number(toks[i]);
                    }
                    sel += (sd & 2) ? " 1 " : " 0 ";
                    sel += side.score_text[static_cast<size_t>(t)].empty() ? std::string("0.000") : side.score_text[static_cast<size_t>(t)];
                    sel += (sd & 1) ? " 1\n" : " 0\n";
                } else if ((fl & 3) == 2) {
                    resc.append(strip(raw_seg[static_cast<size_t>(t)]));
                    resc += " 0 1.0 0\n";
                }
            }
        });
        size_t total = static_cast<size_t>(counts[2] + counts[3]) * 110;
        for (size_t k = 0; k < n_parts; k++) total += sel_part[k].size() + resc_part[k].size();
        pre.reserve(total);
        for (const std::string &x : sel_part) pre += x;                   // selected SEG lines in graph order, then the rescued ones
        for (const std::string &x : resc_part) pre += x;
        for (const std::string &x : hit_part) hits += x;
    }
    char line[1024];
    auto junc_line = [&](const palace_graph_edge &e) {
        const uint32_t supp = e.counts[0], supp_nf = e.counts[1], span = e.counts[2], span_nf = e.counts[3];
        const std::string &l = c.target_name[static_cast<size_t>(e.left)], &r2 = c.target_name[static_cast<size_t>(e.right)];
        if (l.size() + r2.size() < 900)
            pre.append(line, static_cast<size_t>(std::snprintf(line, sizeof line, "JUNC %s %c %s %c %u %u\n", l.c_str(), e.oL ? '-' : '+', r2.c_str(),
                                                               e.oR ? '-' : '+', supp + span + supp_nf, span_nf)));
        else {
            pre += "JUNC " + l + (e.oL ? " - " : " + ") + r2 + (e.oR ? " - " : " + ");
            pre.append(line, static_cast<size_t>(std::snprintf(line, sizeof line, "%u %u\n", supp + span + supp_nf, span_nf)));
        }
    };
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t i : sorted_index) {
            const uint8_t f = edge_flags[i];
            if (pass == 0 ? (f & 2) : ((f & 6) == 4)) junc_line(edges[i]);
        }
    if (!write_file(o.pre_out, pre)) { err_filtered = "cannot write " + o.pre_out; return; }
    if (!o.filtered_out.empty() && !write_file(o.filtered_out, uniq_lines(pre))) { err_filtered = "cannot write " + o.filtered_out; return; }
    if (!write_file(o.hit_segs_out, hits)) { err_filtered = "cannot write " + o.hit_segs_out; return; }
    });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_filtered{filtered_text};

    // ---- matching's two files, as palace_amd/host/matching_main.cpp writes them ----
    const int64_t n_comp = palace_match_result_count(res);
    const int64_t *off = palace_match_result_offsets(res);
    const int32_t *verts = palace_match_result_verts(res), *iter = palace_match_result_iter(res), *open_at = palace_match_result_open_at(res);
    const uint8_t *kind = palace_match_result_kind(res);
    const uint64_t *bare = palace_match_result_bare(res);
    std::string lin, cyc, selfs;
    // lines seen so far (views of the ranges' buffers and of `opened`): flat tables fed with the hashes the formatting threads made
    struct LineSet {
        std::vector<uint64_t> hash;
        std::vector<sv> line;
        size_t mask = 0;
        explicit LineSet(size_t n) { size_t cap = 1024; while (cap < 2 * n + 16) cap <<= 1; hash.assign(cap, 0); line.resize(cap); mask = cap - 1; }
        void prefetch(uint64_t h) const { __builtin_prefetch(&hash[h & mask]); }
        bool insert(sv s, uint64_t h)                                      // true: new
        {
            h |= 1;                                                        // (0 = empty slot)
            size_t at = h & mask;
            while (hash[at]) { if (hash[at] == h && line[at] == s) return false; at = (at + 1) & mask; }
            hash[at] = h; line[at] = s;
            return true;
        }
    };
    std::deque<std::string> opened;
    auto name_of = [&](int32_t v) -> const std::string & { return c.target_name[static_cast<size_t>(contig_of[v >> 1])]; };
    auto comp_line = [&](int64_t k, int64_t first, std::string &s) {
        const int64_t n = off[k + 1] - off[k];
        for (int64_t i = 0; i < n; i++) {
            const int32_t v = verts[off[k] + (first + i) % n];
            if (i) s += '\t';
            s += name_of(v);
            s += (v & 1) ? '-' : '+';
        }
        s += '\n';
    };
    // the lines of all components, formatted on threads (ranges of components; one buffer and the line ends per range)
    const size_t n_parts = static_cast<size_t>(std::max(1, threads)) * 2;
    std::vector<std::string> text(n_parts);
    std::vector<std::vector<uint32_t>> ends(n_parts);                     // end offset of every component's line in its range's buffer
    std::vector<std::vector<uint64_t>> hashes(n_parts);                   // ... and the line's hash
    pool_for(n_parts, threads, [&](size_t part) {
        const int64_t k0 = n_comp * static_cast<int64_t>(part) / static_cast<int64_t>(n_parts), k1 = n_comp * static_cast<int64_t>(part + 1) / static_cast<int64_t>(n_parts);
        ends[part].reserve(static_cast<size_t>(k1 - k0));
        hashes[part].reserve(static_cast<size_t>(k1 - k0));
        for (int64_t k = k0; k < k1; k++) {
            const size_t a0 = text[part].size();
            comp_line(k, 0, text[part]);
            ends[part].push_back(static_cast<uint32_t>(text[part].size()));
            hashes[part].push_back(hash_bytes(sv(text[part]).substr(a0)));
        }
    });
    LineSet lin_seen(static_cast<size_t>(n_comp) * (o.break_cycles ? 2 : 1)), cyc_seen(static_cast<size_t>(n_comp));
    size_t part_of = 0;
    int64_t part_first = 0;
    uint64_t hash_of_line = 0;                                             // hash of the line line_of() returned last
    auto line_of = [&](int64_t k) -> sv {                                  // (k ascends)
        while (k >= n_comp * static_cast<int64_t>(part_of + 1) / static_cast<int64_t>(n_parts)) { part_of++; part_first = n_comp * static_cast<int64_t>(part_of) / static_cast<int64_t>(n_parts); }
        const size_t i = static_cast<size_t>(k - part_first);
        const uint32_t a0 = i ? ends[part_of][i - 1] : 0;
        hash_of_line = hashes[part_of][i];
        if (i + 8 < hashes[part_of].size()) { lin_seen.prefetch(hashes[part_of][i + 8] | 1); cyc_seen.prefetch(hashes[part_of][i + 8] | 1); }
        return sv(text[part_of]).substr(a0, ends[part_of][i] - a0);
    };
    // round 0 lists the bare segments (one-vertex paths) between the components, in first-vertex order; names are distinct,
    // so a bare line cannot repeat another line (a component vertex has an arc, a bare segment has none)
    int64_t next_bare = 0;                                              // next filtered segment id to test for bareness
    auto bare_until = [&](int64_t seg_end) {                             // bare segments with id < seg_end
        for (; next_bare < seg_end; next_bare++)
            if ((bare[next_bare >> 6] >> (next_bare & 63)) & 1) { lin += c.target_name[static_cast<size_t>(contig_of[next_bare])]; lin += "+\n"; }
    };
    bool past_round0 = false;
    for (int64_t k = 0; k < n_comp; k++) {
        if (!past_round0 && iter[k] != 0) { bare_until(n_f); past_round0 = true; }
        if (!past_round0) bare_until((static_cast<int64_t>(verts[off[k]]) + 1) >> 1);   // a component goes behind every bare s with 2 s < its first vertex
        const int64_t n = off[k + 1] - off[k];
        const sv s = line_of(k);
        if (!kind[k]) {
            if (n == 1 && iter[k] > 0) continue;                          // a bare segment is reported once, in round 0
            if (lin_seen.insert(s, hash_of_line)) lin.append(s);
            continue;
        }
        if (!cyc_seen.insert(s, hash_of_line)) continue;
        if (n == 1 && o.self_loops) { selfs += "self\n"; selfs.append(s); }
        else { cyc += "iter " + std::to_string(iter[k]) + "\n"; cyc.append(s); }
        if (o.break_cycles) {
            opened.emplace_back();
            comp_line(k, open_at[k], opened.back());
            if (lin_seen.insert(sv(opened.back()), hash_bytes(opened.back()))) lin += opened.back(); else opened.pop_back();
        }
    }
    if (!past_round0) bare_until(n_f);
    cyc += selfs;
    const std::string nodup = cycle_without_duplicates(cyc);
    if (!write_file(o.linear_out, lin) || !write_file(o.cycle_out, cyc) || !write_file(o.nodup_out, nodup) || !write_file(o.result_out, lin + nodup))
        return fail("cannot write the matching outputs");
    tr.lap("stage 04: result text");
    filtered_text.join();
    if (!err_filtered.empty()) return fail(err_filtered);
    tr.lap("stage 04: filtered graph text (joined)");
    return 0;
}

}  // namespace palace_host
