// hostdump -- prints what the host-side parsers hand to the GPU, as text (no GPU needed).
// Test tool for the CPU test suite:
//   hostdump bam   <file.bam>          header + one line per record (decoded columns + parsed SA items)
//   hostdump bamtime <file.bam> <threads>   load only, prints the record count (PALACE_TRACE=1: laps of the loader)
//   hostdump fastq <file.fq> <threads> [part_bytes]  one line per sequence line
//   hostdump fastqpack <file.fq> <threads> <part_bytes> [keep_every]   the packed form of the sequence lines (pack_fastq_part):
//                                      one line per part "pos0 n_words n_reads", then its words of P0, P1, U (hex, one line per stream);
//                                      keep_every = k: read r is counted iff r % k != 0
//   hostdump fmtg <n> <seed>           format_g6 (textio.hpp) against snprintf("%g") on n values of every kind (random bits, quotients of
//                                      integers as SEG depths are, decimals at and next to rounding ties); prints the number of differences
//   hostdump fasta <file.fa> [threads] one line per record (ordinal, name, length, sequence); with threads: parse_fasta_mt
//   hostdump bamtrace <bam> <fai> [..] what `generateGraph --debug` writes to stderr for the file's records (debug_trace.hpp), on stdout
//   hostdump devicepick x              "<ordinal handed to palace_ctx_create> <ROCR_VISIBLE_DEVICES afterwards>" (device_pick.hpp)
//   hostdump forkcheck x               "1" when the executables would stay one process here (fast_exit.hpp: a profiler / preload in
//                                      the environment, or a GPU runtime already open), else "0"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "bam.hpp"
#include "fast_exit.hpp"
#include "device_pick.hpp"
#include "debug_trace.hpp"
#include "fastx.hpp"

using namespace palace_host;

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: hostdump bam|fastq|fasta <file> [threads]\n"; return 2; }
    const std::string mode = argv[1];
    try {
        if (mode == "forkcheck") {
            std::printf("%d\n", (std::getenv("PALACE_NO_FORK") || gpu_touched_before_main()) ? 1 : 0);
        } else if (mode == "bamtrace") {                  // hostdump bamtrace <bam> <fastg.fai> [max_end min_mapq max_nm enable_paired max_span_frac]
            if (argc < 4) { std::cerr << "usage: hostdump bamtrace <bam> <fastg.fai> [max_end min_mapq max_nm enable_paired max_span_frac]\n"; return 2; }
            BamColumns c;
            c.want_match_segments = false;
            load_bam(argv[2], 4, 1, c);
            palace_graph_params prm{300, 0, 5, 1, 0, 0, 0.80};
            if (argc > 4) prm.max_end = std::atoi(argv[4]);
            if (argc > 5) prm.min_mapq = std::atoi(argv[5]);
            if (argc > 6) prm.max_nm = std::atoi(argv[6]);
            if (argc > 7) prm.enable_paired = std::atoi(argv[7]);
            if (argc > 8) prm.max_span_frac = std::atof(argv[8]);
            const std::string t = debug_trace(c, argv[3], prm);
            std::fwrite(t.data(), 1, t.size(), stdout);
        } else if (mode == "devicepick") {
            const int ord = pick_device();
            const char *v = std::getenv("ROCR_VISIBLE_DEVICES");
            std::printf("%d %s\n", ord, v ? v : "-");
        } else if (mode == "bam") {
            BamColumns c;
            load_bam(argv[2], argc > 3 ? std::atoi(argv[3]) : 4, 1, c);
            for (size_t i = 0; i < c.target_name.size(); i++) std::printf("@SQ\t%s\t%d\n", c.target_name[i].c_str(), c.target_len[i]);
            for (int64_t i = 0; i < c.n(); i++) {
                std::printf("%s\t%u\t%d\t%d\t%u\t%d\t%d\t%d\t%d\t%d\t%d\t%d", c.qname(i).c_str(), c.flag[i], c.tid[i], c.pos[i], c.mapq[i],
                            c.mtid[i], c.mpos[i], c.nm[i], c.ref_len[i], c.read_len[i], c.clip_s[i], c.clip_e[i]);
                for (int32_t k = c.sa_off[i]; k < c.sa_off[i + 1]; k++) {
                    const palace_sa_item &s = c.sa[k];
                    std::printf("\tSA:%d,%d,%d,%d,%d,%d,%d,%d", s.tid2, s.pos2, s.rev2, s.mapq2, s.nm2, s.clip_s2, s.clip_e2, s.len2);
                }
                std::printf("\n");
            }
        } else if (mode == "bamtime") {
            BamColumns c;
            load_bam(argv[2], argc > 3 ? std::atoi(argv[3]) : 4, 1, c);
            std::printf("%lld records, %zu SA items, %zu match segments\n", (long long)c.n(), c.sa.size(), c.mseg_tid.size());
        } else if (mode == "fastq") {
            // the executable's own ingest path: mapped file, parts, two passes (fastx.hpp); optional 4th argument =
            // part size in bytes, so that tests can force many parts on a small file
            const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
            MappedText txt(argv[2]);
            FastqPlan plan;
            plan_fastq(txt, threads, plan, argc > 4 ? static_cast<size_t>(std::atol(argv[4])) : (4u << 20));
            std::vector<uint8_t> bases(static_cast<size_t>(plan.n_bases) + 1);
            std::vector<int64_t> off(static_cast<size_t>(plan.n_reads) + 1, 0);
            pool_for(plan.parts.size(), threads, [&](size_t i) { extract_fastq_part(plan, i, bases.data(), 0, off.data(), 0); });
            for (int64_t i = 0; i < plan.n_reads; i++) {
                std::fwrite(bases.data() + off[i], 1, static_cast<size_t>(off[i + 1] - off[i]), stdout);
                std::fputc('\n', stdout);
            }
        } else if (mode == "fastqpack") {
            const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
            MappedText txt(argv[2]);
            FastqPlan plan;
            plan_fastq(txt, threads, plan, argc > 4 ? static_cast<size_t>(std::atol(argv[4])) : (4u << 20));
            const int every = argc > 5 ? std::atoi(argv[5]) : 0;
            std::vector<uint8_t> keep;
            if (every > 0) { keep.resize(static_cast<size_t>(plan.n_reads)); for (size_t r = 0; r < keep.size(); r++) keep[r] = (r % static_cast<size_t>(every)) != 0; }
            std::vector<int64_t> pos0;
            int64_t n_pos = 0;
            for (const FastqPart &pt : plan.parts) { pos0.push_back(n_pos); n_pos += packed_span(pt.seq_bytes()); }
            std::vector<uint64_t> st[3];
            for (auto &v : st) v.assign(static_cast<size_t>(n_pos / 64) + 1, 0xdeadbeefdeadbeefull);      // every word must be written
            pool_for(plan.parts.size(), threads, [&](size_t i) {
                const size_t o = static_cast<size_t>(pos0[i] / 64);
                pack_fastq_part(plan, i, st[0].data() + o, st[1].data() + o, st[2].data() + o, keep.empty() ? nullptr : keep.data(), 0);
            });
            for (size_t i = 0; i < plan.parts.size(); i++) {
                const size_t o = static_cast<size_t>(pos0[i] / 64), nw = static_cast<size_t>(packed_span(plan.parts[i].seq_bytes()) / 64);
                std::printf("%lld %zu %lld\n", (long long)pos0[i], nw, (long long)plan.parts[i].n_seq());
                for (int q = 0; q < 3; q++) {
                    for (size_t k = 0; k < nw; k++) std::printf("%016llx%c", (unsigned long long)st[q][o + k], k + 1 < nw ? ' ' : '\n');
                    if (!nw) std::fputc('\n', stdout);
                }
            }
        } else if (mode == "fmtg") {
            const long n = std::atol(argv[2]);
            uint64_t st = argc > 3 ? static_cast<uint64_t>(std::atoll(argv[3])) * 0x9E3779B97F4A7C15ull + 1 : 1;
            auto next = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
            long bad = 0, fast = 0;
            char a[64], b[64];
            auto check = [&](double v) {
                const size_t la = format_g6(v, a);
                const int lb = std::snprintf(b, sizeof b, "%g", v);
                if (la != static_cast<size_t>(lb) || std::memcmp(a, b, la) != 0) { if (bad++ < 20) std::printf("differs: %.17g -> '%.*s' vs '%s'\n", v, static_cast<int>(la), a, b); }
                fast += v >= 1e-4 && v < 1e6;
            };
            for (long i = 0; i < n; i++) {
                const uint64_t r = next();
                switch (i % 6) {
                case 0: { double v; const uint64_t bits = (r & 0x800fffffffffffffull) | (static_cast<uint64_t>(1023 - 20 + (next() % 48)) << 52); std::memcpy(&v, &bits, 8); check(v); break; }   // random mantissa, 2^-20 .. 2^27
                case 1: check(static_cast<double>(r % 100000000) / static_cast<double>(std::max<uint64_t>(1, next() % 200000))); break;          // sum / length
                case 2: { const int j = static_cast<int>(next() % 10); check(static_cast<double>(r % 20000000) / std::pow(10.0, j)); break; }  // decimals with up to 8 digits: ties among them
                case 3: check(static_cast<double>((r % 2000000) * 10 + 5) / std::pow(10.0, static_cast<int>(next() % 9))); break;                // xxxxxx5: seven digits ending in 5
                case 4: check(std::nextafter(static_cast<double>((r % 2000000) * 10 + 5) / std::pow(10.0, static_cast<int>(next() % 9)), (next() & 1) ? 1e300 : -1e300)); break;
                default: check(static_cast<double>(r % 3000000)); break;                                                                       // integers up to and past 1e6
                }
            }
            for (double v : {0.0, -0.0, 1e-4, 9.9999949e-5, 9.9999951e-5, 0.1, 0.099999949, 0.09999995, 999999.0, 999999.4, 999999.5, 999999.6, 1e6, 1e-5, 123456.5, 1.5, 2.5, 0.5, 1e5, 1e-300, 1e300})
                check(v);
            std::printf("%ld differences in %ld values (%ld in the fast range)\n", bad, n + 21, fast);
            return bad ? 1 : 0;
        } else if (mode == "fasta") {
            SeqSet db;
            const std::vector<char> txt = read_file(argv[2]);
            if (argc > 3) parse_fasta_mt(txt.data(), txt.size(), db, std::atoi(argv[3]));     // the eref executable's threaded parser
            else parse_fasta(txt, db);
            for (int64_t i = 0; i < db.n(); i++) {
                std::printf("%lld\t%s\t%lld\t", (long long)db.ordinal[i], db.names[i].c_str(), (long long)db.len(i));
                std::fwrite(db.bases.data() + db.offsets[i], 1, static_cast<size_t>(db.len(i)), stdout);
                std::fputc('\n', stdout);
            }
        } else { std::cerr << "unknown mode\n"; return 2; }
    } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
    return 0;
}
