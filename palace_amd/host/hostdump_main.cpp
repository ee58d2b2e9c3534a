// hostdump -- prints what the host-side parsers hand to the GPU, as text (no GPU needed).
// Test tool for the CPU test suite:
//   hostdump bam   <file.bam>          header + one line per record (decoded columns + parsed SA items)
//   hostdump fastq <file.fq> <threads> [part_bytes]  one line per sequence line
//   hostdump fasta <file.fa> [threads] one line per record (ordinal, name, length, sequence); with threads: parse_fasta_mt
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "bam.hpp"
#include "fastx.hpp"

using namespace palace_host;

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: hostdump bam|fastq|fasta <file> [threads]\n"; return 2; }
    const std::string mode = argv[1];
    try {
        if (mode == "bam") {
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
