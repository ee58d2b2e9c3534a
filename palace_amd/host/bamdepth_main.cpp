// bamdepth -- the mean depth the driver computes before generateGraph (palace:538-552):
//     samtools depth -@ T <bam> > <bam>.depth ; first_depth=$(awk '{sum+=$3} END { print sum/NR }' <bam>.depth)
// as one command:   first_depth=$(bamdepth <bam>)
// prints exactly what the awk line prints.
//     bamdepth --per-contig <bam> > contig_depth.tsv
// prints `contig <TAB> depth sum <TAB> covered positions` for every contig with coverage: the two numbers step 5 takes from
// the tabix-indexed depth file per contig (create_sub_graph.py:186-234); palace_amd/scripts/create_sub_graph.py reads it.
//     first_depth=$(bamdepth --depth-gz <bam>.depth.gz <bam>)
// writes the per-base depth file itself -- <bam>.depth.gz (the text of `samtools depth`, BGZF) and <bam>.depth.gz.tbi (the index
// `tabix -s 1 -b 2 -e 2` makes) -- and prints the awk number: the four commands of palace:541-545 as one, host only (depthgz.hpp).
// (`generateGraph <bam> <fai> <out> auto` uses the same number without a second pass over the BAM.)
#include <algorithm>
#include <iostream>
#include <thread>

#include "bam.hpp"
#include "device_pick.hpp"
#include "depth_host.hpp"
#include "depthgz.hpp"

using namespace palace_host;

int main(int argc, char **argv)
{
    const bool per_contig = argc >= 3 && std::string(argv[1]) == "--per-contig";
    const bool depth_gz = argc >= 4 && std::string(argv[1]) == "--depth-gz";
    if (argc < 2 || (per_contig && argc < 3) || (std::string(argv[1]) == "--depth-gz" && argc < 4)) {
        std::cerr << "Usage: " << argv[0] << " [--per-contig | --depth-gz <out.depth.gz>] <bam>\n";
        return 1;
    }
    const char *bam = depth_gz ? argv[3] : per_contig ? argv[2] : argv[1];
    const int threads = static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    BamColumns c;
    try {
        load_bam(bam, threads, 1, c);
    } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
    if (depth_gz) {                                  // no GPU in this mode: text and DEFLATE are host work
        try {
            const DepthGzResult r = write_depth_gz(c, argv[2], threads);
            if (r.lines == 0) { std::cerr << "bamdepth: no position is covered (awk: division by zero)\n"; return 2; }
            std::cout << awk_number(static_cast<double>(r.sum) / static_cast<double>(r.lines)) << "\n";
            return 0;
        } catch (const std::exception &e) { std::cerr << "bamdepth: " << e.what() << "\n"; return 1; }
    }
    palace_ctx *ctx = nullptr;
    if (palace_ctx_create(pick_device(), &ctx)) { std::cerr << "bamdepth: " << palace_last_error() << "\n"; return 1; }
    std::string text;
    std::vector<uint64_t> cs, cc;
    const int rc = per_contig ? first_depth(ctx, c, text, nullptr, nullptr, &cs, &cc) : first_depth(ctx, c, text);
    palace_ctx_destroy(ctx);
    if (rc < 0) { std::cerr << "bamdepth: " << palace_last_error() << "\n"; return 1; }
    if (per_contig) {
        std::string out;
        for (size_t t = 0; t < cs.size(); t++)
            if (cc[t]) out += c.target_name[t] + "\t" + std::to_string(cs[t]) + "\t" + std::to_string(cc[t]) + "\n";
        std::cout << out;
        return 0;
    }
    if (rc > 0) { std::cerr << "bamdepth: no position is covered (awk: division by zero)\n"; return 2; }
    std::cout << text << "\n";
    return 0;
}
