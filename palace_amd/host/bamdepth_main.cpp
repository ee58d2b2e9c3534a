// bamdepth -- the mean depth the driver computes before generateGraph (palace:538-552):
//     samtools depth -@ T <bam> > <bam>.depth ; first_depth=$(awk '{sum+=$3} END { print sum/NR }' <bam>.depth)
// as one command:   first_depth=$(bamdepth <bam>)
// prints exactly what the awk line prints.  (The per-base <bam>.depth.gz that step 5 reads through tabix is not written;
// `generateGraph <bam> <fai> <out> auto` uses the same number without a second pass over the BAM.)
#include <algorithm>
#include <iostream>
#include <thread>

#include "bam.hpp"
#include "depth_host.hpp"

using namespace palace_host;

int main(int argc, char **argv)
{
    if (argc < 2) { std::cerr << "Usage: " << argv[0] << " <bam>\n"; return 1; }
    BamColumns c;
    try {
        load_bam(argv[1], static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency()))), 1, c);
    } catch (const std::exception &e) { std::cerr << e.what() << "\n"; return 1; }
    palace_ctx *ctx = nullptr;
    if (palace_ctx_create(0, &ctx)) { std::cerr << "bamdepth: " << palace_last_error() << "\n"; return 1; }
    std::string text;
    const int rc = first_depth(ctx, c, text);
    palace_ctx_destroy(ctx);
    if (rc < 0) { std::cerr << "bamdepth: " << palace_last_error() << "\n"; return 1; }
    if (rc > 0) { std::cerr << "bamdepth: no position is covered (awk: division by zero)\n"; return 2; }
    std::cout << text << "\n";
    return 0;
}
