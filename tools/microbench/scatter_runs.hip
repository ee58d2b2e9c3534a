// Microbenchmark (GPU box): HBM write bandwidth of short contiguous runs appended to many regions -- the store pattern of the
// partition kernels (a workgroup reserves a run in each of R regions with an atomic cursor and writes it).  Sweeps run
// length and region count; prints GB/s.   hipcc -O3 --offload-arch=gfx950 scatter_runs.hip -o scatter_runs && ./scatter_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// one wave per "row": reserves run_words u32 in region (row id), writes them (lane-contiguous)
__global__ __launch_bounds__(512) void scatter(uint32_t *buf, unsigned int *cursor, uint64_t region_words, int n_regions, int run_words,
                                               int rows_per_wave, int replicas, int mode)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int replica = blockIdx.x % replicas;
    for (int j = 0; j < rows_per_wave; j++) {
        const int row = (wave + j * 8) % n_regions;
        const int region = row * replicas + replica;
        unsigned int g = 0;
        if (mode == 0) {                                   // reserve with a returning atomic, one row at a time
            if (lane == 0) g = atomicAdd(&cursor[region], static_cast<unsigned int>(run_words));
            g = __shfl(g, 0);
        } else {                                           // mode 1: no atomics, the slot follows from the block number
            g = static_cast<unsigned int>((blockIdx.x / replicas) * static_cast<unsigned int>(rows_per_wave / (16 / 2) ) * run_words / 2 + (j / 8) * run_words);
            g = static_cast<unsigned int>(((blockIdx.x / replicas) * 2u + (j >> 3)) * run_words);
        }
        if (g + run_words > region_words) continue;
        uint32_t *dst = buf + static_cast<uint64_t>(region) * region_words + g;
        for (int q = lane; q < run_words; q += 64) dst[q] = q + blockIdx.x;
    }
}

__global__ void stream_write(uint4 *buf, size_t n16)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        buf[i] = uint4{1, 2, 3, 4};
}

int main()
{
    const size_t total_bytes = 8ull << 30;
    uint32_t *buf; unsigned int *cursor;
    hipMalloc(&buf, total_bytes);
    hipMalloc(&cursor, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        stream_write<<<256 * 16, 256>>>(reinterpret_cast<uint4 *>(buf), total_bytes / 16);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    printf("streaming write 16 B/lane: %.0f GB/s\n", total_bytes / ms / 1e6);
    const int rows = 128;
    for (int replicas : {8, 32}) {
        for (int run_words : {16, 32, 48, 64, 128, 256, 512, 1024, 4096}) {
            const int n_regions = rows * replicas;
            const uint64_t region_words = total_bytes / 4 / n_regions;
            const uint64_t runs = total_bytes / 4 / run_words * 9 / 10;       // fill 90 %
            const int rows_per_wave = 16;
            const uint64_t blocks = runs / (8 * rows_per_wave);
            for (int rep = 0; rep < 2; rep++) {
                hipMemset(cursor, 0, 1 << 20);
                hipEventRecord(a);
                scatter<<<static_cast<unsigned>(blocks), 512>>>(buf, cursor, region_words, rows, run_words, rows_per_wave, replicas, 0);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            float ms1 = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                scatter<<<static_cast<unsigned>(blocks), 512>>>(buf, cursor, region_words, rows, run_words, rows_per_wave, replicas, 1);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms1, a, b);
            }
            printf("replicas %2d  run %5d B  : with atomics %6.0f GB/s (%.2f ms)   without %6.0f GB/s (%.2f ms)   %llu runs\n", replicas, run_words * 4,
                   static_cast<double>(runs) * run_words * 4 / ms / 1e6, ms, static_cast<double>(runs) * run_words * 4 / ms1 / 1e6, ms1,
                   static_cast<unsigned long long>(runs));
        }
    }
    return 0;
}
