// Microbenchmark (diagnostic, not part of the product): cost of the LDS operations the bin kernels lean on.
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_ops lds_ops.hip ; run: ./lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t *out, int iters)
{
    __shared__ uint32_t stage[128 * 64];
    __shared__ unsigned int cnt[128 * 8];
    for (int i = threadIdx.x; i < 128 * 8; i += 512) cnt[i] = 0;
    for (int i = threadIdx.x; i < 128 * 64; i += 512) stage[i] = 0;
    __syncthreads();
    uint32_t h = mix(blockIdx.x * 512 + threadIdx.x + 1), acc = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        h = mix(h + it);
        const uint32_t b = h >> 25;                    // 0..127
        if (MODE == 0) acc += atomicAdd(&cnt[b], 1u);                              // returning, random of 128
        if (MODE == 1) atomicAdd(&cnt[b], 1u);                                     // non-returning
        if (MODE == 2) acc += atomicAdd(&cnt[lane], 1u);                           // returning, conflict-free
        if (MODE == 3) acc += atomicAdd(&cnt[b * 8 + (lane & 7)], 1u);             // returning, 8 replicas per bucket
        if (MODE == 4) stage[b * 64 + ((it + (h & 3)) & 63)] = h;                  // rows filling in lockstep
        if (MODE == 5) stage[b * 64 + ((it + (h & 3) + b) & 63)] = h;              // same, rows skewed by bucket
        if (MODE == 6) stage[(h >> 8) & 8191] = h;                                 // random word
        if (MODE == 7) { uint32_t p = atomicAdd(&cnt[b], 1u); stage[b * 64 + (p & 63)] = h; }            // the bin append
        if (MODE == 8) { uint32_t p = atomicAdd(&cnt[b], 1u); stage[b * 64 + ((p + b) & 63)] = h; }      // skewed append
        if (MODE == 9) acc += h;                                                  // loop overhead only
    }
    if (acc == 0x12345678u) out[0] = acc + stage[lane] + cnt[lane];
}

template <int MODE>
static void run(const char *name, uint32_t *d, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * 4;
    k<MODE><<<blocks, 512>>>(d, 8);
    hipEventRecord(a);
    k<MODE><<<blocks, 512>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave-ops per CU: blocks * 8 waves * iters / 256 CUs
    const double ops = double(blocks) * 8 * iters / 256.0;
    printf("%-44s %8.3f ms  %7.1f cycles per wave-op per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / ops);
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 4096);
    const int it = 2000;
    run<9>("loop overhead only", d, it);
    run<0>("ds_add_rtn random of 128 counters", d, it);
    run<1>("ds_add (no return) random of 128", d, it);
    run<2>("ds_add_rtn conflict-free (lane)", d, it);
    run<3>("ds_add_rtn 128 x 8 replicas", d, it);
    run<4>("ds_write rows in lockstep", d, it);
    run<5>("ds_write rows skewed by bucket", d, it);
    run<6>("ds_write random word", d, it);
    run<7>("append: add_rtn + write", d, it);
    run<8>("append skewed: add_rtn + write", d, it);
    return 0;
}
