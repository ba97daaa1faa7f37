// rocPRIM onesweep configurations for the index build's keys-only sort (50 M packed 64-bit words, 36 key bits above 26 index bits):
// the library default (8 bits per pass: 5 passes) against 9 bits per pass (4 passes) and other block shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/ubench/sort_cfg.hip -o /tmp/sort_cfg && /tmp/sort_cfg
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_fill(unsigned long long *k, long long n, int pack_bits, int key_bits) {
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long x = (unsigned long long) i * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    k[i] = ((x & ((1ULL << key_bits) - 1ULL)) << pack_bits) | (unsigned long long) i;
}
__global__ void k_check(const unsigned long long *k, long long n, int pack_bits, int *bad) {
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    if ((k[i] >> pack_bits) > (k[i + 1] >> pack_bits)) atomicAdd(bad, 1);
    else if ((k[i] >> pack_bits) == (k[i + 1] >> pack_bits) && k[i] > k[i + 1]) atomicAdd(bad, 1);  // stable: index order kept
}

template <class Config>
float run(const char *name, unsigned long long *in, unsigned long long *out, long long n, int b0, int b1, int *d_bad) {
    size_t bytes = 0;
    rocprim::radix_sort_keys<Config>(nullptr, bytes, in, out, (size_t) n, (unsigned) b0, (unsigned) b1, 0);
    void *tmp; hipMalloc(&tmp, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a, 0);
        rocprim::radix_sort_keys<Config>(tmp, bytes, in, out, (size_t) n, (unsigned) b0, (unsigned) b1, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (r) best = std::min(best, ms);
    }
    hipMemset(d_bad, 0, 4);
    k_check<<<(unsigned) ((n + 255) / 256), 256>>>(out, n, b0, d_bad);
    int bad = 0; hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("%-34s %7.3f ms  tmp %6.1f MB  %s\n", name, best, bytes / 1e6, bad ? "NOT SORTED" : "ok");
    hipFree(tmp);
    return best;
}

template <unsigned BS, unsigned IPT, unsigned BITS, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::match>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 50000000LL;
    const int pack_bits = 26, key_bits = argc > 2 ? atoi(argv[2]) : 36;
    unsigned long long *in, *out; int *d_bad;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMalloc(&d_bad, 4);
    k_fill<<<(unsigned) ((n + 255) / 256), 256>>>(in, n, pack_bits, key_bits);
    hipDeviceSynchronize();
    printf("n = %lld, bits [%d, %d)\n", n, pack_bits, pack_bits + key_bits);
    run<rocprim::default_config>("default", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<512, 12, 8>>("512x12 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 6, 8>>("1024x6 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    // (9 bits per pass — 36 bits in 4 passes — does not fit: block_radix_rank needs 524 KB of LDS with `match`, 262 KB with `basic`)
    run<Cfg<256, 16, 8>>("256x16 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<512, 16, 8>>("512x16 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 8, 8>>("1024x8 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<256, 24, 8>>("256x24 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 10, 8>>("1024x10 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 12, 8>>("1024x12 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 14, 8>>("1024x14 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 16, 8>>("1024x16 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<512, 20, 8>>("512x20 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<512, 24, 8>>("512x24 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 9, 8>>("1024x9 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    run<Cfg<1024, 7, 8>>("1024x7 8b match", in, out, n, pack_bits, pack_bits + key_bits, d_bad);
    return 0;
}
