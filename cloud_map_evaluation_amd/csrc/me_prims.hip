// me_prims.hip — device-wide sort / scan primitives (rocPRIM) behind plain functions, so that the slow-to-compile
// rocPRIM templates are instantiated in exactly one translation unit.
#include <cstdlib>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "me_internal.hpp"

namespace me {

int sort_pairs_u64_u32(me_ctx *ctx, const unsigned long long *k_in, unsigned long long *k_out,
                       const unsigned int *v_in, unsigned int *v_out, long long n, int begin_bit, int end_bit) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t) n,
                                            (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    TimerScope ts(ctx, "sort");
    ME_CHECK(ctx, rocprim::radix_sort_pairs(ctx->tmp[5].p, bytes, k_in, k_out, v_in, v_out, (size_t) n,
                                            (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
    return ME_OK;
}

// The same sort (stable, ascending keys) by rocPRIM's MERGE sort: a block sort and log2(n / block) merge passes, plain kernels.  For the
// voxel run records (~n / 60 of them) — their radix sort is eight onesweep passes, and a onesweep pass (decoupled lookback) that runs
// beside a kernel filling the chip does not finish before that kernel does: 4 - 15 ms per pass under k_mme3 / k_nn_grid
// (profiles/EXPERIMENTS.md "Round 6").
int sort_pairs_merge_u64_u32(me_ctx *ctx, const unsigned long long *k_in, unsigned long long *k_out, const unsigned int *v_in,
                             unsigned int *v_out, long long n) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::merge_sort(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t) n, rocprim::less<unsigned long long>(), ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    TimerScope ts(ctx, "sort");
    ME_CHECK(ctx, rocprim::merge_sort(ctx->tmp[5].p, bytes, k_in, k_out, v_in, v_out, (size_t) n, rocprim::less<unsigned long long>(), ctx->stream));
    return ME_OK;
}

int exclusive_scan_u32(me_ctx *ctx, const unsigned int *in, unsigned int *out, long long n) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t) n, rocprim::plus<unsigned int>(),
                                          ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    ME_CHECK(ctx, rocprim::exclusive_scan(ctx->tmp[5].p, bytes, in, out, 0u, (size_t) n,
                                          rocprim::plus<unsigned int>(), ctx->stream));
    return ME_OK;
}

// An exclusive scan out of PLAIN kernels (no decoupled lookback): per-block sums of 2048 elements, one small block that scans the block
// sums, a second pass that scans inside the blocks.  Twice the reads of rocPRIM's single-pass scan — and immune to what happens to a
// lookback scan that shares the chip with a kernel filling it (its blocks spin on predecessors that are not scheduled).  For the voxel
// build, which me_run_suite_from runs beside the MME / 1-NN kernels of the other lane.
constexpr int kScanChunk = 2048;
__global__ void __launch_bounds__(256) k_scan_block_sums(const unsigned int *__restrict__ in, long long n, unsigned int *__restrict__ bsum) {
    const long long i0 = (long long) blockIdx.x * kScanChunk;
    unsigned int s = 0;
#pragma unroll
    for (int k = 0; k < kScanChunk / 256; ++k) {
        const long long i = i0 + 256 * k + threadIdx.x;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ unsigned int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void __launch_bounds__(256) k_scan_of_sums(unsigned int *__restrict__ bsum, long long nb) {
    __shared__ unsigned int sm[256];
    unsigned int carry = 0;
    for (long long base = 0; base < nb; base += 256) {
        const long long i = base + threadIdx.x;
        const unsigned int v = i < nb ? bsum[i] : 0u;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inside the 256
            const unsigned int a = threadIdx.x >= (unsigned int) o ? sm[threadIdx.x - o] : 0u;
            __syncthreads();
            sm[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + sm[threadIdx.x] - v;  // exclusive
        carry += sm[255];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan_in_blocks(const unsigned int *__restrict__ in, long long n, const unsigned int *__restrict__ boff,
                                                         unsigned int *__restrict__ out) {
    // thread t owns the 8 consecutive elements i0 + 8 t .. : their sum, an exclusive scan of the 256 sums, then the 8 prefixes
    const long long i0 = (long long) blockIdx.x * kScanChunk + 8LL * threadIdx.x;
    unsigned int v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
        s += v[k];
    }
    __shared__ unsigned int sm[256];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned int a = threadIdx.x >= (unsigned int) o ? sm[threadIdx.x - o] : 0u;
        __syncthreads();
        sm[threadIdx.x] += a;
        __syncthreads();
    }
    unsigned int run = boff[blockIdx.x] + sm[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (i0 + k < n) out[i0 + k] = run;
        run += v[k];
    }
}

int exclusive_scan_u32_plain(me_ctx *ctx, const unsigned int *in, unsigned int *out, long long n) {
    if (n <= 0) return ME_OK;
    const long long nb = (n + kScanChunk - 1) / kScanChunk;
    ME_CHECK(ctx, ctx->tmp[5].ensure((size_t) nb * 4 + 64));
    unsigned int *bsum = ctx->tmp[5].as<unsigned int>();
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned int) nb), dim3(256), 0, ctx->stream, in, n, bsum);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(256), 0, ctx->stream, bsum, nb);
    hipLaunchKernelGGL(k_scan_in_blocks, dim3((unsigned int) nb), dim3(256), 0, ctx->stream, in, n, (const unsigned int *) bsum, out);
    return ME_OK;
}

// out[i] = number of cell starts before i, where i starts a cell when (codes[i] >> shift3) differs from its predecessor's:
// the flags are computed inside the scan (no flag array written and read back)
struct CellStartFlag {
    const unsigned long long *codes;
    int shift3;
    __device__ unsigned int operator()(size_t i) const {
        return (i == 0 || (codes[i] >> shift3) != (codes[i - 1] >> shift3)) ? 1u : 0u;
    }
};

int cell_start_ranks(me_ctx *ctx, const unsigned long long *codes, long long n, int shift3, unsigned int *out) {
    if (n <= 0) return ME_OK;
    auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<size_t>(0), CellStartFlag{codes, shift3});
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::exclusive_scan(nullptr, bytes, flags, out, 0u, (size_t) n, rocprim::plus<unsigned int>(), ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    ME_CHECK(ctx, rocprim::exclusive_scan(ctx->tmp[5].p, bytes, flags, out, 0u, (size_t) n, rocprim::plus<unsigned int>(),
                                          ctx->stream));
    return ME_OK;
}

// out[0 .. *count) = the indices i < n with flags[i] != 0, ascending (a stream compaction; the count stays on the device)
int select_flagged_u32(me_ctx *ctx, const unsigned char *flags, long long n, unsigned int *out, unsigned int *count_device) {
    if (n <= 0) return ME_OK;
    auto idx = rocprim::make_counting_iterator<unsigned int>(0u);
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::select(nullptr, bytes, idx, flags, out, count_device, (size_t) n, ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    ME_CHECK(ctx, rocprim::select(ctx->tmp[5].p, bytes, idx, flags, out, count_device, (size_t) n, ctx->stream));
    return ME_OK;
}

// (nine bits per onesweep pass — four passes over a 36-bit key instead of five — does not build: rocPRIM's rank table for 512 digits
// needs 524 KB of LDS with `match`, 262 KB with `basic`.)  Block shape: rocPRIM has no tuned onesweep entry for gfx950; of the shapes
// that fit (profiles/ubench/sort_cfg.hip, 50 M words, 36 key bits) 1024 threads x 8 keys sorts in 1.39 ms against the default's 1.56.
using SortKeysConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 8, rocprim::block_radix_rank_algorithm::match>>;

int sort_keys_u64(me_ctx *ctx, const unsigned long long *in, unsigned long long *out, long long n, int begin_bit, int end_bit) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::radix_sort_keys<SortKeysConfig>(nullptr, bytes, in, out, (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    TimerScope ts(ctx, "sort");
    ME_CHECK(ctx, rocprim::radix_sort_keys<SortKeysConfig>(ctx->tmp[5].p, bytes, in, out, (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
    return ME_OK;
}

int sort_keys_f64(me_ctx *ctx, const double *in, double *out, long long n) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t) n, 0u, 64u, ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    ME_CHECK(ctx, rocprim::radix_sort_keys(ctx->tmp[5].p, bytes, in, out, (size_t) n, 0u, 64u, ctx->stream));
    return ME_OK;
}

}  // namespace me
