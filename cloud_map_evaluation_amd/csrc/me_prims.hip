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
// needs 524 KB of LDS)
int sort_keys_u64(me_ctx *ctx, const unsigned long long *in, unsigned long long *out, long long n, int begin_bit, int end_bit) {
    if (n <= 0) return ME_OK;
    size_t bytes = 0;
    ME_CHECK(ctx, rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
    ME_CHECK(ctx, ctx->tmp[5].ensure(bytes));
    TimerScope ts(ctx, "sort");
    ME_CHECK(ctx, rocprim::radix_sort_keys(ctx->tmp[5].p, bytes, in, out, (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, ctx->stream));
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
