// issue_rates.hip (round 4, VERDICT round 3 "next 1") — VALU / LDS / MFMA issue intervals and MFMA || VALU overlap on gfx950,
// with loop bodies written as ONE `asm volatile` block each: what is timed is exactly the instruction sequence in the source
// (no v_accvgpr_* moves, no SLP-packed v_pk_*, no s_nop the compiler adds between them; the disassembled loops are committed
// next to the numbers: profiles/ubench/issue_rates.isa.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates > r04_issue_rates.txt
//
// Every kernel: each wave reads s_memtime, runs `iters` iterations of its body, reads s_memtime again.  Blocks have 256 threads
// (one wave per SIMD); W blocks per CU give W waves per SIMD.  Reported: shader cycles per instruction PER SIMD
//   = mean wave cycles / (iters * instructions per iteration * W).
// "pair" kernels: 512 / 1024-thread blocks whose waves 0-3 (8-11) run an MFMA-only loop and waves 4-7 (12-15) a VALU-only loop
// on the same four SIMDs (HW_ID is recorded and checked), the wave-specialised arrangement the guide says overlaps.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

struct Rec {
    unsigned long long cyc;
    unsigned int hwid, role;
};

__device__ __forceinline__ unsigned int hw_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

// ---- loop bodies -------------------------------------------------------------------------------------------------------
// eight independent registers %0..%7, 32 instructions per iteration
#define X8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" \
                     op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"
#define X32(op, tail) X8(op, tail) X8(op, tail) X8(op, tail) X8(op, tail)
#define C8(op) op " vcc, %0, %8\n" op " vcc, %1, %8\n" op " vcc, %2, %8\n" op " vcc, %3, %8\n" op " vcc, %4, %8\n" op " vcc, %5, %8\n" op " vcc, %6, %8\n" op " vcc, %7, %8\n"
#define C32(op) C8(op) C8(op) C8(op) C8(op)

enum Mode {
    M_FMA32, M_PKFMA32, M_FMA64, M_ADD64, M_MUL64, M_CMP32, M_CMP64, M_CNDMASK, M_PERM, M_LSHLOR, M_MAD64, M_MULLO, M_MULHI, M_CVT_F32_F64, M_CNDMASK_SGPR, M_ADDC, M_CMP32_SGPR, M_BFE, M_MUL24, M_AND, M_READLANE, M_FMA32_DEP,
    M_DS128_BCAST, M_DS128_4ADDR, M_DS128_LANE, M_DS64_BCAST, M_DSW8,
    M_MFMA_F32, M_MFMA_I8, M_MFMA_BF16, M_MFMA_F64, M_MFMA_I8_32,
    M_MIX_I8_F32_2, M_MIX_I8_F32_4, M_MIX_I8_F32_6, M_MIX_I8_F64_2, M_MIX_I8_F64_4, M_MIX_F32_F32_4, M_MIX_F32_F32_8, M_MIX_F32_F64_4,
    M_MIX_BF16_F32_4, M_MIX_I8_CMP_4,
    M_COUNT
};

struct Info {
    const char *name;
    int per_iter;      // timed instructions of the first kind per iteration
    int per_iter2;     // fillers per iteration (mixed bodies)
    const char *what;
};
static const Info kInfo[M_COUNT] = {
    {"v_fma_f32", 32, 0, "32 x v_fma_f32, 8 independent chains"},
    {"v_pk_fma_f32", 32, 0, "32 x v_pk_fma_f32 (2 FMAs per lane each)"},
    {"v_fma_f64", 32, 0, "32 x v_fma_f64"},
    {"v_add_f64", 32, 0, "32 x v_add_f64"},
    {"v_mul_f64", 32, 0, "32 x v_mul_f64"},
    {"v_cmp_lt_f32", 32, 0, "32 x v_cmp_lt_f32 vcc"},
    {"v_cmp_lt_f64", 32, 0, "32 x v_cmp_lt_f64 vcc"},
    {"v_cndmask_b32", 32, 0, "32 x v_cndmask_b32 (vcc)"},
    {"v_perm_b32", 32, 0, "32 x v_perm_b32"},
    {"v_lshl_or_b32", 32, 0, "32 x v_lshl_or_b32"},
    {"v_mad_u64_u32", 32, 0, "32 x v_mad_u64_u32"},
    {"v_mul_lo_u32", 32, 0, "32 x v_mul_lo_u32"},
    {"v_mul_hi_u32", 32, 0, "32 x v_mul_hi_u32"},
    {"v_cvt_f32_f64", 32, 0, "32 x v_cvt_f32_f64"},
    {"v_cndmask_b32 (sgpr pair)", 32, 0, "32 x v_cndmask_b32 with the mask in s[20:21] (VOP3)"},
    {"v_addc_co_u32", 32, 0, "32 x v_addc_co_u32 x, vcc, x, x, vcc"},
    {"v_cmp_lt_f32 -> sgpr", 32, 0, "32 x v_cmp_lt_f32 s[20:21] (VOP3)"},
    {"v_bfe_u32", 32, 0, "32 x v_bfe_u32"},
    {"v_mul_u32_u24", 32, 0, "32 x v_mul_u32_u24"},
    {"v_and_b32", 32, 0, "32 x v_and_b32"},
    {"v_readlane_b32", 32, 0, "32 x v_readlane_b32 to 8 sgprs"},
    {"v_fma_f32 dependent", 32, 0, "32 x v_fma_f32, ONE chain"},
    {"ds_read_b128 bcast", 8, 0, "8 x ds_read_b128, wave-uniform address, s_waitcnt lgkmcnt(0) per 8"},
    {"ds_read_b128 4addr", 8, 0, "8 x ds_read_b128, 4 addresses (lane/16), adjacent 16-byte records"},
    {"ds_read_b128 lane", 8, 0, "8 x ds_read_b128, one record per lane (1 KiB per instruction)"},
    {"ds_read_b64 bcast", 8, 0, "8 x ds_read_b64, wave-uniform address"},
    {"ds_write_b8", 8, 0, "8 x ds_write_b8, lane-consecutive bytes"},
    {"v_mfma_f32_16x16x4_f32", 8, 0, "8 MFMAs on 4 named accumulators (VGPR), no moves"},
    {"v_mfma_i32_16x16x64_i8", 8, 0, "8 MFMAs on 4 named accumulators"},
    {"v_mfma_f32_16x16x32_bf16", 8, 0, "8 MFMAs on 4 named accumulators"},
    {"v_mfma_f64_16x16x4_f64", 8, 0, "8 MFMAs on 4 named accumulators"},
    {"v_mfma_i32_32x32x32_i8", 4, 0, "4 MFMAs on 2 named accumulators (16 regs each)"},
    {"i8 MFMA + 2 v_fma_f32", 4, 8, "4 x (v_mfma_i32_16x16x64_i8 ; 2 v_fma_f32)"},
    {"i8 MFMA + 4 v_fma_f32", 4, 16, "4 x (v_mfma_i32_16x16x64_i8 ; 4 v_fma_f32)"},
    {"i8 MFMA + 6 v_fma_f32", 4, 24, "4 x (v_mfma_i32_16x16x64_i8 ; 6 v_fma_f32)"},
    {"i8 MFMA + 2 v_fma_f64", 4, 8, "4 x (v_mfma_i32_16x16x64_i8 ; 2 v_fma_f64)"},
    {"i8 MFMA + 4 v_fma_f64", 4, 16, "4 x (v_mfma_i32_16x16x64_i8 ; 4 v_fma_f64)"},
    {"f32 MFMA + 4 v_fma_f32", 4, 16, "4 x (v_mfma_f32_16x16x4_f32 ; 4 v_fma_f32)"},
    {"f32 MFMA + 8 v_fma_f32", 4, 32, "4 x (v_mfma_f32_16x16x4_f32 ; 8 v_fma_f32)  [round 3's mix, 32 fillers per 4 MFMAs]"},
    {"f32 MFMA + 4 v_fma_f64", 4, 16, "4 x (v_mfma_f32_16x16x4_f32 ; 4 v_fma_f64)"},
    {"bf16 MFMA + 4 v_fma_f32", 4, 16, "4 x (v_mfma_f32_16x16x32_bf16 ; 4 v_fma_f32)"},
    {"i8 MFMA + 4 v_cmp/cndmask", 4, 16, "4 x (v_mfma_i32_16x16x64_i8 ; 2 x (v_cmp_lt_f32 vcc ; v_cndmask_b32))"},
};

#define F4A "v_fma_f32 %4, %4, %12, %13\nv_fma_f32 %5, %5, %12, %13\nv_fma_f32 %6, %6, %12, %13\nv_fma_f32 %7, %7, %12, %13\n"
#define F4B "v_fma_f32 %8, %8, %12, %13\nv_fma_f32 %9, %9, %12, %13\nv_fma_f32 %10, %10, %12, %13\nv_fma_f32 %11, %11, %12, %13\n"
#define F2A "v_fma_f32 %4, %4, %12, %13\nv_fma_f32 %5, %5, %12, %13\n"
#define F2B "v_fma_f32 %6, %6, %12, %13\nv_fma_f32 %7, %7, %12, %13\n"
#define F2C "v_fma_f32 %8, %8, %12, %13\nv_fma_f32 %9, %9, %12, %13\n"
#define F2D "v_fma_f32 %10, %10, %12, %13\nv_fma_f32 %11, %11, %12, %13\n"
#define D4A "v_fma_f64 %4, %4, %12, %13\nv_fma_f64 %5, %5, %12, %13\nv_fma_f64 %6, %6, %12, %13\nv_fma_f64 %7, %7, %12, %13\n"
#define D4B "v_fma_f64 %8, %8, %12, %13\nv_fma_f64 %9, %9, %12, %13\nv_fma_f64 %10, %10, %12, %13\nv_fma_f64 %11, %11, %12, %13\n"
#define D2A "v_fma_f64 %4, %4, %12, %13\nv_fma_f64 %5, %5, %12, %13\n"
#define D2B "v_fma_f64 %6, %6, %12, %13\nv_fma_f64 %7, %7, %12, %13\n"
#define D2C "v_fma_f64 %8, %8, %12, %13\nv_fma_f64 %9, %9, %12, %13\n"
#define D2D "v_fma_f64 %10, %10, %12, %13\nv_fma_f64 %11, %11, %12, %13\n"
#define C2A "v_cmp_lt_f32 vcc, %4, %12\nv_cndmask_b32 %5, %5, %13, vcc\nv_cmp_lt_f32 vcc, %6, %12\nv_cndmask_b32 %7, %7, %13, vcc\n"
#define C2B "v_cmp_lt_f32 vcc, %8, %12\nv_cndmask_b32 %9, %9, %13, vcc\nv_cmp_lt_f32 vcc, %10, %12\nv_cndmask_b32 %11, %11, %13, vcc\n"
#define MI8(n) "v_mfma_i32_16x16x64_i8 %" #n ", %14, %15, %" #n "\n"
#define MF32(n) "v_mfma_f32_16x16x4_f32 %" #n ", %14, %15, %" #n "\n"
#define MBF(n) "v_mfma_f32_16x16x32_bf16 %" #n ", %14, %15, %" #n "\n"

template <int MODE>
__device__ __forceinline__ void body(int iters, float a, float b, float *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int lane = threadIdx.x & 63;
    if constexpr (MODE == M_FMA32 || MODE == M_CMP32 || MODE == M_CNDMASK || MODE == M_PERM || MODE == M_LSHLOR || MODE == M_MULLO ||
                  MODE == M_MULHI || MODE == M_CNDMASK_SGPR || MODE == M_ADDC || MODE == M_CMP32_SGPR || MODE == M_BFE || MODE == M_MUL24 || MODE == M_AND || MODE == M_READLANE || MODE == M_FMA32_DEP) {
        float x0 = a + lane, x1 = a * 2 + lane, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == M_FMA32)
                asm volatile(X32("v_fma_f32", ", %8, %9") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_CMP32)
                asm volatile(C32("v_cmp_lt_f32") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "vcc");
            if constexpr (MODE == M_CNDMASK)
                asm volatile(X32("v_cndmask_b32", ", %8, vcc") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "vcc");
            if constexpr (MODE == M_PERM)
                asm volatile(X32("v_perm_b32", ", %8, %9") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_LSHLOR)
                asm volatile(X32("v_lshl_or_b32", ", 1, %9") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_MULLO)
                asm volatile(X32("v_mul_lo_u32", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_CNDMASK_SGPR)
                asm volatile(X32("v_cndmask_b32", ", %8, s[20:21]") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "s20", "s21");
            if constexpr (MODE == M_ADDC)
                asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_addc_co_u32 %2, vcc, %2, %2, vcc\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                             "v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_addc_co_u32 %6, vcc, %6, %6, vcc\nv_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                             "v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_addc_co_u32 %2, vcc, %2, %2, vcc\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                             "v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_addc_co_u32 %6, vcc, %6, %6, vcc\nv_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                             "v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_addc_co_u32 %2, vcc, %2, %2, vcc\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                             "v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_addc_co_u32 %6, vcc, %6, %6, vcc\nv_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                             "v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_addc_co_u32 %2, vcc, %2, %2, vcc\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                             "v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_addc_co_u32 %6, vcc, %6, %6, vcc\nv_addc_co_u32 %7, vcc, %7, %7, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "vcc");
            if constexpr (MODE == M_CMP32_SGPR)
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cmp_lt_f32 s[20:21], %2, %8\nv_cmp_lt_f32 s[22:23], %3, %8\nv_cmp_lt_f32 s[20:21], %4, %8\nv_cmp_lt_f32 s[22:23], %5, %8\nv_cmp_lt_f32 s[20:21], %6, %8\nv_cmp_lt_f32 s[22:23], %7, %8\n"
                             "v_cmp_lt_f32 s[20:21], %0, %8\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cmp_lt_f32 s[20:21], %2, %8\nv_cmp_lt_f32 s[22:23], %3, %8\nv_cmp_lt_f32 s[20:21], %4, %8\nv_cmp_lt_f32 s[22:23], %5, %8\nv_cmp_lt_f32 s[20:21], %6, %8\nv_cmp_lt_f32 s[22:23], %7, %8\n"
                             "v_cmp_lt_f32 s[20:21], %0, %8\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cmp_lt_f32 s[20:21], %2, %8\nv_cmp_lt_f32 s[22:23], %3, %8\nv_cmp_lt_f32 s[20:21], %4, %8\nv_cmp_lt_f32 s[22:23], %5, %8\nv_cmp_lt_f32 s[20:21], %6, %8\nv_cmp_lt_f32 s[22:23], %7, %8\n"
                             "v_cmp_lt_f32 s[20:21], %0, %8\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cmp_lt_f32 s[20:21], %2, %8\nv_cmp_lt_f32 s[22:23], %3, %8\nv_cmp_lt_f32 s[20:21], %4, %8\nv_cmp_lt_f32 s[22:23], %5, %8\nv_cmp_lt_f32 s[20:21], %6, %8\nv_cmp_lt_f32 s[22:23], %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "s20", "s21", "s22", "s23");
            if constexpr (MODE == M_BFE)
                asm volatile(X32("v_bfe_u32", ", 4, 4") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_MUL24)
                asm volatile(X32("v_mul_u32_u24", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_AND)
                asm volatile(X32("v_and_b32", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_READLANE)
                asm volatile("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\nv_readlane_b32 s24, %4, 11\nv_readlane_b32 s25, %5, 13\nv_readlane_b32 s26, %6, 15\nv_readlane_b32 s27, %7, 17\n"
                             "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\nv_readlane_b32 s24, %4, 11\nv_readlane_b32 s25, %5, 13\nv_readlane_b32 s26, %6, 15\nv_readlane_b32 s27, %7, 17\n"
                             "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\nv_readlane_b32 s24, %4, 11\nv_readlane_b32 s25, %5, 13\nv_readlane_b32 s26, %6, 15\nv_readlane_b32 s27, %7, 17\n"
                             "v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 7\nv_readlane_b32 s23, %3, 9\nv_readlane_b32 s24, %4, 11\nv_readlane_b32 s25, %5, 13\nv_readlane_b32 s26, %6, 15\nv_readlane_b32 s27, %7, 17\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if constexpr (MODE == M_FMA32_DEP)
                asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n"
                             "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n"
                             "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n"
                             "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            if constexpr (MODE == M_MULHI)
                asm volatile(X32("v_mul_hi_u32", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
        }
        *sink = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if constexpr (MODE == M_PKFMA32) {
        f32x2 x0 = {a, a}, x1 = x0 * 2, x2 = x0 * 3, x3 = x0 * 4, x4 = x0 * 5, x5 = x0 * 6, x6 = x0 * 7, x7 = x0 * 8, bb = {b, b}, aa = {a, a};
        for (int it = 0; it < iters; ++it)
            asm volatile(X32("v_pk_fma_f32", ", %8, %9") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bb), "v"(aa));
        *sink = x0[0] + x1[0] + x2[0] + x3[0] + x4[0] + x5[0] + x6[0] + x7[1];
    } else if constexpr (MODE == M_FMA64 || MODE == M_ADD64 || MODE == M_MUL64 || MODE == M_CMP64 || MODE == M_MAD64) {
        double x0 = a + lane, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8, bd = b, ad = a;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == M_FMA64)
                asm volatile(X32("v_fma_f64", ", %8, %9") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bd), "v"(ad));
            if constexpr (MODE == M_ADD64)
                asm volatile(X32("v_add_f64", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bd), "v"(ad));
            if constexpr (MODE == M_MUL64)
                asm volatile(X32("v_mul_f64", ", %8") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bd), "v"(ad));
            if constexpr (MODE == M_CMP64)
                asm volatile(C32("v_cmp_lt_f64") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bd), "v"(ad) : "vcc");
            if constexpr (MODE == M_MAD64) {
                unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             "v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             "v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             "v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ua), "v"(ub) : "vcc");
            }
        }
        *sink = (float) (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    } else if constexpr (MODE == M_CVT_F32_F64) {
        double d0 = a, d1 = a * 2, d2 = a * 3, d3 = a * 4;
        float y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0;
        for (int it = 0; it < iters; ++it)
            asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %9\nv_cvt_f32_f64 %6, %10\nv_cvt_f32_f64 %7, %11\n"
                         "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %9\nv_cvt_f32_f64 %6, %10\nv_cvt_f32_f64 %7, %11\n"
                         "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %9\nv_cvt_f32_f64 %6, %10\nv_cvt_f32_f64 %7, %11\n"
                         "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %8\nv_cvt_f32_f64 %5, %9\nv_cvt_f32_f64 %6, %10\nv_cvt_f32_f64 %7, %11\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        *sink = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
    } else if constexpr (MODE == M_DS128_BCAST || MODE == M_DS128_4ADDR || MODE == M_DS128_LANE) {
        // the wave's own 1 KiB of LDS
        const unsigned int base = (unsigned int) (size_t) lds + (threadIdx.x >> 6 & 3) * 1024;
        const unsigned int addr = base + (MODE == M_DS128_BCAST ? 0u : MODE == M_DS128_4ADDR ? (unsigned int) (lane >> 4) * 16u : (unsigned int) lane * 16u);
        f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
        float s = 0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8\nds_read_b128 %2, %8\nds_read_b128 %3, %8\n"
                         "ds_read_b128 %4, %8\nds_read_b128 %5, %8\nds_read_b128 %6, %8\nds_read_b128 %7, %8\ns_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        }
        s += r0[0] + r1[1] + r2[2] + r3[3] + r4[0] + r5[1] + r6[2] + r7[3];
        *sink = s;
    } else if constexpr (MODE == M_DS64_BCAST) {
        const unsigned int addr = (unsigned int) (size_t) lds + (threadIdx.x >> 6 & 3) * 1024;
        double r0, r1, r2, r3, r4, r5, r6, r7;
        for (int it = 0; it < iters; ++it)
            asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:8\nds_read_b64 %2, %8 offset:16\nds_read_b64 %3, %8 offset:24\n"
                         "ds_read_b64 %4, %8 offset:32\nds_read_b64 %5, %8 offset:40\nds_read_b64 %6, %8 offset:48\nds_read_b64 %7, %8 offset:56\ns_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        *sink = (float) (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);
    } else if constexpr (MODE == M_DSW8) {
        const unsigned int addr = (unsigned int) (size_t) lds + (threadIdx.x >> 6 & 3) * 1024 + lane;
        const unsigned int v = lane;
        for (int it = 0; it < iters; ++it)
            asm volatile("ds_write_b8 %0, %1\nds_write_b8 %0, %1 offset:64\nds_write_b8 %0, %1 offset:128\nds_write_b8 %0, %1 offset:192\n"
                         "ds_write_b8 %0, %1 offset:256\nds_write_b8 %0, %1 offset:320\nds_write_b8 %0, %1 offset:384\nds_write_b8 %0, %1 offset:448\ns_waitcnt lgkmcnt(0)\n"
                         :: "v"(addr), "v"(v) : "memory");
        *sink = lds[lane];
    } else if constexpr (MODE == M_MFMA_F32) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float av = a + lane * 1e-6f;
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\nv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\nv_mfma_f32_16x16x4_f32 %2, %4, %5, %2\nv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                         "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\nv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\nv_mfma_f32_16x16x4_f32 %2, %4, %5, %2\nv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(b));
        *sink = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (MODE == M_MFMA_I8 || MODE == M_MFMA_BF16) {
        i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const i32x4 av = {lane, 0x01010101, lane * 3, 0}, bv = {0x01000100, lane, 1, 2};
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == M_MFMA_I8)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\nv_mfma_i32_16x16x64_i8 %1, %4, %5, %1\nv_mfma_i32_16x16x64_i8 %2, %4, %5, %2\nv_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n"
                             "v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\nv_mfma_i32_16x16x64_i8 %1, %4, %5, %1\nv_mfma_i32_16x16x64_i8 %2, %4, %5, %2\nv_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\nv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\nv_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\nv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n"
                             "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\nv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\nv_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\nv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
        }
        *sink = (float) (c0[0] + c1[1] + c2[2] + c3[3]);
    } else if constexpr (MODE == M_MFMA_F64) {
        f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double av = a + lane * 1e-9, bv = b;
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\nv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\nv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\nv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
                         "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\nv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\nv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\nv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
        *sink = (float) (c0[0] + c1[1] + c2[2] + c3[3]);
    } else if constexpr (MODE == M_MFMA_I8_32) {
        typedef int i32x16 __attribute__((ext_vector_type(16)));
        i32x16 c0 = {0}, c1 = {0};
        const i32x4 av = {lane, 0x01010101, lane * 3, 0}, bv = {0x01000100, lane, 1, 2};
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %2, %3, %0\nv_mfma_i32_32x32x32_i8 %1, %2, %3, %1\nv_mfma_i32_32x32x32_i8 %0, %2, %3, %0\nv_mfma_i32_32x32x32_i8 %1, %2, %3, %1\n"
                         : "+v"(c0), "+v"(c1) : "v"(av), "v"(bv));
        *sink = (float) (c0[0] + c1[5]);
    } else if constexpr (MODE == M_MIX_I8_F32_2 || MODE == M_MIX_I8_F32_4 || MODE == M_MIX_I8_F32_6 || MODE == M_MIX_BF16_F32_4 || MODE == M_MIX_I8_CMP_4) {
        i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const i32x4 av = {lane, 0x01010101, lane * 3, 0}, bv = {0x01000100, lane, 1, 2};
        float x0 = a + lane, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
        for (int it = 0; it < iters; ++it) {
#define MIXOPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a), "v"(av), "v"(bv)
            if constexpr (MODE == M_MIX_I8_F32_2) asm volatile(MI8(0) F2A MI8(1) F2B MI8(2) F2C MI8(3) F2D MIXOPS);
            if constexpr (MODE == M_MIX_I8_F32_4) asm volatile(MI8(0) F4A MI8(1) F4B MI8(2) F4A MI8(3) F4B MIXOPS);
            if constexpr (MODE == M_MIX_I8_F32_6) asm volatile(MI8(0) F4A F2C MI8(1) F2D F4A MI8(2) F4B F2A MI8(3) F2B F4B MIXOPS);
            if constexpr (MODE == M_MIX_BF16_F32_4) asm volatile(MBF(0) F4A MBF(1) F4B MBF(2) F4A MBF(3) F4B MIXOPS);
            if constexpr (MODE == M_MIX_I8_CMP_4) asm volatile(MI8(0) C2A MI8(1) C2B MI8(2) C2A MI8(3) C2B MIXOPS : "vcc");
        }
        *sink = (float) (c0[0] + c1[1] + c2[2] + c3[3]) + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if constexpr (MODE == M_MIX_I8_F64_2 || MODE == M_MIX_I8_F64_4) {
        i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const i32x4 av = {lane, 0x01010101, lane * 3, 0}, bv = {0x01000100, lane, 1, 2};
        double x0 = a + lane, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8, bd = b, ad = a;
        for (int it = 0; it < iters; ++it) {
#define MIXOPSD : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(bd), "v"(ad), "v"(av), "v"(bv)
            if constexpr (MODE == M_MIX_I8_F64_2) asm volatile(MI8(0) D2A MI8(1) D2B MI8(2) D2C MI8(3) D2D MIXOPSD);
            if constexpr (MODE == M_MIX_I8_F64_4) asm volatile(MI8(0) D4A MI8(1) D4B MI8(2) D4A MI8(3) D4B MIXOPSD);
        }
        *sink = (float) (c0[0] + c1[1] + c2[2] + c3[3]) + (float) (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    } else if constexpr (MODE == M_MIX_F32_F32_4 || MODE == M_MIX_F32_F32_8) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float av = a + lane * 1e-6f, bv = b;
        float x0 = a + lane, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == M_MIX_F32_F32_4) asm volatile(MF32(0) F4A MF32(1) F4B MF32(2) F4A MF32(3) F4B MIXOPS);
            if constexpr (MODE == M_MIX_F32_F32_8) asm volatile(MF32(0) F4A F4B MF32(1) F4A F4B MF32(2) F4A F4B MF32(3) F4A F4B MIXOPS);
        }
        *sink = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if constexpr (MODE == M_MIX_F32_F64_4) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float av = a + lane * 1e-6f, bv = b;
        double x0 = a + lane, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8, bd = b, ad = a;
        for (int it = 0; it < iters; ++it) asm volatile(MF32(0) D4A MF32(1) D4B MF32(2) D4A MF32(3) D4B MIXOPSD);
        *sink = c0[0] + c1[1] + c2[2] + c3[3] + (float) (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_one(int iters, float a, float b, Rec *rec, float *sink) {
    float s = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    body<MODE>(iters, a, b, &s);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        Rec r{t1 - t0, hw_id(), 0u};
        rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
    if (s == 12345.678f) sink[0] = s;
}

// wave-specialised pair: first half of the block's waves run body A (matrix), second half body B (vector) — the waves of a
// 512-thread block land two per SIMD (checked through HW_ID on the host side)
template <int MA, int MB, int THREADS>
__global__ void __launch_bounds__(THREADS) k_pair(int iters_a, int iters_b, float a, float b, Rec *rec, float *sink) {
    float s = 0;
    const int wave = threadIdx.x >> 6;
    const unsigned int role = wave >= THREADS / 128 ? 1u : 0u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) body<MA>(iters_a, a, b, &s);
    else body<MB>(iters_b, a, b, &s);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        Rec r{t1 - t0, hw_id(), role};
        rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
    if (s == 12345.678f) sink[0] = s;
}

static Rec *d_rec;
static float *d_sink;
static std::vector<Rec> h_rec;
static int g_cus = 256;

template <int MODE>
static double run_one(int wps, int iters, double *wall_ms = nullptr) {
    const int blocks = g_cus * wps;
    hipLaunchKernelGGL(k_one<MODE>, dim3(blocks), dim3(256), 0, 0, 16, 1.0f, 0.5f, d_rec, d_sink);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_one<MODE>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, 0.5f, d_rec, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (wall_ms) *wall_ms = ms;
    const int nw = blocks * 4;
    CK(hipMemcpy(h_rec.data(), d_rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < nw; ++i) sum += (double) h_rec[i].cyc;
    return sum / nw;  // mean cycles per wave
}

static double g_mhz = 2400.0;  // shader clock for wall -> cycles (hipDeviceProp clockRate; the s_memtime / wall ratio is printed beside it)
template <int MODE>
static void report(int iters) {
    // cycles per instruction per SIMD from the WALL clock: (t(3 * iters) - t(iters)) removes launch overhead and ramp; all
    // SIMDs hold `wps` waves for the whole kernel or the slowest decides — either way this is what a kernel pays.
    // In brackets: the same from the waves' own s_memtime deltas (mean wave ticks / (iters * n * wps)): equal to the wall
    // figure when every wave is resident from start to end and s_memtime ticks at the shader clock.
    const Info &in = kInfo[MODE];
    std::printf("%-28s", in.name);
    for (int wps : {1, 2, 4, 8}) {
        double ms1, ms3;
        run_one<MODE>(wps, iters, &ms1);
        const double cyc = run_one<MODE>(wps, 3 * iters, &ms3);
        const double n = (double) (2 * iters) * (in.per_iter + in.per_iter2) * wps;
        std::printf("  %6.2f [%5.2f]", (ms3 - ms1) * 1e-3 * g_mhz * 1e6 / n, cyc / ((double) 3 * iters * (in.per_iter + in.per_iter2) * wps));
        if (wps == 8) std::printf("   (tick/wall %.0f MHz)", cyc / ms3 * 1e-3);
    }
    std::printf("   | %s\n", in.what);
    std::fflush(stdout);
}

// SIMD cycles per loop iteration from the wall clock (difference of two iteration counts)
template <int MODE>
static double wall_iter(int wps, int iters) {
    double ms1, ms3;
    run_one<MODE>(wps, iters, &ms1);
    run_one<MODE>(wps, 3 * iters, &ms3);
    return (ms3 - ms1) * 1e-3 * g_mhz * 1e6 / ((double) (2 * iters) * wps);
}
// mixed body vs its two legs alone AT THE SAME occupancy; MM / MF: the kernels that time the MFMA (8 per iteration) and the
// filler (32 per iteration) alone
template <int MODE, int MM, int MF>
static void report_mix(int iters) {
    const Info &in = kInfo[MODE];
    std::printf("%-28s", in.name);
    for (int wps : {1, 2, 4, 8}) {
        const double both = wall_iter<MODE>(wps, iters);
        const double m = wall_iter<MM>(wps, iters) / kInfo[MM].per_iter * in.per_iter;
        const double f = wall_iter<MF>(wps, iters) / kInfo[MF].per_iter * in.per_iter2;
        std::printf("  %6.1f (%5.1f + %5.1f, hid %3.0f%%)", both, m, f, 100.0 * (m + f - both) / std::min(m, f));
    }
    std::printf("   | %s\n", in.what);
    std::fflush(stdout);
}

template <int MA, int MB, int THREADS>
static void report_pair(const char *name, int iters_a, int iters_b) {
    const int blocks = g_cus;
    const int nw = blocks * (THREADS / 64);
    auto launch = [&](int ia, int ib) { hipLaunchKernelGGL((k_pair<MA, MB, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, ia, ib, 1.0f, 0.5f, d_rec, d_sink); };
    double res[3][2];
    int shared_ok = 0, shared_n = 0;
    for (int cfg = 0; cfg < 3; ++cfg) {  // 0: A alone (B waves idle), 1: B alone, 2: both
        const int ia = cfg == 1 ? 0 : iters_a, ib = cfg == 0 ? 0 : iters_b;
        launch(cfg == 1 ? 0 : 16, cfg == 0 ? 0 : 16);
        CK(hipDeviceSynchronize());
        launch(ia, ib);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_rec.data(), d_rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
        double s[2] = {0, 0};
        int c[2] = {0, 0};
        for (int i = 0; i < nw; ++i) {
            s[h_rec[i].role] += (double) h_rec[i].cyc;
            c[h_rec[i].role]++;
        }
        res[cfg][0] = s[0] / c[0];
        res[cfg][1] = s[1] / c[1];
        if (cfg == 2) {  // do wave w (role 0) and wave w + half (role 1) of a block share a SIMD?  HW_ID: SIMD_ID = bits 5:4, CU_ID = 11:8, SE 15:13
            const int wpb = THREADS / 64, half = wpb / 2;
            for (int b = 0; b < blocks; ++b)
                for (int w = 0; w < half; ++w) {
                    const unsigned int h0 = h_rec[b * wpb + w].hwid, h1 = h_rec[b * wpb + half + (w % half)].hwid;
                    ++shared_n;
                    if (((h0 >> 4) & 3) == ((h1 >> 4) & 3) && ((h0 >> 8) & 0xf) == ((h1 >> 8) & 0xf)) ++shared_ok;
                }
        }
    }
    const double a_alone = res[0][0], b_alone = res[1][1], a_both = res[2][0], b_both = res[2][1];
    const double both = std::max(a_both, b_both), sum = a_alone + b_alone, shorter = std::min(a_alone, b_alone);
    std::printf("%-44s matrix waves alone %9.0f cyc | vector waves alone %9.0f | together: matrix %9.0f vector %9.0f -> %3.0f %% of the shorter leg hidden (same-SIMD pairs %d / %d)\n",
                name, a_alone, b_alone, a_both, b_both, 100.0 * (sum - both) / shorter, shared_ok, shared_n);
    std::fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    g_mhz = p.clockRate * 1e-3;
    std::printf("# %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, g_cus, p.clockRate);
    CK(hipMalloc(&d_rec, sizeof(Rec) * g_cus * 8 * 16));
    CK(hipMalloc(&d_sink, 64));
    h_rec.resize((size_t) g_cus * 8 * 16);
    const int IT = 4000;
    std::printf("\n== issue interval: cycles per instruction PER SIMD at 1 / 2 / 4 / 8 waves per SIMD (256-thread blocks, one wave per SIMD each): wall clock at clockRate [from s_memtime deltas]\n");
    std::printf("%-28s  %14s  %14s  %14s  %14s\n", "instruction", "1", "2", "4", "8");
    report<M_FMA32>(IT);
    report<M_PKFMA32>(IT);
    report<M_FMA64>(IT);
    report<M_ADD64>(IT);
    report<M_MUL64>(IT);
    report<M_CMP32>(IT);
    report<M_CMP64>(IT);
    report<M_CNDMASK>(IT);
    report<M_PERM>(IT);
    report<M_LSHLOR>(IT);
    report<M_MAD64>(IT);
    report<M_MULLO>(IT);
    report<M_MULHI>(IT);
    report<M_CVT_F32_F64>(IT);
    report<M_CNDMASK_SGPR>(IT);
    report<M_ADDC>(IT);
    report<M_CMP32_SGPR>(IT);
    report<M_BFE>(IT);
    report<M_MUL24>(IT);
    report<M_AND>(IT);
    report<M_READLANE>(IT);
    report<M_FMA32_DEP>(IT);
    report<M_DS128_BCAST>(IT);
    report<M_DS128_4ADDR>(IT);
    report<M_DS128_LANE>(IT);
    report<M_DS64_BCAST>(IT);
    report<M_DSW8>(IT);
    std::printf("\n== matrix instructions, same units\n");
    report<M_MFMA_F32>(IT);
    report<M_MFMA_I8>(IT);
    report<M_MFMA_BF16>(IT);
    report<M_MFMA_F64>(IT / 2);
    report<M_MFMA_I8_32>(IT);
    std::printf("\n== one wave issues both: SIMD cycles per ITERATION (4 MFMAs + fillers) at 1 / 2 / 4 / 8 waves per SIMD, wall clock; in brackets the two legs alone at the same occupancy; hid = (sum - measured) / shorter leg\n");
    report_mix<M_MIX_I8_F32_2, M_MFMA_I8, M_FMA32>(IT);
    report_mix<M_MIX_I8_F32_4, M_MFMA_I8, M_FMA32>(IT);
    report_mix<M_MIX_I8_F32_6, M_MFMA_I8, M_FMA32>(IT);
    report_mix<M_MIX_I8_F64_2, M_MFMA_I8, M_FMA64>(IT);
    report_mix<M_MIX_I8_F64_4, M_MFMA_I8, M_FMA64>(IT);
    report_mix<M_MIX_I8_CMP_4, M_MFMA_I8, M_CMP32>(IT);
    report_mix<M_MIX_BF16_F32_4, M_MFMA_BF16, M_FMA32>(IT);
    report_mix<M_MIX_F32_F32_4, M_MFMA_F32, M_FMA32>(IT);
    report_mix<M_MIX_F32_F32_8, M_MFMA_F32, M_FMA32>(IT);
    report_mix<M_MIX_F32_F64_4, M_MFMA_F32, M_FMA64>(IT);
    std::printf("\n== wave-specialised: matrix-only waves and vector-only waves on the same SIMDs (one block per CU)\n");
    // iteration counts chosen so that both legs take about the same time alone
    report_pair<M_MFMA_F32, M_FMA32, 512>("f32 MFMA || v_fma_f32, 1+1 waves per SIMD", IT, IT * 2);
    report_pair<M_MFMA_F32, M_FMA64, 512>("f32 MFMA || v_fma_f64, 1+1 waves per SIMD", IT, IT * 2);
    report_pair<M_MFMA_I8, M_FMA32, 512>("i8 MFMA  || v_fma_f32, 1+1 waves per SIMD", IT * 2, IT * 2);
    report_pair<M_MFMA_I8, M_FMA64, 512>("i8 MFMA  || v_fma_f64, 1+1 waves per SIMD", IT * 2, IT * 2);
    report_pair<M_MFMA_BF16, M_FMA32, 512>("bf16 MFMA || v_fma_f32, 1+1 waves per SIMD", IT * 2, IT * 2);
    report_pair<M_MFMA_F64, M_FMA64, 512>("f64 MFMA || v_fma_f64, 1+1 waves per SIMD", IT / 2, IT * 2);
    report_pair<M_MFMA_F32, M_FMA32, 1024>("f32 MFMA || v_fma_f32, 2+2 waves per SIMD", IT, IT * 2);
    report_pair<M_MFMA_I8, M_FMA32, 1024>("i8 MFMA  || v_fma_f32, 2+2 waves per SIMD", IT * 2, IT * 2);
    report_pair<M_MFMA_I8, M_DS128_4ADDR, 512>("i8 MFMA  || ds_read_b128 (4 addr), 1+1", IT * 2, IT * 2);
    return 0;
}
