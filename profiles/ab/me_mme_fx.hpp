// me_mme_fx.hpp — exact fixed-point moments for the matrix-pipe MME kernel (k_mme7, me_mme.hip).
//
// The covariance of a radius neighbourhood (map_eval.cpp:1684-1689) needs k, sum(p) and sum(p p^T) over the ACCEPTED
// candidates of every query: D[query][feature] += accept[query][candidate] * F[candidate][feature] — a contraction over the
// candidates.  fp64 MFMA is slower than the vector unit on gfx950 (profiles/r04_issue_rates.txt: 64.5 cycles per 16x16x4),
// the int8 MFMA is not (16.3 cycles per 16x16x64, and it overlaps with vector work).  So the features are INTEGERS, split
// into signed base-256 digits; the accept mask is 0 / -1; v_mfma_i32_16x16x64_i8 adds the digit columns exactly in int32,
// and the sums are put together again per query at the end of a round.  Everything here is exact integer arithmetic; the
// only roundings are (a) a coordinate of magnitude < 2^(52-s) m to the 2^-s m lattice (s = 51: 4.4e-16 m), (b) the low 40
// bits of a second-moment product (2^-63 relative to r^2 2^2s) and (c) the final int -> double conversions.
//
// Fixed point.  X = fix(p.x) - fix(origin.x) in units of 2^-s m, 0 <= X < 2^62 (the cloud's extent fits: `choose_scale`).
// Features of a point, 80 digit columns (a column = one signed byte per candidate):
//   cols  0..23  the 8 digits of X, of Y, of Z
//   cols 24..71  digits 0..7 of V_xx, V_xy, V_xz, V_yy, V_yz, V_zz,   V_ab = floor((A B + 2^39) / 2^40) mod 2^72
//   cols 72..77  digit 8 of the six V;  col 78 = 1 (the count);  col 79 = 0
// Digits are BALANCED: d_i = byte_i(W + 0x80..80) - 128, so that sum d_i 256^i = W exactly for X (no carry out: X < 2^62)
// and = W mod 2^72, as a number in [-2^71, 2^71), for the V.
// Why mod 2^72 is enough: the moments are wanted about the QUERY, sum (A - Aq)(B - Bq) = S_ab - Aq S_b - Bq S_a + k Aq Bq,
// whose magnitude is < k (r 2^s)^2 < 2^111 (`choose_scale` keeps it there); computed mod 2^112 from S_ab mod 2^112 =
// 2^40 (sum V_ab mod 2^72) it is therefore exact up to the rounding (b).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define ME_FX_HD __host__ __device__ __forceinline__
#else
#define ME_FX_HD inline
#endif

namespace me {
namespace fx {

typedef unsigned __int128 u128;
typedef __int128 i128;

constexpr int kCols = 80;          // digit columns per point
constexpr int kTiles = kCols / 16; // MFMA tiles of 16 columns
constexpr int kChunk = 16;         // candidates per feature chunk (cells are padded to whole chunks)
constexpr int kChunkBytes = kCols * kChunk;  // [80 columns][16 candidates] signed bytes

struct Frame {
    long long ox, oy, oz;  // fix(origin) per axis
    int s;                 // binary scale: 1 unit = 2^-s m
};

// round(v * 2^s) for a finite double with |v| * 2^s < 2^63, in integer arithmetic (exact whenever ulp(v) >= 2^-s)
ME_FX_HD long long fix(double v, int s) {
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (unsigned long long) __double_as_longlong(v);
#else
    std::memcpy(&b, &v, 8);
#endif
    const int ex = (int) ((b >> 52) & 0x7ff);
    if (ex == 0) return 0;  // zero / denormal: below every lattice this file uses
    const unsigned long long m = (b & 0xfffffffffffffULL) | (1ULL << 52);  // v = +- m * 2^(ex - 1075)
    const int sh = ex - 1075 + s;
    unsigned long long r;
    if (sh >= 0) r = m << sh;                                   // (caller: |v| 2^s < 2^63, so sh <= 10)
    else if (sh <= -64) r = 0;
    else r = (m + (1ULL << (-sh - 1))) >> (-sh);                // round half away from zero
    return (b >> 63) ? -(long long) r : (long long) r;
}

// Largest scale s <= 51 such that (i) every |coordinate| 2^s < 2^62, (ii) every X = fix(p) - fix(origin) < 2^62 and
// (iii) kmax (radius 2^s)^2 < 2^110 and kmax radius 2^s < 2^62 (the mod-2^112 / mod-2^64 arguments with a factor 2 to spare;
// kmax bounds the accepted neighbours of one query).  Returns -1 when no s >= 30 does (the caller falls back to the vector kernel).
inline int choose_scale(double max_abs_coord, double extent, double radius, double kmax) {
    for (int s = 51; s >= 30; --s) {
        const double f = std::ldexp(1.0, s);
        if (max_abs_coord * f >= 0x1p62 || (extent + 1.0) * f >= 0x1p62) continue;
        const double rr = (radius * 1.01) * f;
        if (kmax * rr * rr >= 0x1p110 || kmax * rr >= 0x1p62) continue;
        return s;
    }
    return -1;
}

// 128-bit product of two 64-bit values
ME_FX_HD u128 mul64(unsigned long long a, unsigned long long b) { return (u128) a * (u128) b; }

// V = floor((A B + 2^39) / 2^40) mod 2^72: lo = its low 64 bits, hi8 = bits 64..71
ME_FX_HD void moment72(unsigned long long a, unsigned long long b, unsigned long long &lo, unsigned int &hi8) {
    const u128 p = mul64(a, b) + ((u128) 1 << 39);
    lo = (unsigned long long) (p >> 40);
    hi8 = (unsigned int) (p >> 104) & 0xffu;
}

// The 80 feature bytes of one point (signed digits, stored as their two's-complement byte)
ME_FX_HD void point_features(unsigned long long X, unsigned long long Y, unsigned long long Z, unsigned char *out) {
    const unsigned long long B8 = 0x8080808080808080ULL;
    auto put8 = [&](unsigned long long w, unsigned char *o) {
        w = (w + B8) ^ B8;  // balanced digits: byte_i(w + bias) - 128 == byte_i(w + bias) ^ 0x80 as a signed byte
        for (int i = 0; i < 8; ++i) o[i] = (unsigned char) (w >> (8 * i));
    };
    put8(X, out + 0);
    put8(Y, out + 8);
    put8(Z, out + 16);
    const unsigned long long A[6] = {X, X, X, Y, Y, Z}, Bv[6] = {X, Y, Z, Y, Z, Z};
    for (int q = 0; q < 6; ++q) {
        unsigned long long lo;
        unsigned int hi8;
        moment72(A[q], Bv[q], lo, hi8);
        // 72-bit balanced digits: (V + bias72) mod 2^72, bias72 = 0x80 x 9
        const unsigned long long w = lo + B8;
        const unsigned int carry = w < lo ? 1u : 0u;
        const unsigned long long wl = w ^ B8;
        for (int i = 0; i < 8; ++i) out[24 + 8 * q + i] = (unsigned char) (wl >> (8 * i));
        out[72 + q] = (unsigned char) (((hi8 + 0x80u + carry) & 0xffu) ^ 0x80u);
    }
    out[78] = 1;
    out[79] = 0;
}

// One query's column sums -> moments about the query, in metres.
// Input: k; S1lo[3] = sum X, Y, Z mod 2^64; M[6] = sum of the 72-bit digit strings of V_xx, xy, xz, yy, yz, zz (exact sums of
// numbers in [-2^71, 2^71): congruent to sum V mod 2^72); the query's own fixed-point coordinates.
struct Moments {
    double s1[3];
    double s2[6];  // xx, xy, xz, yy, yz, zz about the query
};
ME_FX_HD double i128_to_double(i128 v) {
    // magnitude first: hi * 2^64 + lo of a small NEGATIVE number is -2^64 + (2^64 - |v|), and the second term rounds
    const bool neg = v < 0;
    const u128 m = neg ? (u128) 0 - (u128) v : (u128) v;
    const double d = (double) (unsigned long long) (m >> 64) * 18446744073709551616.0 + (double) (unsigned long long) m;
    return neg ? -d : d;
}
ME_FX_HD Moments moments_about_query(long long k, const unsigned long long S1lo[3], const i128 M[6], unsigned long long Xq,
                                     unsigned long long Yq, unsigned long long Zq, int s) {
    // S1lo = the LOW 64 bits of sum X, Y, Z: sum (A - Aq) = S_a - k Aq has magnitude < k r 2^s < 2^62 (`choose_scale`), so it is
    // exact mod 2^64.  With D_a = sum (A - Aq):   sum (A - Aq)(B - Bq) = S_ab - Aq D_b - Bq D_a - k Aq Bq   (mod 2^128, read as a
    // signed 112-bit number): three 64 x 64 -> 128 products per moment instead of 128-bit ones.
    Moments o;
    const unsigned long long Q[3] = {Xq, Yq, Zq};
    const double u1 = ldexp(1.0, -s), u2 = ldexp(1.0, -2 * s);
    long long D[3];
    for (int a = 0; a < 3; ++a) {
        D[a] = (long long) (S1lo[a] - (unsigned long long) k * Q[a]);
        o.s1[a] = (double) D[a] * u1;
    }
    const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
    for (int q = 0; q < 6; ++q) {
        const int a = ia[q], b = ib[q];
        u128 t = (u128) M[q] << 40;
        t -= (u128) ((i128) (long long) Q[a] * (i128) D[b]);
        t -= (u128) ((i128) (long long) Q[b] * (i128) D[a]);
        t -= (u128) (unsigned long long) k * mul64(Q[a], Q[b]);
        const i128 v = (i128) (t << 16) >> 16;
        o.s2[q] = i128_to_double(v) * u2;
    }
    return o;
}

}  // namespace fx
}  // namespace me
