// Host-side check of profiles/ab/me_mme_fx.hpp (the integer arithmetic of the matrix-pipe MME kernel):
// digit features -> column sums over an accepted subset (what v_mfma_i32_16x16x64_i8 accumulates) -> moments about the query,
// against (a) exact __int128 arithmetic on the same fixed-point coordinates and (b) the fp64 sums the vector kernel forms.
// Built and run by tests/test_mme_fx_cpu.py (g++, no GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../profiles/ab/me_mme_fx.hpp"

using namespace me::fx;

static int fails = 0;
#define CHECK(c, ...)                                  \
    do {                                               \
        if (!(c)) {                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                  \
            std::printf("\n");                         \
            if (++fails > 20) std::exit(1);            \
        }                                              \
    } while (0)

int main() {
    std::mt19937_64 rng(12345);
    // --- fix(): exact for coordinates on the lattice, correctly rounded otherwise
    for (int s : {30, 44, 51}) {
        std::uniform_real_distribution<double> U(-1500.0, 1500.0);
        for (int i = 0; i < 40000; ++i) {
            double v = U(rng);
            if (i % 7 == 0) v = std::ldexp(v, -(int) (rng() % 40));  // small magnitudes
            const long long f = fix(v, s);
            const long double want = (long double) v * std::ldexp(1.0L, s);
            CHECK(std::fabs((double) ((long double) f - want)) <= 0.5000001, "fix(%a, %d) = %lld", v, s, f);
            if (std::fabs(v) >= std::ldexp(1.0, 52 - s)) CHECK((long double) f == want, "fix not exact for %a at s = %d", v, s);
        }
    }
    CHECK(fix(0.0, 51) == 0 && fix(-0.0, 51) == 0 && fix(5e-324, 51) == 0, "zeros");
    CHECK(fix(1.0, 51) == (1LL << 51) && fix(-2047.5, 51) == -(4095LL << 50), "powers");

    // --- scenes: a cloud somewhere within +-1100 m, a query, neighbours within r
    for (int scene = 0; scene < 300; ++scene) {
        const double r = scene % 3 == 0 ? 0.1 : (scene % 3 == 1 ? 0.025 : 1.0);
        std::uniform_real_distribution<double> C(-1000.0, 1000.0), D(-1.0, 1.0);
        const double cx = scene % 5 == 0 ? D(rng) * 1e-3 : C(rng), cy = C(rng), cz = C(rng) * 0.02;
        const double origin[3] = {-1100.0 - 0.37, -1100.0 + 0.11, -40.0};
        const int kmax = 1 << 12;
        const int s = choose_scale(1101.0, 2201.0, r, kmax);
        CHECK(s >= 44 && s <= 51, "scale %d", s);
        Frame fr{fix(origin[0], s), fix(origin[1], s), fix(origin[2], s), s};
        const int n = 20 + (int) (rng() % 400);
        std::vector<double> px(n), py(n), pz(n);
        std::vector<unsigned long long> X(n), Y(n), Z(n);
        std::vector<unsigned char> feat((size_t) n * kCols);
        const bool planar = scene % 2 == 0;
        for (int i = 0; i < n; ++i) {
            px[i] = cx + D(rng) * r * 1.2;
            py[i] = cy + D(rng) * r * 1.2;
            pz[i] = cz + (planar ? D(rng) * 1e-3 : D(rng) * r * 1.2);
            X[i] = (unsigned long long) (fix(px[i], s) - fr.ox);
            Y[i] = (unsigned long long) (fix(py[i], s) - fr.oy);
            Z[i] = (unsigned long long) (fix(pz[i], s) - fr.oz);
            CHECK(X[i] < (1ULL << 62) && Y[i] < (1ULL << 62) && Z[i] < (1ULL << 62), "range");
            point_features(X[i], Y[i], Z[i], &feat[(size_t) i * kCols]);
            // the digit string of X is X itself; of a V it is V mod 2^72 in [-2^71, 2^71)
            i128 v = 0;
            for (int d = 7; d >= 0; --d) v = v * 256 + (signed char) feat[(size_t) i * kCols + d];
            CHECK(v == (i128) X[i], "X digits");
            unsigned long long lo;
            unsigned int hi8;
            moment72(X[i], Y[i], lo, hi8);
            i128 w = (signed char) feat[(size_t) i * kCols + 72 + 1];
            for (int d = 7; d >= 0; --d) w = w * 256 + (signed char) feat[(size_t) i * kCols + 24 + 8 * 1 + d];
            const u128 V = ((u128) hi8 << 64) | lo;
            CHECK((((u128) w - V) & (((u128) 1 << 72) - 1)) == 0, "V_xy digits mod 2^72");
            CHECK(w >= -((i128) 129 << 64) && w < ((i128) 1 << 71), "V_xy digit range");  // balanced digits: [-128 (256^9 - 1) / 255, 127 (...) / 255]
        }
        // query = point 0; accepted = strict d2 < r2 like the kernel (the set itself does not matter here)
        const int q = 0;
        long long k = 0;
        int col[kCols] = {0};
        double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
        i128 e1[3] = {0, 0, 0}, e2[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const double dx = px[i] - px[q], dy = py[i] - py[q], dz = pz[i] - pz[q];
            if (!((dx * dx + dy * dy) + dz * dz < r * r)) continue;
            ++k;
            for (int c = 0; c < kCols; ++c) col[c] += (signed char) feat[(size_t) i * kCols + c];  // mask 1 (the kernel: -1, negated)
            s1[0] += dx; s1[1] += dy; s1[2] += dz;
            s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz; s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
            const i128 ex = (i128) X[i] - (i128) X[q], ey = (i128) Y[i] - (i128) Y[q], ez = (i128) Z[i] - (i128) Z[q];
            e1[0] += ex; e1[1] += ey; e1[2] += ez;
            e2[0] += ex * ex; e2[1] += ex * ey; e2[2] += ex * ez; e2[3] += ey * ey; e2[4] += ey * ez; e2[5] += ez * ez;
        }
        CHECK(col[78] == k && col[79] == 0, "count column");
        i128 S1[3], M[6];
        for (int a = 0; a < 3; ++a) {
            S1[a] = 0;
            for (int d = 7; d >= 0; --d) S1[a] = S1[a] * 256 + col[8 * a + d];
        }
        for (int m = 0; m < 6; ++m) {
            M[m] = col[72 + m];
            for (int d = 7; d >= 0; --d) M[m] = M[m] * 256 + col[24 + 8 * m + d];
        }
        const unsigned long long S1lo[3] = {(unsigned long long) S1[0], (unsigned long long) S1[1], (unsigned long long) S1[2]};
        const Moments mo = moments_about_query(k, S1lo, M, X[q], Y[q], Z[q], s);
        const double u1 = std::ldexp(1.0, -s), u2 = std::ldexp(1.0, -2 * s);
        for (int a = 0; a < 3; ++a) {
            CHECK(mo.s1[a] == (double) (long long) e1[a] * u1, "s1[%d] not exact", a);  // (|e1| < 2^63 here; an independent conversion)
            // (the fp64 sums see the coordinates exactly; the lattice moves a coordinate of magnitude < 2^(52-s) by <= 2^-(s+1))
            CHECK(std::fabs(mo.s1[a] - s1[a]) <= (double) k * std::ldexp(1.0, 1 - s) + 1e-13 * std::fabs(s1[a]), "s1[%d] vs fp64: %.17g vs %.17g", a, mo.s1[a], s1[a]);
        }
        for (int m = 0; m < 6; ++m) {
            const double exact = (double) ((long double) (long long) (e2[m] >> 40) * 0x1p40L + (long double) (long long) (e2[m] & (((i128) 1 << 40) - 1))) * u2;  // independent of i128_to_double
            // rounding (b): each product loses at most 2^39 units
            CHECK(std::fabs(mo.s2[m] - exact) <= (double) k * std::ldexp(1.0, 39) * u2 * 1.0001 + std::fabs(exact) * 2.3e-16,
                  "s2[%d]: %g vs exact %g (k = %lld, s = %d)", m, mo.s2[m], exact, k, s);
            CHECK(std::fabs(mo.s2[m] - s2[m]) <= (double) k * 1.2 * r * std::ldexp(1.0, 3 - s) + 1e-13 * r * r * (double) k, "s2[%d] vs fp64: %.17g vs %.17g", m, mo.s2[m], s2[m]);
        }
    }
    // --- choose_scale refuses what does not fit
    CHECK(choose_scale(1e9, 2e9, 0.1, 1000) >= 30 || choose_scale(1e9, 2e9, 0.1, 1000) == -1, "huge extent");
    CHECK(choose_scale(1100, 2200, 0.1, 5e4) <= 51, "dense");
    CHECK(choose_scale(1e12, 2e12, 0.1, 100) == -1, "absurd extent must be refused");
    if (fails == 0) std::printf("OK\n");
    return fails ? 1 : 0;
}
