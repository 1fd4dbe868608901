// det.hpp — device-side noise: counter-based Philox4x32-10 streams and bit-reproducible samplers.
//
// Replaces the reference's process-global np.random draws — np.random.dirichlet (MCTS.py:243-245),
// np.random.randint (MCTS.py:208), np.random.choice (MCTS.py:612), np.random.gumbel
// (MCTS_Gumbel.py:593) — with streams keyed by (seed, global game slot, game sequence number, tree,
// event index), so any game can be replayed on any GPU and in any wave order.  Only IEEE-754
// + - * / sqrt on f64 and integer ops are used (compile with -ffp-contract=off), which makes every
// variate bit-identical to the CPU checker's.  log/exp are argument-reduction + polynomial kernels of
// the classic fdlibm form with a fixed operation order.
#pragma once
#include "wave.hpp"

namespace gaz {
namespace det {

enum : uint32_t { P_DIRICHLET = 0, P_TERMINAL_PICK = 1, P_MOVE = 2, P_GUMBEL = 3, P_OPENING = 4 };

struct Event {
    uint32_t key0, key1;  // 64-bit seed
    uint32_t slot;        // global game slot
    uint32_t game_seq;    // k-th game in that slot
    uint32_t event;       // per-tree event counter
    uint32_t tree;        // 0/1 PUCT trees, 2 game-level / Gumbel
    uint32_t purpose;
};

struct U4 { uint32_t x, y, z, w; };

GAZ_DEV U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

GAZ_DEV U4 draw(const Event& e, uint32_t lane, uint32_t attempt) {
    uint32_t c3 = (e.tree << 30) | (e.purpose << 27) | ((lane & 1023u) << 17) | (attempt & 0x1FFFFu);
    return philox(e.slot, e.game_seq, e.event, c3, e.key0, e.key1);
}

GAZ_DEV uint64_t k52(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6); }
GAZ_DEV double u_open(uint32_t a, uint32_t b) { return (double)(2 * k52(a, b) + 1) * (1.0 / 9007199254740992.0); }
GAZ_DEV double u_half(uint32_t a, uint32_t b) { return (double)k52(a, b) * (1.0 / 4503599627370496.0); }

GAZ_DEV double bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
GAZ_DEV uint64_t d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

// natural log, x > 0 finite
GAZ_DEV double dlog(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = 0;
    uint64_t ix = d2bits(x);
    if ((ix >> 52) == 0) { x = x * 18014398509481984.0; ix = d2bits(x); k -= 54; }
    int e = (int)(ix >> 52) - 1023;
    uint64_t m = ix & 0x000FFFFFFFFFFFFFull;
    if (m >= 0x6A09E667F3BCDull) { e += 1; ix = m | 0x3FE0000000000000ull; }
    else { ix = m | 0x3FF0000000000000ull; }
    k += e;
    double f = bits2d(ix) - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

GAZ_DEV double dexp(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 709.0) x = 709.0;
    if (x < -745.0) return 0.0;
    double fk = x * invln2;
    int k = (int)(fk + (fk < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double hi = x - dk * ln2_hi;
    double lo = dk * ln2_lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021) return y * bits2d((uint64_t)(k + 1023) << 52);
    y = y * bits2d((uint64_t)(k + 1023 + 1000) << 52);
    return y * bits2d((uint64_t)(1023 - 1000) << 52);
}

// x ** y for x >= 0 (np.float64 power, MCTS.py:607-608): exp(y * log x); x == 0 -> 0 for y > 0.  Not libm's pow bit for bit
// (relative error ~ |y ln x| * 2^-52); used only for move-sampling weights, where that moves a cdf step by < 1e-13.
GAZ_DEV double dpow(double x, double y) {
    if (x == 0.0) return y > 0.0 ? 0.0 : 1.0;
    if (x == 1.0 || y == 0.0) return 1.0;
    return dexp(y * dlog(x));
}

// standard normal, Marsaglia polar method; attempt = running Philox sub-counter of this lane's variate
GAZ_DEV double normal(const Event& e, uint32_t lane, uint32_t& attempt) {
    for (;;) {
        U4 r = draw(e, lane, attempt++);
        double v1 = 2.0 * u_open(r.x, r.y) - 1.0;
        double v2 = 2.0 * u_open(r.z, r.w) - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s >= 1.0 || s == 0.0) continue;
        return v1 * dsqrt((-2.0 * dlog(s)) / s);
    }
}

// Gamma(alpha, 1): Marsaglia & Tsang with the alpha < 1 boost
GAZ_DEV double gamma(const Event& e, uint32_t lane, double alpha) {
    uint32_t attempt = 0;
    double boost = 1.0, a = alpha;
    if (a < 1.0) {
        U4 r = draw(e, lane, attempt++);
        boost = dexp(dlog(u_open(r.x, r.y)) / alpha);
        a = alpha + 1.0;
    }
    const double d = a - (1.0 / 3.0);
    const double c = 1.0 / dsqrt(9.0 * d);
    for (;;) {
        double x = normal(e, lane, attempt);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        U4 r = draw(e, lane, attempt++);
        double u = u_open(r.x, r.y);
        const double x2 = x * x;                     // squeeze (same accepted pairs as the log test, see oracle/gaz_det.h)
        if (u < 1.0 - 0.0331 * (x2 * x2)) return (d * v) * boost;
        if (dlog(u) < ((0.5 * x) * x + d) - d * v + d * dlog(v)) return (d * v) * boost;
    }
}

// uniform integer in [0, n): floor(k52 * n / 2^52)
GAZ_DEV uint32_t pick(const Event& e, uint32_t n) {
    U4 r = draw(e, 0, 0);
    return (uint32_t)((k52(r.x, r.y) * (uint64_t)n) >> 52);
}
GAZ_DEV double uniform(const Event& e) { U4 r = draw(e, 0, 0); return u_half(r.x, r.y); }
GAZ_DEV double gumbel(const Event& e, uint32_t lane) { U4 r = draw(e, lane, 0); return -dlog(-dlog(u_open(r.x, r.y))); }

// numpy's pairwise add-reduce order (what np.sum does on a contiguous array), serial form, n <= 256.
template <class T>
GAZ_DEV T np_sum_block(const T* a, int n) {   // n <= 128
    if (n < 8) {
        // every caller's array has at least 8 elements: seven independent reads, then the additions in numpy's order
        // (a dependent read -> add loop costs one LDS round trip per element)
        const T v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = a[4], v5 = a[5], v6 = a[6];
        T res = (T)0;
        if (n > 0) res = res + v0;
        if (n > 1) res = res + v1;
        if (n > 2) res = res + v2;
        if (n > 3) res = res + v3;
        if (n > 4) res = res + v4;
        if (n > 5) res = res + v5;
        if (n > 6) res = res + v6;
        return res;
    }
    T r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 = r0 + a[i + 0]; r1 = r1 + a[i + 1]; r2 = r2 + a[i + 2]; r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4]; r5 = r5 + a[i + 5]; r6 = r6 + a[i + 6]; r7 = r7 + a[i + 7];
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
template <class T>
GAZ_DEV T np_pairwise_sum(const T* a, int n) {
    if (n <= 128) return np_sum_block(a, n);
    int n2 = n / 2; n2 -= n2 % 8;                       // numpy splits once for 128 < n <= 256
    return np_sum_block(a, n2) + np_sum_block(a + n2, n - n2);
}

}  // namespace det
}  // namespace gaz
