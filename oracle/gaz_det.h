/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported
 * or executed by the product path (grok_alpha_zero_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * gaz_det.h — the *injected noise* specification, CPU side.
 *
 * The reference draws every random number from the process-global numpy
 * MT19937 stream (np.random.dirichlet at MCTS.py:244, np.random.randint at
 * MCTS.py:208, np.random.choice at MCTS.py:612 and Self_Play.py:139,
 * np.random.gumbel at MCTS_Gumbel.py:593) after np.random.seed() with OS entropy
 * (Self_Play.py:221), so it has no reproducible stream a device engine could
 * replay.  Parity is therefore defined through *injection*: the functions below
 * define a counter-based stream keyed by (seed, game slot, game sequence number,
 * tree, event index); the fixture generator patches the reference's np.random
 * calls to draw from this stream (tools/ref_shim.py), and the HIP engine
 * implements the same functions on device (csrc/det.hpp — a separate
 * restatement of this spec, not an include of this file).
 *
 * Everything here is built from IEEE-754 +,-,*,/,sqrt on doubles and 32/64-bit
 * integer arithmetic only (no libm, no FMA contraction: compile with
 * -ffp-contract=off), so CPU and GPU agree bit for bit.
 */
#ifndef GAZ_DET_H
#define GAZ_DET_H
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- Philox4x32-10
 * Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11). */
static inline void gaz_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Stream purposes (3 bits of counter word 3). */
enum { GAZ_P_DIRICHLET = 0, GAZ_P_TERMINAL_PICK = 1, GAZ_P_MOVE = 2, GAZ_P_GUMBEL = 3, GAZ_P_OPENING = 4 };

/* One random "event" of one tree of one game.  counter = (slot, game_seq, event,
 * tree<<30 | purpose<<27 | lane<<17 | attempt); key = 64-bit seed. */
typedef struct {
    uint32_t key[2];
    uint32_t slot;      /* global game slot (rank*G + local slot) */
    uint32_t game_seq;  /* k-th game played in that slot */
    uint32_t event;     /* per-tree event counter, +1 per np.random call replaced */
    uint32_t tree;      /* 0,1 = the two PUCT trees (Self_Play.py:39-57); 2 = game-level / Gumbel */
    uint32_t purpose;
} gaz_event;

static inline void gaz_draw(const gaz_event* e, uint32_t lane, uint32_t attempt, uint32_t out[4]) {
    uint32_t ctr[4];
    ctr[0] = e->slot; ctr[1] = e->game_seq; ctr[2] = e->event;
    ctr[3] = (e->tree << 30) | (e->purpose << 27) | ((lane & 1023u) << 17) | (attempt & 0x1FFFFu);
    gaz_philox(ctr, e->key, out);
}

/* 52-bit integer from two words; uniforms derived from it are exact doubles. */
static inline uint64_t gaz_k52(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 6) << 26) | (uint64_t)(b >> 6); }
/* open interval (0,1): (2k+1) / 2^53 */
static inline double gaz_u_open(uint32_t a, uint32_t b) {
    return (double)(2 * gaz_k52(a, b) + 1) * (1.0 / 9007199254740992.0);
}
/* half-open [0,1): k / 2^52 */
static inline double gaz_u_half(uint32_t a, uint32_t b) {
    return (double)gaz_k52(a, b) * (1.0 / 4503599627370496.0);
}

/* ---------------------------------------------------------------- det_log / det_exp
 * Argument reduction + polynomial in the style of the classic public-domain
 * fdlibm algorithms, written with explicit operation order.  < 1 ulp. */
static inline double gaz_bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t gaz_d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

static inline double gaz_log(double x) {
    /* x > 0 finite assumed (callers guarantee it); handles subnormals. */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = 0;
    uint64_t ix = gaz_d2bits(x);
    if ((ix >> 52) == 0) { /* subnormal: scale by 2^54 */
        x = x * 18014398509481984.0; ix = gaz_d2bits(x); k -= 54;
    }
    int e = (int)(ix >> 52) - 1023;
    uint64_t m = ix & 0x000FFFFFFFFFFFFFull;
    /* normalise mantissa into [sqrt(1/2), sqrt(2)) */
    if (m >= 0x6A09E667F3BCDull) { e += 1; ix = m | 0x3FE0000000000000ull; }
    else { ix = m | 0x3FF0000000000000ull; }
    k += e;
    double f = gaz_bits2d(ix) - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    /* log(1+f) = f - hfsq + s*(hfsq+R) */
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

static inline double gaz_exp(double x) {
    /* finite x assumed; underflows to 0 below -745, saturates above 709. */
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
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
    /* scale by 2^k in two steps so subnormal results round once at the end */
    if (k >= -1021) {
        return y * gaz_bits2d((uint64_t)(k + 1023) << 52);
    } else {
        y = y * gaz_bits2d((uint64_t)(k + 1023 + 1000) << 52);
        return y * gaz_bits2d((uint64_t)(1023 - 1000) << 52);
    }
}

static inline double gaz_sqrt(double x) { return __builtin_sqrt(x); } /* IEEE correctly rounded */
/* x ** y, x >= 0: the device's stand-in for libm pow (csrc/det.hpp dpow), same operation order */
static inline double gaz_pow(double x, double y) {
    if (x == 0.0) return y > 0.0 ? 0.0 : 1.0;
    if (x == 1.0 || y == 0.0) return 1.0;
    return gaz_exp(y * gaz_log(x));
}

/* ---------------------------------------------------------------- samplers */
/* Standard normal by the Marsaglia polar method; *attempt is the running Philox
 * sub-counter of this lane. */
static inline double gaz_normal(const gaz_event* e, uint32_t lane, uint32_t* attempt) {
    for (;;) {
        uint32_t r[4]; gaz_draw(e, lane, (*attempt)++, r);
        double v1 = 2.0 * gaz_u_open(r[0], r[1]) - 1.0;
        double v2 = 2.0 * gaz_u_open(r[2], r[3]) - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s >= 1.0 || s == 0.0) continue;
        return v1 * gaz_sqrt((-2.0 * gaz_log(s)) / s);
    }
}

/* Gamma(alpha, 1), Marsaglia & Tsang (2000) with the alpha<1 boost. */
static inline double gaz_gamma(const gaz_event* e, uint32_t lane, double alpha) {
    uint32_t attempt = 0;
    double boost = 1.0, a = alpha;
    if (a < 1.0) {
        uint32_t r[4]; gaz_draw(e, lane, attempt++, r);
        boost = gaz_exp(gaz_log(gaz_u_open(r[0], r[1])) / alpha);
        a = alpha + 1.0;
    }
    const double d = a - (1.0 / 3.0);
    const double c = 1.0 / gaz_sqrt(9.0 * d);
    for (;;) {
        double x = gaz_normal(e, lane, &attempt);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        uint32_t r[4]; gaz_draw(e, lane, attempt++, r);
        double u = gaz_u_open(r[0], r[1]);
        /* the paper's squeeze: inside the acceptance region of the log test below, so the accepted (x, u) pairs are the same
           (235 M random pairs, 0 disagreements in this arithmetic); it spares two logs for 3 of 4 draws */
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2)) return (d * v) * boost;
        if (gaz_log(u) < ((0.5 * x) * x + d) - d * v + d * gaz_log(v)) return (d * v) * boost;
    }
}

/* Dirichlet(alpha * 1_n): g_i / sum_i g_i, sum taken sequentially i = 0..n-1. */
static inline void gaz_dirichlet(const gaz_event* e, double alpha, int n, double* out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) { out[i] = gaz_gamma(e, (uint32_t)i, alpha); s = s + out[i]; }
    for (int i = 0; i < n; ++i) out[i] = out[i] / s;
}

/* Uniform integer in [0, n): floor(k52 * n / 2^52); replaces np.random.randint(0, n). */
static inline uint32_t gaz_pick(const gaz_event* e, uint32_t n) {
    uint32_t r[4]; gaz_draw(e, 0, 0, r);
    return (uint32_t)((gaz_k52(r[0], r[1]) * (uint64_t)n) >> 52);
}

/* Uniform in [0,1) for categorical sampling; replaces the uniform inside
 * np.random.choice(p=...) (inverse-cdf, searchsorted side='right'). */
static inline double gaz_uniform(const gaz_event* e) {
    uint32_t r[4]; gaz_draw(e, 0, 0, r);
    return gaz_u_half(r[0], r[1]);
}

/* Standard Gumbel(0,1) for lane i: -log(-log(u)), u in (0,1). */
static inline double gaz_gumbel(const gaz_event* e, uint32_t lane) {
    uint32_t r[4]; gaz_draw(e, lane, 0, r);
    return -gaz_log(-gaz_log(gaz_u_open(r[0], r[1])));
}

/* numpy's pairwise float sum (numpy/_core/src/umath/loops_utils.h.src,
 * @TYPE@_pairwise_sum): what np.sum does on a contiguous array; the reference's
 * legal_policy /= np.sum(legal_policy) (Connect4.py:294, Gomoku.py:134,
 * Tictactoe.py:202) goes through it when run without Numba. */
static inline float gaz_np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float res = 0.0f; /* numpy starts from -0.0 for the <8 loop? it uses 0. with identity; adding +0 keeps value */
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    } else {
        int n2 = n / 2; n2 -= n2 % 8;
        return gaz_np_sum_f32(a, n2) + gaz_np_sum_f32(a + n2, n - n2);
    }
}
static inline double gaz_np_sum_f64(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    } else {
        int n2 = n / 2; n2 -= n2 % 8;
        return gaz_np_sum_f64(a, n2) + gaz_np_sum_f64(a + n2, n - n2);
    }
}

#ifdef __cplusplus
}
#endif
#endif
