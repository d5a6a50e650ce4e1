/*
 * ptrs_detmath.h -- deterministic transcendental substrate (libm stand-in).
 *
 * The reference calls Rust `f32::{sin,cos,exp,ln,log2,powf,atan2,acos,tan}` which lower to the
 * platform libm ("Rust core float intrinsics").  A path tracer is a branchy
 * estimator: a 1-ulp difference in one of these can flip a discrete decision and change a whole
 * path, so the CPU oracle and the gfx950 kernels must agree bit for bit.  This header is the ONE
 * definition of those functions, shared as a numerical substrate (like libm itself) by
 *   - oracle/            (g++,   -ffp-contract=off)
 *   - pathtracer-rs_amd/ (hipcc, -ffp-contract=off, device and host)
 * It is not a restatement of any reference algorithm and holds no renderer logic.
 *
 * Method: evaluate in IEEE binary64 using only + - * / (no FMA contraction, no reassociation,
 * no table lookups that depend on the platform), then round once to binary32.  The binary64
 * error is < 2^-48 relative, so the result is the correctly rounded f32 except for inputs whose
 * true value lies within ~2^-48 of a rounding boundary (about 1 in 1e7 arguments).
 *
 * Build requirement on every compiler: -ffp-contract=off and no -ffast-math.
 */
#ifndef PTRS_DETMATH_H
#define PTRS_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ inline
#else
#define PT_HD static inline
#endif

/* ---- bit helpers -------------------------------------------------------------------------- */
PT_HD uint64_t ptd_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PT_HD double ptd_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PT_HD uint32_t ptf_bits(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
PT_HD float ptf_from_bits(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }
PT_HD int ptd_isnan(double x) { return x != x; }
PT_HD int ptd_isinf(double x) { return (ptd_bits(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
PT_HD double ptd_floor(double x) { return __builtin_floor(x); }
PT_HD double ptd_nan(void) { return ptd_from_bits(0x7ff8000000000000ULL); }
PT_HD double ptd_inf(void) { return ptd_from_bits(0x7ff0000000000000ULL); }

/* 2^k for -1022 <= k <= 1023 */
PT_HD double ptd_pow2i(int k) { return ptd_from_bits((uint64_t)(k + 1023) << 52); }

/* ---- sin / cos ---------------------------------------------------------------------------- */
/* reduce x to r in [-pi/4, pi/4] and quadrant q (0..3); Cody-Waite with a 33-bit head */
PT_HD double ptd_reduce_pio2(double x, int *q) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    double k = ptd_floor(x * TWO_OVER_PI + 0.5);
    double r = (x - k * PIO2_1) - k * PIO2_1T;
    long long ki = (long long)k;
    *q = (int)(((ki % 4) + 4) % 4);
    return r;
}

PT_HD double ptd_sin_poly(double r) {
    const double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0, S3 = -1.0 / 5040.0, S4 = 1.0 / 362880.0,
                 S5 = -1.0 / 39916800.0, S6 = 1.0 / 6227020800.0, S7 = -1.0 / 1307674368000.0,
                 S8 = 1.0 / 355687428096000.0;
    double z = r * r;
    double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * (S6 + z * (S7 + z * S8))))));
    return r + r * (z * p);
}

PT_HD double ptd_cos_poly(double r) {
    const double C1 = -0.5, C2 = 1.0 / 24.0, C3 = -1.0 / 720.0, C4 = 1.0 / 40320.0,
                 C5 = -1.0 / 3628800.0, C6 = 1.0 / 479001600.0, C7 = -1.0 / 87178291200.0,
                 C8 = 1.0 / 20922789888000.0, C9 = -1.0 / 6402373705728000.0;
    double z = r * r;
    double p = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * (C6 + z * (C7 + z * (C8 + z * C9)))))));
    return 1.0 + z * p;
}

PT_HD double ptd_sin(double x) {
    if (ptd_isnan(x) || ptd_isinf(x)) return ptd_nan();
    int q;
    double r = ptd_reduce_pio2(x, &q);
    switch (q) {
        case 0: return ptd_sin_poly(r);
        case 1: return ptd_cos_poly(r);
        case 2: return -ptd_sin_poly(r);
        default: return -ptd_cos_poly(r);
    }
}

PT_HD double ptd_cos(double x) {
    if (ptd_isnan(x) || ptd_isinf(x)) return ptd_nan();
    int q;
    double r = ptd_reduce_pio2(x, &q);
    switch (q) {
        case 0: return ptd_cos_poly(r);
        case 1: return -ptd_sin_poly(r);
        case 2: return -ptd_cos_poly(r);
        default: return ptd_sin_poly(r);
    }
}

/* sin and cos of one argument: the same reduction and polynomials as ptd_sin / ptd_cos, evaluated once -- each result is
 * bit for bit what the separate functions return */
PT_HD void ptd_sincos(double x, double *sn, double *cs) {
    if (ptd_isnan(x) || ptd_isinf(x)) { *sn = ptd_nan(); *cs = ptd_nan(); return; }
    int q;
    double r = ptd_reduce_pio2(x, &q);
    const double ps = ptd_sin_poly(r), pc = ptd_cos_poly(r);
    *sn = (q & 1) ? pc : ps; if (q & 2) *sn = -*sn;         /* q: 0 sin, 1 cos, 2 -sin, 3 -cos */
    *cs = (q & 1) ? ps : pc; if (q == 1 || q == 2) *cs = -*cs; /* q: 0 cos, 1 -sin, 2 -cos, 3 sin */
}

/* ---- log ---------------------------------------------------------------------------------- */
/* x = m * 2^e, m in [sqrt(1/2), sqrt(2)); returns log(m), writes e.  x must be finite, > 0. */
PT_HD double ptd_log_mant(double x, int *e_out) {
    uint64_t b = ptd_bits(x);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) { /* subnormal double: scale up (not reachable from f32 inputs, kept for safety) */
        x = x * 18014398509481984.0; /* 2^54 */
        b = ptd_bits(x);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = ptd_from_bits((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 3.0 + z * (1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 +
               z * (1.0 / 13.0 + z * (1.0 / 15.0 + z * (1.0 / 17.0 + z * (1.0 / 19.0 +
               z * (1.0 / 21.0 + z * (1.0 / 23.0))))))))));
    *e_out = e;
    return 2.0 * s + 2.0 * s * (z * p);
}

PT_HD double ptd_log(double x) {
    if (ptd_isnan(x) || x < 0.0) return ptd_nan();
    if (x == 0.0) return -ptd_inf();
    if (ptd_isinf(x)) return x;
    int e;
    double lm = ptd_log_mant(x, &e);
    return (double)e * 6.93147180559945286227e-01 + lm;
}

PT_HD double ptd_log2(double x) {
    if (ptd_isnan(x) || x < 0.0) return ptd_nan();
    if (x == 0.0) return -ptd_inf();
    if (ptd_isinf(x)) return x;
    int e;
    double lm = ptd_log_mant(x, &e);
    return (double)e + lm * 1.44269504088896338700e+00;
}

/* ---- exp ---------------------------------------------------------------------------------- */
PT_HD double ptd_exp(double x) {
    if (ptd_isnan(x)) return x;
    if (x > 709.0) return ptd_inf();
    if (x < -745.0) return 0.0;
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double k = ptd_floor(x * INV_LN2 + 0.5);
    double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = 1.0 + r * (1.0 + r * (1.0 / 2.0 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0 +
               r * (1.0 / 720.0 + r * (1.0 / 5040.0 + r * (1.0 / 40320.0 + r * (1.0 / 362880.0 +
               r * (1.0 / 3628800.0 + r * (1.0 / 39916800.0 + r * (1.0 / 479001600.0 +
               r * (1.0 / 6227020800.0 + r * (1.0 / 87178291200.0))))))))))))));
    int ki = (int)k;
    /* split the scaling so that both factors stay normal */
    if (ki > 1000) return p * ptd_pow2i(1000) * ptd_pow2i(ki - 1000);
    if (ki < -1000) return p * ptd_pow2i(-1000) * ptd_pow2i(ki + 1000);
    return p * ptd_pow2i(ki);
}

PT_HD double ptd_pow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (ptd_isnan(x) || ptd_isnan(y)) return ptd_nan();
    if (x == 0.0) return y > 0.0 ? 0.0 : ptd_inf();
    if (x < 0.0) return ptd_nan(); /* integer y with negative base is not needed on this path */
    if (x == 1.0) return 1.0;
    return ptd_exp(y * ptd_log(x));
}

/* ---- atan / atan2 / acos ------------------------------------------------------------------ */
/* atan for z >= 0 */
PT_HD double ptd_atan_pos(double z) {
    const double PIO2 = 1.57079632679489655800e+00;
    int inv = 0;
    if (z > 1.0) { z = 1.0 / z; inv = 1; }
    /* z in [0,1]: subtract the nearest of atan(0), atan(.25), atan(.5), atan(.75), atan(1) */
    int i = (int)ptd_floor(4.0 * z + 0.5);
    double c = 0.25 * (double)i;
    double base;
    switch (i) {
        case 0: base = 0.0; break;
        case 1: base = 2.44978663126864143473e-01; break;
        case 2: base = 4.63647609000806093515e-01; break;
        case 3: base = 6.43501108793284370660e-01; break;
        default: base = 7.85398163397448278999e-01; break;
    }
    double t = (z - c) / (1.0 + z * c);
    double w = t * t;
    double p = -1.0 / 3.0 + w * (1.0 / 5.0 + w * (-1.0 / 7.0 + w * (1.0 / 9.0 + w * (-1.0 / 11.0 +
               w * (1.0 / 13.0 + w * (-1.0 / 15.0 + w * (1.0 / 17.0 + w * (-1.0 / 19.0))))))));
    double a = base + (t + t * (w * p));
    return inv ? PIO2 - a : a;
}

PT_HD double ptd_atan2(double y, double x) {
    const double PI = 3.14159265358979311600e+00;
    const double PIO2 = 1.57079632679489655800e+00;
    if (ptd_isnan(x) || ptd_isnan(y)) return ptd_nan();
    int ysign = (int)(ptd_bits(y) >> 63);
    int xsign = (int)(ptd_bits(x) >> 63);
    if (y == 0.0) { /* +-0 */
        if (!xsign) return y;  /* atan2(+-0, +x or +0) = +-0 */
        return ysign ? -PI : PI;
    }
    if (x == 0.0) return ysign ? -PIO2 : PIO2;
    if (ptd_isinf(x)) {
        if (ptd_isinf(y)) {
            double a = xsign ? 3.0 * (PI / 4.0) : PI / 4.0;
            return ysign ? -a : a;
        }
        if (xsign) return ysign ? -PI : PI;
        return ysign ? -0.0 : 0.0;
    }
    if (ptd_isinf(y)) return ysign ? -PIO2 : PIO2;
    double ay = ysign ? -y : y;
    double ax = xsign ? -x : x;
    double a = ptd_atan_pos(ay / ax);
    if (xsign) a = PI - a;
    return ysign ? -a : a;
}

/* sqrt in binary64: the hardware/OCML operation is correctly rounded on x86-64 and gfx950 */
PT_HD double ptd_sqrt(double x) { return __builtin_sqrt(x); }

PT_HD double ptd_acos(double x) {
    if (ptd_isnan(x) || x > 1.0 || x < -1.0) return ptd_nan();
    /* acos(x) = atan2(sqrt((1-x)(1+x)), x) */
    return ptd_atan2(ptd_sqrt((1.0 - x) * (1.0 + x)), x);
}

/* ---- binary32 entry points (what the renderer calls) -------------------------------------- */
PT_HD float pt_sinf(float x) { return (float)ptd_sin((double)x); }
PT_HD float pt_cosf(float x) { return (float)ptd_cos((double)x); }
PT_HD void pt_sincosf(float x, float *sn, float *cs) { double s, c; ptd_sincos((double)x, &s, &c); *sn = (float)s; *cs = (float)c; }
PT_HD float pt_tanf(float x) { return (float)(ptd_sin((double)x) / ptd_cos((double)x)); }
PT_HD float pt_logf(float x) { return (float)ptd_log((double)x); }
PT_HD float pt_log2f(float x) { return (float)ptd_log2((double)x); }
PT_HD float pt_expf(float x) { return (float)ptd_exp((double)x); }
PT_HD float pt_powf(float x, float y) { return (float)ptd_pow((double)x, (double)y); }
PT_HD float pt_atan2f(float y, float x) { return (float)ptd_atan2((double)y, (double)x); }
PT_HD float pt_acosf(float x) { return (float)ptd_acos((double)x); }

#endif /* PTRS_DETMATH_H */
