// gs_log.h -- the strict-order mode's natural logarithm: evaluated in double-double arithmetic (about 100 bits) and rounded
// once, i.e. correctly rounded except when the exact value lies within ~2^-90 of a rounding boundary.  The reference ran on a
// libm whose log was correctly rounded (glibc's IBM Accurate Mathematical Library, R 3.5 / 2018); the device library's log
// is faithful, not correctly rounded (tools/ubench/libm_bits.hip: 0.16 % of arguments differ from the host's in the last bit),
// and on the duplicated-column ties of the real-R tables one last bit of a dML decides an action (gm_strict.h).
//   x = 2^e m, m in [sqrt(1/2), sqrt(2));  c = centre of m's 1/128-wide interval;  r = (m - c) / c  (m - c is exact);
//   log x = e ln 2 + log c + log1p(r),  log1p(r) = r - r^2/2 + r^3 (1/3 - r/4 + ...),  |r| < 2^-7.4
// with e ln 2, log c, 1/c from tables of double-double constants (tools/make_log_table.py).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef PAREBEN_HOST_EMUL
#define GS_TABLE_QUAL static const
#else
#define GS_TABLE_QUAL __device__ const
#endif
#include "gs_log_table.h"

struct GsDD { double hi, lo; };
DEV GsDD gs_two_sum(double a, double b) { GS_FP const double s = a + b, bb = s - a; return GsDD{s, (a - (s - bb)) + (b - bb)}; }
DEV GsDD gs_quick_two_sum(double a, double b) { GS_FP const double s = a + b; return GsDD{s, b - (s - a)}; }
DEV GsDD gs_two_prod(double a, double b) { GS_FP const double p = a * b; return GsDD{p, fma(a, b, -p)}; }
DEV GsDD gs_dd_add(GsDD a, GsDD b)
{
    GS_FP
    GsDD s = gs_two_sum(a.hi, b.hi);
    const GsDD t = gs_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = gs_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return gs_quick_two_sum(s.hi, s.lo);
}
DEV GsDD gs_dd_mul_d(GsDD a, double b)
{
    GS_FP
    GsDD p = gs_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return gs_quick_two_sum(p.hi, p.lo);
}
DEV GsDD gs_dd_mul(GsDD a, GsDD b)
{
    GS_FP
    GsDD p = gs_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return gs_quick_two_sum(p.hi, p.lo);
}

DEV double gs_log(double x)
{
    GS_FP
    uint64_t u;
    memcpy(&u, &x, 8);
    const int be = (int)((u >> 52) & 0x7ff);
    if (!(x > 0) || be == 0 || be == 0x7ff) return log(x);      // zero, negative, NaN, infinity, subnormal: the library's answer
    int e = be - 1023;
    uint64_t mu = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &mu, 8);                                          // m in [1, 2)
    if (m >= 1.4140625) { m *= 0.5; e += 1; }                    // m in [0.70703125, 1.4140625)
    int i = (int)(m * 128.0) - 90;
    if (i < 0) i = 0;
    if (i > 90) i = 90;
    // around 1 the centre is 1 itself (log c = 0, r = m - 1 exactly): no cancellation between log c and log1p(r)
    const bool unit = fabs(m - 1.0) < 1.0 / 256;
    const double c = unit ? 1.0 : GS_LOG_TAB[i][0];
    const double d = m - c;                                      // exact
    const GsDD r = unit ? GsDD{d, 0.0} : gs_dd_mul_d(GsDD{GS_LOG_TAB[i][1], GS_LOG_TAB[i][2]}, d);
    // log1p(r): r - r^2/2 + r^3/3 in double-double, the tail -r^4/4 + ... - r^12/12 in double
    const double rh = r.hi;
    const double tail = rh * rh * rh * rh * (-1.0 / 4 + rh * (1.0 / 5 + rh * (-1.0 / 6 + rh * (1.0 / 7 + rh * (-1.0 / 8 + rh * (1.0 / 9 + rh * (-1.0 / 10 + rh * (1.0 / 11 + rh * (-1.0 / 12)))))))));
    const GsDD r2 = gs_dd_mul(r, r);
    const GsDD r3 = gs_dd_mul(gs_dd_mul(r2, r), GsDD{0x1.5555555555555p-2, 0x1.5555555555555p-56});   // r^3 / 3
    GsDD acc = gs_dd_add(GsDD{r2.hi * -0.5, r2.lo * -0.5}, r3);
    acc = gs_dd_add(acc, GsDD{tail, 0.0});
    acc = gs_dd_add(r, acc);
    if (!unit) acc = gs_dd_add(GsDD{GS_LOG_TAB[i][3], GS_LOG_TAB[i][4]}, acc);
    if (e != 0) acc = gs_dd_add(gs_dd_mul_d(GsDD{GS_LN2_HI, GS_LN2_LO}, (double)e), acc);
    return acc.hi + acc.lo;
}
