/*
 * div_check.c -- CPU check of the division used on the back-substitution chain of the coarse solve
 * (csrc/mg3d_kernels.hip, lu_div): with r = RN(1/d) the twice-refined quotient
 *     q = n*r;  q += (n - d*q)*r;  q += (n - d*q)*r      (residuals by FMA)
 * must equal the IEEE quotient n/d bit for bit.  Operands: random significands and exponents inside
 * the window lu_div accepts, plus the classic hard cases (significands of all ones, just above a power
 * of two, quotients next to a rounding boundary built as n = RN(q*d) +- ulps).
 * usage: div_check <pairs>   prints "checked N mismatches M".  Test infrastructure only.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void)
{ /* xorshift128+ */
    uint64_t a = s[0], b = s[1];
    s[0] = b;
    a ^= a << 23;
    s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
}
static double from_bits(uint64_t u)
{
    double x;
    memcpy(&x, &u, 8);
    return x;
}
static uint64_t to_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
static double make(uint64_t mant, int exp2, int neg)
{
    return from_bits(((uint64_t)neg << 63) | ((uint64_t)(exp2 + 1023) << 52) | (mant & 0xFFFFFFFFFFFFFull));
}
static double lu_div(double n, double d, double r)
{
    double q = n * r;
    double rem = fma(-d, q, n);
    q = fma(rem, r, q);
    rem = fma(-d, q, n);
    return fma(rem, r, q);
}
static long bad = 0, checked = 0;
static void check(double n, double d)
{
    const double r = 1.0 / d;
    const double want = n / d, got = lu_div(n, d, r);
    checked++;
    if (to_bits(want) != to_bits(got)) {
        if (bad < 10)
            printf("MISMATCH n=%a d=%a want=%a got=%a\n", n, d, want, got);
        bad++;
    }
}
int main(int argc, char **argv)
{
    const long pairs = argc > 1 ? atol(argv[1]) : 1000000;
    const uint64_t hard[] = {0xFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFEull, 0x0ull, 0x1ull, 0x2ull, 0x8000000000000ull,
                             0x7FFFFFFFFFFFFull, 0x8000000000001ull, 0x5555555555555ull, 0xAAAAAAAAAAAAAull};
    const int nh = (int)(sizeof hard / sizeof hard[0]);
    for (int a = 0; a < nh; a++)
        for (int b = 0; b < nh; b++)
            for (int en = -498; en <= 498; en += 83)
                for (int ed = -460; ed <= 460; ed += 92)
                    for (int sg = 0; sg < 4; sg++)
                        check(make(hard[a], en, sg & 1), make(hard[b], ed, sg >> 1));
    for (long i = 0; i < pairs; i++) {
        const uint64_t u = rnd(), v = rnd(), w = rnd();
        const int en = (int)(w % 997) - 498, ed = (int)((w >> 16) % 921) - 460;
        double n = make(u, en, (int)(w >> 40) & 1), d = make(v, ed, (int)(w >> 41) & 1);
        check(n, d);
        /* quotient next to a rounding boundary: n' = q*d rounded, nudged by -2..2 ulps */
        const double q = make(rnd(), (int)((w >> 24) % 61) - 30, 0);
        double np = q * d;
        const int k = (int)((w >> 44) % 5) - 2;
        np = from_bits(to_bits(np) + (uint64_t)(int64_t)k);
        const unsigned e = (unsigned)(to_bits(np) >> 52) & 0x7ffu;
        if (e - 525u < 997u)
            check(np, d);
        /* hard significand against a random one */
        check(make(hard[i % nh], en, 0), d);
        check(n, make(hard[(i / nh) % nh], ed, 0));
    }
    printf("checked %ld mismatches %ld\n", checked, bad);
    return bad != 0;
}
