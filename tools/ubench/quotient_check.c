/* The short quotient of ilqg_device.hpp (div_plain) against the C division, on the host:
 *     y = 1.0 / b (correctly rounded);  q = a * y;  q' = fma(fma(-b, q, a), y, q)
 * Markstein's theorem: q' == a / b unless something underflows.  Random operands with random exponents in
 * [emin, emax], every 7th divisor with an all-ones or a nearly-all-zeros significand.
 *     gcc -O2 -ffp-contract=off -o quotient_check quotient_check.c -lm
 *     ./quotient_check 200000000 -30 30       -> 0 mismatches       (also -300 300 and -1 0)
 *     ./quotient_check 2000000 -1022 -940 -100 100   numerators swept from the smallest normal numbers upwards:
 *                                              mismatches only below 2^-998 (66 000, 1 400, 18, 1, 0, 0, ... of 2e6)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ULL;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double mk(int emin, int emax) {
    uint64_t m = rnd() & ((1ULL << 52) - 1);
    int e = emin + (int)(rnd() % (uint64_t)(emax - emin + 1));
    uint64_t b = ((uint64_t)(e + 1023) << 52) | m | ((rnd() & 1) << 63);
    double d;
    memcpy(&d, &b, 8);
    return d;
}
static int differs(double a, double b) {
    volatile double y = 1.0 / b;
    double q = a * y, r = fma(-b, q, a), q1 = fma(r, y, q);
    return q1 != a / b;
}
int main(int argc, char **argv) {
    if(argc < 4) return 1;
    long n = atol(argv[1]);
    int emin = atoi(argv[2]), emax = atoi(argv[3]);
    if(argc >= 6) { /* numerators in bands of 6 exponents from emin to emax, divisors in [argv[4], argv[5]] */
        for(int ea = emin; ea <= emax; ea += 6) {
            long bad = 0;
            for(long i = 0; i < n; i++) bad += differs(mk(ea, ea + 5), mk(atoi(argv[4]), atoi(argv[5])));
            printf("numerator 2^[%d,%d]: %ld of %ld differ\n", ea, ea + 5, bad, n);
        }
        return 0;
    }
    long bad = 0;
    for(long i = 0; i < n; i++) {
        double a = mk(emin, emax), b = mk(emin, emax);
        if(i % 7 == 0) {
            uint64_t u;
            memcpy(&u, &b, 8);
            u |= (1ULL << 52) - 1;
            if(i % 14 == 0) u &= ~((1ULL << 52) - 1) | 1;
            memcpy(&b, &u, 8);
        }
        bad += differs(a, b);
    }
    printf("n=%ld, exponents [%d,%d]: %ld differ\n", n, emin, emax, bad);
    return 0;
}
