/* bh_cols.hip raw_bits_inv(): rank / m for integers 1 <= rank <= m <= 2^18 as q0 = rank * (1 / m), q = fma(fma(-q0, m, rank), 1 / m, q0)
 * must be the IEEE quotient (double)rank / (double)m -- the ecdf factor of statsmodels' fdr_bh -- for EVERY rank of the column
 * lengths tried: 21 chosen ones (powers of two and their neighbours, the BASELINE sizes, primes) and `n_random` random ones.
 * Built and run by tests/test_abi_and_host.py::test_rank_over_m_by_reciprocal (gcc, -ffp-contract=off). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static long check(int mi, long* bad) {
    const double m = mi, y = 1.0 / m;
    for (int a = 1; a <= mi; ++a) {
        const double want = (double)a / m, q0 = (double)a * y, r = fma(-q0, m, (double)a), q1 = fma(r, y, q0);
        if (q1 != want) { if (*bad < 5) printf("mismatch rank=%d m=%d\n", a, mi); ++*bad; }
    }
    return mi;
}

int main(int argc, char** argv) {
    const int n_random = argc > 1 ? atoi(argv[1]) : 300;
    long bad = 0, tot = 0;
    const int ms[] = {1, 2, 3, 7, 1000, 1023, 1024, 1025, 4099, 24999, 25000, 25001, 65537, 131071, 199999, 200000, 262143, 262144,
                      77777, 100003, 250007};
    for (unsigned i = 0; i < sizeof ms / sizeof *ms; ++i) tot += check(ms[i], &bad);
    srand(1);
    for (int t = 0; t < n_random; ++t) tot += check(1 + rand() % 262144, &bad);
    printf("checked %ld bad %ld\n", tot, bad);
    return bad != 0;
}
