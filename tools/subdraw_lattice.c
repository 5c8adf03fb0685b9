// exhaustive joint distribution of the four sub-draws of a 32-bit word: x_j = W * A^j mod 2^32, idx_j = (x_j * n) >> 32
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
int main(int argc, char** argv) {
    uint32_t A = (uint32_t)strtoul(argv[1], 0, 0);
    int n = atoi(argv[2]); int m0 = argc > 3 ? atoi(argv[3]) : n, m1 = argc > 4 ? atoi(argv[4]) : n, m2 = argc > 5 ? atoi(argv[5]) : n, m3 = argc > 6 ? atoi(argv[6]) : n;
    uint32_t A2 = A * A, A3 = A2 * A;
    int cells = n * n * n * n;
    uint64_t* hist = calloc(cells, sizeof(uint64_t));
    #pragma omp parallel
    {
        uint32_t* h = calloc(cells, sizeof(uint32_t));
        #pragma omp for schedule(static)
        for (int64_t w = 0; w < (1ll << 32); ++w) {
            uint32_t W = (uint32_t)w;
            uint32_t i0 = ((uint64_t)W * m0) >> 32, i1 = ((uint64_t)(W * A) * m1) >> 32, i2 = ((uint64_t)(W * A2) * m2) >> 32, i3 = ((uint64_t)(W * A3) * m3) >> 32;
            h[((i0 * n + i1) * n + i2) * n + i3]++;
        }
        #pragma omp critical
        for (int c = 0; c < cells; ++c) hist[c] += h[c];
        free(h);
    }
    double expct = 4294967296.0 / ((double)m0*m1*m2*m3), worst = 0, chi = 0;
    for (int c = 0; c < cells; ++c) { int dd[4] = {c / (n*n*n), (c / (n*n)) % n, (c / n) % n, c % n}; if (dd[0]>=m0||dd[1]>=m1||dd[2]>=m2||dd[3]>=m3) continue; double d = hist[c] / expct - 1.0; if (d < 0) d = -d; if (d > worst) worst = d; chi += (hist[c]-expct)*(hist[c]-expct)/expct; }
    // pair marginals
    double worst2 = 0;
    for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) {
        uint64_t* p = calloc(n * n, sizeof(uint64_t));
        for (int c = 0; c < cells; ++c) { int d[4] = {c / (n*n*n), (c / (n*n)) % n, (c / n) % n, c % n}; p[d[a] * n + d[b]] += hist[c]; }
        const int mm[4] = {m0, m1, m2, m3};   // (mixed action counts: a pair's cells are m_a x m_b)
        double e2 = 4294967296.0 / ((double)mm[a] * mm[b]);
        for (int c = 0; c < n * n; ++c) { if (c / n >= mm[a] || c % n >= mm[b]) continue; double d = p[c] / e2 - 1.0; if (d < 0) d = -d; if (d > worst2) worst2 = d; }
        free(p);
    }
    printf("A=%u n=%d counts=%d,%d,%d,%d cells=%d expected/cell=%.0f  max |rel dev| 4-tuple=%.3e  pairs=%.3e  chi2/cells=%.3f\n", A, n, m0, m1, m2, m3, cells, expct, worst, worst2, chi / cells);
    return 0;
}
