/*
 * kmer_oracle.c -- plain-C CPU restatement of the PhaMers count + score path.
 *
 * TEST INFRASTRUCTURE ONLY.  Loaded (ctypes) by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * timed CPU baseline.  Never linked into, or called from, phamers_amd/.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here
 * against tests/golden/ (vectors produced by executing the reference's own
 * function bodies, see tools/gen_golden.py).
 *
 * Reference citations are relative to the PhaMers tree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* scripts/kmer.py:183-196 -- symbol i of `symbols` -> code i, any other byte
 * (case-sensitive) -> -1 (the reference's '-'). */
static void build_lut(const char *symbols, int n_sym, int8_t lut[256]) {
    memset(lut, -1, 256);
    for (int i = 0; i < n_sym; ++i) lut[(unsigned char)symbols[i]] = (int8_t)i;
}

/* scripts/kmer.py:32-50 -- literal sliding window; first base is the most
 * significant digit (int(window, n_sym)); a window touching an invalid
 * character is skipped.  out[n_sym^k] must be zeroed by the caller or here. */
int oracle_count_string(const char *seq, uint64_t L, int k, const char *symbols,
                        int n_sym, int64_t *out) {
    if (k < 1 || n_sym < 1 || n_sym > 9) return -1;
    uint64_t D = 1;
    for (int i = 0; i < k; ++i) D *= (uint64_t)n_sym;
    memset(out, 0, D * sizeof(int64_t));
    int8_t lut[256];
    build_lut(symbols, n_sym, lut);
    if (L < (uint64_t)k) return 0;
    for (uint64_t i = 0; i + (uint64_t)k <= L; ++i) {
        uint64_t idx = 0;
        int ok = 1;
        for (int j = 0; j < k; ++j) {
            int8_t c = lut[(unsigned char)seq[i + j]];
            if (c < 0) { ok = 0; break; }
            idx = idx * (uint64_t)n_sym + (uint64_t)c;
        }
        if (ok) out[idx] += 1;
    }
    return 0;
}

/* scripts/kmer.py:100-105 -- list form: concatenated sequences + offsets[n+1]. */
int oracle_count(const char *bases, const uint64_t *offsets, uint64_t n, int k,
                 const char *symbols, int n_sym, int64_t *out) {
    uint64_t D = 1;
    for (int i = 0; i < k; ++i) D *= (uint64_t)n_sym;
    for (uint64_t c = 0; c < n; ++c) {
        int rc = oracle_count_string(bases + offsets[c], offsets[c + 1] - offsets[c], k,
                                     symbols, n_sym, out + c * D);
        if (rc) return rc;
    }
    return 0;
}

/* scripts/kmer.py:209-221 -- float64 row / row-sum; zero row -> NaN. */
int oracle_normalize(const int64_t *counts, uint64_t n, uint64_t D, double *out) {
    for (uint64_t r = 0; r < n; ++r) {
        double s = 0.0;
        for (uint64_t j = 0; j < D; ++j) s += (double)counts[r * D + j];
        for (uint64_t j = 0; j < D; ++j) out[r * D + j] = (double)counts[r * D + j] / s;
    }
    return 0;
}

static double sqdist(const double *a, const double *b, uint64_t D) {
    double s = 0.0;
    for (uint64_t j = 0; j < D; ++j) { double d = a[j] - b[j]; s += d * d; }
    return s;
}

/* scripts/learning.py:118-128 (+ scripts/phamer.py:186-187,268-273) --
 * brute-force Euclidean k-NN over R[M][D], uniform majority vote over labels
 * {0,1}; result 2*(pred-0.5).  Ties in distance go to the lower index. */
int oracle_knn_score(const double *Q, uint64_t N, const double *R, uint64_t M,
                     const uint8_t *labels, uint64_t D, int kn, double *out) {
    if (kn < 1 || (uint64_t)kn > M || kn > 64) return -1;
    double bd[64];
    uint64_t bi[64];
    for (uint64_t q = 0; q < N; ++q) {
        int have = 0;
        for (uint64_t r = 0; r < M; ++r) {
            double d = sqdist(Q + q * D, R + r * D, D);
            if (have < kn || d < bd[have - 1]) {
                int p = have < kn ? have : kn - 1;
                while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                bd[p] = d; bi[p] = r;
                if (have < kn) ++have;
            }
        }
        int votes = 0;
        for (int j = 0; j < kn; ++j) votes += labels[bi[j]] ? 1 : 0;
        out[q] = (2 * votes > kn) ? 1.0 : -1.0;
    }
    return 0;
}

/* scripts/phamer.py:198-210,250-256 + scripts/learning.py:47-66 -- nearest
 * positive / negative centroid (first index wins ties), then
 * tanh((e- - e+)/(e+ + e-)). */
int oracle_centroid_score(const double *Q, uint64_t N, const double *Cpos, uint64_t nPos,
                          const double *Cneg, uint64_t nNeg, uint64_t D, double *out) {
    for (uint64_t q = 0; q < N; ++q) {
        double bp = INFINITY, bn = INFINITY;
        for (uint64_t c = 0; c < nPos; ++c) {
            double d = sqdist(Q + q * D, Cpos + c * D, D);
            if (d < bp) bp = d;
        }
        for (uint64_t c = 0; c < nNeg; ++c) {
            double d = sqdist(Q + q * D, Cneg + c * D, D);
            if (d < bn) bn = d;
        }
        double ep = sqrt(bp), en = sqrt(bn);
        out[q] = tanh((en - ep) / (ep + en));
    }
    return 0;
}
