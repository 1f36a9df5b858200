/* oracle/sanitize_selftest.c -- TEST INFRASTRUCTURE ("host -fsanitize=address build").
 *
 * Drives every entry point of oracle/lmi_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer
 * (`make -C oracle asan`, run by tests/test_oracle_golden.py) over the shapes the parity tests use and the edge
 * cases the reference's behaviour defines: buckets shorter than k (faiss pads idx = -1, LearnedIndex.py:360-370),
 * empty buckets, empty query sets, d not a multiple of the vector width, 1 and many threads, nb = all classes.
 * Also checks, where it is free, that results do not depend on the thread count (the determinism the GPU tests
 * rely on when they use this oracle as the checker).  Exit code 0 = clean. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void lmi_oracle_linear(const float *x, int64_t n, int din, const float *W, const float *b, int dout, int relu,
                       float *out, int nthreads);
void lmi_oracle_rank_classes(const float *logits, int64_t n, int L, int nb, int32_t *classes);
void lmi_oracle_softmax(const float *logits, int64_t n, int L, float *probs);
void lmi_oracle_knn_ip(const float *xq, int64_t nq, const float *xb, int64_t nb, int d, int k, float *D, int64_t *I,
                       int nthreads);
void lmi_oracle_sqnorms(const float *x, int64_t n, int d, float *out);
void lmi_oracle_l2_finish(const float *key, const int64_t *idx, const float *qn, int64_t nq, int k, float *dist);
float lmi_oracle_dot(const float *a, const float *b, int d, float c0);
float lmi_oracle_expf(float x);

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static float frand(void)
{
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (float)((double)(rng >> 11) / (double)(1ull << 53)) * 2.0f - 1.0f;
}
static float *fill(size_t n)
{
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));   /* exact size: any over-read is an ASan report */
    for (size_t i = 0; i < n; ++i) p[i] = frand();
    return p;
}
static int fails = 0;
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void mlp_case(int64_t n, int din, int H, int L, int nb)
{
    float *x = fill((size_t)n * din), *W1 = fill((size_t)H * din), *b1 = fill(H), *W2 = fill((size_t)L * H), *b2 = fill(L);
    float *h1 = (float *)malloc((size_t)(n ? n : 1) * H * 4), *h1b = (float *)malloc((size_t)(n ? n : 1) * H * 4);
    float *lg = (float *)malloc((size_t)(n ? n : 1) * L * 4), *pr = (float *)malloc((size_t)(n ? n : 1) * L * 4);
    int32_t *cls = (int32_t *)malloc((size_t)(n ? n : 1) * nb * 4);
    lmi_oracle_linear(x, n, din, W1, b1, H, 1, h1, 1);
    lmi_oracle_linear(x, n, din, W1, b1, H, 1, h1b, 5);
    EXPECT(memcmp(h1, h1b, (size_t)n * H * 4) == 0);
    lmi_oracle_linear(h1, n, H, W2, b2, L, 0, lg, 3);
    lmi_oracle_rank_classes(lg, n, L, nb, cls);
    lmi_oracle_softmax(lg, n, L, pr);
    for (int64_t i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < L; ++j) s += pr[i * L + j];
        EXPECT(fabs(s - 1.0) < 1e-4);
        for (int j = 0; j < nb; ++j) EXPECT(cls[i * nb + j] >= 0 && cls[i * nb + j] < L);
        for (int j = 1; j < nb; ++j) EXPECT(lg[i * L + cls[i * nb + j - 1]] >= lg[i * L + cls[i * nb + j]]);
        if (n && din) EXPECT(lmi_oracle_dot(x + i * din, W1, din, b1[0]) == (h1[i * H] > 0 ? h1[i * H] : lmi_oracle_dot(x + i * din, W1, din, b1[0])));
    }
    free(x); free(W1); free(b1); free(W2); free(b2); free(h1); free(h1b); free(lg); free(pr); free(cls);
}

static void knn_case(int64_t nq, int64_t nb, int d, int k)
{
    float *q = fill((size_t)nq * d), *xb = fill((size_t)nb * d);
    size_t m = (size_t)(nq ? nq : 1) * k;
    float *D1 = (float *)malloc(m * 4), *D8 = (float *)malloc(m * 4), *qn = (float *)malloc((nq ? nq : 1) * 4), *dist = (float *)malloc(m * 4);
    int64_t *I1 = (int64_t *)malloc(m * 8), *I8 = (int64_t *)malloc(m * 8);
    lmi_oracle_knn_ip(q, nq, xb, nb, d, k, D1, I1, 1);
    lmi_oracle_knn_ip(q, nq, xb, nb, d, k, D8, I8, 8);
    EXPECT(memcmp(D1, D8, (size_t)nq * k * 4) == 0 && memcmp(I1, I8, (size_t)nq * k * 8) == 0);
    for (int64_t i = 0; i < nq; ++i)
        for (int j = 0; j < k; ++j) {
            const int64_t r = I1[i * k + j];
            if (j < nb) {
                EXPECT(r >= 0 && r < nb);
                EXPECT(D1[i * k + j] == lmi_oracle_dot(q + i * d, xb + r * d, d, 0.0f));
                if (j) EXPECT(D1[i * k + j - 1] > D1[i * k + j] || (D1[i * k + j - 1] == D1[i * k + j] && I1[i * k + j - 1] < r));
            } else {
                EXPECT(r == -1 && D1[i * k + j] == -FLT_MAX);   /* faiss padding */
            }
        }
    lmi_oracle_sqnorms(q, nq, d, qn);
    lmi_oracle_l2_finish(D1, I1, qn, nq, k, dist);
    free(q); free(xb); free(D1); free(D8); free(qn); free(dist); free(I1); free(I8);
}

int main(void)
{
    mlp_case(37, 64, 128, 12, 3);
    mlp_case(5, 45, 512, 256, 4);
    mlp_case(1, 768, 512, 120, 120);    /* nb = all classes */
    mlp_case(0, 32, 16, 4, 2);          /* empty batch */
    mlp_case(9, 7, 8, 3, 1);
    knn_case(33, 1000, 64, 10);
    knn_case(7, 3, 45, 10);             /* bucket shorter than k */
    knn_case(4, 0, 96, 10);             /* empty bucket */
    knn_case(0, 50, 96, 10);            /* no query routed here */
    knn_case(17, 257, 768, 10);
    knn_case(3, 9, 1, 5);
    knn_case(65, 4099, 33, 10);         /* row count and width off every block size */
    /* the softmax's domain is x <= 0 (row maximum subtracted); x <= -87 flushes to 0 by definition */
    for (float x = -86.9f; x <= 0.0f; x += 0.0137f) EXPECT(fabsf(lmi_oracle_expf(x) - expf(x)) <= 1e-6f * expf(x));
    EXPECT(lmi_oracle_expf(-200.0f) == 0.0f && lmi_oracle_expf(-87.0f) == 0.0f && lmi_oracle_expf(0.0f) == 1.0f);
    if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    puts("oracle sanitize selftest: clean");
    return 0;
}
