/*
 * oracle/lmi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic on the LearnedMetricIndex query hot path
 * (SURVEY.md section 8a rows A2, A3, A6).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (learnedmetricindex_amd/) never does.
 *
 * What is restated (reference file:line, all under /root/reference/search/li/):
 *   lmi_oracle_linear       model.py:45-49, 97-99, 232   torch Linear (+ReLU): y = b + x.W^T
 *   lmi_oracle_rank_classes model.py:238-239             softmax + topk(L) == full descending sort
 *   lmi_oracle_softmax      model.py:238                 row softmax (multi-level priorities)
 *   lmi_oracle_knn_ip       LearnedIndex.py:360-365      faiss.knn(xq, xb, k, METRIC_INNER_PRODUCT)
 *
 * PARITY STATUS: "parity unpinned" at the two third-party boundaries.
 *   The scan arithmetic of the reference lives in faiss-cpu==1.7.4 (requirements.txt:8) and the
 *   MLP arithmetic in torch==2.1.1 (requirements-cpu.txt:3); neither source is under
 *   /root/reference and the reference has no tests or golden vectors.  This file therefore
 *   DEFINES the canonical arithmetic and both the oracle and the HIP kernels are held to it:
 *
 *     dot(a, b) = acc_{d},  acc_0 = c0,  acc_{k+1} = fmaf(a[k], b[k], acc_k)      (k ascending)
 *
 *   i.e. a k-ordered chain of IEEE-754 binary32 fused multiply-adds, one rounding per step --
 *   what `for (k) s += a[k]*b[k]` compiles to with FMA contraction, and bit-for-bit what
 *   gfx950's v_mfma_f32_32x32x2_f32 computes.  c0 is the bias for a Linear layer (torch addmm
 *   starts from the bias) and 0 for the scan.  Ties are broken towards the lower class index /
 *   lower in-bucket row (faiss keeps the earlier row when a later score is not strictly
 *   better; pandas groupby keeps rows in index order).  The reference's own Python logic around
 *   these calls IS pinned by import: tests/golden/make_golden.py runs the unmodified reference
 *   (torch for the MLP, BLAS sgemm + top-k in place of the absent faiss wheel) and the oracle
 *   is checked against those fixtures: identical ids, distances within 1e-4 relative.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VB 8 /* vectors per transposed block */
#define QB 8  /* queries per register block   */

int lmi_oracle_abi_version(void) { return 1; }

/* ---------------------------------------------------------------------------------------------
 * A2  torch.nn.Linear (+ optional ReLU):  out[i][j] = act(chain(b[j]; x[i][:], W[j][:]))
 * x [n][din] row-major, W [dout][din] row-major (torch layout, model.py:45-49), b [dout].
 * ------------------------------------------------------------------------------------------- */
void lmi_oracle_linear(const float *x, int64_t n, int din, const float *W, const float *b,
                       int dout, int relu, float *out, int nthreads)
{
    /* W^T so that the j loop is contiguous and vectorises; each acc[j] is still a k-ordered chain */
    float *Wt = (float *)malloc((size_t)din * dout * sizeof(float));
    for (int j = 0; j < dout; ++j)
        for (int k = 0; k < din; ++k) Wt[(size_t)k * dout + j] = W[(size_t)j * din + k];
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float *acc = out + (size_t)i * dout;
        const float *xi = x + (size_t)i * din;
        for (int j = 0; j < dout; ++j) acc[j] = b ? b[j] : 0.0f;
        for (int k = 0; k < din; ++k) {
            const float xv = xi[k];
            const float *w = Wt + (size_t)k * dout;
            for (int j = 0; j < dout; ++j) acc[j] = fmaf(xv, w[j], acc[j]);
        }
        if (relu)
            for (int j = 0; j < dout; ++j) acc[j] = acc[j] > 0.0f ? acc[j] : 0.0f;
    }
    free(Wt);
}

/* ---------------------------------------------------------------------------------------------
 * A3  predict_proba's `prob.topk(prob.shape[1])`: the full descending order of the classes.
 * Ranked on the logits (softmax is monotone), ties -> lower class index.
 * classes [n][nb] receives the first nb entries of that order.
 * ------------------------------------------------------------------------------------------- */
void lmi_oracle_rank_classes(const float *logits, int64_t n, int L, int nb, int32_t *classes)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *l = logits + (size_t)i * L;
        int32_t *c = classes + (size_t)i * nb;
        int cnt = 0;
        for (int j = 0; j < L; ++j) {
            /* insertion into a descending list; strict > keeps the earlier index first on ties */
            int p = cnt < nb ? cnt : nb;
            while (p > 0 && l[j] > l[c[p - 1]]) --p;
            if (p >= nb) continue;
            int last = cnt < nb ? cnt : nb - 1;
            for (int t = last; t > p; --t) c[t] = c[t - 1];
            c[p] = j;
            if (cnt < nb) ++cnt;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * lmi_expf: the exponential used by the canonical softmax.  Defined here (not libm) so that the
 * HIP kernel can run the identical sequence of binary32 operations: Cody-Waite reduction with
 * fmaf, degree-6 Horner polynomial with fmaf, scale by 2^n through the exponent field.
 * Domain used: x <= 0 (softmax subtracts the row maximum).  x < -87 returns 0.
 * ------------------------------------------------------------------------------------------- */
float lmi_oracle_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) return INFINITY;
    const float log2e = 1.44269502162933349609375f;    /* 0x3FB8AA3B */
    const float ln2hi = 0.693145751953125f;            /* 0x3F317200 */
    const float ln2lo = 1.42860677279532e-06f;         /* 0x35BFBE8E */
    float nf = rintf(x * log2e);
    float r = fmaf(-nf, ln2hi, x);
    r = fmaf(-nf, ln2lo, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t ni = (int32_t)nf;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(ni + 127) << 23; /* ni in [-126, 127] given the clamps above */
    return p * s.f;
}

/* A3  softmax(dim=1), model.py:238.  max, exp(l - max), sequential sum in index order, divide. */
void lmi_oracle_softmax(const float *logits, int64_t n, int L, float *probs)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *l = logits + (size_t)i * L;
        float *p = probs + (size_t)i * L;
        float m = l[0];
        for (int j = 1; j < L; ++j) m = l[j] > m ? l[j] : m;
        float s = 0.0f;
        for (int j = 0; j < L; ++j) {
            p[j] = lmi_oracle_expf(l[j] - m);
            s += p[j];
        }
        for (int j = 0; j < L; ++j) p[j] = p[j] / s;
    }
}

/* ---------------------------------------------------------------------------------------------
 * A6  faiss.knn(xq, xb, k, metric=METRIC_INNER_PRODUCT)  (call site LearnedIndex.py:360-365)
 * xq [nq][d], xb [nb][d] row-major.  D [nq][k] similarities descending, I [nq][k] row indices.
 * nb < k: tail padded with I = -1, D = -FLT_MAX (faiss' heap initial value).
 * Ties: the earlier row wins (strict > on insertion).
 * ------------------------------------------------------------------------------------------- */
static inline void topk_insert(float *D, int64_t *I, int k, float s, int64_t idx)
{
    if (!(s > D[k - 1])) return;
    int p = k - 1;
    while (p > 0 && s > D[p - 1]) {
        D[p] = D[p - 1];
        I[p] = I[p - 1];
        --p;
    }
    D[p] = s;
    I[p] = idx;
}

/* (score desc, row asc): the order every list is kept in */
static inline int better_sr(float s, int64_t r, float s2, int64_t r2) { return s > s2 || (s == s2 && r < r2); }

/* one contiguous range of row-blocks against all queries; rows ascend, so strict > keeps the earlier */
static void knn_range(const float *xq, int64_t nq, const float *xt, int64_t blk0, int64_t blk1, int64_t nb,
                      int d, int k, float *D, int64_t *I)
{
    for (int64_t i = 0; i < nq * (int64_t)k; ++i) { D[i] = -FLT_MAX; I[i] = -1; }
    const int64_t nqb = (nq + QB - 1) / QB;
    for (int64_t qb = 0; qb < nqb; ++qb) {
        const int64_t q0 = qb * QB;
        const int nqq = (int)((nq - q0) < QB ? (nq - q0) : QB);
        const float *qp[QB];
        for (int t = 0; t < QB; ++t) qp[t] = xq + (size_t)(q0 + (t < nqq ? t : 0)) * d;
        for (int64_t blk = blk0; blk < blk1; ++blk) {
            float acc[QB][VB];
            memset(acc, 0, sizeof(acc));
            const float *xp = xt + (size_t)blk * d * VB;
            for (int kk = 0; kk < d; ++kk) {
                const float *xv = xp + (size_t)kk * VB;
                for (int t = 0; t < QB; ++t) {
                    const float qv = qp[t][kk];
                    for (int v = 0; v < VB; ++v) acc[t][v] = fmaf(qv, xv[v], acc[t][v]);
                }
            }
            const int64_t r0 = blk * VB;
            const int nv = (int)((nb - r0) < VB ? (nb - r0) : VB);
            for (int t = 0; t < nqq; ++t)
                for (int v = 0; v < nv; ++v)
                    topk_insert(D + (size_t)(q0 + t) * k, I + (size_t)(q0 + t) * k, k, acc[t][v], r0 + v);
        }
    }
}

void lmi_oracle_knn_ip(const float *xq, int64_t nq, const float *xb, int64_t nb, int d, int k,
                       float *D, int64_t *I, int nthreads)
{
    for (int64_t i = 0; i < nq * (int64_t)k; ++i) {
        D[i] = -FLT_MAX;
        I[i] = -1;
    }
    /* -FLT_MAX sentinels sit below every finite score; a real score of exactly -FLT_MAX is not
       representable by unit-norm inputs. */
    if (nq == 0 || nb == 0) return;
    if (nthreads < 1) nthreads = 1;
    const int64_t nblk = (nb + VB - 1) / VB;
    if (nthreads > nblk) nthreads = (int)nblk;
    /* transposed copy of xb in blocks of VB rows: xt[blk][k][v] */
    float *xt = (float *)calloc((size_t)nblk * d * VB, sizeof(float));
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t r = 0; r < nb; ++r) {
        float *dst = xt + (size_t)(r / VB) * d * VB + (r % VB);
        const float *src = xb + (size_t)r * d;
        for (int kk = 0; kk < d; ++kk) dst[(size_t)kk * VB] = src[kk];
    }
    if (nthreads == 1) {
        knn_range(xq, nq, xt, 0, nblk, nb, d, k, D, I);
    } else {
        /* threads split the ROWS (a bucket is visited by few queries); per-thread lists are merged
           in (score desc, row asc) order, so the result does not depend on the thread count */
        float *Dt = (float *)malloc((size_t)nthreads * nq * k * sizeof(float));
        int64_t *It = (int64_t *)malloc((size_t)nthreads * nq * k * sizeof(int64_t));
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
        for (int t = 0; t < nthreads; ++t) {
            const int64_t b0 = nblk * t / nthreads, b1 = nblk * (t + 1) / nthreads;
            knn_range(xq, nq, xt, b0, b1, nb, d, k, Dt + (size_t)t * nq * k, It + (size_t)t * nq * k);
        }
        for (int64_t q = 0; q < nq; ++q) {
            int head[256];
            for (int t = 0; t < nthreads && t < 256; ++t) head[t] = 0;
            for (int j = 0; j < k; ++j) {
                int bt = -1;
                float bs = -FLT_MAX;
                int64_t br = -1;
                for (int t = 0; t < nthreads && t < 256; ++t) {
                    if (head[t] >= k) continue;
                    const float s = Dt[((size_t)t * nq + q) * k + head[t]];
                    const int64_t r = It[((size_t)t * nq + q) * k + head[t]];
                    if (r < 0) continue; /* that thread's list is exhausted */
                    if (bt < 0 || better_sr(s, r, bs, br)) { bt = t; bs = s; br = r; }
                }
                if (bt < 0) break;
                D[(size_t)q * k + j] = bs;
                I[(size_t)q * k + j] = br;
                head[bt]++;
            }
        }
        free(Dt);
        free(It);
    }
    free(xt);
}

/* L2 metric (no counterpart in the reference, which scans with METRIC_INNER_PRODUCT only, LearnedIndex.py:364;
 * this is what faiss.knn(..., METRIC_L2) would be in its place).  Canonical arithmetic, shared with the HIP
 * kernels (include/lmi_hip.h: lmi_set_metric): xn = chain <x,x>, qn = chain <q,q>, key = chain <q,x> continued by
 * one step fmaf(1, -xn/2, .) -- i.e. the inner product of [q, 1] with [x, -xn/2], which lmi_oracle.py computes with
 * lmi_oracle_knn_ip on the augmented vectors -- and dist = fmaf(-2, key, qn). */
void lmi_oracle_sqnorms(const float *x, int64_t n, int d, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int k = 0; k < d; ++k) acc = fmaf(x[(size_t)i * d + k], x[(size_t)i * d + k], acc);
        out[i] = acc;
    }
}

void lmi_oracle_l2_finish(const float *key, const int64_t *idx, const float *qn, int64_t nq, int k, float *dist)
{
    for (int64_t q = 0; q < nq; ++q)
        for (int j = 0; j < k; ++j)
            dist[q * k + j] = idx[q * k + j] < 0 ? FLT_MAX : fmaf(-2.0f, key[q * k + j], qn[q]);
}

/* Single canonical dot product, exposed for spot checks at full size. */
float lmi_oracle_dot(const float *a, const float *b, int d, float c0)
{
    float acc = c0;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}
