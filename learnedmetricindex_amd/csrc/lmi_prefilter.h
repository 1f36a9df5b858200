// lmi_prefilter.h -- fp16-MFMA prefilter + exact fp32 re-rank for the bucket scan (gfx950).
//
// Why: the exact scan is bound by the f32 MFMA rate (157 TFLOP/s); v_mfma_f32_32x32x16_f16 runs 16x
// faster.  The prefilter computes APPROXIMATE similarities with fp16 operands (f32 accumulate),
// keeps every row that could still be among a (query, bucket)'s 10 best given a PROVEN error bound,
// and the survivors (typically 10-20 per slot) are re-scored with the canonical k-ordered binary32
// fmaf chain -- so ids and distances are bit-identical to the exact path (tests assert equality of
// the two modes and of both with the oracle).  Slots whose candidate sets overflow fall back to an
// exact brute-force kernel; nothing is ever approximate in the output.
//
// Scaled units.  Index vectors are stored as x^ = fp16(x'), x' = sx * x (sx = one power of two per index
// with max|x'| in [0.5, 1)), queries as q^ = fp16(q'), q' = sq * q (one power of two per QUERY: a slot's
// scores are only compared with scores of the same query); shat = sum q^_k x^_k approximates s' = sx*sq*s.
//   fp16 x fp16 products are exact in binary32; a round-to-nearest binary32 summation of d terms errs by at
//   most d*2^-24 * sum|terms| (the canonical chain), one whose additions may truncate by d*2^-23 * sum|terms|
//   (allowed for the MFMA's accumulation); and
//   |<q^,x^> - <q',x'>| = |<q^-q', x^> + <q', x^-x'>| <= ||q^-q'|| ||x^|| + ||q'|| ||x^-x'||     (Cauchy-Schwarz)
//   where the rounding-error norms are MEASURED (q^_k - q'_k is exact in binary32): per query by
//   query_norm_kernel, the largest per bucket by bucket_norm_kernel at build time, all rounded up.  So
//   |shat - s'_canonical| <= eps' := dq (xn + dx) + qn dx + 4 d 2^-24 (qn + dq)(xn + dx)            (slot_bound_kernel)
//   with qn = ||q'||, dq = ||q^-q'||, xn / dx the bucket's largest ||x'|| / ||x^-x'||.  Subnormal fp16
//   operands are covered by the measured norms, provided the hardware converts and multiplies them
//   un-flushed: pf_selftest_kernel checks that once per process.
// Candidate rule.  Let That be the 10th largest shat of the bucket.  The 10 rows with shat >= That have
// s'_c >= That - eps', hence the canonical 10th best T_c >= That - eps', hence every row of the
// canonical top-10 has shat >= T_c - eps' >= That - 2 eps'.  Any lower bound of That may replace it:
// the 10th best shat of any SUBSET of the bucket's rows is (pass 1 uses every 16th 256-row tile).
#pragma once
#include "lmi_kernels.h"


namespace lmi {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int PF_RB = 2;            // row-blocks per wave: block tile = 4 waves x 2 x 32 = 256 vectors
constexpr int PF_TILE_ROWS = 256;
#ifndef LMI_PF_STAGE_G
#define LMI_PF_STAGE_G 2
#endif
constexpr int PF_STAGE_G = LMI_PF_STAGE_G;  // k16-groups per stage -> BK = 16 * PF_STAGE_G
#ifndef LMI_PF_CAP
#define LMI_PF_CAP 1024  // 2048: duplicate-heavy data (60-100 copies of a vector) stops overflowing the buffers (tools/dup_cliff.py:
                         // 11 -> 1.7 ms at 100 copies) for +1.2 % on the benchmark (the buffers' stride doubles)
#endif
constexpr int PF_CAP = LMI_PF_CAP;        // candidate slots per (query, rank); overflow -> exact fallback
constexpr int PF_KEEP = 64;         // survivors re-scored per slot; more -> exact fallback

// ---- ingest (prefilter mode): the index keeps a bucket-contiguous ROW-MAJOR f32 copy (exact
//      re-ranking reads whole rows: 3 KiB contiguous instead of 192 scattered 16-B pieces of the
//      fragment-major layout, which cost 4x the bytes in 64-B sectors) and the fp16 fragments ----
__global__ void scatter_rows_kernel(const float* __restrict__ src, int d, const int* __restrict__ pos,
                                    long long row0, const long long* __restrict__ index, long long n_total, long long nrows,
                                    float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * d) return;
    const long long i = idx / d;
    const int k = (int)(idx - i * d);
    const long long o = index ? index[i] : row0 + i;  // index: the objects' original row numbers (owned-only ingest)
    if (o < 0 || o >= n_total) return;
    const long long p = pos[o];
    if (p >= 0) dst[p * d + k] = src[idx];
}

// global max |x| (bits of a non-negative float order like unsigned ints)
__global__ void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// scale[0] <- power of two s with max*s in [0.5, 1) (1 if max == 0); scale[1] <- 1/s
__device__ __forceinline__ float scale_of_max(unsigned maxbits) {
    const float m = __uint_as_float(maxbits);
    int e = 0;
    float s = 1.0f;
    if (m > 0.0f && m < INFINITY) { (void)frexpf(m, &e); s = ldexpf(1.0f, -e); }
    return s;
}
__global__ void make_scale_kernel(const unsigned* __restrict__ maxbits, float* __restrict__ scale) {
    const float s = scale_of_max(*maxbits);
    scale[0] = s;
    scale[1] = 1.0f / s;
}

// row-major f32 -> fp16 fragment-major (x scale): H[rb][k/16][((k>>3)&1)*32 + r][k&7].
// One thread per (slab row p, k16-group, half).
__global__ void convert16_kernel(const float* __restrict__ rows, int d, long long n_rows, int KG16,
                                 const float* __restrict__ scale, uint4* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * KG16 * 2) return;
    const int hh = (int)(idx & 1);
    const int g = (int)((idx >> 1) % KG16);
    const long long p = (idx >> 1) / KG16;
    const float s = scale[0];
    const float* x = rows + p * d;
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + 8 * hh + j;
        h[j] = (_Float16)(k < d ? x[k] * s : 0.0f);
    }
    dst[((size_t)(p >> 5) * KG16 + g) * 64 + hh * 32 + (p & 31)] = *reinterpret_cast<uint4*>(&h);
}

// Rounding-up factor of a binary32 norm: a sum of d non-negative squares accumulated in ANY order errs by at
// most d 2^-24 relative, its square root by half that (+ one rounding); the factor covers twice the bound for
// every d (a fixed 1.0002 only did up to d ~ 6 000).
__device__ __forceinline__ float norm_guard(int d) { return 1.0001f + (float)d * 5.96046448e-8f; }

// per-bucket max of the scaled row norm ||x'|| and of the norm of the row's fp16 rounding error
// ||x^ - x'|| (x' = x * scale exactly, x^ = the _Float16 image convert16_kernel stores; the difference of
// the two is exact in binary32), both rounded up (bits of non-negative floats order like ints)
__global__ void bucket_norm_kernel(const float* __restrict__ rows, int d, const int* __restrict__ rb_start,
                                   const int* __restrict__ nb_rows, const float* __restrict__ scale,
                                   unsigned* __restrict__ bnorm_bits, unsigned* __restrict__ bdelta_bits) {
    const int b = blockIdx.y;
    const int n_b = nb_rows[b];
    const float s = scale[0];
    const float guard = norm_guard(d);
    float best = 0.0f, bestd = 0.0f;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_b; row += gridDim.x * blockDim.x) {
        const float* x = rows + ((size_t)rb_start[b] * 32 + row) * d;
        float acc = 0.0f, dl = 0.0f;
        for (int k = 0; k < d; ++k) {
            const float xs = x[k] * s;
            const float e = (float)(_Float16)xs - xs;
            acc += xs * xs;
            dl += e * e;
        }
        best = fmaxf(best, sqrtf(acc) * guard);
        bestd = fmaxf(bestd, sqrtf(dl) * guard);
    }
    if (best > 0.0f) atomicMax(bnorm_bits + b, __float_as_uint(best));
    if (bestd > 0.0f) atomicMax(bdelta_bits + b, __float_as_uint(bestd));
}

// ---- hardware self-test: the bound assumes that fp16 SUBNORMAL operands enter the MFMA and the
//      f32 -> f16 conversion un-flushed (|q^ - q'| <= u|q'| + 2^-25).  One wave multiplies subnormal
//      A values (j+1)*2^-24 by B = 1024 over k = 0..15 and compares with the exact sum; lmi_create
//      runs it once and the prefilter is only enabled when it passes. ----
__global__ void pf_selftest_kernel(int* __restrict__ ok) {
    const int lane = threadIdx.x;
    half8 a, b;
    float expect = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float tiny = (float)(8 * (lane >> 5) + j + 1) * 5.9604644775390625e-8f;  // (k+1) * 2^-24
        a[j] = (_Float16)tiny;       // conversion must keep the subnormal
        b[j] = (_Float16)1024.0f;
    }
    for (int k = 0; k < 16; ++k) expect += (float)(k + 1) * 5.9604644775390625e-8f * 1024.0f;  // exact in binary32
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    bool good = true;
#pragma unroll
    for (int r = 0; r < 16; ++r) good = good && (c[r] == expect);
    const unsigned long long all = __ballot(good);
    if (lane == 0) *ok = (all == ~0ull) ? 1 : 0;
}

// ---- per batch: one launch initialises every per-call array (eight hipMemsetAsync calls took 37 us) ----
struct FillRanges {
    static constexpr int MAXR = 12;
    unsigned* p[MAXR];
    long long n[MAXR];   // 32-bit words
    unsigned v[MAXR];
    int count;
    // host side: false (and nothing written) when the list is full -- callers turn that into an error, never a stray write
    bool add(void* ptr, long long words, unsigned value) {
        if (count >= MAXR) return false;
        p[count] = static_cast<unsigned*>(ptr); n[count] = words; v[count] = value; ++count;
        return true;
    }
};
__global__ void fill_ranges_kernel(FillRanges F) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (int r = 0; r < F.count; ++r)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < F.n[r]; i += stride) F.p[r][i] = F.v[r];
}

// ---- per batch: query scale, norms, fp16 packing, per-slot bound ---------------------------------
__global__ __launch_bounds__(256) void query_norm_kernel(const float* __restrict__ q, int nq, int d,
                                                         float* __restrict__ qnorm, float* __restrict__ qdelta,
                                                         float* __restrict__ qscale) {
    // One wave per query (coalesced 16-byte loads).  Every query gets its own power-of-two scale s
    // (max|q| * s in [0.5, 1): a slot's scores are only ever compared with scores of the same query), its
    // scaled norm ||q'|| and the norm of its fp16 rounding error ||q^ - q'|| (q^ - q' is exact in binary32);
    // norm_guard(d) covers the binary32 error of the sums: both are upper bounds.
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wv;
    if (i >= nq) return;
    const float* row = q + (size_t)i * d;
    const bool vec = (d & 3) == 0;
    float m = 0.0f;
    if (vec) {
        for (int k = lane * 4; k < d; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int k = lane; k < d; k += 64) m = fmaxf(m, fabsf(row[k]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float s = scale_of_max(__float_as_uint(m));
    float acc = 0.0f, dl = 0.0f;
    auto term = [&](float v) {
        const float vs = v * s;
        const float e = (float)(_Float16)vs - vs;
        acc += vs * vs;
        dl += e * e;
    };
    if (vec) {
        for (int k = lane * 4; k < d; k += 256) {  // second pass over the row: L1/L2 hits
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            term(v.x); term(v.y); term(v.z); term(v.w);
        }
    } else {
        for (int k = lane; k < d; k += 64) term(row[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); dl += __shfl_xor(dl, o); }
    if (lane == 0) {
        qnorm[i] = sqrtf(acc) * norm_guard(d);
        qdelta[i] = sqrtf(dl) * norm_guard(d);
        qscale[i] = s;
    }
}

// colmap gather of row-major queries -> fp16 fragments (x qscale); one thread per (col-block, k16-group,
// lane): a wave writes one whole 1-KiB fragment (coalesced); each lane reads 32 contiguous bytes of its row
__global__ void pack_queries16_kernel(const float* __restrict__ q, int d, const int* __restrict__ colmap,
                                      long long ncols, int KG16, const float* __restrict__ qscale,
                                      uint4* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ncols * KG16 * 2) return;
    const int lane = (int)(idx & 63);
    const int g = (int)((idx >> 6) % KG16);
    const long long cb = (idx >> 6) / KG16;
    const int hh = lane >> 5;
    const int qi = colmap[cb * 32 + (lane & 31)];
    const float s = qi >= 0 ? qscale[qi] : 1.0f;  // the query's power-of-two scale (query_norm_kernel)
    half8 h;
    const int k0 = 16 * g + 8 * hh;
    if (qi >= 0 && k0 + 8 <= d && (d & 3) == 0) {
        const float4 lo = *reinterpret_cast<const float4*>(q + (size_t)qi * d + k0);
        const float4 hi = *reinterpret_cast<const float4*>(q + (size_t)qi * d + k0 + 4);
        h[0] = (_Float16)(lo.x * s); h[1] = (_Float16)(lo.y * s); h[2] = (_Float16)(lo.z * s); h[3] = (_Float16)(lo.w * s);
        h[4] = (_Float16)(hi.x * s); h[5] = (_Float16)(hi.y * s); h[6] = (_Float16)(hi.z * s); h[7] = (_Float16)(hi.w * s);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            h[j] = (_Float16)((qi >= 0 && k < d) ? q[(size_t)qi * d + k] * s : 0.0f);
        }
    }
    dst[idx] = *reinterpret_cast<uint4*>(&h);
}

// eps2[col] = 2*eps' of the slot occupying column `col` (header); -1 for idle columns
__global__ void slot_bound_kernel(const int* __restrict__ bucket_order, const int* __restrict__ slot_col, int nslots,
                                  int nb, int dpad, const float* __restrict__ qnorm, const float* __restrict__ qdelta,
                                  const unsigned* __restrict__ bnorm_bits, const unsigned* __restrict__ bdelta_bits,
                                  float* __restrict__ eps2) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nslots) return;
    const int col = slot_col[p];
    if (col < 0) return;
    const int b = bucket_order[p];
    const float qn = qnorm[p / nb], dq = qdelta[p / nb];
    const float xn = __uint_as_float(bnorm_bits[b]), dx = __uint_as_float(bdelta_bits[b]);
    // |<q^,x^> - <q',x'>| = |<q^-q', x^> + <q', x^-x'>| <= dq (xn + dx) + qn dx   (Cauchy-Schwarz, measured norms)
    // + the two binary32 summations: the canonical chain rounds to nearest (<= d 2^-24 sum|terms|), the MFMA's
    // internal additions are allowed to truncate (<= d 2^-23 sum|terms|): 4 d 2^-24 (qn + dq)(xn + dx) with slack
    const float e = dq * (xn + dx) + qn * dx + 4.0f * (float)dpad * 5.96046448e-8f * (qn + dq) * (xn + dx);
    eps2[col] = 2.0f * e * 1.001f;
}

// ------------------------------------------------------------------------------------------------
// Prefilter kernels: fp16 operands; block tile 256 vectors x 256 queries (8 waves = two groups of four, one
// block per CU; build option LMI_PF_NG=1: 256 x 128, 4 waves, two blocks per CU); wave = 64 vectors x <= 128
// queries, 8 accumulator tiles; both operands by LDS-DMA into a ring of three stages (32 k each, 32 KiB),
// one barrier per stage, stage u+2 in flight while stage u computes; inside a stage the fragment reads are
// inline asm, the DMA pieces sit between the MFMA groups and the last group is deferred across the barrier
// (step_fused).  (A/B on MI355X, 10M x 768: a one-stage register pipeline (A to VGPRs, B via VGPR ->
// ds_write) spent 71 % of its wave time parked on waits -- an fp16 stage is 8x shorter than an f32 one,
// shorter than the memory latency; the ring is 25 % faster.)  Two passes over the same code (template SAMPLE):
//   pass 1 (SAMPLE):  P.parts (4, 8 or 16) items per (bucket, query tile) scan every PF_SAMPLE-th 256-row
//                     tile of the whole bucket (part p takes sampled tiles p, p+parts, ..) with per-lane
//                     VALUES-ONLY top-PF_LK lists, merge the 8 lists of a column and store the part's 10
//                     best; the consumer's 10th best of the union of the parts is a lower bound of the
//                     bucket's 10th best That (-inf if the lists hold < 10 values);
//   pass 2 (!SAMPLE): items (bucket, query tile, chunk) from the XCD-affine queues; no lists, no
//                     inter-item traffic: every row with shat >= bound[col] - 2 eps' is appended to
//                     the slot's candidate buffer.  About 10 * PF_SAMPLE rows per slot pass.
// ------------------------------------------------------------------------------------------------
// (LMI_PF_SAMPLE, LMI_PF_SAMPLE_ROWS, sample_stride: lmi_kernels.h -- the routing kernels count the sampled tiles too)
#ifndef LMI_PF_RING2
#define LMI_PF_RING2 3  // ring slots of the NG 2 pass-2 kernel (3 or 4; A/B on MI355X: 4 is 1-3 % slower, more in flight only raised the load latency)
#endif
#ifndef LMI_PF_RING2_SAMPLE
#define LMI_PF_RING2_SAMPLE 4  // .. of pass 1, which waits 3-4 x longer for a stage to land (stamps) and has the LDS: -3.4 %
#endif
#ifndef LMI_PF_A_AUX
#define LMI_PF_A_AUX 0  // cache policy of the vector-fragment DMA (2 = nt: measured 3 % slower, a chunk is read by its 2 query tiles)
#endif
#ifndef LMI_PF_FUSED
#define LMI_PF_FUSED 1  // NG 2: DMA pieces interleaved with the MFMA groups of the stage
#endif
constexpr int PF_FUSED = LMI_PF_FUSED;
#ifndef LMI_PF_NG
#define LMI_PF_NG 2
#endif
constexpr int PF_NG = LMI_PF_NG;  // wave groups per prefilter block (PreItem): 1 -> 128-query tiles, 2 -> 256-query tiles
constexpr int PF_PARTS_MAX = 16;  // ... split over P.parts = 4, 8 or 16 items per (bucket, query tile) (the host picks: enough
                                 // items to fill the chip also when a rank owns 1/8 of the buckets), merged by the consumers

struct PrefilterParams {
    const uint4* slab16;
    const uint4* qfrag16;
    int KG16;  // k16-groups per row-block (multiple of PF_STAGE_G)
    int L;
    int chunk_rb;
    const int* rb_start;
    const int* nb_rows;
    const int* nch;
    const int* m;
    const int* m0;        // lmi_pass2.h pass 1: the bucket's primary columns [0, m0) are the sampled ones
    const int* cb_start;
    const int* qt_base;   // [L+1] prefix of the query-tile counts of the buckets taken heaviest first (pass-1 items)
    const int* by_work;   // [L] that order
    const int* grp_bucket;
    const int* grp_base;
    const int* grp_n;
    const int* grp_total;
    const int* grp_base1;  // lmi_pass2.h pass 1: the XCD-affine queues of its (bucket, query tile, sampled tile) items
    const int* grp_total1;
    long long ncols;       // columns of the batch (stride of the pass-1 lists of lmi_pass2.h)
    unsigned* head;       // [NGRP] pass-2 queue heads; [NGRP] = pass-1 head ([NGRP..2 NGRP): lmi_pass2.h's pass-1 queues)
    int parts;            // pass-1 items per (bucket, query tile): 4, 8 or 16
    float* bound;         // [columns][parts][4 row-waves][PF_LK] pass 1: the best sampled shat of every 64-row strip
    float* bound1;        // [columns] bound_merge_kernel: 10th best of the union of the parts -> pass 2
    const float* eps2;    // 2 eps' per column
    unsigned* cand_cnt;   // [columns]
    unsigned* cand_row;   // [columns][PF_CAP]
    float* cand_s;        // [columns][PF_CAP]
    unsigned long long* stamps;  // LMI_PF_STAMPS builds: phase cycles (behind the pass-1 lists in pf_bound)
    // second run of pass 2 for the columns whose candidate buffer overflowed (overflow_rebound_kernel); all null in the first
    const unsigned* redo_count;      // [1] columns to redo: 0 -> the launch returns at once
    const int* redo_bucket;          // [L] the bucket has such a column: its items are run again, the others skipped
    const unsigned char* redo_col;   // [columns] only these columns keep a finite threshold
};

// Pass 1 keeps the PF_LK best values per lane and column.  The bound only has to be the 10th best of SOME
// subset of the bucket's scores: the union of the 8 (row-wave, half) lists x parts of a column holds the
// sample's ten best unless five of them fall into one list; a shorter list is a cheaper insert (the
// epilogue of pass 1 is VALU-bound) and a higher entry threshold.
#ifndef LMI_PF_EPI_G
#define LMI_PF_EPI_G 1  // pass 2: score registers tested per branch of the epilogue (A/B on MI355X: 1: -2.7 %, 2: -1.6 %, 4: +11 % -- its 4-entry reservations overflow the 64-entry list)
#endif
constexpr int PF_LIST = 64 + LMI_PF_EPI_G;  // compaction-list entries per wave (+ slack for the group that overflows it)
#ifndef LMI_PF_LK
#define LMI_PF_LK 4
#endif
constexpr int PF_LK = LMI_PF_LK;
// one instruction each (fmaxf / fminf also canonicalise both operands: three); a quiet NaN operand yields the other one
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void vlist_insert(float (&v)[PF_LK], float s) {  // values-only sorted insert (descending)
    // compare-exchange chain: the carried value sinks through the list; 2 PF_LK - 1 min/max, no branch, no predicate
    // (v_max/v_min return the other operand for a NaN: a NaN never enters the list)
#pragma unroll
    for (int t = 0; t < PF_LK; ++t) {
        const float hi = vmaxf(v[t], s);
        if (t + 1 < PF_LK) s = vminf(v[t], s);
        v[t] = hi;
    }
}

// NG = wave groups per block.  NG 1: 4 waves, 256 vectors x 128 queries, two blocks per CU.  NG 2: 8 waves,
// 256 vectors x 256 queries, one block per CU: group g (waves 4g..4g+3) owns its share of the tile's
// col-blocks and both groups read the SAME staged vector fragments, so a CU fetches every vector stage
// once instead of twice and a bucket has half as many query tiles re-reading its chunks.
// In-kernel phase timing of pass 2 (developer builds, -DLMI_PF_STAMPS): shader-clock cycles per wave and phase summed into
// P.bound (free once bound_merge_kernel has run) as u64 [8 waves][12 phases]: 0 landed-wait, 1 barrier, 2 stage (MFMA + DMA
// issue), 3 epilogue rest (clearing the accumulators), 4 item start, 5 item end, 7 = tiles, 8 / 9 / 10 = the epilogue's flush /
// threshold pass / list read + position atomics.  tools/pf_stamps.py prints the table.
#ifdef LMI_PF_STAMPS
#ifndef LMI_PF_STAMPS_SAMPLE
#define LMI_PF_STAMPS_SAMPLE 0  // 1: stamp pass 1 (prefilter_kernel<true, .>) instead of pass 2
#endif
#define PF_STAMP(PH) if (SAMPLE == (LMI_PF_STAMPS_SAMPLE != 0)) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[PH] += t_ - st_last; st_last = t_; }
#define PF_STAMPS_WRITE(NVT)                                                                         \
        if (SAMPLE == (LMI_PF_STAMPS_SAMPLE != 0)) {                                                 \
            st_acc[7] = (unsigned long long)(NVT);                                                   \
            if (lane == 0) {                                                                         \
                unsigned long long* g = P.stamps + w * 12;                                           \
                for (int i = 0; i < 12; ++i) atomicAdd(g + i, st_acc[i]);                            \
            }                                                                                        \
        }
#else
#define PF_STAMP(PH)
#define PF_STAMPS_WRITE(NVT)
#endif

template <int NCB, bool SAMPLE, int NG>
struct PreItem {
    static constexpr int NLIST = SAMPLE ? NCB : 1;
    // ring slots: what is in flight (RING - 1 stages) over the load latency bounds the stage rate; NG 2's
    // one block per CU leaves LDS for a fourth slot (3 x 32 KiB in flight instead of 2 x 24 KiB x 2 blocks)
    static constexpr int RING = NG == 1 ? 3 : (SAMPLE ? LMI_PF_RING2_SAMPLE : LMI_PF_RING2);
    const PrefilterParams& P;
    uint4* sB0;
    uint4* sB1;  // three DISTINCT __shared__ B arrays [4 NG col-blocks][PF_STAGE_G][64] uint4 = 8 NG KiB each ...
    uint4* sB2;  // ... and three A arrays [4 row-waves][PF_RB][PF_STAGE_G][64] (16 KiB each): the LDS-DMA ring
    uint4 *sA0, *sA1, *sA2;
    uint4 *sB3, *sA3;       // fourth ring slot (RING 4 only)
    uint2* sList;           // pass 2: [4 NG waves][64] candidate compaction lists
    int lane, w, h, c;
    int wr, grp;            // row-wave (0..3) and wave group (0..NG-1): w = 4 grp + wr
    int cbofs;              // first col-block of this wave's group inside the tile
    float lv[NLIST][PF_LK]; // pass 1 only
    float thr[NCB];         // pass 2: bound - 2 eps' of this lane's column in col-block n (+inf: idle column)
    half8 p_a, p_b[NCB];    // NG 2: operands of the stage's deferred last MFMA group (step_fused)
    unsigned pend_pos, pend_row, pend_col;  // pass 2: this lane's candidate of the previous tile ...
    float pend_s;                           // ... whose position atomic is in flight
    f32x16 acc[PF_RB][NCB];
#ifdef LMI_PF_STAMPS
    unsigned long long st_acc[12], st_last;
#endif

    template <int SLOT>
    __device__ __forceinline__ void issue_dma(const uint4* ap0, const uint4* ap1, const uint4* qp) {
#ifdef LMI_ABL_NOLOAD  // timing-only ablation builds (tools/scan_ab.py --no-check)
        return;
#endif
#ifdef LMI_ABL_HOTA    // every block streams the same 64 KiB of A: all L2 hits
        ap0 = P.slab16 + (((size_t)(ap0 - P.slab16)) & 2047);
        ap1 = P.slab16 + (((size_t)(ap1 - P.slab16)) & 2047) + 2048;
#endif
        uint4* sA = SLOT == 0 ? sA0 : SLOT == 1 ? sA1 : SLOT == 2 ? sA2 : sA3;
        uint4* sB = SLOT == 0 ? sB0 : SLOT == 1 ? sB1 : SLOT == 2 ? sB2 : sB3;
        // this wave stages row-blocks j = grp, grp + NG, .. of its row-wave's PF_RB (ap0 [, ap1]) and col-block w
#pragma unroll
        for (int g = 0; g < PF_STAGE_G; ++g) {
            glds16(reinterpret_cast<const float4*>(ap0 + g * 64 + lane),
                   reinterpret_cast<float4*>(sA + ((wr * PF_RB + (NG == 1 ? 0 : grp)) * PF_STAGE_G + g) * 64));
            if (NG == 1)
                glds16(reinterpret_cast<const float4*>(ap1 + g * 64 + lane), reinterpret_cast<float4*>(sA + ((wr * PF_RB + 1) * PF_STAGE_G + g) * 64));
            glds16(reinterpret_cast<const float4*>(qp + g * 64 + lane), reinterpret_cast<float4*>(sB + (w * PF_STAGE_G + g) * 64));
        }
    }

    template <int SLOT>
    __device__ __forceinline__ void compute_dma() {
        const uint4* sA = (SLOT == 0 ? sA0 : SLOT == 1 ? sA1 : SLOT == 2 ? sA2 : sA3) + (wr * PF_RB) * PF_STAGE_G * 64 + lane;
        const uint4* sB = (SLOT == 0 ? sB0 : SLOT == 1 ? sB1 : SLOT == 2 ? sB2 : sB3) + cbofs * PF_STAGE_G * 64 + lane;
#pragma unroll
        for (int g = 0; g < PF_STAGE_G; ++g) {
            half8 bq[NCB];
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                const uint4 t = sB[(n * PF_STAGE_G + g) * 64];
                bq[n] = *reinterpret_cast<const half8*>(&t);
            }
#pragma unroll
            for (int j = 0; j < PF_RB; ++j) {
                const uint4 ta = sA[(j * PF_STAGE_G + g) * 64];
                const half8 av = *reinterpret_cast<const half8*>(&ta);
#pragma unroll
                for (int n = 0; n < NCB; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bq[n], acc[j][n], 0, 0, 0);
            }
        }
    }

    // NG 2: stage SLOT's MFMAs with the 4 DMA pieces of stage SLOT+RING-1 (slot DST) issued one before each
    // group of NCB MFMAs.  The fragment reads are inline asm: hipcc orders every ds_read it can see behind
    // ALL pending LDS-DMA (`s_waitcnt vmcnt(0)`: it cannot tell the ring slots apart once the loop's back
    // edge merges its bookkeeping), which drains the look-ahead; the ring's own `s_waitcnt vmcnt(N)` +
    // barrier is the real ordering.  Reads run one MFMA group ahead of their use; every fragment of the
    // stage has registers of its own (nothing an in-flight MFMA still reads is overwritten).
    template <int OFF>
    static __device__ __forceinline__ void lds_rd(half8& r, unsigned addr) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    }
    template <int G>
    __device__ __forceinline__ void rd_b(half8 (&b)[NCB], unsigned aB) {
        lds_rd<(0 * PF_STAGE_G + G) * 1024>(b[0], aB);
        if (NCB > 1) lds_rd<(1 * PF_STAGE_G + G) * 1024>(b[NCB > 1 ? 1 : 0], aB);
        if (NCB > 2) lds_rd<(2 * PF_STAGE_G + G) * 1024>(b[NCB > 2 ? 2 : 0], aB);
        if (NCB > 3) lds_rd<(3 * PF_STAGE_G + G) * 1024>(b[NCB > 3 ? 3 : 0], aB);
    }
    // the reads issued so far have landed; the operands are "defined" here for the compiler
    static __device__ __forceinline__ void lds_wait(half8& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory"); }
    static __device__ __forceinline__ void lds_wait(half8& a, half8 (&b)[NCB]) {
        if (NCB == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b[0]) :: "memory");
        if (NCB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b[0]), "+v"(b[NCB > 1 ? 1 : 0]) :: "memory");
        if (NCB == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b[0]), "+v"(b[NCB > 1 ? 1 : 0]), "+v"(b[NCB > 2 ? 2 : 0]) :: "memory");
        if (NCB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b[0]), "+v"(b[NCB > 1 ? 1 : 0]), "+v"(b[NCB > 2 ? 2 : 0]), "+v"(b[NCB > 3 ? 3 : 0]) :: "memory");
    }
    template <int J>
    __device__ __forceinline__ void mma(const half8& a, const half8 (&b)[NCB]) {
#pragma unroll
        for (int n = 0; n < NCB; ++n)
            acc[J][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n], acc[J][n], 0, 0, 0);
    }

    // The stage's last MFMA group is deferred across the next barrier (its operands p_a, p_b are already
    // in registers, so the ring slot is free): it runs while the next stage's first fragment reads are in
    // flight -- right after a barrier both waves of a SIMD wait for LDS at the same time.
    template <int SLOT, int DST>
    __device__ __forceinline__ void step_fused(const uint4* ap0, const uint4* qp, bool pending) {
        static_assert(PF_STAGE_G == 2 && PF_RB == 2, "written out for 2 k-groups x 2 row-blocks");
        const uint4* sA = (SLOT == 0 ? sA0 : SLOT == 1 ? sA1 : SLOT == 2 ? sA2 : sA3) + (wr * PF_RB) * PF_STAGE_G * 64 + lane;
        const uint4* sB = (SLOT == 0 ? sB0 : SLOT == 1 ? sB1 : SLOT == 2 ? sB2 : sB3) + cbofs * PF_STAGE_G * 64 + lane;
        const unsigned aA = (unsigned)reinterpret_cast<uintptr_t>(sA);  // generic address of LDS: low 32 bits = LDS offset
        const unsigned aB = (unsigned)reinterpret_cast<uintptr_t>(sB);
        float4* dA = reinterpret_cast<float4*>((DST == 0 ? sA0 : DST == 1 ? sA1 : DST == 2 ? sA2 : sA3) + (wr * PF_RB + grp) * PF_STAGE_G * 64);
        float4* dB = reinterpret_cast<float4*>((DST == 0 ? sB0 : DST == 1 ? sB1 : DST == 2 ? sB2 : sB3) + w * PF_STAGE_G * 64);
        half8 a00, a10, a01, a11, b0[NCB], b1[NCB];  // a<j><g>
        // fragment (j, g) of A at (j * G + g) KiB, (n, g) of B at (n * G + g) KiB
        rd_b<0>(b0, aB);
        lds_rd<(0 * PF_STAGE_G + 0) * 1024>(a00, aA);
        lds_rd<(1 * PF_STAGE_G + 0) * 1024>(a10, aA);
        if (pending) mma<1>(p_a, p_b);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait(a00, b0);
        lds_wait(a10);
        rd_b<1>(b1, aB);
        lds_rd<(0 * PF_STAGE_G + 1) * 1024>(a01, aA);
        lds_rd<(1 * PF_STAGE_G + 1) * 1024>(a11, aA);
#ifndef LMI_ABL_NOLOAD
        glds16o<0, LMI_PF_A_AUX>(reinterpret_cast<const float4*>(ap0 + lane), dA);
#endif
        mma<0>(a00, b0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef LMI_ABL_NOLOAD
        glds16o<1024, LMI_PF_A_AUX>(reinterpret_cast<const float4*>(ap0 + lane), dA);
#endif
        mma<1>(a10, b0);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait(a01, b1);
        lds_wait(a11);
#ifndef LMI_ABL_NOLOAD
        glds16(reinterpret_cast<const float4*>(qp + lane), dB);
#endif
        mma<0>(a01, b1);
        __builtin_amdgcn_sched_barrier(0);
#ifndef LMI_ABL_NOLOAD
        glds16o<1024>(reinterpret_cast<const float4*>(qp + lane), dB);
#endif
        p_a = a11;
#pragma unroll
        for (int n = 0; n < NCB; ++n) p_b[n] = b1[n];
    }

    // pass 1: per-lane values-only top-10 lists (predicated insert; a ballot + branch per score was 20 % slower)
    __device__ __forceinline__ void epilogue_sample(int rb_tile0, int n_b) {
        const bool ragged = (unsigned)((rb_tile0 + (wr + 1) * PF_RB) * 32) > (unsigned)n_b;  // wave-uniform: rows past the bucket's end
#pragma unroll
        for (int j = 0; j < PF_RB; ++j) {
            const unsigned rowbase = (unsigned)((rb_tile0 + wr * PF_RB + j) * 32);
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = acc[j][n][r];
                    if (ragged && rowbase + acc_row(r, h) >= (unsigned)n_b) s = -INFINITY;  // zero-padded / clamped rows
                    vlist_insert(lv[SAMPLE ? n : 0], s);
                    acc[j][n][r] = 0.0f;
                }
            }
        }
    }

    // pass 2: a candidate's (column, row, shat) goes to the slot's buffer at position atomicAdd(cand_cnt).
    // A returning global atomic inside the K pipeline drains the LDS-DMA look-ahead (vmcnt is one
    // in-order counter), so a tile with few candidates (the wave counts them first: <= 64 in its
    // 64 x 128 scores; expected ~16 at 83 000-row buckets) splits the emission over two tile ends:
    // the candidates are compacted through a wave-private 64-entry LDS list (ballot + mbcnt, no LDS
    // atomics) so that lane i owns entry i and issues ITS atomic; position and entry stay in 4
    // registers while the next tile computes; the stores go out at the next tile end (flush_pending),
    // long after the atomic returned.  Dense tiles (small buckets: every row is a candidate) take the
    // direct path, one returning atomic per candidate.
    __device__ __forceinline__ void flush_pending() {
        if (pend_pos < (unsigned)PF_CAP) {
            P.cand_row[(size_t)pend_col * PF_CAP + pend_pos] = pend_row;
            P.cand_s[(size_t)pend_col * PF_CAP + pend_pos] = pend_s;
        }
        pend_pos = 0xffffffffu;
    }

    __device__ __forceinline__ void epilogue_emit(int rb_tile0, int n_b, size_t col0) {
        flush_pending();
        PF_STAMP(8)
        const unsigned row0 = (unsigned)((rb_tile0 + wr * PF_RB) * 32);  // first of this wave's 64 rows
        if (row0 + 32u * PF_RB > (unsigned)n_b) {  // wave-uniform: the bucket's ragged end (zero-padded / clamped rows)
#pragma unroll
            for (int j = 0; j < PF_RB; ++j)
#pragma unroll
                for (int n = 0; n < NCB; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (row0 + j * 32 + acc_row(r, h) >= (unsigned)n_b) acc[j][n][r] = __builtin_nanf("");  // fails every >= (a bound may be -inf)
        }
        // optimistic compaction: (column, row, shat) of every passing score -> list[0..cnt); more than 64
        // discards the list and re-emits the whole tile on the direct path below
        uint2* list = sList + w * PF_LIST;
        int tot = 0;  // wave-uniform
        {
            // entry key = (column in tile) << 8 | (row in the wave's 64 rows); opaque so that the 128
            // per-(j,n,r) keys are formed where they are used (base + literal), not hoisted out of the K loop
            unsigned kb = (unsigned)((c << 8) | (4 * h));
            asm volatile("" : "+v"(kb));
            // Scores that pass are rare (~16 of the wave's 8 192), so the registers are tested in groups of LMI_PF_EPI_G
            // through their maximum and the per-register pass (ballot + compaction) runs for a group with a hit only;
            // the no-hit case is the FALL-THROUGH path (a taken branch per register cost ~40 cycles x 128).
#pragma unroll
            for (int j = 0; j < PF_RB; ++j)
#pragma unroll
                for (int n = 0; n < NCB; ++n)
#pragma unroll
                    for (int r0 = 0; r0 < 16; r0 += LMI_PF_EPI_G) {
                        // one v_max / v_max3 + v_max (fmaxf() also canonicalises both operands: 3 instructions per pair); a quiet
                        // NaN (ragged rows) never wins
                        float gm = acc[j][n][r0];
                        if (LMI_PF_EPI_G == 2) asm("v_max_f32 %0, %1, %2" : "=v"(gm) : "v"(acc[j][n][r0]), "v"(acc[j][n][r0 + (LMI_PF_EPI_G > 1 ? 1 : 0)]));
                        if (LMI_PF_EPI_G == 4) {
                            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(gm) : "v"(acc[j][n][r0]), "v"(acc[j][n][r0 + (LMI_PF_EPI_G > 1 ? 1 : 0)]), "v"(acc[j][n][r0 + (LMI_PF_EPI_G > 2 ? 2 : 0)]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(gm) : "v"(gm), "v"(acc[j][n][r0 + (LMI_PF_EPI_G > 3 ? 3 : 0)]));
                        }
                        static_assert(LMI_PF_EPI_G == 1 || LMI_PF_EPI_G == 2 || LMI_PF_EPI_G == 4, "group maximum written out for 1, 2, 4");
                        bool gpass = gm >= thr[n];  // thr = +inf for idle columns
#ifdef LMI_ABL_NOEMIT
                        gpass = gpass && thr[n] == 12345.678f;  // never true, keeps the compares alive
#endif
                        const unsigned long long gmask = __ballot(gpass);
                        if (__builtin_expect(gmask != 0ull, 0)) {
                            // every lane of the group with a hit takes LMI_PF_EPI_G consecutive list entries (one ballot per
                            // group, no branch per register); a register below the threshold leaves its entry invalid
                            if (gpass) {
                                const int my = tot + LMI_PF_EPI_G * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(gmask >> 32),
                                                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)gmask, 0u));
                                uint2* at = list + min(my, 64);  // past the list: its slack entries (the tile then takes the direct path)
#pragma unroll
                                for (int u = 0; u < LMI_PF_EPI_G; ++u) {
                                    const int r = r0 + u;
                                    const unsigned key = kb + (unsigned)(((n * 32) << 8) | (j * 32 + (r & 3) + 8 * (r >> 2)));
                                    at[u] = make_uint2(acc[j][n][r] >= thr[n] ? key : 0xffffffffu, __float_as_uint(acc[j][n][r]));
                                }
                            }
                            tot += LMI_PF_EPI_G * (int)__popcll(gmask);
                        }
                    }
        }
        PF_STAMP(9)
        if (tot > 0 && tot <= 64) {
            if (lane < tot) {
                const uint2 e = list[lane];
                if (e.x != 0xffffffffu) {
                    pend_col = (unsigned)(col0 + (e.x >> 8));
                    pend_row = row0 + (e.x & 255u);
                    pend_s = __uint_as_float(e.y);
#ifdef LMI_ABL_NOATOMIC  // timing-only ablation: no returning atomic in the K pipeline (positions collide)
                    pend_pos = (unsigned)lane;
#else
                    pend_pos = atomicAdd(P.cand_cnt + pend_col, 1u);
#endif
                }
            }
        } else if (tot > 64) {
            float t3[NCB];
#pragma unroll
            for (int n = 0; n < NCB; ++n) { t3[n] = thr[n]; asm volatile("" : "+v"(t3[n])); }
#pragma unroll
            for (int j = 0; j < PF_RB; ++j)
#pragma unroll
                for (int n = 0; n < NCB; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (acc[j][n][r] >= t3[n]) {
                            const size_t col = col0 + n * 32 + c;
                            const unsigned pos = atomicAdd(P.cand_cnt + col, 1u);
                            if (pos < (unsigned)PF_CAP) {
                                P.cand_row[col * PF_CAP + pos] = row0 + j * 32 + acc_row(r, h);
                                P.cand_s[col * PF_CAP + pos] = acc[j][n][r];
                            }
                        }
        }
        PF_STAMP(10)
#pragma unroll
        for (int j = 0; j < PF_RB; ++j)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;
    }

    // The tile = col-blocks [cbt0, cbt0 + ncb_tile) of bucket b; this wave's group owns NCB of them from cbofs
    // (`idle`: none -- the wave stages its share, computes a duplicate and emits nothing).
    // SAMPLE: `ch` is the part p in [0, P.parts): the item covers the tiles (p + P.parts*i)*PF_SAMPLE, i = 0,1,..
    // of the whole bucket and writes its 10 best values; !SAMPLE: chunk `ch`, every tile.
    __device__ __forceinline__ void run(int b, int cbt0, int ncb_tile, int cbofs_, bool idle, int ch) {
#ifdef LMI_PF_STAMPS
        for (int i = 0; i < 12; ++i) st_acc[i] = 0;
        st_last = __builtin_readcyclecounter();
#endif
        const int tid = threadIdx.x;
        lane = tid & 63; h = lane >> 5; c = lane & 31;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: stage pointers and LDS destinations stay in SGPRs
        wr = w & 3; grp = w >> 2; cbofs = cbofs_;
        const int KG = P.KG16, NS = KG / PF_STAGE_G;
        const int n_b = P.nb_rows[b];
        const int nrb_b = (n_b + 31) >> 5;
        const int nrb_all = SAMPLE ? nrb_b : min(P.chunk_rb, nrb_b - ch * P.chunk_rb);
        const int nvt_all = (nrb_all + 4 * PF_RB - 1) / (4 * PF_RB);
        // Sampling stride of the bucket: every PF_SAMPLE-th tile for large buckets, denser for small ones, so that
        // pass 2 never emits more than ~0.5 % of a bucket's rows per slot (10 x stride of them): at 160 candidates
        // per slot a 5 000-row bucket puts > 64 candidates into every 64 x 128 wave tile and the epilogue
        // leaves its fast path for one returning atomic per candidate (a few such buckets cost 7 % of pass 2).
        const int stride = SAMPLE ? sample_stride(n_b) : 1;
        const int TSTEP = SAMPLE ? stride * P.parts : 1;          // tile stride
        const int t0 = SAMPLE ? ch * stride : 0;                  // first tile
        const int nvt = nvt_all > t0 ? (nvt_all - t0 + TSTEP - 1) / TSTEP : 0;  // tiles this item processes
        const int rb_in_b0 = SAMPLE ? t0 * 4 * PF_RB : ch * P.chunk_rb;
        const int cb_tile = P.cb_start[b] + cbt0;                 // the tile's first col-block (global)
        const int m_left = idle ? 0 : P.m[b] - (cbt0 + cbofs) * 32;  // live columns from this group's first one
        const size_t col0 = (size_t)(cb_tile + cbofs) * 32;
        // wave-uniform bases; the lane's 16 bytes are added at the DMA (SGPR base + 32-bit lane offset)
        const uint4* aslab = P.slab16 + ((size_t)P.rb_start[b] * KG) * 64;
        const uint4* bbase = P.qfrag16 + ((size_t)(cb_tile + min(w, ncb_tile - 1)) * KG) * 64;
        const size_t rb_stride = (size_t)KG * 64;
        const int rb_last = nrb_b - 1;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            if (SAMPLE) {
#pragma unroll
                for (int j = 0; j < PF_LK; ++j) lv[SAMPLE ? n : 0][j] = -INFINITY;
            } else {
                const float v10 = P.bound1[col0 + n * 32 + c];
                const bool wanted = !P.redo_col || (n * 32 + c < m_left && P.redo_col[col0 + n * 32 + c]);
                thr[n] = n * 32 + c < m_left && wanted ? v10 - P.eps2[col0 + n * 32 + c] : INFINITY;
            }
#pragma unroll
            for (int j = 0; j < PF_RB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;
        }
        pend_pos = 0xffffffffu;
        // running pointers of the NEXT stage to load: this wave's share of its row-wave's vectors, col-block w
        int vt_n = 0, t_n = 0;  // vt_n counts tiles; tile index = vt * TSTEP
        const int j0 = NG == 1 ? 0 : grp;
        const uint4* ap0 = aslab + (size_t)min(rb_in_b0 + wr * PF_RB + j0, rb_last) * rb_stride;
        const uint4* ap1 = aslab + (size_t)min(rb_in_b0 + wr * PF_RB + 1, rb_last) * rb_stride;  // NG 1 only
        const uint4* qp = bbase;
        // LDS-DMA ring of RING slots: stage u+RING-1 is issued while stage u computes, so a load has
        // RING-1 stage times to land (a one-stage register pipeline left 71 % of the wave time parked on
        // waits; bytes in flight / load latency is what bounds the stage rate).  Every wave issues exactly
        // (PF_RB/NG + 1) * PF_STAGE_G DMAs per stage (6 or 4; waves whose col-block is past the tile stage
        // a duplicate) and the stream never stops (past the end the last stage is re-loaded), so "stage u
        // has landed" is a constant `s_waitcnt vmcnt((RING-2) x that)`: only the younger stages may be
        // pending (the epilogue's few stores/atomics are younger still: the wait only gets more
        // conservative).  A tile is NSR = NS rounded up to a multiple of RING stages (the extra ones load,
        // compute nothing), so every tile starts in ring slot 0 and the epilogue has ONE call site.
        static_assert((PF_RB / NG + 1) * PF_STAGE_G == (NG == 1 ? 6 : 4) && (NG == 2 || RING == 3), "vmcnt literals below");
        const int NSR = (NS + RING - 1) / RING * RING;
#define PF_ADVANCE                                                                                \
        if (++t_n < NS) { ap0 += PF_STAGE_G * 64; ap1 += PF_STAGE_G * 64; qp += PF_STAGE_G * 64; } \
        else if (t_n == NSR) {                                                                    \
            t_n = 0;                                                                              \
            if (vt_n + 1 < nvt) {                                                                 \
                ++vt_n; qp = bbase;                                                               \
                ap0 = aslab + (size_t)min(rb_in_b0 + (vt_n * TSTEP * 4 + wr) * PF_RB + j0, rb_last) * rb_stride; \
                ap1 = aslab + (size_t)min(rb_in_b0 + (vt_n * TSTEP * 4 + wr) * PF_RB + 1, rb_last) * rb_stride; \
            }                                                                                     \
        }
#ifdef LMI_ABL_NOWAIT  // timing-only ablation builds: garbage results
#define PF_WAIT_LANDED
#else
#define PF_WAIT_LANDED                                                                            \
        if (NG == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                             \
        else if (RING == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
#ifdef LMI_ABL_NOBAR
#define PF_BARRIER
#else
#define PF_BARRIER __builtin_amdgcn_s_barrier();
#endif
        // LMI_PF_FUSED (NG 2): the stage's 4 DMA pieces are issued one before each group of 4 MFMAs instead of
        // all at once after the barrier (where both waves of a SIMD sit in DMA issue with the MFMA pipe idle).
#define PF_STEP(SLOT, LIVE)                                                                       \
        PF_WAIT_LANDED                                                                            \
        PF_STAMP(0)                                                                               \
        PF_BARRIER                                                                                \
        PF_STAMP(1)                                                                               \
        if (NG == 2 && PF_FUSED && (LIVE)) {                                                      \
            step_fused<SLOT, (SLOT + RING - 1) % RING>(ap0, qp, SLOT > 0 || t > 0);               \
            PF_ADVANCE                                                                            \
        } else {                                                                                  \
            issue_dma<(SLOT + RING - 1) % RING>(ap0, ap1, qp);                                    \
            PF_ADVANCE                                                                            \
            if (LIVE) compute_dma<SLOT>();                                                        \
        }                                                                                         \
        PF_STAMP(2)
        PF_STAMP(4)
        if (nvt > 0) {
            issue_dma<0>(ap0, ap1, qp);
            PF_ADVANCE
            issue_dma<1>(ap0, ap1, qp);
            PF_ADVANCE
            if (RING == 4) {
                issue_dma<2>(ap0, ap1, qp);
                PF_ADVANCE
            }
        }
        for (int vt = 0; vt < nvt; ++vt) {
            for (int t = 0; t < NSR; t += RING) {
                PF_STEP(0, true)
                PF_STEP(1, t + 1 < NS)
                PF_STEP(2, t + 2 < NS)
                if (RING == 4) {
                    PF_STEP(3, t + 3 < NS)
                }
            }
            if (NG == 2 && PF_FUSED) mma<1>(p_a, p_b);  // the tile's last deferred group
            if (SAMPLE) epilogue_sample(rb_in_b0 + vt * TSTEP * 4 * PF_RB, n_b);
            else epilogue_emit(rb_in_b0 + vt * TSTEP * 4 * PF_RB, n_b, col0);
            PF_STAMP(3)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead before the LDS is reused
        __syncthreads();
        PF_STAMP(5)
#undef PF_STEP
#undef PF_ADVANCE
#undef PF_WAIT_LANDED
#undef PF_BARRIER
        if (!SAMPLE) {
            flush_pending();
            PF_STAMPS_WRITE(nvt)
            return;
        }
        // ---- pass 1: the lists go to bound[col][part][row-wave][PF_LK]; bound_merge_kernel takes the 10th best of a column's
        //      parts x 4 lists.  The two halves of a wave (lanes l, l ^ 32: same column, other rows) are merged here, by
        //      shuffles: any subset of the bucket's scores gives a valid bound, and one top-PF_LK list per 64 rows x sampled
        //      tiles still holds the sample's ten best unless five of them fall into the same 64 rows.  (The block-wide
        //      merge through LDS this replaces -- 8 barriers, 32 threads merging serially -- was 15 % of pass 1.) ----
        if (!idle) {
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                float other[PF_LK];
#pragma unroll
                for (int j = 0; j < PF_LK; ++j) other[j] = __shfl_xor(lv[SAMPLE ? n : 0][j], 32, 64);
#pragma unroll
                for (int j = 0; j < PF_LK; ++j) vlist_insert(lv[SAMPLE ? n : 0], other[j]);
                if (h == 0) {
                    float* bl = P.bound + (((col0 + n * 32 + c) * (size_t)P.parts + ch) * 4 + wr) * PF_LK;
#pragma unroll
                    for (int j = 0; j < PF_LK; ++j) bl[j] = lv[SAMPLE ? n : 0][j];
                }
            }
        }
        PF_STAMP(6)   // pass 1: the merge of the value lists
        PF_STAMPS_WRITE(nvt)
    }
};

// 10th best of the union of a column's nparts x 4 sampled lists (each PF_LK values, sorted descending).  One lane per list
// (LPC = 4 nparts lanes per column: 16, 32 or 64), the list in registers (one 16-byte load); ten steps of "largest head wins
// and advances", the maximum over the column's lanes by butterfly steps.
template <int LPC>
__global__ __launch_bounds__(256) void bound_merge_kernel(const float* __restrict__ lists, long long ncols, float* __restrict__ bound1) {
    const long long col = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPC;
    const int li = threadIdx.x & (LPC - 1);
    float v[PF_LK];
#pragma unroll
    for (int j = 0; j < PF_LK; ++j) v[j] = -INFINITY;
    if (col < ncols) {
        const float* src = lists + (col * LPC + li) * PF_LK;
#pragma unroll
        for (int j = 0; j < PF_LK; ++j) v[j] = src[j];
    }
    float pv = -INFINITY;
#pragma unroll
    for (int step = 0; step < KPB; ++step) {
        float m = v[0];  // the lane's head; a winner shifts its list up by one
        int who = li;
#pragma unroll
        for (int o = 1; o < LPC; o <<= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int ow = __shfl_xor(who, o, 64);
            if (om > m || (om == m && ow < who)) { m = om; who = ow; }
        }
        pv = m;  // -inf once the sample is exhausted: fewer than 10 sampled rows
        if (who == li && m > -INFINITY) {
#pragma unroll
            for (int j = 0; j + 1 < PF_LK; ++j) v[j] = v[j + 1];
            v[PF_LK - 1] = -INFINITY;
        }
    }
    if (col < ncols && li == 0) bound1[col] = pv;
}

template <bool SAMPLE, int NG>
__global__ __launch_bounds__(256 * NG, NG == 1 ? 2 : 1) void prefilter_kernel(PrefilterParams P) {
    __shared__ __attribute__((aligned(16))) uint4 sB0[4 * NG * PF_STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) uint4 sB1[4 * NG * PF_STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) uint4 sB2[4 * NG * PF_STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) uint4 sA0[4 * PF_RB * PF_STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) uint4 sA1[4 * PF_RB * PF_STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) uint4 sA2[4 * PF_RB * PF_STAGE_G * 64];
    constexpr bool RING4 = NG == 2 && (SAMPLE ? LMI_PF_RING2_SAMPLE : LMI_PF_RING2) == 4;
    __shared__ __attribute__((aligned(16))) uint4 sB3[RING4 ? 4 * NG * PF_STAGE_G * 64 : 1];
    __shared__ __attribute__((aligned(16))) uint4 sA3[RING4 ? 4 * PF_RB * PF_STAGE_G * 64 : 1];
    __shared__ uint2 sList[SAMPLE ? 1 : 4 * NG * PF_LIST];
#define PF_ITEM_ARGS P, sB0, sB1, sB2, sA0, sA1, sA2, sB3, sA3, sList
    int* s_item = reinterpret_cast<int*>(sB1);
    int grp = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (NGRP - 1));
    if (!SAMPLE && P.redo_count && *P.redo_count == 0u) return;  // the redo launch of a batch without overflowed columns
    for (;;) {
        if (threadIdx.x == 0) {
            int b = -1, local = 0;
          do {
            b = -1;
            if (SAMPLE) {  // one plain queue of (bucket, query tile) items
                const int tot = P.qt_base[P.L] * P.parts;
                const int it = (int)atomicAdd(&P.head[NGRP], 1u);
                if (it < tot) {
                    const int pair = it / P.parts;  // (bucket, query tile) pair; parts adjacent in the queue
                    int lo = 0, hi = P.L;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (P.qt_base[mid] <= pair) lo = mid; else hi = mid;
                    }
                    b = P.by_work[lo];
                    local = (pair - P.qt_base[lo]) * P.parts + (it % P.parts);
                }
            } else {
                for (int tries = 0; tries < NGRP; ++tries) {
                    const int tot = P.grp_total[grp];
                    if (__hip_atomic_load(&P.head[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tot) {
                        const int it = (int)atomicAdd(&P.head[grp], 1u);
                        if (it < tot) {
                            const int* base = P.grp_base + grp * (P.L + 1);
                            int lo = 0, hi = P.grp_n[grp];
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (base[mid] <= it) lo = mid; else hi = mid;
                            }
                            b = P.grp_bucket[grp * P.L + lo];
                            local = it - base[lo];
                            break;
                        }
                    }
                    grp = (grp + 1) & (NGRP - 1);
                }
            }
          } while (!SAMPLE && P.redo_bucket && b >= 0 && !P.redo_bucket[b]);  // redo launch: items of untouched buckets are dropped
            s_item[0] = b;
            s_item[1] = local;
        }
        __syncthreads();
        const int b = s_item[0], local = s_item[1];
        __syncthreads();
        if (b < 0) return;
        // query tiles of the bucket: its col-blocks split evenly over nqt = ceil(col-blocks / (4 NG)) tiles
        // (route_scan_kernel / route_group_kernel count the same nqt); a tile's col-blocks split over the groups
        const int ncb_b = (P.m[b] + 31) >> 5;
        const int nqt = (ncb_b + 4 * NG - 1) / (4 * NG);
        const int per = (ncb_b + nqt - 1) / nqt;
        const int qt = SAMPLE ? local / P.parts : local % nqt, ch = SAMPLE ? local % P.parts : local / nqt;
        const int cbt0 = qt * per;
        const int ncb_tile = min(per, ncb_b - cbt0);
        const int wgrp = (int)(threadIdx.x >> 8);
        const int ncb_g0 = NG == 1 ? ncb_tile : (ncb_tile + 1) >> 1;
        int ncb_w = wgrp ? ncb_tile - ncb_g0 : ncb_g0;
        int cbofs = wgrp ? ncb_g0 : 0;
        const bool idle = ncb_w == 0;
        if (idle) { ncb_w = 1; cbofs = 0; }
        switch (ncb_w) {
            case 1: { PreItem<1, SAMPLE, NG> it{PF_ITEM_ARGS}; it.run(b, cbt0, ncb_tile, cbofs, idle, ch); break; }
            case 2: { PreItem<2, SAMPLE, NG> it{PF_ITEM_ARGS}; it.run(b, cbt0, ncb_tile, cbofs, idle, ch); break; }
            case 3: { PreItem<3, SAMPLE, NG> it{PF_ITEM_ARGS}; it.run(b, cbt0, ncb_tile, cbofs, idle, ch); break; }
            default: { PreItem<4, SAMPLE, NG> it{PF_ITEM_ARGS}; it.run(b, cbt0, ncb_tile, cbofs, idle, ch); break; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Select + exact re-rank: one wave per (query, rank) slot.
//   That = 10th largest shat of the slot's candidates (every row of the shat-top-10 was emitted);
//   survivors = candidates with shat >= That - 2 eps' (a superset of the canonical top-10, header);
//   each survivor is re-scored by one lane with the canonical chain acc = fmaf(q[k], x[k], acc)
//   on the f32 slab; the 10 best by (score desc, row asc) become the slot's rank list, in the same
//   form merge_kernel's phase A writes (dist = 1 - s, ids, faiss padding for short buckets).
// Overflow: a column with more than PF_CAP emitted candidates first gets a tighter bound and a second run of pass 2
// (overflow_rebound_kernel); a slot still over PF_CAP, or with more survivors than the re-rank holds (PF_KEEP here, RC_KEEP
// on the streamed path), sets fallback[p]: fallback_kernel re-scores its candidates (buffer complete) or its whole bucket.
// ------------------------------------------------------------------------------------------------
struct RescoreParams {
    const int* bucket_order;
    const int* slot_col;
    int nslots, nb, d, raw;
    const int* rb_start;
    const int* nb_rows;
    const unsigned* cand_cnt;
    const unsigned* cand_row;
    const float* cand_s;
    const float* eps2;
    const float* rows;  // bucket-contiguous row-major f32 [slab rows][d]
    const float* q;     // row-major [nq][d]
    const float* qn2;   // L2 metric: |q|^2 per query (nullptr: inner product)
    const unsigned* ids_slab;
    float* rank_d;
    unsigned* rank_id;
    int* fallback;
    int* nkeep;  // [nslots] survivors re-scored (statistics; summed on request)
};

// canonical similarity of slab row p with query qv[0..d): acc = fmaf(q[k], x[k], acc), k ascending
__device__ __forceinline__ float exact_score(const float* __restrict__ rows, size_t p, const float* __restrict__ qv, int d) {
    const float* x = rows + p * d;
    float acc = 0.0f;
    int k = 0;
    if ((d & 3) == 0) {  // rows are 16-byte aligned: vector loads, same k order
        for (; k < d; k += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k);
            const float4 qq = *reinterpret_cast<const float4*>(qv + k);
            acc = __builtin_fmaf(qq.x, xv.x, acc); acc = __builtin_fmaf(qq.y, xv.y, acc);
            acc = __builtin_fmaf(qq.z, xv.z, acc); acc = __builtin_fmaf(qq.w, xv.w, acc);
        }
    } else {
        for (; k < d; ++k) acc = __builtin_fmaf(qv[k], x[k], acc);
    }
    return acc;
}

// lanes 0..9 hold the sorted (score, row) results; writes the slot's rank list (merge phase A form)
__device__ __forceinline__ void write_rank_list(int lane, float my_s, unsigned my_r, int n_b, int rb0, int raw,
                                                const unsigned* __restrict__ ids_slab, float* rd, unsigned* ri,
                                                const float* qn2, int q) {
    const float FMAXV = 3.402823466e+38f;
    if (lane < KPB) {
        const bool real = my_r != NOROW && lane < n_b;
        float dv;
        unsigned iv;
        if (raw) { dv = real ? my_s : -FMAXV; iv = real ? my_r : NOROW; }
        else if (real) { dv = sim_to_dist(my_s, qn2, q); iv = ids_slab[(size_t)rb0 * 32 + my_r]; }
        else { dv = pad_dist(qn2); iv = ids_slab[(size_t)rb0 * 32 + (n_b - 1)]; }  // faiss padding (Q4)
        rd[lane] = dv;
        ri[lane] = iv;
    }
}

// 4 slots per block (one wave each).  The candidates live in registers (<= 16 per lane) for the ten
// selection passes; the query is staged in LDS once per wave; every survivor's row is streamed in
// 128-byte pieces (8 independent 16-byte loads in flight per lane) through the k-ordered fmaf chain.
constexpr int RS_WAVES = 4;
#ifndef LMI_RS_AHEAD
#define LMI_RS_AHEAD 8  // 128-byte lines of a survivor's row requested ahead of the chain (select_rescore_kernel)
#endif
#ifndef LMI_RS_WAVES_PER_EU
#define LMI_RS_WAVES_PER_EU 6  // the kernel is latency-bound (random 3-KiB rows, a serial chain): occupancy is its parallelism
#endif
constexpr int RS_MAXD = 1024;  // queries up to this many dims are staged in LDS (else read from L2)

__global__ __launch_bounds__(64 * RS_WAVES) __attribute__((amdgpu_waves_per_eu(LMI_RS_WAVES_PER_EU))) void select_rescore_kernel(RescoreParams P) {
    __shared__ unsigned keep_row[RS_WAVES][PF_KEEP];
    __shared__ __attribute__((aligned(16))) float qs[RS_WAVES][RS_MAXD];
    __shared__ unsigned sink[RS_WAVES][64];  // destination of the prefetching LDS-DMA loads (never read)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * RS_WAVES + wv;
    if (p >= P.nslots) return;
    float* rd = P.rank_d + (size_t)p * KPB;
    unsigned* ri = P.rank_id + (size_t)p * KPB;
    const int col = P.slot_col[p];
    const float FMAXV = 3.402823466e+38f;
    if (lane == 0) { P.fallback[p] = 0; P.nkeep[p] = 0; }
    if (col < 0) {  // unvisited (LearnedIndex.py:340-341)
        if (lane < KPB) { rd[lane] = P.raw ? -FMAXV : INFINITY; ri[lane] = P.raw ? NOROW : 0u; }
        return;
    }
    const unsigned cnt = P.cand_cnt[col];
    if (cnt > (unsigned)PF_CAP) {
        if (lane == 0) P.fallback[p] = 1;
        return;
    }
    // stage the query (coalesced) while the candidates load
    const float* qg = P.q + (size_t)(p / P.nb) * P.d;
    const bool q_lds = P.d <= RS_MAXD;
    if (q_lds)
        for (int k = lane; k < P.d; k += 64) qs[wv][k] = qg[k];
    const float* cs = P.cand_s + (size_t)col * PF_CAP;
    const unsigned* cr = P.cand_row + (size_t)col * PF_CAP;
    // The kernel is latency-bound (random 3-KiB rows, one serial chain per survivor), so its speed is its
    // occupancy: registers are kept to the candidates' order-preserving integer images (the rows are
    // re-read for the ~14 survivors only) and one 128-byte piece of the row in flight per lane.
    constexpr int PER = PF_CAP / 64;
    const int nper = (int)((cnt + 63u) >> 6);  // wave-uniform: register slots in use (~4 of 16 at C2)
    unsigned key[PER];  // monotone image of shat; 0 = no candidate
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = lane + 64 * i;
        key[i] = 0u;
        if (i < nper && e < (int)cnt) {
            const unsigned bits = __float_as_uint(cs[e]);
            key[i] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
        }
    }
    // That = 10th largest shat: bisection on the keys, counting with ballots (32 x nper compares and
    // scalar popcounts; ten argmax passes over the wave cost 120 ds_bpermute round trips per slot).
    // Fewer than 10 candidates: -inf, everything survives.
    float pv = -INFINITY;
    if (cnt >= (unsigned)KPB) {
        unsigned T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned probe = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (i < nper) c += (int)__popcll(__ballot(key[i] >= probe));
            if (c >= KPB) T = probe;
        }
        pv = __uint_as_float(T ^ ((T >> 31) ? 0x80000000u : 0xffffffffu));
    }
    const float cut = pv - P.eps2[col];
    const unsigned cbits = __float_as_uint(cut);
    // key of the cut; (-inf) - eps = -inf maps below every candidate's key
    const unsigned kcut = cut != cut ? 1u : cbits ^ ((cbits >> 31) ? 0xffffffffu : 0x80000000u);
    // survivors -> keep_row[] (order is irrelevant: the final sort is by (score, row))
    unsigned nk = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (i < nper) {
            const bool keep = key[i] != 0u && key[i] >= kcut;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const unsigned k = nk + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                if (k < (unsigned)PF_KEEP) keep_row[wv][k] = cr[lane + 64 * i];
            }
            nk += (unsigned)__popcll(bal);
        }
    }
    if (nk > (unsigned)PF_KEEP) {
        if (lane == 0) P.fallback[p] = 1;
        return;
    }
    if (lane == 0) P.nkeep[p] = (int)nk;
    // qs[wv] / keep_row[wv] are private to this wave (other waves of the block may have returned):
    // LDS is in-order per wave, a wave-level fence is all that is needed
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int b = P.bucket_order[p];
    const int rb0 = P.rb_start[b], n_b = P.nb_rows[b];
    float s = -INFINITY;
    unsigned row = NOROW;
    if (lane < (int)nk) {
        row = keep_row[wv][lane];
        const float* x = P.rows + ((size_t)rb0 * 32 + row) * P.d;
        const float* qv = q_lds ? qs[wv] : qg;
        float acc = 0.0f;
        int k = 0;
        if (q_lds && (P.d & 31) == 0) {
            // One 128-byte line of the row per step and lane.  With only that line outstanding the DRAM
            // sees 24 isolated accesses per 3-KiB row (2.1 TB/s measured).  Registers for more lines cost
            // occupancy, so the lines RS_AHEAD steps ahead are pulled into L2/MALL by 4-byte-per-lane
            // LDS-DMA loads into a sink (no VGPRs; issued AFTER the step's own loads, since vmcnt returns in
            // order).  The query is read from LDS by inline asm: hipcc would order a visible ds_read
            // behind every pending LDS-DMA.
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const unsigned qaddr = (unsigned)reinterpret_cast<uintptr_t>(&qs[wv][0]);
            constexpr int RS_AHEAD = LMI_RS_AHEAD;
#pragma unroll
            for (int j = 1; j <= RS_AHEAD; ++j)
                if (j * 32 < P.d)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + j * 32),
                                                     (__attribute__((address_space(3))) void*)&sink[wv][0], 4, 0, 0);
            for (; k < P.d; k += 32) {
                // the step's line by inline asm, then exactly one prefetch (past the end: the last line again),
                // then `vmcnt(1)`: the line has landed, the prefetch may still be out (hipcc waits for vmcnt(0))
                f32x4 xv[8];
                const float* xk = x + k;
                asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
                             "global_load_dwordx4 %2, %8, off offset:32\n\tglobal_load_dwordx4 %3, %8, off offset:48\n\t"
                             "global_load_dwordx4 %4, %8, off offset:64\n\tglobal_load_dwordx4 %5, %8, off offset:80\n\t"
                             "global_load_dwordx4 %6, %8, off offset:96\n\tglobal_load_dwordx4 %7, %8, off offset:112"
                             : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(xv[4]), "=&v"(xv[5]), "=&v"(xv[6]), "=&v"(xv[7])
                             : "v"(xk) : "memory");
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + min(k + (RS_AHEAD + 1) * 32, P.d - 32)),
                                                 (__attribute__((address_space(3))) void*)&sink[wv][0], 4, 0, 0);
                asm volatile("s_waitcnt vmcnt(1)"
                             : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                             :: "memory");
                f32x4 qq[8];
                const unsigned qa = qaddr + (unsigned)k * 4u;
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                             "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                             "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(qq[0]), "=&v"(qq[1]), "=&v"(qq[2]), "=&v"(qq[3]), "=&v"(qq[4]), "=&v"(qq[5]), "=&v"(qq[6]), "=&v"(qq[7])
                             : "v"(qa) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = __builtin_fmaf(qq[i].x, xv[i].x, acc); acc = __builtin_fmaf(qq[i].y, xv[i].y, acc);
                    acc = __builtin_fmaf(qq[i].z, xv[i].z, acc); acc = __builtin_fmaf(qq[i].w, xv[i].w, acc);
                }
            }
        } else if ((P.d & 3) == 0) {
            for (; k + 32 <= P.d; k += 32) {  // 8 independent 16-byte loads in flight, then 32 chained fmas
                float4 xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4*>(x + k + 4 * i);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 qq = *reinterpret_cast<const float4*>(qv + k + 4 * i);
                    acc = __builtin_fmaf(qq.x, xv[i].x, acc); acc = __builtin_fmaf(qq.y, xv[i].y, acc);
                    acc = __builtin_fmaf(qq.z, xv[i].z, acc); acc = __builtin_fmaf(qq.w, xv[i].w, acc);
                }
            }
        }
        for (; k < P.d; ++k) acc = __builtin_fmaf(qv[k], x[k], acc);
        s = acc;
    }
    // 10 best by (score desc, row asc): a survivor's output position is the number of survivors that beat
    // it (rows are distinct, so positions are too); the owner lane writes the entry itself.
    int pos = 0;
    for (int l = 0; l < (int)nk; ++l) {
        const float os = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), l));
        const unsigned orr = (unsigned)__builtin_amdgcn_readlane((int)row, l);
        pos += better(os, orr, s, row) ? 1 : 0;
    }
    const int nreal = min(min((int)nk, KPB), n_b);
    if (lane < (int)nk && pos < KPB) {
        rd[pos] = P.raw ? s : sim_to_dist(s, P.qn2, p / P.nb);
        ri[pos] = P.raw ? row : P.ids_slab[(size_t)rb0 * 32 + row];
    }
    if (lane >= nreal && lane < KPB) {  // faiss padding (Q4): -FLT_MAX similarity, last id of the bucket
        rd[lane] = P.raw ? -FMAXV : pad_dist(P.qn2);
        ri[lane] = P.raw ? NOROW : P.ids_slab[(size_t)rb0 * 32 + (n_b - 1)];
    }
}

// statistics on request: out[0] += survivors, out[1] += fallback slots
// Candidate-buffer overflow (more than PF_CAP rows passed a column's threshold: the sampled bound was loose, typically many
// copies of the same vectors near the top).  The PF_CAP candidates that were stored are a subset of the bucket: the 10th best
// of their scores is a valid -- and much tighter -- lower bound of That.  The column gets that bound, an empty buffer and a
// flag; prefilter_kernel<false> is launched once more and re-runs the buckets that hold such columns with every other
// column's threshold at +inf.  Without this such a slot took the exact fallback (one block brute-forcing the bucket).
__global__ void overflow_rebound_kernel(const int* __restrict__ slot_col, const int* __restrict__ bucket_order, int nslots,
                                        unsigned* __restrict__ cand_cnt, const float* __restrict__ cand_s, float* __restrict__ bound1,
                                        unsigned* __restrict__ redo_count, int* __restrict__ redo_bucket, unsigned char* __restrict__ redo_col) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nslots) return;
    const int col = slot_col[p];
    if (col < 0 || cand_cnt[col] <= (unsigned)PF_CAP) return;
    float v[KPB];
#pragma unroll
    for (int j = 0; j < KPB; ++j) v[j] = -INFINITY;
    const float* cs = cand_s + (size_t)col * PF_CAP;
    for (int i = 0; i < PF_CAP; ++i) {
        float s = cs[i];
        if (s > v[KPB - 1]) {
#pragma unroll
            for (int t = 0; t < KPB; ++t) {  // sorted insert (descending)
                const float hi = fmaxf(v[t], s);
                s = fminf(v[t], s);
                v[t] = hi;
            }
        }
    }
    if (v[KPB - 1] > bound1[col]) bound1[col] = v[KPB - 1];
    cand_cnt[col] = 0u;
    redo_col[col] = 1;
    redo_bucket[bucket_order[p]] = 1;
    atomicAdd(redo_count, 1u);
}

__global__ void prefilter_stats_kernel(const int* __restrict__ nkeep, const int* __restrict__ fallback, int nslots,
                                       unsigned long long* __restrict__ out) {
    unsigned long long a = 0, b = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nslots; p += gridDim.x * blockDim.x) { a += nkeep[p]; b += fallback[p]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, a); atomicAdd(out + 1, b); }
}

// Exact fallback for overflowed slots: one block per slot, brute force over the whole bucket with the
// canonical chain on the VALU (slow, rare, always correct).
__global__ __launch_bounds__(256) void fallback_kernel(RescoreParams P) {
    __shared__ float fs[256 * KPB];
    __shared__ unsigned fr[256 * KPB];
    const int tid = threadIdx.x, lane = tid & 63;
    // a block looks at 256 slots' flags at once and works through the flagged ones (a launch of one block
    // per slot spent 13 us finding nothing to do)
    __shared__ int todo[256], ntodo;
    if (tid == 0) ntodo = 0;
    __syncthreads();
    {
        const int mine = blockIdx.x * 256 + tid;
        if (mine < P.nslots && P.fallback[mine]) todo[atomicAdd(&ntodo, 1)] = mine;
    }
    __syncthreads();
    const int nt = ntodo;
    for (int ti = 0; ti < nt; ++ti) {
    const int p = todo[ti];
    const int b = P.bucket_order[p];
    const int rb0 = P.rb_start[b], n_b = P.nb_rows[b];
    const float* qv = P.q + (size_t)(p / P.nb) * P.d;
    float v[KPB];
    unsigned id[KPB];
#pragma unroll
    for (int j = 0; j < KPB; ++j) { v[j] = -INFINITY; id[j] = NOROW; }
    // A slot whose candidate buffer did NOT overflow is here for its survivor count only (more rows within 2 eps' of its
    // top-10 than the re-rank holds: hundreds of copies of a vector): its candidates are a complete superset of the
    // canonical top-10, so re-scoring those <= PF_CAP rows is exact and ~100 x cheaper than the whole bucket.
    const int ccol = P.slot_col[p];
    const unsigned ccnt = ccol >= 0 ? P.cand_cnt[ccol] : 0xffffffffu;
    if (ccnt <= (unsigned)PF_CAP) {
        const unsigned* cr = P.cand_row + (size_t)ccol * PF_CAP;
        for (unsigned it = tid; it < ccnt; it += 256) {
            const unsigned row = cr[it];
            const float s = exact_score(P.rows, (size_t)rb0 * 32 + row, qv, P.d);
            if (better(s, row, v[KPB - 1], id[KPB - 1])) {  // candidates come in no order: ties by row here
#pragma unroll
                for (int t = KPB - 1; t > 0; --t) {   // list_insert with the (score desc, row asc) order
                    const bool shift = better(s, row, v[t - 1], id[t - 1]);
                    const bool here = better(s, row, v[t], id[t]);
                    id[t] = shift ? id[t - 1] : (here ? row : id[t]);
                    v[t] = shift ? v[t - 1] : (here ? s : v[t]);
                }
                if (better(s, row, v[0], id[0])) { v[0] = s; id[0] = row; }
            }
        }
    } else {
    for (unsigned row = tid; row < (unsigned)n_b; row += 256) {
        const float s = exact_score(P.rows, (size_t)rb0 * 32 + row, qv, P.d);
        if (s > v[KPB - 1]) list_insert(v, id, s, row);  // rows ascend per thread: strict > keeps the earlier
    }
    }
#pragma unroll
    for (int j = 0; j < KPB; ++j) { fs[tid * KPB + j] = v[j]; fr[tid * KPB + j] = id[j]; }
    __syncthreads();
    if (tid < 64) {
    // wave 0: lane owns lists lane, lane+64, lane+128, lane+192 (heads 4 bits each)
    unsigned heads = 0;
    float my_s = -INFINITY;
    unsigned my_r = NOROW;
    for (int j = 0; j < KPB; ++j) {
        float bs = -INFINITY;
        unsigned br = NOROW;
        int bl = -1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hd = (heads >> (4 * t)) & 15;
            if (hd < KPB) {
                const float s = fs[(lane + 64 * t) * KPB + hd];
                const unsigned r = fr[(lane + 64 * t) * KPB + hd];
                if (better(s, r, bs, br)) { bs = s; br = r; bl = t; }
            }
        }
        int wl = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o);
            const unsigned orr = __shfl_xor(br, o);
            const int ol = __shfl_xor(wl, o);
            if (better(os, orr, bs, br) || (os == bs && orr == br && ol < wl)) { bs = os; br = orr; wl = ol; }
        }
        if (wl == lane && bl >= 0) heads += 1u << (4 * bl);
        if (lane == j) { my_s = bs; my_r = br; }
    }
    write_rank_list(lane, my_s, my_r, n_b, rb0, P.raw, P.ids_slab, P.rank_d + (size_t)p * KPB,
                    P.rank_id + (size_t)p * KPB, P.qn2, p / P.nb);
    }
    __syncthreads();  // fs / fr are reused by the block's next slot
    }
}

}  // namespace lmi
