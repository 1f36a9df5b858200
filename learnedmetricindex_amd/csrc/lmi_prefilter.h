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

constexpr int PF_STAGE_G = 2;       // k16-groups per stage of the scan kernels (lmi_pass2.h): the slab pads K to a multiple of 32
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
                                    float* __restrict__ dst, int pitch) {   // pitch: floats per row of dst (d rounded up to 4, zero-filled)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * d) return;
    const long long i = idx / d;
    const int k = (int)(idx - i * d);
    const long long o = index ? index[i] : row0 + i;  // index: the objects' original row numbers (owned-only ingest)
    if (o < 0 || o >= n_total) return;
    const long long p = pos[o];
    if (p >= 0) dst[p * pitch + k] = src[idx];
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

// row-major f32 -> fp16 fragment-major (x scale).  Two fragment shapes (one 1-KiB fragment = one MFMA operand, a lane's 16 bytes = 8
// consecutive k of one row):
// (the host picks by the kernel family that will read them: lmi_hip.hip, frag16x16())
//   K <= 128 (lmi_pass2_small.h, v_mfma_f32_32x32x16_f16):  H[rb][k/16][lane = 32 ((k >> 3) & 1) + r][k & 7]              (32 rows x 16 k)
//   K  > 128 (lmi_pass2.h, v_mfma_f32_16x16x32_f16):        H[rb][k/32][r >> 4][lane = 16 ((k >> 3) & 3) + (r & 15)][k & 7]  (16 rows x 32 k)
// (r = row in its 32-row block.)  Either way a row-block's fragments of 32 consecutive k are 2 KiB side by side.
// One thread per (slab row p, k16-group, half).
__global__ void convert16_kernel(const float* __restrict__ rows, int d, int pitch, long long n_rows, int KG16,
                                 const float* __restrict__ scale, uint4* __restrict__ dst, int f16x16 /* 1: the K > 128 shape */) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * KG16 * 2) return;
    const int hh = (int)(idx & 1);
    const int g = (int)((idx >> 1) % KG16);
    const long long p = (idx >> 1) / KG16;
    const float s = scale[0];
    const float* x = rows + p * pitch;
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + 8 * hh + j;
        h[j] = (_Float16)(k < d ? x[k] * s : 0.0f);
    }
    const int r = (int)(p & 31);
    if (f16x16) {
        const int kq = 2 * (g & 1) + hh;   // which 8 of the 32-k step
        dst[(((size_t)(p >> 5) * (KG16 / 2) + (g >> 1)) * 2 + (r >> 4)) * 64 + 16 * kq + (r & 15)] = *reinterpret_cast<uint4*>(&h);
    } else {
        dst[((size_t)(p >> 5) * KG16 + g) * 64 + hh * 32 + r] = *reinterpret_cast<uint4*>(&h);
    }
}

// Rounding-up factor of a binary32 norm: a sum of d non-negative squares accumulated in ANY order errs by at
// most d 2^-24 relative, its square root by half that (+ one rounding); the factor covers twice the bound for
// every d (a fixed 1.0002 only did up to d ~ 6 000).
__device__ __forceinline__ float norm_guard(int d) { return 1.0001f + (float)d * 5.96046448e-8f; }

// per-bucket max of the scaled row norm ||x'|| and of the norm of the row's fp16 rounding error
// ||x^ - x'|| (x' = x * scale exactly, x^ = the _Float16 image convert16_kernel stores; the difference of
// the two is exact in binary32), both rounded up (bits of non-negative floats order like ints)
__global__ void bucket_norm_kernel(const float* __restrict__ rows, int d, int pitch, const int* __restrict__ rb_start,
                                   const int* __restrict__ nb_rows, const float* __restrict__ scale,
                                   unsigned* __restrict__ bnorm_bits, unsigned* __restrict__ bdelta_bits) {
    const int b = blockIdx.y;
    const int n_b = nb_rows[b];
    const float s = scale[0];
    const float guard = norm_guard(d);
    float best = 0.0f, bestd = 0.0f;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_b; row += gridDim.x * blockDim.x) {
        const float* x = rows + ((size_t)rb_start[b] * 32 + row) * pitch;
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
    unsigned long long* ts;   // nullable: device stamp of the launch's start (the first kernel of a scan on the separate-kernels route)
    // host side: false (and nothing written) when the list is full -- callers turn that into an error, never a stray write
    bool add(void* ptr, long long words, unsigned value) {
        if (count >= MAXR) return false;
        p[count] = static_cast<unsigned*>(ptr); n[count] = words; v[count] = value; ++count;
        return true;
    }
};
__global__ void fill_ranges_kernel(FillRanges F) {
    ts_first(F.ts);
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
                                      uint4* __restrict__ dst, int f16x16 /* 1: the K > 128 shape */) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ncols * KG16 * 2) return;
    const int lane = (int)(idx & 63);
    const int g = (int)((idx >> 6) % KG16);
    const long long cb = (idx >> 6) / KG16;
    // the fragment shapes of convert16_kernel: 32 columns x 16 k (K <= 128) or 16 columns x 32 k (K > 128: fragment (k/32, column half))
    const bool f16 = f16x16 != 0;
    const int colb = f16 ? 16 * (g & 1) + (lane & 15) : (lane & 31);
    const int k0 = f16 ? 32 * (g >> 1) + 8 * (lane >> 4) : 16 * g + 8 * (lane >> 5);
    const int qi = colmap[cb * 32 + colb];
    const float s = qi >= 0 ? qscale[qi] : 1.0f;  // the query's power-of-two scale (query_norm_kernel)
    half8 h;
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
// The prefilter's scan kernels (pass 1: sampled bounds, pass 2: candidate emission) are in lmi_pass2.h.  (Rounds 1-2 ran
// `prefilter_kernel<SAMPLE, NG>` from this file: 256 x 256 block tiles, a wave = 64 vectors x <= 128 queries, two query tiles per
// bucket; removed in round 3 -- git history and DESIGN.md sections 5b / 5e keep its measurements.)
// ------------------------------------------------------------------------------------------------
#ifndef LMI_PF_A_AUX
#define LMI_PF_A_AUX 0  // cache policy of the vector-fragment DMA (2 = nt)
#endif

// Where a column's candidates past its own PF_CAP entries go (round 4): ONE append-only log shared by all columns -- an entry is
// (column, index past the buffer | launch << 31, row, score), a position comes from one bump atomic: a dozen instructions at the
// emission sites (anything heavier inlined into their unrolled register loops sent the accumulators to scratch).  Behind pass 2
// overflow_rebound_kernel gives every overflowed column a range of the sorted array (one bump atomic per column), the selection
// kernel's blocks first scatter the log into those ranges (pf_x_scatter: nothing to do on an empty log -- every batch of the
// bench), and fallback_kernel re-scores a slot with thousands of rows inside 2 eps' of its top ten (duplicate-heavy data) from
// its candidates instead of brute-forcing its bucket.  A launch that found the log full sets its fail flag and is handled as
// in round 3 (tighter bound + second run of pass 2 without a log, then the exact fallback).
#ifndef LMI_PF_X_LOG2
#define LMI_PF_X_LOG2 22   // 4 M entries: 64 MiB of log + 32 MiB sorted, per handle, allocated when first needed... (lmi_hip.hip)
#endif
struct OverflowLog {
    uint4* log;          // [cap]
    unsigned cap;        // 0: no log
    unsigned* head;      // [1] entries appended by this batch's launches (may run past cap)
    unsigned* fail;      // [1] this launch's flag: an append found the log full
    unsigned launch;     // 0: pass 2, 1: its redo launch (the columns it emits again start from index 0)
};
__device__ __forceinline__ void pf_x_append(const OverflowLog& X, size_t col, unsigned e, unsigned row, float s) {
    // (one bump atomic per lane; aggregating them per wave-instruction -- ballot, leader, shuffle -- was measured on duplicate-heavy
    // batches of ~480 k appends: 2.9 against 3.0 ms, and it costs the scan kernels registers around every emission site)
    const unsigned i = X.cap ? atomicAdd(X.head, 1u) : 0xffffffffu;
    if (i < X.cap) X.log[i] = make_uint4((unsigned)col, e | (X.launch << 31), row, __float_as_uint(s));
    else atomicOr(X.fail, 1u);
}

struct PrefilterParams {
    const uint4* slab16;
    const uint4* qfrag16;
    int KG16;  // k16-groups per row-block (d > 128: a multiple of 2, a stage of pass2_kernel holds two)
    int L;
    int chunk_rb;
    const int* chunk_rb_b;  // nullable [L]: the call's row-blocks per chunk of each bucket (lmi_kernels.h: graded pass-2 items); null: chunk_rb
    int sample_max;  // pass 1's largest sampling stride (lmi_kernels.h: sample_stride)
    int tile_cb;  // col-blocks per query tile (12; the low-dimensional kernels at K = 65..96: 8, lmi_pass2_small.h)
    const int* rb_start;
    const int* nb_rows;
    const int* nch;
    const int* m;
    const int* m0;        // pass 1: the bucket's primary columns [0, m0) are sampled at the full rate (route_count_kernel)
    const int* cb_start;
    const int* grp_bucket;
    const int* grp_base;
    const int* grp_n;
    const int* grp_total;
    const int* grp_base1;  // pass 1: the XCD-affine queues of its (bucket, query tile, sampled tile) items
    const int* grp_total1;
    long long ncols;       // columns of the batch (stride of the pass-1 lists)
    unsigned* head;       // [0, NGRP) pass-2 queue heads; [16, 16 + NGRP): the heads of pass 2's redo launch; [24, 24 + NGRP): pass 1
    float* bound;         // [P2_NSL lists][16 slots][columns] pass 1: slot maxima of the sampled tiles (lmi_pass2.h)
    float* bound1;        // [columns] bound_merge2_kernel (+ query_bound_kernel): the lower bound of That pass 2 emits against
    const float* eps2;    // 2 eps' per column
    unsigned* cand_cnt;   // [columns]
    unsigned* cand_row;   // [columns][PF_CAP]
    float* cand_s;        // [columns][PF_CAP]
    unsigned long long* stamps;  // LMI_P2_STAMPS builds: phase cycles (behind the pass-1 lists in pf_bound)
    unsigned long long* ts_start;     // nullable: device stamp of the launch's start (lmi_kernels.h: ts_first)
    unsigned long long* ts_end_cell;  // nullable: the launch's end as the maximum over its workgroups (a per-call zeroed cell; ts_max)
    // second run of pass 2 for the columns whose candidate buffer overflowed (overflow_rebound_kernel); all null in the first
    const unsigned* redo_count;      // [1] columns to redo: 0 -> the launch returns at once
    const int* redo_bucket;          // [L] the bucket has such a column: its items are run again, the others skipped
    const unsigned char* redo_col;   // [columns] only these columns keep a finite threshold
    OverflowLog x;                   // where a column's candidates past PF_CAP go
};
// a candidate of column `col` whose position atomic returned `pos`
__device__ __forceinline__ void cand_store(const PrefilterParams& P, size_t col, unsigned pos, unsigned row, float s) {
    if (__builtin_expect(pos < (unsigned)PF_CAP, 1)) {
        P.cand_row[col * PF_CAP + pos] = row;
        P.cand_s[col * PF_CAP + pos] = s;
    } else {
        pf_x_append(P.x, col, pos - (unsigned)PF_CAP, row, s);
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
    const float* rows;  // bucket-contiguous row-major f32 [slab rows][dp], columns d .. dp - 1 zero
    int dp;             // floats per row of `rows`: d rounded up to a multiple of 4 (16-byte rows for the streamed re-rank)
    const float* q;     // row-major [nq][d]
    const float* qn2;   // L2 metric: |q|^2 per query (nullptr: inner product)
    const unsigned* ids_slab;
    float* rank_d;
    unsigned* rank_id;
    int* fallback;
    int* nkeep;  // [nslots] survivors re-scored (statistics; summed on request)
    // fallback slots (select_kernel / rescore appends; fallback_kernel's blocks stride over the list) and where their candidates
    // past PF_CAP are (the shared overflow table: epoch of the first pass-2 launch, + 1 for the redo launch's columns)
    int* fb_list;              // [nslots]
    int* fb_count;             // [1]
    const unsigned* x_fail;    // [2] fail flags of pass 2 / its redo launch (an append found the overflow log full)
    const unsigned* x_off;     // [columns] nullable: offset of the column's sorted overflow entries in x_ext (overflow_rebound_kernel)
    uint2* x_ext;              // (row, score) of the candidates past PF_CAP, grouped by column, in emission order
    const uint4* x_log;        // the batch's overflow log and its length (pf_x_scatter)
    const unsigned* x_head;
    unsigned x_cap;
    const unsigned char* redo_col;   // [columns] nullable: columns emitted again by the redo launch
    unsigned long long* ts;               // nullable: the call's stamp set (lmi_kernels.h)
    const unsigned long long* p2_end;     // pass 2's end cell: the selection kernel copies it to ST_P2END
    // the fused tail (lmi_tail.h; all null / 0 on the five-launch route): fallback_kernel merges a query once its last flagged slot is re-scored
    unsigned* host_oflag;                 // nullable: a word of PINNED HOST memory fallback_kernel sets when this batch put candidates into the overflow log
                                          // (or filled it): the host then arms the sort-by-column machinery (overflow_rebound_kernel + pass 2's redo
                                          // launch) for its next calls -- two launches every batch would otherwise pay for (lmi_hip.hip, scan_enqueue)
    int* merge_pending;                   // [nq] flagged slots of the query not yet re-scored (tail_kernel writes it for every query)
    int m_kout;
    float* m_out_d;                       // [nq][kout]
    unsigned* m_out_id;
    unsigned* m_out_key;                  // nullable
};
// the first workgroup of the selection launch: its own start + pass 2's end (known now: the stream ran pass 2 to completion)
__device__ __forceinline__ void select_stamps(const RescoreParams& P) {
    if (P.ts && threadIdx.x == 0 && blockIdx.x == 0) {
        P.ts[ST_TAIL] = wall_clock64();
        if (P.p2_end) P.ts[ST_P2END] = *P.p2_end;
    }
}

// The overflow log -> the columns' ranges of x_ext (every block of the selection kernel's grid calls this first; the ranges were
// handed out by overflow_rebound_kernel; fallback_kernel, a later launch, reads them).  Empty log: one cached scalar load.
__device__ __forceinline__ void pf_x_scatter(const RescoreParams& P, unsigned first, unsigned stride) {
    if (!P.x_off || P.x_fail[0]) return;
    const unsigned n = min(*P.x_head, P.x_cap);
    for (unsigned i = first; i < n; i += stride) {
        const uint4 e = P.x_log[i];
        const unsigned col = e.x, idx = e.y & 0x7fffffffu;
        if (idx + (unsigned)PF_CAP < P.cand_cnt[col]) P.x_ext[P.x_off[col] + idx] = make_uint2(e.z, e.w);
    }
}

// flags slot p for fallback_kernel (called by one lane)
__device__ __forceinline__ void flag_fallback(const RescoreParams& P, int p) {
    P.fallback[p] = 1;
    P.fb_list[atomicAdd(P.fb_count, 1)] = p;
}


// canonical similarity of slab row p with query qv[0..d): acc = fmaf(q[k], x[k], acc), k ascending
__device__ __forceinline__ float exact_score(const float* __restrict__ rows, size_t p, const float* __restrict__ qv, int d, int pitch) {
    const float* x = rows + p * pitch;
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
    select_stamps(P);
    pf_x_scatter(P, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
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
        if (lane == 0) flag_fallback(P, p);
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
        if (lane == 0) flag_fallback(P, p);
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
        const float* x = P.rows + ((size_t)rb0 * 32 + row) * P.dp;
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
// flag; pass 2 (pass2_kernel<false> / pass2_small_kernel<KG, false>) is launched once more and re-runs the buckets that hold such columns with every other
// column's threshold at +inf.  Without this such a slot took the exact fallback (one block brute-forcing the bucket).
// Round 4: when the launch's overflow log did not run full (x_cap != 0, fail flag clear) none of this is needed -- every column's
// candidates are complete and fallback_kernel re-scores them.
__global__ void overflow_rebound_kernel(const int* __restrict__ slot_col, const int* __restrict__ bucket_order, int nslots,
                                        unsigned* __restrict__ cand_cnt, const float* __restrict__ cand_s, float* __restrict__ bound1,
                                        unsigned* __restrict__ redo_count, int* __restrict__ redo_bucket, unsigned char* __restrict__ redo_col,
                                        const unsigned* __restrict__ x_fail, unsigned x_mask /* the log's capacity: 0 = none */,
                                        unsigned* __restrict__ x_off, unsigned* __restrict__ x_total) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nslots) return;
    const int col = slot_col[p];
    if (col < 0 || cand_cnt[col] <= (unsigned)PF_CAP) return;
    if (x_mask != 0u && x_fail[0] == 0u) {   // everything past the buffer is in the overflow log: the column's range of the sorted array
        x_off[col] = atomicAdd(x_total, cand_cnt[col] - (unsigned)PF_CAP);
        return;
    }
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

__global__ void sum_u32_kernel(const unsigned* __restrict__ v, long long n, unsigned long long* __restrict__ out) {
    unsigned long long a = 0;
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}

__global__ void prefilter_stats_kernel(const int* __restrict__ nkeep, const int* __restrict__ fallback, int nslots,
                                       unsigned long long* __restrict__ out) {
    unsigned long long a = 0, b = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < nslots; p += gridDim.x * blockDim.x) { a += nkeep[p]; b += fallback[p]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, a); atomicAdd(out + 1, b); }
}

// N = G * KPB entries (distance, id), entry e = rank (e / KPB), position (e % KPB): lane e's output position is the number of
// entries in front of it by (distance, e) -- every list is sorted by distance, so this is the heads-cursor merge of
// merge_ranks_kernel (ties: the lower rank, then the earlier position).  raw (lmi_knn_ip): larger first.
template <int G>
__device__ __forceinline__ void merge_entries(float dv, unsigned iv, int lane, int raw, int kout, size_t q, float* out_d, unsigned* out_id,
                                              unsigned* out_key) {
    constexpr int N = G * KPB;
    int pos = 0;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const float od = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv), e));
        const bool front = raw ? od > dv : od < dv;
        pos += (front || (od == dv && e < lane)) ? 1 : 0;
    }
    if (lane < N && pos < kout) {
        const size_t o = q * (size_t)kout + pos;
        out_d[o] = dv;
        out_id[o] = iv;
        if (out_key) out_key[o] = (unsigned)(lane / KPB) * 16u + (unsigned)(lane % KPB);
    }
}

// Exact fallback for overflowed slots: one block per slot, brute force over the whole bucket with the
// canonical chain on the VALU (slow, rare, always correct).
__global__ __launch_bounds__(256) void fallback_kernel(RescoreParams P) {
    __shared__ float fs[256 * KPB];
    __shared__ unsigned fr[256 * KPB];
    const int tid = threadIdx.x, lane = tid & 63;
    if (P.ts) ts_first(P.ts + ST_FB);
    if (P.host_oflag && blockIdx.x == 0 && tid == 0 && (*P.x_head > 0u || P.x_fail[0] != 0u)) *P.host_oflag = 1u;
    // the flagged slots are on a list (select_kernel / the re-rank append): the blocks stride over it -- one block per slot at a
    // time, every block of the grid busy when thousands of slots are flagged (duplicate-heavy data); an empty list costs one
    // round of blocks reading the count
    const int nt = *P.fb_count;
    for (int ti = blockIdx.x; ti < nt; ti += gridDim.x) {
    const int p = P.fb_list[ti];
    const int b = P.bucket_order[p];
    const int rb0 = P.rb_start[b], n_b = P.nb_rows[b];
    const float* qv = P.q + (size_t)(p / P.nb) * P.d;
    float v[KPB];
    unsigned id[KPB];
#pragma unroll
    for (int j = 0; j < KPB; ++j) { v[j] = -INFINITY; id[j] = NOROW; }
    // A slot whose candidates are COMPLETE is here for its survivor count only (more rows within 2 eps' of its top-10 than the
    // re-rank holds: hundreds of copies of a vector): its candidates are a superset of the canonical top-10, so re-scoring
    // them is exact and far cheaper than the whole bucket.  Complete = everything past the column's own PF_CAP entries made it
    // into the overflow log (the launch that emitted the column never found it full) and was sorted by column behind pass 2.
    const int ccol = P.slot_col[p];
    const unsigned ccnt = ccol >= 0 ? P.cand_cnt[ccol] : 0xffffffffu;
    const int launch = (P.redo_col && ccol >= 0 && P.redo_col[ccol]) ? 1 : 0;   // which pass-2 launch filled the column last
    const bool in_table = ccol >= 0 && ccnt > (unsigned)PF_CAP && P.x_off != nullptr && launch == 0 && P.x_fail[0] == 0u;
    // The log was NOT sorted by column this batch (the machinery is armed only after a batch that needed it: host_oflag): the column's
    // entries are picked out of the log as it lies -- complete as long as the log did not run full; every flagged slot reads the whole log.
    const bool in_log = ccol >= 0 && ccnt > (unsigned)PF_CAP && P.x_off == nullptr && P.x_log != nullptr && P.x_cap != 0u && P.x_fail[0] == 0u;
    auto consider = [&](unsigned row) {
        const float s = exact_score(P.rows, (size_t)rb0 * 32 + row, qv, P.d, P.dp);
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
    };
    if (ccnt <= (unsigned)PF_CAP || in_table || in_log) {
        const unsigned* cr = P.cand_row + (size_t)ccol * PF_CAP;
        for (unsigned it = tid; it < min(ccnt, (unsigned)PF_CAP); it += 256) consider(cr[it]);
        if (in_table) {
            const uint2* ext = P.x_ext + P.x_off[ccol];
            for (unsigned e = tid; e < ccnt - (unsigned)PF_CAP; e += 256) consider(ext[e].x);
        } else if (in_log) {
            const unsigned n_log = min(*P.x_head, P.x_cap);
            for (unsigned e = tid; e < n_log; e += 256) {
                const uint4 en = P.x_log[e];
                if (en.x == (unsigned)ccol) consider(en.z);
            }
        }
    } else {
    if (tid == 0) atomicAdd(P.fb_count + 5, 1);   // (statistics: slots that scan their whole bucket)
    for (unsigned row = tid; row < (unsigned)n_b; row += 256) {
        const float s = exact_score(P.rows, (size_t)rb0 * 32 + row, qv, P.d, P.dp);
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
    if (P.merge_pending) {
        // The fused tail (lmi_tail.h) left this query unmerged: whoever re-scores its LAST flagged slot merges it from the rank lists in
        // global memory (the others': written by earlier launches or by other workgroups of this one -- agent-scope release before the
        // counter, acquire behind it; the waits after the fences are asm: ROCm 7.2 can drop the fence's own).
        const int q = p / P.nb;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0) old = atomicSub(&P.merge_pending[q], 1);
        old = __shfl(old, 0);
        if (old == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int n = P.nb * KPB;   // nb <= 4
            const float dv = lane < n ? P.rank_d[(size_t)q * n + lane] : 0.0f;
            const unsigned iv = lane < n ? P.rank_id[(size_t)q * n + lane] : 0u;
            switch (P.nb) {
                case 1: merge_entries<1>(dv, iv, lane, P.raw, P.m_kout, (size_t)q, P.m_out_d, P.m_out_id, P.m_out_key); break;
                case 2: merge_entries<2>(dv, iv, lane, P.raw, P.m_kout, (size_t)q, P.m_out_d, P.m_out_id, P.m_out_key); break;
                case 3: merge_entries<3>(dv, iv, lane, P.raw, P.m_kout, (size_t)q, P.m_out_d, P.m_out_id, P.m_out_key); break;
                default: merge_entries<4>(dv, iv, lane, P.raw, P.m_kout, (size_t)q, P.m_out_d, P.m_out_id, P.m_out_key); break;
            }
        }
    }
    }
    __syncthreads();  // fs / fr are reused by the block's next slot
    }
    if (P.ts && P.merge_pending) ts_last(P.ts + ST_END);   // the fused tail: this is the search's last launch
}

}  // namespace lmi
