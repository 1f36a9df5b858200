// lmi_kernels.h -- gfx950 (MI355X / CDNA4) device code for the LearnedMetricIndex query hot path.
//
// Data layout ("fragment-major", the one layout every GEMM-shaped kernel here consumes):
//   a matrix M[R][K] (R rows, K-contiguous) is stored as float4 F[R/32][K/8][64] with
//       F[rb][g][lane].s = M[rb*32 + (lane & 31)][8*g + 2*s + (lane >> 5)],   s = 0..3
//   i.e. one 1-KiB fragment (rb, g) is exactly the A (or B) operand registers of four consecutive
//   v_mfma_f32_32x32x2_f32 steps: lane l feeds row/col (l & 31) and k = k0 + (l >> 5).
//   - a wave reads a fragment with ONE 16-byte load per lane, 1 KiB fully coalesced;
//   - `global_load_lds_dwordx4` (LDS-DMA) can stage it: its LDS image is lane-linear, which is this
//     layout, and the later ds_read_b128 at lane*16 is conflict-free;
//   - the k order inside every accumulator stays 0,1,2,...: the result is the canonical k-ordered
//     fmaf chain (bit-identical to oracle/lmi_oracle.c).
//   R is padded to a multiple of 32 and K to a multiple of 32 with zeros (fmaf(0,0,acc) == acc).
//
// MFMA orientation: S^T = X . Q^T  (A = index vectors / weight rows, B = queries).  In the 32x32
// accumulator a lane then owns ONE query column (lane & 31) and 16 rows, so per-query top-k is a
// per-lane register list with no cross-lane traffic in the hot loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KPB = 10;         // results per (query, bucket): LearnedIndex.py:334
#ifndef LMI_SCAN_RB
#define LMI_SCAN_RB 1
#endif
constexpr int RB = LMI_SCAN_RB;       // row-blocks (of 32 vectors) per wave: 2 -> 1 block/CU, 1 -> 2 blocks/CU
constexpr int TILE_ROWS = 128 * RB;   // vectors per block tile (4 waves x RB x 32)
constexpr int TILE_COLS = 128;  // queries per block tile (4 col-blocks x 32)
constexpr int STAGE_G = 4;      // k-groups (of 8) per LDS stage -> BK = 32
constexpr int SCAN_LDS = 0;           // all LDS is static: 2 x (A 16*RB KiB + B 16 KiB)
constexpr unsigned NOROW = 0xFFFFFFFFu;
constexpr int NGRP = 8;               // work-queue groups = XCDs of the MI355X

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// Device-side phase stamps (lmi_set_timing 2).  A hipEvent between two kernels is a ~5 us bubble on the stream (eight of them:
// 6 % of a C5 step); instead the kernels of a search write the chip's constant 100 MHz clock (s_memrealtime) into the call's set
// of the handle's stamp ring: the FIRST workgroup of a launch its start (workgroups are dispatched in order), the LAST one of a
// short launch its end; a persistent kernel's end is the maximum over its workgroups, folded into a per-call zeroed cell and
// copied to the set by the next kernel on the stream.  Phases are differences of stamps (lmi_hip.hip, read_stamp_set); nothing is
// synchronised, polled or waited for.  All pointers are null when stamps are off.
// ------------------------------------------------------------------------------------------------
enum { ST_MLP0 = 0, ST_MLP1, ST_FRONT, ST_P1, ST_P2, ST_P2END, ST_TAIL, ST_FB, ST_MERGE, ST_END, ST_SCAN0, ST_SCAN1,
       ST_CLK_WALL, ST_CLK_CYC,   // the dominant kernel's block 0: its own life in 100 MHz ticks and in shader-clock cycles (s_memtime): cycles / ticks x 100 = the held clock in MHz
       ST_COUNT = 16 };
// block 0 of a persistent kernel: (s_memrealtime, s_memtime) at its start -> call clk_end with them at its end
__device__ __forceinline__ void clk_begin(unsigned long long* ts_start, unsigned long long& w0, unsigned long long& c0) {
    if (ts_start && blockIdx.x == 0 && threadIdx.x == 0) { w0 = wall_clock64(); c0 = __builtin_readcyclecounter(); }
}
__device__ __forceinline__ void clk_end(unsigned long long* ts_start, int st_index, unsigned long long w0, unsigned long long c0) {
    if (ts_start && blockIdx.x == 0 && threadIdx.x == 0) {   // ts_start = set + st_index: the set's base is ts_start - st_index
        unsigned long long* set = ts_start - st_index;
        set[ST_CLK_WALL] = wall_clock64() - w0;
        set[ST_CLK_CYC] = __builtin_readcyclecounter() - c0;
    }
}
__device__ __forceinline__ void ts_first(unsigned long long* p) {   // the launch's first workgroup
    if (p && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) *p = wall_clock64();
}
__device__ __forceinline__ void ts_last(unsigned long long* p) {    // the launch's last workgroup (1-D grids)
    if (p && threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) *p = wall_clock64();
}
__device__ __forceinline__ void ts_max(unsigned long long* cell) {  // every workgroup of a persistent kernel, at its end
    if (cell && threadIdx.x == 0) atomicMax(cell, (unsigned long long)wall_clock64());
}
__global__ void stamp_kernel(unsigned long long* p) { *p = wall_clock64(); }   // the end of a call whose last kernel carries no stamp

// ------------------------------------------------------------------------------------------------
// pack: row-major -> fragment-major.  One thread per (destination row, k-group).
//   gather form  (rowmap != nullptr or identity): dst row p <- src row rowmap[p] (-1 -> zeros)
//   scatter form (pos != nullptr): src row i of this chunk -> dst row pos[row0 + i] (-1 -> dropped)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8(const float* __restrict__ src, int d, int k0, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k0 + j < d) ? src[k0 + j] : 0.0f;
}

__global__ void pack_gather_kernel(const float* __restrict__ src, int d, const int* __restrict__ rowmap,
                                   int n_src_rows, long long n_dst_rows, int KG, float4* __restrict__ dst,
                                   unsigned long long* ts = nullptr) {
    ts_first(ts);
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_dst_rows * KG) return;
    int g = (int)(idx % KG);
    long long p = idx / KG;
    long long srow = rowmap ? (long long)rowmap[p] : (p < n_src_rows ? p : -1);
    float v[8];
    if (srow >= 0) load8(src + srow * d, d, g * 8, v);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
    float4* f = dst + ((size_t)(p >> 5) * KG + g) * 64 + (p & 31);
    f[0] = make_float4(v[0], v[2], v[4], v[6]);
    f[32] = make_float4(v[1], v[3], v[5], v[7]);
}

__global__ void pack_scatter_kernel(const float* __restrict__ src, int d, const int* __restrict__ pos,
                                    long long row0, const long long* __restrict__ index, long long n_total, long long nrows, int KG,
                                    float4* __restrict__ dst) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * KG) return;
    int g = (int)(idx % KG);
    long long i = idx / KG;
    const long long o = index ? index[i] : row0 + i;
    if (o < 0 || o >= n_total) return;
    long long p = pos[o];
    if (p < 0) return;
    float v[8];
    load8(src + i * d, d, g * 8, v);
    float4* f = dst + ((size_t)(p >> 5) * KG + g) * 64 + (p & 31);
    f[0] = make_float4(v[0], v[2], v[4], v[6]);
    f[32] = make_float4(v[1], v[3], v[5], v[7]);
}

// inverse of pack: fragment-major rows [p0, p0+n) -> row-major [n][d]  (lmi_bucket_read)
__global__ void unpack_kernel(const float4* __restrict__ src, int KG, long long p0, long long n, int d,
                              float* __restrict__ dst) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KGd = (d + 7) >> 3;
    if (idx >= n * KGd) return;
    int g = (int)(idx % KGd);
    long long i = idx / KGd, p = p0 + i;
    const float4* f = src + ((size_t)(p >> 5) * KG + g) * 64 + (p & 31);
    const float4 e = f[0], o = f[32];
    const float v[8] = {e.x, o.x, e.y, o.y, e.z, o.z, e.w, o.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (g * 8 + j < d) dst[i * d + g * 8 + j] = v[j];
}

// L2 metric (lmi_set_metric): rows [n][d] -> [n][da] with column d = -|x|^2 / 2 (|x|^2 by the canonical chain
// fmaf(x[k], x[k], acc), k ascending) and zeros after it; queries -> [q, 1, 0..] and qn2 = |q|^2 by the same chain.
// One thread per (row, column) for the copy, one thread per row for the chain.
__global__ void augment_copy_kernel(const float* __restrict__ src, int d, int da, long long n, float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * da) return;
    const long long i = idx / da;
    const int k = (int)(idx - i * da);
    dst[idx] = k < d ? src[i * d + k] : 0.0f;
}
__global__ void augment_norm_kernel(const float* __restrict__ src, int d, int da, long long n, float* __restrict__ dst,
                                    float* __restrict__ qn2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* x = src + i * d;
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = __builtin_fmaf(x[k], x[k], acc);
    if (qn2) { qn2[i] = acc; dst[i * da + d] = 1.0f; }   // a query
    else dst[i * da + d] = -0.5f * acc;                   // an indexed vector
}

// plain copy by a kernel (16 bytes per lane): results -> pinned host memory without a hipMemcpyAsync
// (lmi_copy_out); dst/src 16-byte aligned, the tail is copied bytewise
__global__ void copy_bytes_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long bytes) {
    const long long n16 = bytes >> 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    const long long t = (n16 << 4) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < bytes) dst[t] = src[t];
}

// up to 4 ranges in one launch: blockIdx.y picks the range (lmi_copy_out_many)
struct CopyRanges { const unsigned char* src[4]; unsigned char* dst[4]; long long bytes[4]; };
__global__ void copy_ranges_kernel(CopyRanges C) {
    const unsigned char* __restrict__ src = C.src[blockIdx.y];
    unsigned char* __restrict__ dst = C.dst[blockIdx.y];
    const long long bytes = C.bytes[blockIdx.y], n16 = bytes >> 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    const long long t = (n16 << 4) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < bytes) dst[t] = src[t];
}

// ------------------------------------------------------------------------------------------------
// MLP layer:  Y^T = W . X^T + b  (ReLU unless last).   model.py:45-49, 97-99, 232
//   Wf   [n_rb][KG][64]  weights, rows = output features (padded with zero rows)
//   Xf   [ncb][KG][64]   activations, "rows" = queries
//   grid (ceil(ncb/CBW), ceil(n_rb/4)), block 256: wave w -> feature block blockIdx.y*4+w, CBW col-blocks
//   (CBW = 4, or 1 when the layer has few feature blocks and would otherwise fill a third of the CUs).
//   Accumulators start at the bias (torch addmm starts from the bias): chain = b + sum_k.
//   !LAST: output written as the next layer's fragment-major activations (KGn = n_rb_pad*4 groups).
//    LAST: logits row-major [nq][L].
// ------------------------------------------------------------------------------------------------
template <bool LAST, int CBW = 4>
__global__ __launch_bounds__(256) void mlp_layer_kernel(const float4* __restrict__ Wf,
                                                        const float* __restrict__ bias,
                                                        const float4* __restrict__ Xf, int KG, int n_rb,
                                                        int ncb, float* __restrict__ out, int KGn,
                                                        int nq, int L) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    const int rb = blockIdx.y * 4 + w;
    const int cb0 = blockIdx.x * CBW;
    if (rb >= n_rb) return;
    f32x16 acc[CBW];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float bv = bias[rb * 32 + acc_row(r, h)];
#pragma unroll
        for (int n = 0; n < CBW; ++n) acc[n][r] = bv;
    }
    const float4* ap = Wf + (size_t)rb * KG * 64 + lane;
    const float4* bp[CBW];
#pragma unroll
    for (int n = 0; n < CBW; ++n) {
        int cb = cb0 + n < ncb ? cb0 + n : ncb - 1;  // clamp: duplicates are computed and discarded
        bp[n] = Xf + (size_t)cb * KG * 64 + lane;
    }
    // operands straight from L2 (the weights are shared by every block, the activations by the block row):
    // k-group g+1 is requested before the MFMAs of group g
    float4 a = ap[0];
    float4 b[CBW];
#pragma unroll
    for (int n = 0; n < CBW; ++n) b[n] = bp[n][0];
    for (int g = 0; g < KG; ++g) {
        const int gn = g + 1 < KG ? g + 1 : g;
        const float4 a_next = ap[(size_t)gn * 64];
        float4 b_next[CBW];
#pragma unroll
        for (int n = 0; n < CBW; ++n) b_next[n] = bp[n][(size_t)gn * 64];
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int n = 0; n < CBW; ++n) {
                const float bv = s == 0 ? b[n].x : s == 1 ? b[n].y : s == 2 ? b[n].z : b[n].w;
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[n], 0, 0, 0);
            }
        }
        a = a_next;
#pragma unroll
        for (int n = 0; n < CBW; ++n) b[n] = b_next[n];
    }
#pragma unroll
    for (int n = 0; n < CBW; ++n) {
        const int cb = cb0 + n;
        if (cb >= ncb) continue;
        const int q = cb * 32 + c;
        if (LAST) {
            if (q < nq) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int j = rb * 32 + acc_row(r, h);
                    if (j < L) out[(size_t)q * L + j] = acc[n][r];
                }
            }
        } else {
            // feature j = rb*32 + 8*qd + 4*h + i  ->  group rb*4+qd, step s = 2h + (i>>1), half i&1
            float* o = out + ((size_t)cb * KGn) * 256;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float v0 = fmaxf(acc[n][4 * qd + 0], 0.0f), v1 = fmaxf(acc[n][4 * qd + 1], 0.0f);
                float v2 = fmaxf(acc[n][4 * qd + 2], 0.0f), v3 = fmaxf(acc[n][4 * qd + 3], 0.0f);
                float* og = o + (size_t)(rb * 4 + qd) * 256;
                *reinterpret_cast<float2*>(og + (0 * 32 + c) * 4 + 2 * h) = make_float2(v0, v2);
                *reinterpret_cast<float2*>(og + (1 * 32 + c) * 4 + 2 * h) = make_float2(v1, v3);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Class ranking: first nb entries of `prob.topk(L)` (model.py:239), on the logits,
// ties -> lower class index.  One wave per query, nb selection passes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rank_classes_kernel(const float* __restrict__ logits, int nq, int L,
                                                         int nb, int* __restrict__ order) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    const float* l = logits + (size_t)q * L;
    float pv = INFINITY;
    int pi = -1;
    for (int t = 0; t < nb; ++t) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < L; j += 64) {
            float v = l[j];
            bool after = (v < pv) || (v == pv && j > pi);
            if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o);
            int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) order[(size_t)q * nb + t] = (bi == 0x7fffffff) ? -1 : bi;
        pv = bv;
        pi = bi;
    }
}

// ------------------------------------------------------------------------------------------------
// Canonical softmax (model.py:238) + gather in ranked order -> predict_proba's `probs`.
// lmi_expf is the SAME sequence of binary32 operations as oracle/lmi_oracle.c:lmi_oracle_expf
// (Cody-Waite reduction and a degree-6 Horner polynomial, all fmaf), so probabilities are
// bit-identical on both sides; the row sum runs in class-index order.  One thread per query.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lmi_expf(float x) {
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) return INFINITY;
    const float log2e = 1.44269502162933349609375f;
    const float ln2hi = 0.693145751953125f;
    const float ln2lo = 1.42860677279532e-06f;
    const float nf = __builtin_rintf(x * log2e);
    float r = __builtin_fmaf(-nf, ln2hi, x);
    r = __builtin_fmaf(-nf, ln2lo, r);
    float p = 1.0f / 720.0f;
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int ni = (int)nf;
    return p * __uint_as_float((unsigned)(ni + 127) << 23);
}

__global__ void softmax_ranked_kernel(const float* __restrict__ logits, const int* __restrict__ order, int nq,
                                      int L, float* __restrict__ probs) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float* l = logits + (size_t)q * L;
    float m = l[0];
    for (int j = 1; j < L; ++j) m = l[j] > m ? l[j] : m;
    float s = 0.0f;
    for (int j = 0; j < L; ++j) s += lmi_expf(l[j] - m);
    for (int t = 0; t < L; ++t) {
        const int cls = order[(size_t)q * L + t];  // -1: rank_classes_kernel ran out of comparable logits (NaN)
        probs[(size_t)q * L + t] = cls >= 0 ? lmi_expf(l[cls] - m) / s : __builtin_nanf("");
    }
}

// ------------------------------------------------------------------------------------------------
// Routing: the GPU twin of the `for bucket: filter_path_idxs(bucket_path, path)` loop
// (LearnedIndex.py:350-353, utils.py:61-65): queries grouped bucket-major (CSR).
// ------------------------------------------------------------------------------------------------
struct RouteArrays {
    const int* nb_rows;   // [L] objects per bucket (0: empty / not owned)
    const int* nch;       // [L] scan chunks per bucket
    int* m;               // [L] queries routed to the bucket (zeroed before route_count)
    int* m0;              // [L] of them: PRIMARY slots -- the columns [0, m0) of the bucket, the only ones the prefilter's pass 1
                          // samples when one bound per query is enough (primary_nb > 0; otherwise m0 == m).  [L..2L): the others' counter
    int* cb_start;        // [L+1] col-block prefix
    int* item_base;       // [L+1] work-item prefix
    long long* part_base; // [L+1] partial-list prefix
    long long* stats;     // [0] = sum m_b * n_b (pairs), [1] = items
    // XCD-affine work queues (route_group_kernel): NGRP groups of buckets, heaviest first
    int* grp_bucket;      // [NGRP][L]   bucket ids of the group, in processing order
    int* grp_base;        // [NGRP][L+1] prefix of the buckets' item counts inside the group
    int* grp_n;           // [NGRP]      buckets in the group
    int* grp_total;       // [NGRP]      items in the group
    int* grp_base1;       // [NGRP][L+1] the same prefix for the pass-1 items (query tiles x sampled tiles) of lmi_pass2.h
    int* grp_total1;      // [NGRP]
    int tile_cb;          // col-blocks per query tile: 4 (exact scan), 12 (prefilter: lmi_pass2.h), 8 (the round-2 prefilter kernel)
    int primary_nb;       // > 0 (= n_buckets): a slot is primary iff no lower rank of its query holds a bucket of >= 64 rows here
    int sample_items;     // 1: also build the pass-1 queues of lmi_pass2.h (grp_base1 / grp_total1: query tiles x SAMPLED tiles)
    int sample_max;       // pass 1's largest sampling stride (PF_SAMPLE; PF_SAMPLE_LOWD at K <= 64)
    unsigned long long* dbg;  // nullable developer aid: route_group_body stamps the clock after its key pass [2] and its sort [3]
    // Graded pass-2 items (round 5; prefilter only, <= 1 024 buckets): a bucket's rows are handed out in chunks of chunk_rb_b[b] row-blocks,
    // chosen per CALL from where the bucket sits in the work-sorted order: the buckets a queue serves last get short chunks, the others
    // long ones -- few item starts over most of the launch, a fine-grained end (C2: workgroups idle 4 % of the launch at 2 048 rows per
    // item everywhere, 1 % at 512 with 9 % more time spent on item starts)
    int* chunk_rb_b;          // nullable [L] out: row-blocks per chunk of bucket b (null: nch / the index's static chunk)
    int chunk_rb;             // the index's static chunk (= nch): thousands of buckets keep it
    int chunk_lvl[3];         // row-blocks per chunk: buckets ahead of the last chunk_frac[0] of the work, the next ones, the last chunk_frac[1]
    float chunk_frac[2];
};
// chunks of a bucket of nrb row-blocks at crb row-blocks per chunk
__device__ __forceinline__ int chunks_of(int n_rows, int crb) { return (((n_rows + 31) >> 5) + crb - 1) / crb; }

// ---- prefilter pass 1: which tiles of a bucket are sampled (shared by the routing kernels and lmi_prefilter.h / lmi_pass2.h) ----
#ifndef LMI_PF_SAMPLE_ROWS
#define LMI_PF_SAMPLE_ROWS 512   // a bucket is sampled at stride s only if it has >= this many rows per unit of s.  Round 4: 2 000 -> 512 -- with 2 000 a
                                 // bucket under 4 000 rows was "sampled" whole (pass 1 = a second pass 2: at 2 000 leaves over 4M rows pass 1 1.32 ms
                                 // against pass 2's 1.22); 512: +7..22 % queries/s on 500-2 000-leaf shapes, C2 / hard / C1 / C5 unchanged (their
                                 // buckets sit at stride 16 either way; 256 loses at d <= 128: profiles/r04_pass2_experiments.txt section 12)
#endif
#ifndef LMI_PF_SAMPLE
#define LMI_PF_SAMPLE 16
#endif
constexpr int PF_SAMPLE = LMI_PF_SAMPLE;  // pass 1 looks at every PF_SAMPLE-th tile of a large bucket ...
// ... every PF_SAMPLE_LOWD-th at K <= 64 (RouteArrays / PrefilterParams::sample_max carry the call's value: there a candidate costs as much
// as the rows that would rule it out -- C5: stride 16 / 8 / 4: pass 1 0.100 / 0.120 / 0.167, pass 2 0.315 / 0.267 / 0.250 ms, 15.0 / 15.75 /
// 15.1 M q/s; at d = 768 stride 8 loses 3 %, 32 loses 6 %) ...
#ifndef LMI_PF_SAMPLE_LOWD
#define LMI_PF_SAMPLE_LOWD 8
#endif
constexpr int PF_SAMPLE_LOWD = LMI_PF_SAMPLE_LOWD, PF_SAMPLE_LOWD_KG = 4;
// ... and at every 8th, 4th, 2nd or every tile of buckets below LMI_PF_SAMPLE_ROWS x the stride
__device__ __forceinline__ int sample_stride(int n_b, int smax) {
    int s = smax;
    while (s > 1 && n_b < LMI_PF_SAMPLE_ROWS * s) s >>= 1;
    return s;
}
// Pass-1 items of a bucket (lmi_pass2.h): its sampled tiles j = 0, 1, ..; every P1_ALL_EVERY-th of them is run over ALL the
// bucket's columns (query tiles of m), the others over the primary columns only (query tiles of m0): a primary column is sampled
// at the full rate, the others at 1 / P1_ALL_EVERY of it -- enough of an own bound that their candidate buffers cannot overflow
// when the query's bound (query_bound_kernel) turns out loose for their bucket (10 x the stride x P1_ALL_EVERY rows pass).
constexpr int P1_ALL_EVERY = 2;
// An item takes up to P1_TPI sampled tiles of ONE class (the even ones: all columns; the odd ones: primary columns), two sampled tiles apart:
// the item's fixed costs (queue ticket, cold ring / query tile staged in LDS, barriers) are paid once for them (round 4; 1 = round 3's items).
#ifndef LMI_P1_TPI
#define LMI_P1_TPI 2
#endif
constexpr int P1_TPI = LMI_P1_TPI;
static_assert(P1_ALL_EVERY == 2 && P1_TPI >= 1, "the two classes are the even and the odd sampled tiles");
__device__ __forceinline__ int pass1_items(int nst, int nqt_all, int nqt_primary) {
    const int n_all = (nst + 1) / 2, n_pri = nst / 2;
    return ((n_all + P1_TPI - 1) / P1_TPI) * nqt_all + ((n_pri + P1_TPI - 1) / P1_TPI) * nqt_primary;
}
// item `local` of a bucket with nst sampled tiles -> its first sampled tile j0, the number of tiles nt (j0, j0 + 2, ..) and the query tile
// qt; returns whether the item runs over all columns
__device__ __forceinline__ bool pass1_decode(int local, int nst, int nqt_all, int nqt_primary, int* j0, int* nt, int* qt) {
    const int n_all = (nst + 1) / 2, n_pri = nst / 2;
    const int g_all = (n_all + P1_TPI - 1) / P1_TPI;
    if (local < g_all * nqt_all) {
        const int g = local / nqt_all;
        *qt = local - g * nqt_all;
        *j0 = g * P1_TPI * 2;
        *nt = min(P1_TPI, n_all - g * P1_TPI);
        return true;
    }
    local -= g_all * nqt_all;
    const int g = local / nqt_primary;
    *qt = local - g * nqt_primary;
    *j0 = g * P1_TPI * 2 + 1;
    *nt = min(P1_TPI, n_pri - g * P1_TPI);
    return false;
}
__device__ __forceinline__ int sample_tiles256(int n_b, int smax) {  // sampled 256-row tiles of a bucket
    const int nt = (((n_b + 31) >> 5) + 7) / 8;
    const int s = sample_stride(n_b, smax);
    return (nt + s - 1) / s;
}

// query tiles of a bucket with m routed queries: its col-blocks over tiles of tile_cb (tile_cb 4: ceil(m / 128))
__device__ __forceinline__ int query_tiles(int m, int tile_cb) { return (((m + 31) >> 5) + tile_cb - 1) / tile_cb; }

// Positions are handed out per block through an LDS histogram (one global atomic per bucket and
// block instead of one per slot: 40 000 returning atomics on 120 hot words took 66 us).
// Primary slots (RouteArrays::primary_nb) take the bucket's first columns, the others follow: slot_local = position among its
// kind, bit 30 set for the others (route_fill_kernel adds m0).
constexpr int ROUTE_LDS_BUCKETS = 4096;
constexpr int ROUTE_OTHER = 1 << 30;
__global__ __launch_bounds__(256) void route_count_kernel(const int* __restrict__ bucket_order, int nslots, int L,
                                                          RouteArrays R, int* __restrict__ slot_local) {
    __shared__ int cnt_s[2 * ROUTE_LDS_BUCKETS];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int b = -1;
    bool other = false;
    if (p < nslots) {
        b = bucket_order[p];
        if (!(b >= 0 && b < L && R.nb_rows[b] > 0)) b = -1;
        if (b >= 0 && R.primary_nb > 0) {
            const int r = p % R.primary_nb;
            for (int j = p - r; j < p; ++j) {   // a lower rank of this query with a bucket large enough to give the query its bound
                const int bj = bucket_order[j];
                if (bj >= 0 && bj < L && R.nb_rows[bj] >= 64) other = true;
            }
        }
    }
    int* m1 = R.m0 + L;
    if (L > ROUTE_LDS_BUCKETS) {  // huge fan-out: plain global atomics
        if (p < nslots) {
            if (b >= 0) atomicAdd(&R.m[b], 1);
            slot_local[p] = b < 0 ? -1 : other ? (atomicAdd(&m1[b], 1) | ROUTE_OTHER) : atomicAdd(&R.m0[b], 1);
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) cnt_s[i] = 0;
    __syncthreads();
    const int loc = b >= 0 ? atomicAdd(&cnt_s[other ? L + b : b], 1) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int c0 = cnt_s[i], c1 = cnt_s[L + i];
        if (c0 + c1 > 0) atomicAdd(&R.m[i], c0 + c1);
        cnt_s[i] = c0 > 0 ? atomicAdd(&R.m0[i], c0) : 0;       // block's base position among the bucket's primary slots
        cnt_s[L + i] = c1 > 0 ? atomicAdd(&m1[i], c1) : 0;      // ... among the others
    }
    __syncthreads();
    if (p < nslots) slot_local[p] = b < 0 ? -1 : other ? ((cnt_s[L + b] + loc) | ROUTE_OTHER) : cnt_s[b] + loc;
}

__global__ __launch_bounds__(256) void route_scan_kernel(int L, RouteArrays R) {
    // one block: thread t owns buckets [t*per, (t+1)*per); exclusive scan of 4 running sums
    __shared__ long long sh[4][256];
    const int t = threadIdx.x;
    const int per = (L + 255) / 256;
    const int b0 = min(t * per, L), b1 = min(b0 + per, L);
    long long cb = 0, items = 0, part = 0, pairs = 0;
    for (int b = b0; b < b1; ++b) {
        const int m = R.m[b];
        cb += (m + 31) >> 5;
        items += (long long)query_tiles(m, R.tile_cb) * R.nch[b];
        part += (long long)m * R.nch[b];
        pairs += (long long)m * R.nb_rows[b];
    }
    sh[0][t] = cb; sh[1][t] = items; sh[2][t] = part; sh[3][t] = pairs;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        long long v[4] = {0, 0, 0, 0};
        if (t >= o)
            for (int i = 0; i < 4; ++i) v[i] = sh[i][t - o];
        __syncthreads();
        for (int i = 0; i < 4; ++i) sh[i][t] += v[i];
        __syncthreads();
    }
    long long ecb = sh[0][t] - cb, eit = sh[1][t] - items, epart = sh[2][t] - part;
    for (int b = b0; b < b1; ++b) {
        const int m = R.m[b];
        R.cb_start[b] = (int)ecb;
        R.item_base[b] = (int)eit;
        R.part_base[b] = epart;
        ecb += (m + 31) >> 5;
        eit += (long long)query_tiles(m, R.tile_cb) * R.nch[b];
        epart += (long long)m * R.nch[b];
    }
    if (t == 255) {
        R.cb_start[L] = (int)sh[0][255];
        R.item_base[L] = (int)sh[1][255];
        R.part_base[L] = sh[2][255];
        R.stats[0] = sh[3][255];
        R.stats[1] = sh[1][255];
    }
}

// Buckets -> NGRP work queues.  All items of a bucket go to ONE queue and each queue is served
// first by the blocks of one XCD (scan_kernel), so that the blocks sharing an XCD's 4 MiB L2 work
// on the same few buckets: their query tiles (384 KiB each) then stay L2-resident and a chunk's
// vectors are fetched once for all of its query tiles.  Heaviest buckets first (LPT) balances the
// queues; blocks of a drained queue steal from the others, so placement only affects speed.
// dynamic LDS of route_group_kernel: the sort keys (8 bytes per bucket, padded to a power of two) + two item counts per bucket
__host__ __device__ inline int route_group_pow2(int L) { int p = 64; while (p < L) p <<= 1; return p; }
__host__ __device__ inline size_t route_group_lds(int L) { return (size_t)route_group_pow2(L) * 8 + (size_t)L * 8; }
constexpr int ROUTE_MAX_BUCKETS = 8000;   // up to here the sort runs in LDS (8 x 8 192 + 8 x 8 000 bytes); beyond (GLOBAL: fan-outs like
                                          // [100, 100]) in a global scratch buffer of route_group_lds(L) + 4 L bytes: slower, rare
constexpr int ROUTE_ID_BITS = 20;         // bucket ids in the sort key: lmi_buckets_begin refuses 2^20 buckets or more
// (the body: also the queue-building block of front_kernel, lmi_front.h, which passes the per-bucket counts it holds in LDS as R.m / R.m0;
// `base`: route_group_lds(L) bytes of LDS, or the global scratch; `active_sp`: one LDS word; 1 024 threads)
template <bool GLOBAL, int NT = 1024>   // NT: threads of the block (1 024: route_group_kernel; 512: pack_kernel's queue block, lmi_front.h)
__device__ __forceinline__ void route_group_body(int L, const RouteArrays& R, char* base, int* active_sp) {
    // Everything in LDS: the ranking is a bitonic sort of one 64-bit key per bucket (work descending, id ascending), O(L log^2 L / 1024)
    // per thread.  (Until round 4 every thread counted the buckets ahead of its own: O(L^2 / 1024) -- a few us at L = 120, 196 us
    // at L = 2 000, on the critical path in front of pass 1.)
    int& active_s = *active_sp;
    const int P = route_group_pow2(L);
    unsigned long long* key_s = reinterpret_cast<unsigned long long*>(base);
    int* items_s = reinterpret_cast<int*>(key_s + P);
    int* items1_s = items_s + L;
    // the sorted bucket ids: written over the keys once they are read (LDS), or behind the item counts (GLOBAL)
    int* order_s = GLOBAL ? items1_s + L : reinterpret_cast<int*>(base);
    const int t = threadIdx.x;
    if (t == 0) active_s = 0;
    __syncthreads();
    {
        int mine = 0;
        for (int b = t; b < P; b += NT) {
            unsigned long long key = ~0ull;   // padding sorts last
            if (b < L) {
                const int m = R.m[b];
                long long work = (long long)m * R.nb_rows[b];   // queries x rows: a fine proxy of the bucket's scan time
                mine += work > 0;
                items_s[b] = query_tiles(m, R.tile_cb) * R.nch[b];
                const int mp = R.sample_items ? R.m0[b] : 0;   // pass 1 runs over the primary columns only
                items1_s[b] = (R.sample_items && m > 0) ? pass1_items(sample_tiles256(R.nb_rows[b], R.sample_max), query_tiles(m, R.tile_cb), query_tiles(mp, R.tile_cb)) : 0;
                if (work > (1ll << 43) - 1) work = (1ll << 43) - 1;   // (the heaviest of the heavy then rank by id: placement only affects speed)
                key = ((unsigned long long)((1ll << 43) - 1 - work) << ROUTE_ID_BITS) | (unsigned long long)b;
            }
            key_s[b] = key;
        }
        if (mine) atomicAdd(&active_s, mine);
    }
    __syncthreads();
    if (R.dbg && t == 0) R.dbg[2] = wall_clock64();
    bool ranked = false;
    int my_rank[2] = {0, 0}, my_id[2] = {0, 0};
    if (!GLOBAL && P <= 2 * NT && P <= 1024) {
        // Few buckets (C2: 120, C5: 256): RANK BY COUNTING -- a thread per key counts the smaller keys (the keys are distinct: work | id):
        // P broadcast reads from LDS, no barrier per step (the bitonic network below: log^2 P steps with a block barrier each, ~9 us of
        // a 12-us block at 256 buckets; a single-wave register network with 8 lane shuffles per step measured 8-11 us; this: ~1 us)
        ranked = true;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = t + u * NT;
            if (i < P) {
                const unsigned long long mine = key_s[i];
                int r = 0;
                long long w_all = 0, w_tail = 0;   // work of all buckets / of those ranked at or behind this one (graded chunks)
#pragma unroll 8
                for (int o = 0; o < P; ++o) {
                    const unsigned long long ko = key_s[o];
                    r += ko < mine ? 1 : 0;
                    if (R.chunk_rb_b) {
                        const long long wo = ko == ~0ull ? 0ll : (1ll << 43) - 1 - (long long)(ko >> ROUTE_ID_BITS);
                        w_all += wo;
                        w_tail += ko >= mine ? wo : 0ll;
                    }
                }
                my_rank[u] = r;
                my_id[u] = (int)(mine & ((1ull << ROUTE_ID_BITS) - 1));
                if (R.chunk_rb_b && mine != ~0ull) {
                    const float f = w_all > 0 ? (float)((double)w_tail / (double)w_all) : 0.0f;
                    const int crb = f > R.chunk_frac[0] ? R.chunk_lvl[0] : f > R.chunk_frac[1] ? R.chunk_lvl[1] : R.chunk_lvl[2];
                    const int b = my_id[u];
                    R.chunk_rb_b[b] = crb;
                    items_s[b] = query_tiles(R.m[b], R.tile_cb) * chunks_of(R.nb_rows[b], crb);   // (items_s lies behind the keys: no overlap)
                }
            }
        }
        __syncthreads();   // every key is read: the ids may go over them
    } else {
    if (R.chunk_rb_b)   // thousands of buckets: items are short and many anyway -- the index's static chunk
        for (int b = t; b < L; b += NT) R.chunk_rb_b[b] = R.chunk_rb;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key_s[i], c = key_s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { key_s[i] = c; key_s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    }
    if (R.dbg && t == 0) R.dbg[3] = wall_clock64();
    if (ranked) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = t + u * NT;
            if (i < P && my_rank[u] < L) order_s[my_rank[u]] = my_id[u];   // (over the keys: every one was read before the barrier above)
        }
    } else
    if constexpr (GLOBAL) {
        for (int i = t; i < L; i += NT) order_s[i] = (int)(key_s[i] & ((1ull << ROUTE_ID_BITS) - 1));
    } else {
        // rank -> bucket id, in place: every thread reads its keys before anyone writes an id over a key
        constexpr int PER = (8192 + NT - 1) / NT;
        int ids[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * NT;
            ids[u] = i < P ? (int)(key_s[i] & ((1ull << ROUTE_ID_BITS) - 1)) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * NT;
            if (i < L) order_s[i] = ids[u];
        }
    }
    if (t < NGRP) { R.grp_base[t * (L + 1)] = 0; if (R.sample_items) R.grp_base1[t * (L + 1)] = 0; }
    __syncthreads();
    // Buckets -> queues in "snake" order over the work-sorted list (ranks 0..7 -> queues 0..7, ranks 8..15 -> queues 7..0, ..).  (Exact
    // LPT is serial: 54-73 us at L = 120 whichever way it was written; the snake's queue loads differ by a few per cent and draining
    // queues steal anyway.)  Wave g scans queue g's item counts (its members are rank 8 r + (r odd ? 7 - g : g), r = 0, 1, ..), wave
    // NGRP + g its pass-1 item counts: 64 members per step, a shuffle scan + a carry.
    const int A = active_s;  // buckets with work: ranks 0 .. A-1 (the others have zero work and sort last)
    const int ln = t & 63;
    for (int wv = t >> 6; wv < 2 * NGRP; wv += NT / 64) {   // (a wave per queue and kind of item; a block of fewer waves: several each)
        const int g = wv % NGRP;
        const bool second = wv >= NGRP;   // the pass-1 items
        const int rows = A / NGRP, rem = A % NGRP;
        const int ph_last = (rows & 1) ? NGRP - 1 - g : g;          // this queue's position in the partial last row
        const int n_g = rows + (ph_last < rem ? 1 : 0);
        if (!second || R.sample_items) {
            const int* cnt = second ? items1_s : items_s;
            int carry = 0;
            for (int r0 = 0; r0 < n_g; r0 += 64) {
                const int r = r0 + ln;
                int b = -1, x = 0;
                if (r < n_g) {
                    b = order_s[r * NGRP + ((r & 1) ? NGRP - 1 - g : g)];
                    x = cnt[b];
                }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(x, o, 64);
                    if (ln >= o) x += up;
                }
                if (r < n_g) {
                    if (!second) {
                        R.grp_bucket[g * L + r] = b;
                        R.grp_base[g * (L + 1) + r + 1] = carry + x;
                    } else {
                        R.grp_base1[g * (L + 1) + r + 1] = carry + x;
                    }
                }
                carry += __shfl(x, 63, 64);
            }
            if (ln == 0) {
                if (!second) { R.grp_n[g] = n_g; R.grp_total[g] = carry; }
                else R.grp_total1[g] = carry;
            }
        }
    }
}
template <bool GLOBAL>
__global__ __launch_bounds__(1024) void route_group_kernel(int L, RouteArrays R, char* scratch) {
    extern __shared__ __attribute__((aligned(16))) char grp_smem[];
    __shared__ int active_s;
    route_group_body<GLOBAL>(L, R, GLOBAL ? scratch : grp_smem, &active_s);
}

__global__ void route_fill_kernel(const int* __restrict__ bucket_order, const int* __restrict__ slot_local,
                                  int nslots, int nb, const int* __restrict__ cb_start, const int* __restrict__ m0,
                                  int* __restrict__ colmap, int* __restrict__ slot_col) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nslots) return;
    int loc = slot_local[p];
    int col = -1;
    if (loc >= 0) {
        const int b = bucket_order[p];
        col = cb_start[b] * 32 + ((loc & ROUTE_OTHER) ? m0[b] + (loc & (ROUTE_OTHER - 1)) : loc);
        colmap[col] = p / nb;
    }
    slot_col[p] = col;
}

// ------------------------------------------------------------------------------------------------
// Bucket scan (the hot kernel): replaces `faiss.knn(queries_for_this_bucket, data_in_this_bucket)`
// for every (rank, bucket) at once (LearnedIndex.py:107-117, 350-365).
//
// Persistent grid; work item = (bucket b, query tile qt of 128 columns, chunk ch of chunk_rb
// row-blocks), query tile fastest so that concurrently running items share the chunk's vectors
// in L2/MALL.  Per item: for each 128-row vector tile, S^T = X.Q^T by v_mfma_f32_32x32x2_f32 over
// K in 32-wide stages, both operands staged by LDS-DMA (global_load_lds_dwordx4) into a 2-deep
// ring; epilogue = threshold test against the lane's running 10th-best + rare sorted insert.
// Block = 4 waves stacked along the vectors (wave w owns row-block 4*vt+w, all 4 col-blocks), so
// dead col-blocks of a ragged query tile are skipped uniformly by the whole block.
// ------------------------------------------------------------------------------------------------
struct ScanParams {
    const float4* slab;
    const float4* qfrag;
    int KG;  // k-groups per row-block (multiple of STAGE_G)
    int L;
    int chunk_rb;
    const int* rb_start;
    const int* nb_rows;
    const int* nch;
    const int* m;
    const int* cb_start;
    const long long* part_base;
    const int* grp_bucket;
    const int* grp_base;
    const int* grp_n;
    const int* grp_total;
    unsigned* head;  // [NGRP] queue heads, zeroed before the launch
    float* col_thr;  // [columns] best known lower bound of each (query, rank) slot's 10th-best
                     // similarity, shared by the chunks of a bucket; -inf before the launch
    float* part_score;
    unsigned* part_row;
    unsigned long long* ts_start;   // nullable: device stamps (ts_first / ts_max above)
    unsigned long long* ts_end_cell;
};

__device__ __forceinline__ void glds16(const float4* gsrc, float4* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16; the source address is per lane.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with the instruction's immediate offset OFF (bytes), which the hardware adds to BOTH addresses:
// consecutive 1-KiB fragments of one stream share the address registers and the M0 base
template <int OFF, int AUX = 0>  // AUX 2 = nt (non-temporal): a stream that is read once
__device__ __forceinline__ void glds16o(const float4* gsrc, float4* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, AUX);
}

// sorted (descending) insert of (s,row) into a 10-entry register list; caller checked s > v[9].
__device__ __forceinline__ void list_insert(float (&v)[KPB], unsigned (&id)[KPB], float s, unsigned row) {
#pragma unroll
    for (int t = KPB - 1; t > 0; --t) {
        const bool shift = s > v[t - 1];  // strict: an equal, earlier row stays in front
        const bool here = s > v[t];
        id[t] = shift ? id[t - 1] : (here ? row : id[t]);
        v[t] = shift ? v[t - 1] : (here ? s : v[t]);
    }
    const bool top = s > v[0];
    id[0] = top ? row : id[0];
    v[0] = top ? s : v[0];
}

// similarity -> returned distance.  IP (the reference's only metric): 1 - s in binary32 (LearnedIndex.py:368).
// L2 (lmi_set_metric): the scanned vectors carry the extra column -|x|^2/2 and the queries a 1 there, so the
// "similarity" is key = <q,x> - |x|^2/2 (ranking by it == ranking by |q - x|^2) and dist = |q|^2 - 2 key, one fmaf.
// Padding of short buckets (faiss: worst value of the metric): 1 - (-FLT_MAX) resp. FLT_MAX.
__device__ __forceinline__ float sim_to_dist(float s, const float* qn2, int q) {
    return qn2 ? __builtin_fmaf(-2.0f, s, qn2[q]) : 1.0f - s;
}
__device__ __forceinline__ float pad_dist(const float* qn2) { return qn2 ? 3.402823466e+38f : 1.0f - (-3.402823466e+38f); }

__device__ __forceinline__ bool better(float s, unsigned r, float s2, unsigned r2) {
    return s > s2 || (s == s2 && r < r2);
}

// monotone float max on a global word (works for any sign; -inf start value)
__device__ __forceinline__ void atomic_max_float(float* p, float v) {
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(p), __float_as_uint(v));
}

// The four stage buffers are FOUR DISTINCT __shared__ arrays (declared in scan_kernel) and the
// buffer parity is a template constant: the compiler can then prove that the LDS-DMA writes of
// stage u+1 (parity 1-PAR) do not alias the ds_reads of stage u (parity PAR) and does not drain
// vmcnt(0) in front of them -- with one array + runtime offsets it did, serialising load and MFMA.
struct ScanLds {
    float4 *A0, *A1;  // each [4 waves][RB][STAGE_G][64] float4
    float4 *B0, *B1;  // each [4 col-blocks][STAGE_G][64] float4
};

template <int NCB>
struct ScanItem {
    const ScanParams& P;
    const ScanLds& S;
    int lane, w, h, c;
    int KG, NS, n_b, nrb_b, rb_in_b0, total;
    const float4* aslab;   // bucket's first row-block, + lane
    const float4* bbase;
    float lv[NCB][KPB];
    unsigned li[NCB][KPB];
    const float* cthr;    // &col_thr[first column of this item] + c
    f32x16 acc[RB][NCB];

    template <int PAR>
    __device__ __forceinline__ void issue(int u) {  // DMA stage u into the parity-PAR buffers
#ifdef LMI_ABL_NOLOAD  // timing-only ablation build: no operand traffic (results are garbage)
        return;
#endif
        float4* sA = PAR ? S.A1 : S.A0;
        float4* sB = PAR ? S.B1 : S.B0;
        const int vt = u / NS, t = u - vt * NS;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            // row-blocks past the bucket's end are clamped to its last one: their scores are
            // discarded by the row < n_b test of the epilogue (logical rows >= n_b)
#ifdef LMI_ABL_HOTA  // timing-only ablation: every block streams the same 16 row-blocks (L2-resident)
            const int rb = (rb_in_b0 + (vt * 4 + w) * RB + j) & 15;
#else
            const int rb = min(rb_in_b0 + (vt * 4 + w) * RB + j, nrb_b - 1);
#endif
            const float4* a = aslab + ((size_t)rb * KG + t * STAGE_G) * 64;
#ifndef LMI_ABL_NOA
#pragma unroll
            for (int g = 0; g < STAGE_G; ++g)
                glds16(a + (size_t)g * 64, sA + ((w * RB + j) * STAGE_G + g) * 64);
#endif
        }
#ifndef LMI_ABL_NOB
        if (w < NCB) {
            const float4* q = bbase + (size_t)(t * STAGE_G) * 64;
#pragma unroll
            for (int g = 0; g < STAGE_G; ++g) glds16(q + (size_t)g * 64, sB + (w * STAGE_G + g) * 64);
        }
#endif
    }

    template <int PAR>
    __device__ __forceinline__ void compute(int u) {  // MFMAs of stage u (parity-PAR buffers)
        const float4* sA = (PAR ? S.A1 : S.A0) + (w * RB) * STAGE_G * 64 + lane;
        const float4* sB = (PAR ? S.B1 : S.B0) + lane;
        const int vt = u / NS, t = u - vt * NS;
        float4 fa[2][RB], fb[2][NCB];  // fragment registers, double-buffered over g
#pragma unroll
        for (int j = 0; j < RB; ++j) fa[0][j] = sA[(j * STAGE_G) * 64];
#pragma unroll
        for (int n = 0; n < NCB; ++n) fb[0][n] = sB[(n * STAGE_G) * 64];
#pragma unroll
        for (int g = 0; g < STAGE_G; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < STAGE_G) {
#pragma unroll
                for (int j = 0; j < RB; ++j) fa[nxt][j] = sA[(j * STAGE_G + g + 1) * 64];
#pragma unroll
                for (int n = 0; n < NCB; ++n) fb[nxt][n] = sB[(n * STAGE_G + g + 1) * 64];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const float4 a = fa[cur][j];
                    const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
#pragma unroll
                    for (int n = 0; n < NCB; ++n) {
                        const float4 b = fb[cur][n];
                        const float bv = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j][n], 0, 0, 0);
                    }
                }
            }
        }
        if (t == NS - 1) {  // tile finished: filter the 16 x RB x NCB scores of this lane
            // Pruning bound (exact): col_thr[col] is the 10th-best similarity of some FINISHED chunk of
            // the same bucket for this query, so a score strictly below it cannot be in the bucket's
            // top-10; equal scores are kept (the row tie-break may still favour them).  Re-read per
            // tile (one L2 load per column), never kept in registers across the MFMA loop.
            float gthr[NCB];
#pragma unroll
            for (int n = 0; n < NCB; ++n)
                gthr[n] = __hip_atomic_load(cthr + n * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const unsigned rowbase = (unsigned)((rb_in_b0 + (vt * 4 + w) * RB + j) * 32);
#pragma unroll
                for (int n = 0; n < NCB; ++n) {
                    const float thr = lv[n][KPB - 1];
                    unsigned mask = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mask |= (unsigned)(acc[j][n][r] > thr && acc[j][n][r] >= gthr[n] && rowbase + acc_row(r, h) < (unsigned)n_b) << r;
                    while (mask) {  // rare after warm-up; ascending r == ascending row
                        const int r = __builtin_ctz(mask);
                        mask &= mask - 1;
                        float s = acc[j][n][0];
#pragma unroll
                        for (int i = 1; i < 16; ++i) s = (r == i) ? acc[j][n][i] : s;
                        if (s > lv[n][KPB - 1]) list_insert(lv[n], li[n], s, rowbase + acc_row(r, h));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;
                }
            }
        }
    }

    template <int PAR>
    __device__ __forceinline__ void step(int u) {
        if (u + 1 < total) issue<1 - PAR>(u + 1);
        compute<PAR>(u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    __device__ __forceinline__ void run(int b, int qt, int ch) {
        const int tid = threadIdx.x;
        lane = tid & 63; w = tid >> 6; h = lane >> 5; c = lane & 31;
        KG = P.KG; NS = KG / STAGE_G;
        n_b = P.nb_rows[b];
        nrb_b = (n_b + 31) >> 5;
        rb_in_b0 = ch * P.chunk_rb;                              // first row-block of the chunk
        const int nrb = min(P.chunk_rb, nrb_b - rb_in_b0);       // row-blocks in this chunk
        const int nvt = (nrb + 4 * RB - 1) / (4 * RB);           // TILE_ROWS-row tiles
        const int cb0 = P.cb_start[b] + qt * 4;
        aslab = P.slab + ((size_t)P.rb_start[b] * KG) * 64 + lane;
        bbase = P.qfrag + ((size_t)(cb0 + w) * KG) * 64 + lane;  // wave w stages col-block w
        total = nvt * NS;
        cthr = P.col_thr + (size_t)cb0 * 32 + c;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
#pragma unroll
            for (int j = 0; j < KPB; ++j) { lv[n][j] = -INFINITY; li[n][j] = NOROW; }
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.0f;
        }

        issue<0>(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int u = 0; u < total; u += 2) {
            step<0>(u);
            if (u + 1 < total) step<1>(u + 1);
        }

        // ---- merge the 8 (wave, half) lists of every column through LDS, two col-blocks per pass:
        //      scores of col-block n2 in A<n2>, rows in B<n2>, each [32][8][KPB] ----
        const int m_b = P.m[b];
        const int nch_b = P.nch[b];
        const long long pbase = P.part_base[b];
#pragma unroll
        for (int pass = 0; pass < (NCB + 1) / 2; ++pass) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int n = pass * 2 + n2;
                if (n < NCB) {
                    float* ms = reinterpret_cast<float*>(n2 ? S.A1 : S.A0);
                    unsigned* mr = reinterpret_cast<unsigned*>(n2 ? S.B1 : S.B0);
                    const int o = (c * 8 + (w * 2 + h)) * KPB;
#pragma unroll
                    for (int j = 0; j < KPB; ++j) { ms[o + j] = lv[n][j]; mr[o + j] = li[n][j]; }
                }
            }
            __syncthreads();
            if (tid < 64) {
                const int n2 = tid >> 5, cc = tid & 31, n = pass * 2 + n2;
                const int col_in_b = qt * TILE_COLS + n * 32 + cc;
                if (n < NCB && col_in_b < m_b) {
                    const float* ms = reinterpret_cast<const float*>(n2 ? S.A1 : S.A0);
                    const unsigned* mr = reinterpret_cast<const unsigned*>(n2 ? S.B1 : S.B0);
                    const int o = cc * 8 * KPB;
                    unsigned heads = 0;  // 4 bits per source list
                    const long long dst = (pbase + (long long)col_in_b * nch_b + ch) * KPB;
                    for (int j = 0; j < KPB; ++j) {
                        float bs = -INFINITY;
                        unsigned br = NOROW;
                        int bsrc = 0;
#pragma unroll
                        for (int src = 0; src < 8; ++src) {
                            const int hd = (heads >> (4 * src)) & 15;
                            if (hd < KPB) {
                                const float s = ms[o + src * KPB + hd];
                                const unsigned r = mr[o + src * KPB + hd];
                                if (better(s, r, bs, br)) { bs = s; br = r; bsrc = src; }
                            }
                        }
                        heads += 1u << (4 * bsrc);
                        P.part_score[dst + j] = bs;
                        P.part_row[dst + j] = br;
                        if (j == KPB - 1 && br != NOROW) atomic_max_float(P.col_thr + (size_t)(cb0 + n) * 32 + cc, bs);
                    }
                }
            }
            __syncthreads();
        }
    }
};

// Alternatives measured and removed (round 1, tools/scan_ab.py, 10M x 768, interleaved in one process):
// a register-staged item (A by global_load_dwordx4 straight to VGPRs, B via VGPR -> ds_write, one
// list per (wave, column)) ran 114 TFLOP/s against this LDS-DMA form's 119; issuing the DMA one
// k-group at a time between the MFMAs -3 %; RB = 2 (256 x 128 tile, one block per CU) -16 %.  With the
// operand loads removed the loop runs 141 TFLOP/s: every 1-KiB operand load costs ~100 SIMD cycles
// of MFMA issue whichever way it is loaded, even when it hits L2.

__global__ __launch_bounds__(256, RB == 1 ? 2 : 1) void scan_kernel(ScanParams P) {
    __shared__ __attribute__((aligned(16))) float4 sA0[4 * RB * STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) float4 sA1[4 * RB * STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) float4 sB0[4 * STAGE_G * 64];
    __shared__ __attribute__((aligned(16))) float4 sB1[4 * STAGE_G * 64];
    const ScanLds S{sA0, sA1, sB0, sB1};
    int* s_item = reinterpret_cast<int*>(sB1);  // item broadcast: no DMA is in flight between items
    ts_first(P.ts_start);
    // home queue = this block's XCD (HW_REG_XCC_ID, id 20, bits [3:0]); any value works: speed only
    int grp = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (NGRP - 1));
    for (;;) {
        if (threadIdx.x == 0) {
            int b = -1, local = 0;
            for (int tries = 0; tries < NGRP; ++tries) {
                const int tot = P.grp_total[grp];
                if (__hip_atomic_load(&P.head[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tot) {
                    const int it = (int)atomicAdd(&P.head[grp], 1u);
                    if (it < tot) {
                        const int* base = P.grp_base + grp * (P.L + 1);
                        int lo = 0, hi = P.grp_n[grp];  // last i with base[i] <= it
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (base[mid] <= it) lo = mid; else hi = mid;
                        }
                        b = P.grp_bucket[grp * P.L + lo];
                        local = it - base[lo];
                        break;
                    }
                }
                grp = (grp + 1) & (NGRP - 1);  // queue drained: steal from the next one
            }
            s_item[0] = b;
            s_item[1] = local;
        }
        __syncthreads();
        const int b = s_item[0], local = s_item[1];
        __syncthreads();
        if (b < 0) { ts_max(P.ts_end_cell); return; }
        const int m_b = P.m[b];
        const int nqt = (m_b + TILE_COLS - 1) / TILE_COLS;
        const int qt = local % nqt, ch = local / nqt;
        const int ncb = min(4, (m_b - qt * TILE_COLS + 31) >> 5);
        switch (ncb) {
            case 1: { ScanItem<1> it{P, S}; it.run(b, qt, ch); break; }
            case 2: { ScanItem<2> it{P, S}; it.run(b, qt, ch); break; }
            case 3: { ScanItem<3> it{P, S}; it.run(b, qt, ch); break; }
            default: { ScanItem<4> it{P, S}; it.run(b, qt, ch); break; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Merge (LearnedIndex.py:125-146, 368-371): one wave per query.
//   phase A, per rank: chunk partials -> top-10 by (sim desc, row asc); dist = 1 - sim (binary32),
//            id = ids[row]; short buckets padded like faiss (-FLT_MAX / last id);
//            empty / unowned / invalid bucket -> (+inf, 0).
//   phase B: ranks merged by (dist asc, rank asc, position asc) == the reference's repeated
//            hstack + stable argsort; first kout entries.
// raw != 0 (lmi_knn_ip): dist <- similarity, id <- row (NOROW on padding), no 1 - x.
// ------------------------------------------------------------------------------------------------
struct MergeParams {
    const int* bucket_order;  // [nq][nb]
    const int* slot_col;      // [nq][nb]
    int nq, nb, L, kout, raw;
    int skip_a;  // rank lists were already written (prefilter path): phase B only
    const int* rb_start;
    const int* nb_rows;
    const int* nch;
    const int* cb_start;
    const long long* part_base;
    const float* part_score;
    const unsigned* part_row;
    const unsigned* ids_slab;
    const float* qn2;   // L2 metric: |q|^2 per query (nullptr: inner product, dist = 1 - sim)
    float* rank_d;      // scratch [nq][nb][KPB]
    unsigned* rank_id;  // scratch [nq][nb][KPB]
    float* out_d;       // [nq][kout]
    unsigned* out_id;
    unsigned* out_key;  // nullable
    unsigned long long* ts;              // nullable: the call's stamp set (ST_MERGE at the start, ST_END by the last workgroup)
    const unsigned long long* scan_end;  // nullable: scan_kernel's end cell, copied to ST_SCAN1
};
__device__ __forceinline__ void merge_stamps_begin(const MergeParams& P) {
    if (P.ts && threadIdx.x == 0 && blockIdx.x == 0) {
        P.ts[ST_MERGE] = wall_clock64();
        if (P.scan_end) P.ts[ST_SCAN1] = *P.scan_end;
    }
}

__global__ __launch_bounds__(64) void merge_kernel(MergeParams P) {
    const int q = blockIdx.x, lane = threadIdx.x;
    merge_stamps_begin(P);
    if (q >= P.nq) return;
    float* rd = P.rank_d + (size_t)q * P.nb * KPB;
    unsigned* ri = P.rank_id + (size_t)q * P.nb * KPB;
    const float FMAXV = 3.402823466e+38f;
    // ---------------- phase A ----------------
    for (int r = 0; r < (P.skip_a ? 0 : P.nb); ++r) {
        const int p = q * P.nb + r;
        const int b = P.bucket_order[p];
        const int col = P.slot_col[p];
        if (col < 0) {  // unvisited: LearnedIndex.py:340-341 initial values
            if (lane < KPB) { rd[r * KPB + lane] = P.raw ? -FMAXV : INFINITY; ri[r * KPB + lane] = P.raw ? NOROW : 0u; }
            continue;
        }
        const int n_b = P.nb_rows[b], nch = P.nch[b];
        const long long l0 = P.part_base[b] + (long long)(col - P.cb_start[b] * 32) * nch;
        // lane owns chunk lists lane, lane+64, ...; heads packed 4 bits each (<= 8 lists per lane)
        unsigned long long heads = 0;
        float my_s = 0.f;
        unsigned my_r = 0;
        for (int j = 0; j < KPB; ++j) {
            float bs = -INFINITY;
            unsigned br = NOROW;
            int bl = -1;
            int t = 0;
            for (int chn = lane; chn < nch; chn += 64, ++t) {
                const int hd = (int)((heads >> (4 * (t & 15))) & 15);
                if (hd < KPB) {
                    const float s = P.part_score[(l0 + chn) * KPB + hd];
                    const unsigned rr = P.part_row[(l0 + chn) * KPB + hd];
                    if (better(s, rr, bs, br)) { bs = s; br = rr; bl = t; }
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
            if (wl == lane && bl >= 0) heads += 1ull << (4 * (bl & 15));
            if (lane == j) { my_s = bs; my_r = br; }
        }
        if (lane < KPB) {
            float dv;
            unsigned iv;
            const bool real = my_r != NOROW && lane < n_b;
            if (P.raw) {
                dv = real ? my_s : -FMAXV;
                iv = real ? my_r : NOROW;
            } else if (real) {
                dv = sim_to_dist(my_s, P.qn2, q);
                iv = P.ids_slab[(size_t)P.rb_start[b] * 32 + my_r];
            } else {  // faiss padding: sim = -FLT_MAX, idx = -1 -> last label of the bucket
                dv = pad_dist(P.qn2);
                iv = P.ids_slab[(size_t)P.rb_start[b] * 32 + (n_b - 1)];
            }
            rd[r * KPB + lane] = dv;
            ri[r * KPB + lane] = iv;
        }
    }
    __syncthreads();  // single wave: orders the global writes above before the reads below
    // ---------------- phase B ----------------
    // lane owns ranks lane, lane+64, ...; candidate key (dist, rank); position is implicit (heads)
    unsigned long long heads = 0;
    for (int j = 0; j < P.kout; ++j) {
        float bd = INFINITY;
        int brk = 0x7fffffff, bt = -1;
        bool any = false;
        int t = 0;
        for (int r = lane; r < P.nb; r += 64, ++t) {
            const int hd = (int)((heads >> (4 * (t & 15))) & 15);
            if (hd < KPB) {
                const float dv = rd[r * KPB + hd];
                const bool bet = P.raw ? (!any || dv > bd) : (!any || dv < bd);
                if (bet) { bd = dv; brk = r; bt = t; any = true; }
            }
        }
        if (!any) { bd = P.raw ? -INFINITY : INFINITY; brk = 0x7fffffff; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(bd, o);
            const int ork = __shfl_xor(brk, o);
            const bool bet = P.raw ? (od > bd) : (od < bd);
            if ((ork != 0x7fffffff) && (brk == 0x7fffffff || bet || (od == bd && ork < brk))) { bd = od; brk = ork; }
        }
        // winner rank brk (same on all lanes); its owner advances
        int pos = 0;
        if (brk != 0x7fffffff && (brk & 63) == lane) {
            pos = (int)((heads >> (4 * (bt & 15))) & 15);
            heads += 1ull << (4 * (bt & 15));
        }
        const int owner = brk == 0x7fffffff ? 0 : (brk & 63);
        pos = __shfl(pos, owner);
        if (lane == 0) {
            const size_t o = (size_t)q * P.kout + j;
            if (brk == 0x7fffffff) {
                P.out_d[o] = P.raw ? -FMAXV : INFINITY;
                P.out_id[o] = P.raw ? NOROW : 0u;
                if (P.out_key) P.out_key[o] = 0xFFFFFFFFu;
            } else {
                P.out_d[o] = rd[brk * KPB + pos];
                P.out_id[o] = ri[brk * KPB + pos];
                if (P.out_key) P.out_key[o] = (unsigned)brk * 16u + (unsigned)pos;
            }
        }
    }
    if (P.ts) ts_last(P.ts + ST_END);
}

// Phase B alone, one THREAD per query, for nb <= 16 rank lists that already exist (the prefilter path): a
// head-cursor merge of nb sorted lists by (dist, rank, position) in registers.  (merge_kernel's wave-wide
// argmin costs 18 ds_bpermute round trips per output: 43 us for 10 000 queries; this one 6 us.)
__global__ __launch_bounds__(64) void merge_ranks_kernel(MergeParams P) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    merge_stamps_begin(P);
    if (q >= P.nq) return;
    const float* rd = P.rank_d + (size_t)q * P.nb * KPB;
    const unsigned* ri = P.rank_id + (size_t)q * P.nb * KPB;
    const float FMAXV = 3.402823466e+38f;
    unsigned long long heads = 0;  // 16 ranks x 4 bits
    for (int j = 0; j < P.kout; ++j) {
        float bd = 0.0f;
        int brk = -1, bpos = 0;
        for (int r = 0; r < P.nb; ++r) {
            const int hd = (int)((heads >> (4 * r)) & 15);
            if (hd < KPB) {
                const float dv = rd[r * KPB + hd];
                if (brk < 0 || (P.raw ? dv > bd : dv < bd)) { bd = dv; brk = r; bpos = hd; }  // ties: the lower rank stays
            }
        }
        const size_t o = (size_t)q * P.kout + j;
        if (brk < 0) {
            P.out_d[o] = P.raw ? -FMAXV : INFINITY;
            P.out_id[o] = P.raw ? NOROW : 0u;
            if (P.out_key) P.out_key[o] = 0xFFFFFFFFu;
        } else {
            heads += 1ull << (4 * brk);
            P.out_d[o] = bd;
            P.out_id[o] = ri[brk * KPB + bpos];
            if (P.out_key) P.out_key[o] = (unsigned)brk * 16u + (unsigned)bpos;
        }
    }
    if (P.ts) ts_last(P.ts + ST_END);   // (thread 0 of the last workgroup always holds a query)
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU: merge of the all-gathered per-GPU results.  gathered [world][nq][kout], every list
// sorted by (dist, key); output = first kout of the union by (dist asc, key asc).  One wave per
// query, lane = source GPU (world <= 64).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void merge_gathered_kernel(const float* __restrict__ gd,
                                                           const unsigned* __restrict__ gi,
                                                           const unsigned* __restrict__ gk, int world,
                                                           long long world_stride, int nq, int kout,
                                                           float* __restrict__ od, unsigned* __restrict__ oi) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    int head = 0;
    const size_t base = (size_t)lane * world_stride + (size_t)q * kout;
    for (int j = 0; j < kout; ++j) {
        float d = INFINITY;
        unsigned key = 0xFFFFFFFFu;
        if (lane < world && head < kout) { d = gd[base + head]; key = gk[base + head]; }
        int wl = lane;
        float bd = d;
        unsigned bk = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float odv = __shfl_xor(bd, o);
            const unsigned okv = __shfl_xor(bk, o);
            const int ol = __shfl_xor(wl, o);
            if (odv < bd || (odv == bd && (okv < bk || (okv == bk && ol < wl)))) { bd = odv; bk = okv; wl = ol; }
        }
        unsigned idv = 0;
        if (lane == wl && lane < world && head < kout) { idv = gi[base + head]; ++head; }
        idv = __shfl(idv, wl);
        if (lane == 0) { od[(size_t)q * kout + j] = bd; oi[(size_t)q * kout + j] = idv; }
    }
}

}  // namespace lmi
