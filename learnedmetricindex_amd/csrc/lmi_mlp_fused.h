// lmi_mlp_fused.h -- one-launch MLP for the navigation models (gfx950), and the device-side multi-level walk.
//
// NeuralNetwork.predict_proba (model.py:226-241: Sequential(Linear, ReLU, .., Linear) -> softmax -> topk(L)) for a
// block of 32 columns, every layer in ONE kernel:
//   * a column is a query (1-level index, `lmi_mlp_topk` / `lmi_mlp_proba`) or a (query, internal node) pair of the
//     multi-level walk (LearnedIndex.py:254-301: all queries that popped the same node are evaluated together, the
//     block's model id selects the weight set);
//   * layer 0 streams the input rows through LDS in chunks of 96 features (coalesced row segments, gathered by the
//     column -> query map), hidden activations never leave LDS (row-major [32][width + 1]: the odd stride makes both
//     the B-operand reads and the accumulator write-back bank-conflict free), ReLU is applied in registers;
//   * every inner product is the canonical chain: v_mfma_f32_32x32x2_f32 fed k in order, one accumulator per
//     output, started at the bias (bit-identical to oracle/lmi_oracle.c and to the unfused mlp_layer_kernel);
//     weights are the fragment-major tiles lmi_set_mlp packs, read straight from L2 one k-group ahead;
//   * epilogue per column, from the logits in LDS: class ranking by selection passes (ties -> lower class index),
//     the canonical softmax (same expf, row sum in class order) and, by mode,
//       FM_TOPK   bucket_order[q][0..nb)                                        (LearnedIndex.py:197-214)
//       FM_PROBA  probs[q][L] descending + classes[q][L]                        (model.py:238-241)
//       FM_NAV    the node's children pushed into the query's priority queue    (LearnedIndex.py:220-227, 289-299)
//     Final layers wider than FM_MAXH keep their logits in global memory and the host runs the separate ranking
//     kernels afterwards.
#pragma once
#include "lmi_kernels.h"

namespace lmi {

constexpr int FM_MAXL = 8;      // Linear layers per model (LMI_MAX_LAYERS)
constexpr int FM_COLS = 32;     // columns per block = one MFMA column block
constexpr int FM_CHUNK = 96;    // input features staged per step of layer 0 (12 k-groups: a multiple of the 3-deep weight prefetch)
constexpr int FM_CHUNK_S = FM_CHUNK + 1;
constexpr int FM_MAXH = 512;    // widest layer whose outputs stay in LDS
enum { FM_TOPK = 0, FM_PROBA = 1, FM_NAV = 2 };

struct ModelDesc {  // device copy of one model's shape and weights (root = model 0, internal nodes 1..)
    int n_layers;
    int dims[FM_MAXL + 1];
    int KG[FM_MAXL];             // k-groups (of 8) of the layer's input: cdiv(d, 8) for layer 0, 4 * n_rb(prev) after
    const float4* W[FM_MAXL];    // fragment-major [n_rb][KG][64]
    const float* b[FM_MAXL];     // [n_rb * 32], zero padded
};

struct FusedParams {
    const ModelDesc* models;
    int n_models;
    const float* x;              // row-major [nq][d]
    int d, nq;
    int s0, s1;                  // strides (floats, odd) of the two activation buffers; act1 starts act0_floats after act0
    int act0_floats;
    int logits_in_lds;           // the last layer's outputs fit the activation buffer (else: global, host ranks)
    // columns: node_count == nullptr -> model 0 over the nq queries; else the pending (query, node) pairs of this
    // step of the walk, model m's queries at col_query[m * nq ..)
    const int* node_count;
    const int* col_query;
    int nb;                      // FM_TOPK
    int* order;                  // FM_TOPK [nq][nb]
    float* logits_out;           // nullable [nq][L] (model 0 only)
    float* probs;                // FM_PROBA [nq][L]
    int* classes;                // FM_PROBA [nq][L]
    // FM_NAV: per-query priority queue (entries are never moved: a popped entry is marked dead)
    float* pq_prob;              // [cap][nq]
    int* pq_ent;                 // [cap][nq] flat child index = child_offset[model] + class; -1 = popped
    int* pq_len;                 // [nq] entries ever pushed
    int cap;
    const int* child_offset;     // [n_models + 1]
    int reverse;                 // root: children pushed least probable first (LearnedIndex.py:220-227)
    int* zero_counts;            // block 0 clears the NEXT step's counters
    int n_zero;
    unsigned long long* ts;      // nullable: device stamp of the launch's start (lmi_kernels.h)
};

template <int MODE>
__global__ __launch_bounds__(256) void mlp_fused_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) float fm_smem[];
    __shared__ int s_hdr[4];
    __shared__ int s_q[FM_COLS];
    __shared__ float s_max[FM_COLS], s_sum[FM_COLS];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the row-block tests below are s_cbranch, not exec masks
    ts_first(P.ts);
    if (tid == 0) {
        int model = -1, ncols = 0, first = 0;
        if (!P.node_count) {
            model = 0;
            first = blockIdx.x * FM_COLS;
            ncols = min(FM_COLS, P.nq - first);
        } else {
            int b = blockIdx.x;
            for (int m = 0; m < P.n_models; ++m) {
                const int cnt = P.node_count[m];
                const int cbs = (cnt + FM_COLS - 1) / FM_COLS;
                if (b < cbs) { model = m; first = b * FM_COLS; ncols = min(FM_COLS, cnt - first); break; }
                b -= cbs;
            }
        }
        s_hdr[0] = ncols > 0 ? model : -1;
        s_hdr[1] = ncols;
        s_hdr[2] = first;
    }
    if (P.zero_counts && blockIdx.x == 0)
        for (int i = tid; i < P.n_zero; i += 256) P.zero_counts[i] = 0;
    __syncthreads();
    const int model = s_hdr[0], ncols = s_hdr[1], first = s_hdr[2];
    if (model < 0) return;
    if (tid < FM_COLS)
        s_q[tid] = tid < ncols ? (P.node_count ? P.col_query[(size_t)model * P.nq + first + tid] : first + tid) : -1;
    __syncthreads();
    const ModelDesc& M = P.models[model];
    const int n_layers = M.n_layers;
    float* chunk0 = fm_smem;                                   // two input chunks: one computes while the next lands
    float* chunk1 = fm_smem + FM_COLS * FM_CHUNK_S;
    float* act0 = fm_smem + 2 * FM_COLS * FM_CHUNK_S;
    float* act1 = act0 + P.act0_floats;
    const int L = M.dims[n_layers];

    for (int li = 0; li < n_layers; ++li) {
        const int J = M.dims[li + 1];
        const int n_rb = (J + 31) >> 5;
        const int KG = M.KG[li];
        const bool last = li + 1 == n_layers;
        float* out = (li & 1) ? act1 : act0;
        const int So = (li & 1) ? P.s1 : P.s0;
        const float* in = (li & 1) ? act0 : act1;   // li > 0: the previous layer's outputs
        const int Si = (li & 1) ? P.s0 : P.s1;
        const bool out_lds = !last || P.logits_in_lds;
        // the descriptor's pointers come out of memory, so hipcc would use FLAT loads for them -- which count on
        // lgkmcnt as well, so that every `s_waitcnt lgkmcnt(0)` for an LDS read also waited for the weight
        // prefetch (measured: no prefetch depth helped until the loads were global_load)
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) f4v* gf4;
        typedef const __attribute__((address_space(1))) float* gf1;
        const gf4 Wl = (gf4)M.W[li];
        const gf1 bl = (gf1)M.b[li];
        for (int pass = 0; pass * 16 < n_rb; ++pass) {
            int rb[4];
            bool ok[4];
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rb[j] = pass * 16 + 4 * j + w;
                ok[j] = rb[j] < n_rb;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = bl[(ok[j] ? rb[j] : 0) * 32 + acc_row(r, h)];
            }
            gf4 ap[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ap[j] = Wl + (size_t)(ok[j] ? rb[j] : 0) * KG * 64 + lane;
            // The weight fragments come straight from L2, three k-groups ahead of their use, in three NAMED register
            // sets used in rotation (group g uses set g % 3 and refills it for group g + 3): with one wave per SIMD
            // nothing else hides the load latency; rotating by register copies would make every copy wait for the
            // newest load, and selecting the set by a run-time branch made hipcc shuffle the accumulators between the
            // branches (880 v_accvgpr_mov, scratch): the loops below are unrolled by three instead.
            f4v aq0[4], aq1[4], aq2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                aq0[j] = ap[j][0];
                aq1[j] = ap[j][(size_t)min(1, KG - 1) * 64];
                aq2[j] = ap[j][(size_t)min(2, KG - 1) * 64];
            }
            // one k-group G with weight set AQ and B from BSRC (this lane's column, k offset applied); row-blocks past
            // the layer's last one are skipped by scalar branches (their loads are clamped)
#define FM_STEP(AQ, BNEXT, G)                                                                              \
            {                                                                                              \
                const float* bs_ = (BNEXT);  /* the NEXT group's B values are requested before this group's MFMAs */ \
                const float n0 = bs_[0 + h], n1 = bs_[2 + h], n2 = bs_[4 + h], n3 = bs_[6 + h];            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
                    if (ok[j]) {                                                                           \
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[j].x, bq0, acc[j], 0, 0, 0);      \
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[j].y, bq1, acc[j], 0, 0, 0);      \
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[j].z, bq2, acc[j], 0, 0, 0);      \
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[j].w, bq3, acc[j], 0, 0, 0);      \
                    }                                                                                      \
                }                                                                                          \
                const int gn_ = min((G) + 3, KG - 1);                                                      \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) AQ[j] = ap[j][(size_t)gn_ * 64];             \
                bq0 = n0; bq1 = n1; bq2 = n2; bq3 = n3;                                                    \
            }
            // groups [G0, G1) (G0 a multiple of 3) with B at BASE + 8 * (g - G0); the look-ahead of the last group
            // re-reads its own values
#define FM_RANGE(BASE, G0, G1)                                                                             \
            {                                                                                              \
                int g_ = (G0);                                                                             \
                const float* bb_ = (BASE);                                                                 \
                const int last_ = ((G1) - (G0) - 1) * 8;                                                   \
                float bq0 = bb_[0 + h], bq1 = bb_[2 + h], bq2 = bb_[4 + h], bq3 = bb_[6 + h];              \
                int o_ = 0;                                                                                \
                for (; g_ + 3 <= (G1); g_ += 3, o_ += 24) {                                                \
                    FM_STEP(aq0, bb_ + min(o_ + 8, last_), g_)                                             \
                    FM_STEP(aq1, bb_ + min(o_ + 16, last_), g_ + 1)                                        \
                    FM_STEP(aq2, bb_ + min(o_ + 24, last_), g_ + 2)                                        \
                }                                                                                          \
                if (g_ < (G1)) FM_STEP(aq0, bb_ + min(o_ + 8, last_), g_)                                  \
                if (g_ + 1 < (G1)) FM_STEP(aq1, bb_ + min(o_ + 16, last_), g_ + 1)                         \
            }
            if (li == 0) {
                // input features in chunks of 96 through two LDS buffers: thread -> (row, 12-float segment); the
                // next chunk's rows are requested before this chunk's MFMAs and written to the other buffer after them
                const int d = P.d;
                const int nck = (KG * 8 + FM_CHUNK - 1) / FM_CHUNK;
                const int row = tid >> 3, seg = tid & 7;
                const int qi = s_q[row];
                const float* src = P.x + (size_t)(qi < 0 ? 0 : qi) * d;
                const bool vec_ok = (d & 3) == 0;
                float4 st[3];
                // loads are unconditional (clamped addresses) and masked afterwards: a load under a per-lane condition
                // makes hipcc branch around it and wait vmcnt(0) per element
                auto fetch = [&](int ck) {
                    const int k0 = ck * FM_CHUNK + seg * 12;
                    if (vec_ok && (ck + 1) * FM_CHUNK <= d) {  // block-uniform: the whole chunk lies inside the rows
#pragma unroll
                        for (int i = 0; i < 3; ++i) st[i] = *reinterpret_cast<const float4*>(src + k0 + 4 * i);
                    } else {
                        float e[12];
#pragma unroll
                        for (int i = 0; i < 12; ++i) e[i] = src[min(k0 + i, d - 1)];
#pragma unroll
                        for (int i = 0; i < 12; ++i) e[i] = k0 + i < d ? e[i] : 0.0f;
#pragma unroll
                        for (int i = 0; i < 3; ++i) st[i] = make_float4(e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]);
                    }
                    if (qi < 0) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                };
                auto put = [&](float* buf) {
                    float* dst = buf + row * FM_CHUNK_S + seg * 12;
#pragma unroll
                    for (int i = 0; i < 3; ++i) { dst[4 * i + 0] = st[i].x; dst[4 * i + 1] = st[i].y; dst[4 * i + 2] = st[i].z; dst[4 * i + 3] = st[i].w; }
                };
                fetch(0);
                put(chunk0);
                __syncthreads();
                for (int ck = 0; ck < nck; ++ck) {
                    const float* cur = (ck & 1) ? chunk1 : chunk0;
                    if (ck + 1 < nck) fetch(ck + 1);
                    const int g_end = min(KG, (ck + 1) * (FM_CHUNK / 8));
                    FM_RANGE(cur + c * FM_CHUNK_S, ck * (FM_CHUNK / 8), g_end)
                    if (ck + 1 < nck) put((ck & 1) ? chunk0 : chunk1);
                    __syncthreads();
                }
            } else {
                FM_RANGE(in + c * Si, 0, KG)
            }
#undef FM_RANGE
#undef FM_STEP
            // outputs: feature f = rb*32 + acc_row(r, h) of column c
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!ok[j]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = rb[j] * 32 + acc_row(r, h);
                    float v = acc[j][r];
                    if (!last) v = fmaxf(v, 0.0f);
                    if (out_lds) out[c * So + f] = v;
                    if (last && P.logits_out && c < ncols && f < L) P.logits_out[(size_t)s_q[c] * L + f] = v;
                }
            }
        }
        __syncthreads();
    }
    if (!P.logits_in_lds) return;  // wide output layer: the host runs rank_classes_kernel / softmax_ranked_kernel
    const float* lg = ((n_layers - 1) & 1) ? act1 : act0;
    const int SL = ((n_layers - 1) & 1) ? P.s1 : P.s0;
    if (MODE != FM_TOPK) {
        // canonical softmax terms: row max, then the sum of expf(l - max) in class order (softmax_ranked_kernel)
        if (tid < ncols) {
            const float* l = lg + tid * SL;
            float m = l[0];
            for (int j = 1; j < L; ++j) m = l[j] > m ? l[j] : m;
            float s = 0.0f;
            for (int j = 0; j < L; ++j) s += lmi_expf(l[j] - m);
            s_max[tid] = m;
            s_sum[tid] = s;
        }
        __syncthreads();
    }
    // Ranking: 8 lanes per column, all 8 columns of a wave side by side (a wave per column, one column after the
    // other, took 27 us of a 106-us block: 32 sequential passes of 12 dependent shuffles each).  Selection pass t
    // finds the t-th class of the descending order, ties -> lower class index (rank_classes_kernel's rule).
    const int T = MODE == FM_TOPK ? P.nb : L;
    const int col = w * 8 + (lane >> 3), sub = lane & 7;
    const bool live = col < ncols;
    const int q = s_q[live ? col : 0];
    const float* l = lg + (live ? col : 0) * SL;
    int base = 0;
    if (MODE == FM_NAV) base = P.pq_len[q];
    float pv = INFINITY;
    int pi = -1;
    for (int t = 0; t < T; ++t) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = sub; j < L; j += 8) {
            const float v = l[j];
            const bool after = (v < pv) || (v == pv && j > pi);
            if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (sub == 0 && live) {
            const int cls = bi == 0x7fffffff ? -1 : bi;
            if (MODE == FM_TOPK) {
                P.order[(size_t)q * P.nb + t] = cls;
            } else {
                const float pr = cls >= 0 ? lmi_expf(l[cls] - s_max[col]) / s_sum[col] : __builtin_nanf("");
                if (MODE == FM_PROBA) {
                    P.classes[(size_t)q * L + t] = cls;
                    P.probs[(size_t)q * L + t] = pr;
                } else {
                    const int pos = base + (P.reverse ? L - 1 - t : t);
                    if (pos < P.cap) {  // entry-major: the pop kernel's threads (one per query) read coalesced
                        P.pq_prob[(size_t)pos * P.nq + q] = pr;
                        P.pq_ent[(size_t)pos * P.nq + q] = cls >= 0 ? P.child_offset[model] + cls : -1;
                    }
                }
            }
        }
        pv = bv;
        pi = bi;
    }
    if (MODE == FM_NAV && sub == 0 && live) P.pq_len[q] = min(P.cap, base + L);
}

// ------------------------------------------------------------------------------------------------
// Multi-level walk, one step (LearnedIndex.py:234-250: `pq.pop` for every unfinished query, then
// `_visit_internal_nodes` / `_visit_buckets`).  One thread per query pops the most probable entries of its
// queue -- ties: the entry pushed LATER (what the reference's ascending stable sort + pop-from-the-tail does) --
// and either records a bucket, or queues the query for its node's model (this step's mlp_fused_kernel<FM_NAV>),
// or drops a path that is neither.  child_bucket: >= 0 slab bucket id, -1 a listed bucket without objects (the
// slot stays unvisited but counts), -2 not a bucket.
// ------------------------------------------------------------------------------------------------
struct NavParams {
    int nq, nb, cap;
    float* pq_prob;
    int* pq_ent;
    const int* pq_len;
    const int* child_model;   // [entries] model id of the child, -1: leaf
    const int* child_bucket;  // [entries]
    int* out_len;             // [nq] buckets recorded
    int* out_slab;            // [nq][nb] slab bucket ids (-1: no objects)
    int* out_ent;             // [nq][nb] flat child index of the bucket (-> its path on the host)
    int* node_count;          // [n_models] this step's counters (zeroed by the previous step)
    int* col_query;           // [n_models][nq]
    int* active;              // queries waiting for an expansion after this step
    const int* prev_active;   // nullable: the same count of the step before -- 0: the walk is over, this launch returns (the host then
                              // enqueues every possible step up front instead of reading the count back)
};

// The queries of a wave that stopped at an internal node queue up for its model: ONE counter atomic per (wave, model) and one for the
// step's count of waiting queries (round 5: 2 atomics per query on 1 + n_models addresses were most of the step -- 10 000 queries: 40-100 us).
// my_cm: the lane's model, -1: none.  Every lane of the wave must call (no early returns in front).
__device__ __forceinline__ void nav_push(const NavParams& P, int q, int my_cm) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long todo = __ballot(my_cm >= 0);
    const int total = (int)__popcll(todo);
    if (total == 0) return;
    const int first = __ffsll((long long)todo) - 1;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int cm = __shfl(my_cm, leader, 64);
        const unsigned long long same = __ballot(my_cm == cm);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0u));
        int base = 0;
        if (lane == leader) base = atomicAdd(&P.node_count[cm], (int)__popcll(same));
        base = __shfl(base, leader, 64);
        if (my_cm == cm) P.col_query[(size_t)cm * P.nq + base + rank] = q;
        todo &= ~same;
    }
    if (lane == first) atomicAdd(P.active, total);
}

__global__ __launch_bounds__(256) void nav_pop_kernel(NavParams P) {
    if (P.prev_active && *P.prev_active == 0) return;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < P.nq;
    int have = live ? P.out_len[q] : P.nb;
    const float* pp = P.pq_prob + (live ? q : 0);   // entry i at [i * nq]
    int* pe = P.pq_ent + (live ? q : 0);
    const int len = (live && have < P.nb) ? P.pq_len[q] : 0;
    // Bucket pops change nothing but the queue, so they continue within this step; the walk pauses at the first
    // internal node (its children's probabilities come from this step's grouped MLP launch) -- the same sequence
    // of pops as the reference's one-pop-per-iteration loop, in fewer launches.
    int my_cm = -1;
    while (len > 0) {
        float best = 0.0f;
        int bi = -1;
        for (int i = 0; i < len; ++i) {
            if (pe[(size_t)i * P.nq] < 0) continue;
            const float v = pp[(size_t)i * P.nq];
            if (bi < 0 || v >= best) { best = v; bi = i; }  // >=: the later entry wins a tie
        }
        if (bi < 0) break;  // queue exhausted: the remaining slots stay EMPTY (the reference would fail here)
        const int ent = pe[(size_t)bi * P.nq];
        pe[(size_t)bi * P.nq] = -1;
        const int cm = P.child_model[ent], cb = P.child_bucket[ent];
        if (cm >= 0) { my_cm = cm; break; }
        if (cb >= -1) {
            P.out_slab[(size_t)q * P.nb + have] = cb;
            P.out_ent[(size_t)q * P.nb + have] = ent;
            P.out_len[q] = ++have;
            if (have >= P.nb) break;
        }
    }
    nav_push(P, q, my_cm);
}

// The same step for trees whose queues fit LDS (cap <= NAV_LDS_CAP entries: [10, 10] has 110): a wave per 64 queries reads their queues
// ONCE (entry-major: a 256-byte row per entry) and every pop scans LDS instead of global memory -- a step's ~10 pops x up to 110
// entries per query took 43-101 us of a 0.96-ms walk at 10 000 queries (round 5 trace, profiles/r05_nav.txt); same pops, same order.
constexpr int NAV_LDS_CAP = 128;
__global__ __launch_bounds__(64) void nav_pop_lds_kernel(NavParams P) {
    extern __shared__ __attribute__((aligned(16))) char nav_smem[];
    if (P.prev_active && *P.prev_active == 0) return;
    float* sp = reinterpret_cast<float*>(nav_smem) + threadIdx.x;      // [cap][64]: this lane's column
    int* se = reinterpret_cast<int*>(nav_smem) + P.cap * 64 + threadIdx.x;
    const int q = blockIdx.x * 64 + threadIdx.x;
    const bool live = q < P.nq;
    int have = live ? P.out_len[q] : P.nb;
    const int len = (live && have < P.nb) ? P.pq_len[q] : 0;
    const float* pp = P.pq_prob + (live ? q : 0);
    int* pe = P.pq_ent + (live ? q : 0);
    int mx = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    for (int i0 = 0; i0 < mx; i0 += 16) {   // 32 loads in flight per lane, then their LDS stores (clamped addresses: no branch per load)
        float v[16];
        int e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t o = (size_t)min(i0 + j, max(len - 1, 0)) * P.nq;
            v[j] = pp[o];
            e[j] = pe[o];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (i0 + j < len) { sp[(i0 + j) * 64] = v[j]; se[(i0 + j) * 64] = e[j]; }
    }
    int my_cm = -1;
    while (len > 0) {
        float best = 0.0f;
        int bi = -1;
        for (int i = 0; i < len; ++i) {
            const int e = se[i * 64];
            const float v = sp[i * 64];
            if (e >= 0 && (bi < 0 || v >= best)) { best = v; bi = i; }  // >=: the later entry wins a tie
        }
        if (bi < 0) break;
        const int ent = se[bi * 64];
        se[bi * 64] = -1;
        pe[(size_t)bi * P.nq] = -1;
        const int cm = P.child_model[ent], cb = P.child_bucket[ent];
        if (cm >= 0) { my_cm = cm; break; }
        if (cb >= -1) {
            P.out_slab[(size_t)q * P.nb + have] = cb;
            P.out_ent[(size_t)q * P.nb + have] = ent;
            P.out_len[q] = ++have;
            if (have >= P.nb) break;
        }
    }
    nav_push(P, q, my_cm);
}

}  // namespace lmi
