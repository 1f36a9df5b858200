// lmi_mlp_fused.h -- one-launch MLP for the navigation models (gfx950), and the device-side multi-level walk.
//
// NeuralNetwork.predict_proba (model.py:226-241: Sequential(Linear, ReLU, .., Linear) -> softmax -> topk(L)) for a
// block of 32 columns, every layer in ONE kernel:
//   * a column is a query (1-level index, `lmi_mlp_topk` / `lmi_mlp_proba`) or a (query, internal node) pair of the
//     multi-level walk (LearnedIndex.py:254-301: all queries that popped the same node are evaluated together, the
//     block's model id selects the weight set);
//   * layer 0 streams the input rows through LDS in chunks of 128 features (coalesced row segments, gathered by the
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
constexpr int FM_CHUNK = 128;   // input features staged per step of layer 0
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
    float* pq_prob;              // [nq][cap]
    int* pq_ent;                 // [nq][cap] flat child index = child_offset[model] + class; -1 = popped
    int* pq_len;                 // [nq] entries ever pushed
    int cap;
    const int* child_offset;     // [n_models + 1]
    int reverse;                 // root: children pushed least probable first (LearnedIndex.py:220-227)
    int* zero_counts;            // block 0 clears the NEXT step's counters
    int n_zero;
};

template <int MODE>
__global__ __launch_bounds__(256) void mlp_fused_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) float fm_smem[];
    __shared__ int s_hdr[4];
    __shared__ int s_q[FM_COLS];
    __shared__ float s_max[FM_COLS], s_sum[FM_COLS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, c = lane & 31;
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
    float* chunk = fm_smem;
    float* act0 = fm_smem + FM_COLS * FM_CHUNK_S;
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
        const float4* Wl = M.W[li];
        const float* bl = M.b[li];
        for (int pass = 0; pass * 16 < n_rb; ++pass) {
            int rb[4];
            bool ok[4];
            f32x16 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rb[j] = pass * 16 + 4 * j + w;
                ok[j] = rb[j] < n_rb;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = ok[j] ? bl[rb[j] * 32 + acc_row(r, h)] : 0.0f;
            }
            const float4* ap[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ap[j] = Wl + (size_t)(ok[j] ? rb[j] : 0) * KG * 64 + lane;
            float4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ok[j] ? ap[j][0] : make_float4(0.f, 0.f, 0.f, 0.f);
            // one k-group: B from `bsrc` (this lane's column, k offset already applied), A one group ahead
            auto group = [&](const float* bsrc, int g) {
                const int gn = g + 1 < KG ? g + 1 : g;
                float4 an[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) an[j] = ok[j] ? ap[j][(size_t)gn * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float b0 = bsrc[0 + h], b1 = bsrc[2 + h], b2 = bsrc[4 + h], b3 = bsrc[6 + h];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ok[j]) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b0, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b2, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b3, acc[j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = an[j];
            };
            if (li == 0) {
                const int d = P.d;
                const int nck = (KG * 8 + FM_CHUNK - 1) / FM_CHUNK;
                for (int ck = 0; ck < nck; ++ck) {
                    {   // stage features [128 ck, 128 ck + 128) of the block's 32 rows: thread -> (row, 16-float segment)
                        const int row = tid >> 3, seg = tid & 7;
                        const int qi = s_q[row];
                        const int k0 = ck * FM_CHUNK + seg * 16;
                        float* dst = chunk + row * FM_CHUNK_S + seg * 16;
                        const float* src = P.x + (size_t)(qi < 0 ? 0 : qi) * d + k0;
                        if (qi >= 0 && k0 + 16 <= d && (d & 3) == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
                                dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; ++i) dst[i] = (qi >= 0 && k0 + i < d) ? src[i] : 0.0f;
                        }
                    }
                    __syncthreads();
                    const int g_end = min(KG, (ck + 1) * (FM_CHUNK / 8));
                    for (int g = ck * (FM_CHUNK / 8); g < g_end; ++g)
                        group(chunk + c * FM_CHUNK_S + (g - ck * (FM_CHUNK / 8)) * 8, g);
                    __syncthreads();
                }
            } else {
                for (int g = 0; g < KG; ++g) group(in + c * Si + g * 8, g);
            }
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
    const int T = MODE == FM_TOPK ? P.nb : L;
    for (int col = w; col < ncols; col += 4) {
        const int q = s_q[col];
        const float* l = lg + col * SL;
        int base = 0;
        if (MODE == FM_NAV) base = P.pq_len[q];
        float pv = INFINITY;
        int pi = -1;
        for (int t = 0; t < T; ++t) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int j = lane; j < L; j += 64) {
                const float v = l[j];
                const bool after = (v < pv) || (v == pv && j > pi);
                if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) {
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
                        if (pos < P.cap) {
                            P.pq_prob[(size_t)q * P.cap + pos] = pr;
                            P.pq_ent[(size_t)q * P.cap + pos] = cls >= 0 ? P.child_offset[model] + cls : -1;
                        }
                    }
                }
            }
            pv = bv;
            pi = bi;
        }
        if (MODE == FM_NAV && lane == 0) P.pq_len[q] = min(P.cap, base + L);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-level walk, one step (LearnedIndex.py:234-250: `pq.pop` for every unfinished query, then
// `_visit_internal_nodes` / `_visit_buckets`).  One thread per query pops the most probable entry of its
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
    int* active;              // queries that popped something and are still short of nb buckets
};

__global__ __launch_bounds__(256) void nav_pop_kernel(NavParams P) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P.nq) return;
    int have = P.out_len[q];
    if (have >= P.nb) return;
    const float* pp = P.pq_prob + (size_t)q * P.cap;
    int* pe = P.pq_ent + (size_t)q * P.cap;
    const int len = P.pq_len[q];
    float best = 0.0f;
    int bi = -1;
    for (int i = 0; i < len; ++i) {
        if (pe[i] < 0) continue;
        const float v = pp[i];
        if (bi < 0 || v >= best) { best = v; bi = i; }  // >=: the later entry wins a tie
    }
    if (bi < 0) return;  // queue exhausted: the remaining slots stay EMPTY (the reference would fail here)
    const int ent = pe[bi];
    pe[bi] = -1;
    const int cm = P.child_model[ent], cb = P.child_bucket[ent];
    if (cm >= 0) {
        const int pos = atomicAdd(&P.node_count[cm], 1);
        P.col_query[(size_t)cm * P.nq + pos] = q;
    } else if (cb >= -1) {
        P.out_slab[(size_t)q * P.nb + have] = cb;
        P.out_ent[(size_t)q * P.nb + have] = ent;
        P.out_len[q] = ++have;
    }
    if (have < P.nb) atomicAdd(P.active, 1);  // (hipcc folds a wave's increments into one atomic)
}

}  // namespace lmi
