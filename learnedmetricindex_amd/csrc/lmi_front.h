// lmi_front.h -- the per-batch preparation of the prefilter path as TWO lean launches (gfx950).
//
// Until round 4 a batch was prepared by eight launches on the critical path in front of pass 1 (fill_ranges, route_count,
// route_scan, route_group on a side stream, route_fill, query_norm, pack_queries16, slot_bound: 57 us at the C5 shape, 79 us at C2's,
// each kernel waiting for a grid-wide result of the one before).  Now:
//
//   route_kernel<NB>   a block per bucket, no communication between blocks.  The block walks the whole bucket_order array (40 000
//                      slots = 160 KB, from L2; a wave's ids of several 64-query steps loaded at once) and counts the slots routed to
//                      ITS bucket, primary and other (RouteArrays) -- ballots, no atomics; publishes the count as a tagged 8-byte granule
//                      and sums the granules of the LOWER buckets (they are dispatched first and wait for nobody) into the bucket's
//                      first col-block -- the one grid-wide quantity routing needs, without a launch boundary, a contended atomic or
//                      anything to reset; then hands every slot of the bucket its column in a FIXED order (wave's query range, step,
//                      rank, lane; from the waves' LDS lists of the slots found, or by a second walk when a list ran over):
//                      slot_col[], colmap[]; writes m[b], m0[b], cb_start[b].
//   pack_kernel<..>    a block per col-block (32 columns): each column's query row is read ONCE, coalesced (8..64 lanes per row, all of
//                      a wave's rows in flight together): max |q| -> the query's power-of-two scale -> fp16 image -> ||q'||, ||q^ - q'||
//                      -> eps' of the slot (the arithmetic of query_norm_kernel / slot_bound_kernel, lmi_prefilter.h); the fp16 values go
//                      through an LDS image of the col-block's fragments and leave as whole 1-KiB fragments; the block also sets its
//                      columns' pass-1 lists to -inf.  Block 0 builds the XCD-affine work queues + statistics from m / m0
//                      (route_group_body: route_group_kernel's code); all blocks share the small per-call fills (the old fill_ranges
//                      list) and the marking of unvisited slots.
//
// A first form (one launch: every block built the batch's full routing histogram in LDS to learn its bucket's col-block prefix, then
// packed its bucket's columns) was correct but slower than the eight launches: 40 000 LDS atomics per block (~10 us), one fat kernel
// with 145 spilled registers, one block per CU (profiles/r05_front_experiments.txt).  Used when the fan-out and the batch are moderate
// (FR_MAX_L buckets, FR_MAX_SLOTS slots, FR_MAX_D dims); beyond, scan_enqueue keeps the separate kernels.  Both routes give every slot a
// private column of its bucket -- which column is irrelevant to the results (tests/test_gpu_front.py runs both).
#pragma once
#include "lmi_prefilter.h"

namespace lmi {

constexpr int FR_THREADS = 1024, FR_WAVES = FR_THREADS / 64;   // route_kernel / the queue block
constexpr int FR_MAX_L = 512;            // buckets: one 1 024-thread block per bucket, each walking the whole batch and reading the lower buckets' counts
                                         // (2 000 leaves: the separate kernels' 115 us are faster)
constexpr int FR_MAX_SLOTS = 1 << 17;    // slots per batch: every block of route_kernel walks all of them twice
constexpr int FR_SLICE_G = 48;           // k16-groups of a col-block staged at a time (48 KiB: d <= 768 in one slice)
constexpr int FR_LIST = 512;             // slots of its bucket a wave of route_kernel keeps in LDS (more: the bucket's block walks a second time)
constexpr int FP_THREADS = 512, FP_WAVES = FP_THREADS / 64;    // pack_kernel
constexpr int FR_MAX_D = 64 * 4 * 8;     // a row's chunks fit a wave's registers (4 per lane)
static_assert(FR_SLICE_G % 2 == 0, "the 16 x 32 fragment shape pairs k16-groups");

struct FrontParams {
    const int* bucket_order;   // [nq][nb]
    int nq, nb, L;
    const unsigned* epoch_dev; // this call's tag of the granules below: a DEVICE word (never 0; stale entries of earlier calls never match),
                               // bumped by the launch behind pack_kernel's consumers (bound_merge2_kernel) -- not a kernel argument, which a
                               // graph replay of the call would freeze
    unsigned long long* gran;  // [L + 1] {epoch << 32 | value}: [b] = queries routed to bucket b, published by its block with ONE 8-byte store
                               // (the data is the flag); [L] = the batch's col-blocks, published by the last bucket's block
    int* cb_bucket;            // [col-blocks] the bucket a col-block belongs to
    int* colmap;               // [columns] the query of a column (-1: idle)
    RouteArrays R;             // out: m, m0, cb_start, stats, grp_* (item_base / part_base: exact mode only, not written)
    FillRanges Z;              // the small per-call arrays (pack_kernel's blocks share them)
    const float* q;            // [nq][d] row-major queries of the scan
    int d, KG16, f16x16;
    float* qnorm; float* qdelta; float* qscale;   // [nq] (every block that packs a query's slot writes the same values)
    uint4* qfrag16;
    int* slot_col;             // [nq * nb]
    float* eps2;               // [columns]
    const unsigned* bnorm; const unsigned* bdelta;   // [L] bits of the buckets' largest ||x'|| / ||x^ - x'||
    float* pf_bound;           // [bound_rows][ncols] pass-1 lists
    long long ncols;
    int bound_rows;
    unsigned long long* ts;    // nullable: device time stamp of the first launch's start (lmi_set_timing 2)
    unsigned long long* dbg;   // nullable (LMI_FR_DEBUG=1 in the environment): clock stamps (tools/front_phases.py)
};
#define FR_DBG(base, slot) do { if (P.dbg && threadIdx.x == 0 && (base) >= 0) P.dbg[(base) + (slot)] = wall_clock64(); } while (0)

// dynamic LDS of route_kernel: the bucket sizes; of pack_kernel's col-block blocks: the fragment image; of its queue block: sizes + sort
__host__ __device__ inline int fr_lpad(int L) { return (L + 3) & ~3; }
__host__ __device__ inline size_t fr_route_lds(int L) { return (size_t)fr_lpad(L) * 4; }
__host__ __device__ inline size_t fr_pack_lds(int L, int KG16) {
    const size_t pack = (size_t)(KG16 < FR_SLICE_G ? KG16 : FR_SLICE_G) * 1024, grp = (size_t)fr_lpad(L) * 12 + route_group_lds(L) + 16;
    return pack > grp ? pack : grp;
}

// 8 consecutive floats of a query row starting at k0, zeros past the row's end or when !ok -- BRANCH-FREE (clamped addresses + selects): a
// wave's loads of several rows / chunks then issue back to back and are waited for once (with a branch per load hipcc waits per load).
// VEC: d % 8 == 0 (16-byte loads).
template <bool VEC>
__device__ __forceinline__ void fr_load8(const float* __restrict__ row, int d, int k0, bool ok, float (&v)[8]) {
    if constexpr (VEC) {
        const int kk = ok ? k0 : 0;
        const float4 lo = *reinterpret_cast<const float4*>(row + kk), hi = *reinterpret_cast<const float4*>(row + kk + 4);
        v[0] = ok ? lo.x : 0.0f; v[1] = ok ? lo.y : 0.0f; v[2] = ok ? lo.z : 0.0f; v[3] = ok ? lo.w : 0.0f;
        v[4] = ok ? hi.x : 0.0f; v[5] = ok ? hi.y : 0.0f; v[6] = ok ? hi.z : 0.0f; v[7] = ok ? hi.w : 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = ok && k0 + j < d;
            const float x = row[in ? k0 + j : 0];
            v[j] = in ? x : 0.0f;
        }
    }
}

// the bucket ids of U consecutive 64-query steps of a wave (NB ranks each) and the sizes of those buckets: every global load issued
// before the first use (clamped addresses, selects: no branch between them), then every LDS lookup, each waited for once
template <int NB, int U>
struct FrChunk {
    int id[U][NB];
    int rows[U][NB];
    __device__ __forceinline__ void load(const int* __restrict__ bo, int q_first, int lane, int q_end) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q_first + 64 * u + lane;
            const int qq = min(q, q_end - 1);
#pragma unroll
            for (int r = 0; r < NB; ++r) id[u][r] = bo[(size_t)qq * NB + r];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = q_first + 64 * u + lane < q_end;
#pragma unroll
            for (int r = 0; r < NB; ++r) id[u][r] = live ? id[u][r] : -1;
        }
    }
    __device__ __forceinline__ void sizes(const int* nbr, int L) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const bool inr = id[u][r] >= 0 && id[u][r] < L;
                const int x = nbr[inr ? id[u][r] : 0];
                rows[u][r] = inr ? x : 0;
            }
    }
};

// One walk over the whole batch's bucket_order by the block of bucket b, in the FIXED order (wave's query range, 64-query step, rank,
// lane).  COUNT: the wave's primary / other slots of the bucket -> own0 / own1.  !COUNT: every slot of the bucket gets its column
// (c0 / c1: the wave's first primary / other position; n0: the bucket's primary slots) -> slot_col[], colmap[].
// COUNT also records the bucket's slots in discovery order into the wave's LDS list `wl` (slot | other << 31; n_list counts them, past
// `cap` too): the caller then places them from the list and needs no second walk unless a wave's list ran over.
template <int NB, bool COUNT>
__device__ __forceinline__ void fr_walk(const FrontParams& P, const int* nbr, int b, int nbv, int lane, int q0w, int q1w, int& c0, int& c1, int n0, int col_base,
                                        unsigned* wl = nullptr, int cap = 0, int* n_list = nullptr) {
    const int L = P.L;
    const bool use_primary = P.R.primary_nb > 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nl = 0;
    auto slot = [&](int q, int r, int br, int rows, bool big) __attribute__((always_inline)) {
        const bool other = use_primary && big;
        const bool mine = rows > 0 && br == b;
        const unsigned long long bal0 = __ballot(mine && !other), bal1 = __ballot(mine && other);
        if (COUNT && wl) {
            if (mine) {
                const int e = nl + (int)__popcll((bal0 | bal1) & lt);
                if (e < cap) wl[e] = (unsigned)(q * nbv + r) | (other ? 0x80000000u : 0u);
            }
            nl += (int)__popcll(bal0 | bal1);
        }
        if (!COUNT && mine) {
            const int cpos = other ? n0 + c1 + (int)__popcll(bal1 & lt) : c0 + (int)__popcll(bal0 & lt);
            P.slot_col[(size_t)q * nbv + r] = col_base + cpos;
            P.colmap[col_base + cpos] = q;
        }
        c0 += (int)__popcll(bal0);
        c1 += (int)__popcll(bal1);
    };
    if constexpr (NB > 0) {
        constexpr int U = NB <= 4 ? 4 : NB <= 8 ? 2 : 1;
        for (int qs = q0w; qs < q1w; qs += 64 * U) {
            FrChunk<NB, U> C;
            C.load(P.bucket_order, qs, lane, q1w);
            C.sizes(nbr, L);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bool big = false;   // a lower rank of this query holds a bucket of >= 64 rows: the slot is not primary (route_count_kernel)
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    slot(qs + 64 * u + lane, r, C.id[u][r], C.rows[u][r], big);
                    big = big || C.rows[u][r] >= 64;
                }
            }
        }
    } else {   // any rank count: one step at a time
        for (int qs = q0w; qs < q1w; qs += 64) {
            const int q = qs + lane;
            bool big = false;
            for (int r = 0; r < nbv; ++r) {
                const int br = q < q1w ? P.bucket_order[(size_t)q * nbv + r] : -1;
                const bool inr = br >= 0 && br < L;
                const int rows = inr ? nbr[br] : 0;
                slot(q, r, br, rows, big);
                big = big || rows >= 64;
            }
        }
    }
    if (COUNT && n_list) *n_list = nl;
}

// ---- launch 1: a block per bucket -- which slots go to the bucket, in which columns --------------------------------------------
template <int NB>
__global__ __launch_bounds__(FR_THREADS) void route_kernel(FrontParams P) {
    extern __shared__ __attribute__((aligned(16))) char fr_smem[];
    __shared__ int misc[2 * FR_WAVES + 2];
    __shared__ unsigned wlist[FR_WAVES][FR_LIST];   // a wave's slots of the bucket, in discovery order
    __shared__ int over_s;
    const int L = P.L, nq = P.nq, nb = NB > 0 ? NB : P.nb;
    const unsigned epoch_v = *P.epoch_dev;
    int* nbr = reinterpret_cast<int*>(fr_smem);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (b == 0 && tid == 0 && P.ts) *P.ts = wall_clock64();
    const int dbg_base = b == 0 ? 8 : b == (int)gridDim.x - 1 ? 16 : -1;
    FR_DBG(dbg_base, 0);
    if (P.R.nb_rows[b] == 0 && b != L - 1) {   // an empty (or unowned) bucket: nothing is routed to it (the last bucket's block stays: it publishes the total)
        if (tid == 0) {
            P.R.m[b] = 0; P.R.m0[b] = 0; P.R.cb_start[b] = 0;
            __hip_atomic_store(P.gran + b, (unsigned long long)epoch_v << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    for (int i = tid; i < L; i += FR_THREADS) nbr[i] = P.R.nb_rows[i];
    __syncthreads();
    const int Qw = (((nq + FR_WAVES - 1) / FR_WAVES) + 63) / 64 * 64;
    const int q0w = w * Qw, q1w = min(nq, q0w + Qw);
    int own0 = 0, own1 = 0, n_mine = 0;   // wave-uniform
    if (tid == 0) over_s = 0;
    fr_walk<NB, true>(P, nbr, b, nb, lane, q0w, q1w, own0, own1, 0, 0, wlist[w], FR_LIST, &n_mine);
    if (lane == 0) { misc[w] = own0; misc[FR_WAVES + w] = own1; }
    __syncthreads();
    if (lane == 0 && n_mine > FR_LIST) over_s = 1;   // (after the barrier that orders thread 0's reset; read behind the next one)
    FR_DBG(dbg_base, 1);
    int base0 = 0, base1 = 0, n0 = 0, n1 = 0;
    for (int i = 0; i < FR_WAVES; ++i) {
        const int a = misc[i], c = misc[FR_WAVES + i];
        if (i < w) { base0 += a; base1 += c; }
        n0 += a; n1 += c;
    }
    const int m_b = n0 + n1, ncb = (m_b + 31) >> 5;
    // The bucket's first col-block = the col-blocks of the buckets before it: every block PUBLISHES its count as a tagged 8-byte granule
    // (relaxed agent-scope store: no fence, nothing to reset -- a stale tag never matches) and one wave sweeps the granules of the LOWER
    // buckets until all carry this call's tag.  Lower-indexed blocks are dispatched first and depend on nobody, so the sweep ends; it
    // is bounded anyway (a give-up leaves a wrong layout and a word in stats[3] for the host to see).
    if (tid == 0) {
        __hip_atomic_store(P.gran + b, ((unsigned long long)epoch_v << 32) | (unsigned long long)(unsigned)m_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        P.R.m[b] = m_b;
        P.R.m0[b] = n0;
    }
    if (w == 0) {
        int sum = 0;
        for (int i0 = 0; i0 < b; i0 += 64) {
            const int i = i0 + lane;
            unsigned long long x = (unsigned long long)epoch_v << 32;
            for (unsigned spins = 0;; ++spins) {
                if (i < b) x = __hip_atomic_load(P.gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(x >> 32) == epoch_v)) break;
                if (spins > (1u << 22)) { if (lane == 0) P.R.stats[3] = -1; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            sum += ((int)(unsigned)x + 31) >> 5;   // (lanes past b carry 0)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) {
            misc[2 * FR_WAVES] = sum;
            P.R.cb_start[b] = sum;
            if (b == L - 1) __hip_atomic_store(P.gran + L, ((unsigned long long)epoch_v << 32) | (unsigned long long)(unsigned)(sum + ncb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (m_b == 0) return;
    const int cbs = misc[2 * FR_WAVES];
    for (int i = tid; i < ncb; i += FR_THREADS) P.cb_bucket[cbs + i] = b;
    for (int i = m_b + tid; i < ncb * 32; i += FR_THREADS) P.colmap[cbs * 32 + i] = -1;   // the idle columns of the last col-block
    int c0 = base0, c1 = base1;
    if (over_s) {   // a wave found more slots than its list holds (a skewed batch): the placing walk
        fr_walk<NB, false>(P, nbr, b, nb, lane, q0w, q1w, c0, c1, n0, cbs * 32);
    } else {        // from the lists: the same order, hence the same columns, as the placing walk
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int e0 = 0; e0 < n_mine; e0 += 64) {
            const int e = e0 + lane;
            const unsigned ent = e < n_mine ? wlist[w][e] : 0u;
            const bool live = e < n_mine, other = (ent >> 31) != 0u;
            const unsigned long long bal0 = __ballot(live && !other), bal1 = __ballot(live && other);
            if (live) {
                const int p = (int)(ent & 0x7fffffffu);
                const int cpos = other ? n0 + c1 + (int)__popcll(bal1 & lt) : c0 + (int)__popcll(bal0 & lt);
                P.slot_col[p] = cbs * 32 + cpos;
                P.colmap[cbs * 32 + cpos] = p / nb;
            }
            c0 += (int)__popcll(bal0);
            c1 += (int)__popcll(bal1);
        }
    }
    FR_DBG(dbg_base, 2);
}

// ---- launch 2: a block per col-block -- rows -> scale, norms, eps', fp16 fragments, pass-1 lists; block 0: the work queues -------
// GS lanes per query row (a power of two >= the row's 8-float chunks, capped at the wave); CP chunks per lane and row: a row is read
// ONCE (a wave's rows of a batch x CP chunks in flight together) and stays in registers through the k-slices.
template <int GS, int CP, bool VEC>
__global__ __launch_bounds__(FP_THREADS) void pack_kernel(FrontParams P) {
    extern __shared__ __attribute__((aligned(16))) char fp_smem[];
    __shared__ int active_s;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const unsigned epoch_v = *P.epoch_dev;
    // ---- the small per-call arrays (the old fill_ranges_kernel list): every block its share, before anything can make it leave ----
    for (int r = 0; r < P.Z.count; ++r)
        for (long long i = (long long)bid * FP_THREADS + tid; i < P.Z.n[r]; i += (long long)gridDim.x * FP_THREADS) P.Z.p[r][i] = P.Z.v[r];
    // ---- unvisited slots (LearnedIndex.py:340-341: a bucket id outside the index, or a bucket without rows here): every block its share
    //      (one block walking all 40 000 slots, two dependent loads each, was 40 us: the launch's long pole) ----
    for (long long p = (long long)bid * FP_THREADS + tid; p < (long long)P.nq * P.nb; p += (long long)gridDim.x * FP_THREADS) {
        const int br = P.bucket_order[p];
        if (!(br >= 0 && br < P.L && P.R.nb_rows[br] > 0)) P.slot_col[p] = -1;
    }
    if (bid == 0) {
        // ---- the XCD-affine work queues + statistics from route_kernel's m / m0 (1 024-thread body: two passes of this block) ----
        FR_DBG(0, 0);
        const int L = P.L;
        int* nbr = reinterpret_cast<int*>(fp_smem);
        int* m_s = nbr + fr_lpad(L);
        int* m0_s = m_s + fr_lpad(L);
        char* sort_s = reinterpret_cast<char*>(m0_s + fr_lpad(L));
        long long pairs = 0, items = 0;
        for (int i = tid; i < L; i += FP_THREADS) {
            const int rows = P.R.nb_rows[i], m = P.R.m[i];
            nbr[i] = rows; m_s[i] = m; m0_s[i] = P.R.m0[i];
            pairs += (long long)m * rows;
            items += (long long)query_tiles(m, P.R.tile_cb) * P.R.nch[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pairs += __shfl_xor(pairs, o); items += __shfl_xor(items, o); }
        __shared__ long long red[2][FP_WAVES];
        if (lane == 0) { red[0][w] = pairs; red[1][w] = items; }
        __syncthreads();
        if (tid == 0) {
            long long ps = 0, is = 0;
            for (int i = 0; i < FP_WAVES; ++i) { ps += red[0][i]; is += red[1][i]; }
            P.R.stats[0] = ps;
            P.R.stats[1] = is;
        }
        FR_DBG(0, 1);
        RouteArrays R2 = P.R;
        R2.m = m_s;
        R2.m0 = m0_s;
        R2.nb_rows = nbr;
        route_group_body<false, FP_THREADS>(L, R2, sort_s, &active_s);
        FR_DBG(0, 4);
        return;
    }
    const unsigned long long alloc = P.gran[P.L];   // (route_kernel is complete: the launch boundary orders it)
    const int n_cb = (unsigned)(alloc >> 32) == epoch_v ? (int)(unsigned)alloc : 0;
    const int cb = bid - 1;
    if (cb >= n_cb) return;
    char* stage = fp_smem;
    const int b = P.cb_bucket[cb];
    const int d = P.d, KG16 = P.KG16;
    const int nchunk_row = (d + 7) >> 3;               // chunks that hold data; the slab pads K to 16 KG16
    constexpr int CPW = 64 / GS;                       // rows per wave instruction
    constexpr int NCOL = (32 + FP_WAVES * CPW - 1) / (FP_WAVES * CPW);   // rows per wave and col-block (4 at GS = 64)
    const int gl = lane % GS, sub = lane / GS;
    const float xn = __uint_as_float(P.bnorm[b]), dx = __uint_as_float(P.bdelta[b]);
    const float guard = norm_guard(d);
    // The pass-1 lists of the live columns -> -inf: [bound_rows] rows of n_cb * 32 floats.  Block cb fills the cb-th 32-KiB piece of that
    // (row-major) region -- CONTIGUOUS memory, not its own columns' 128-byte pieces of every row (48 MB in 128-byte pieces 200 KB apart
    // ran at 1.4 TB/s: most of the launch).  Stores only: they leave while the rows are on their way.
    {
        const float ninf = -INFINITY;
        const float4 f4 = make_float4(ninf, ninf, ninf, ninf);
        const long long per_row = (long long)n_cb * 8;   // float4 per row
        const long long total = (long long)P.bound_rows * per_row;
        const long long lo = (long long)cb * P.bound_rows * 8, hi = min(total, lo + (long long)P.bound_rows * 8);
        for (long long i = lo + tid; i < hi; i += FP_THREADS) {
            const long long row = i / per_row, x = i - row * per_row;
            *reinterpret_cast<float4*>(P.pf_bound + (size_t)row * (size_t)P.ncols + (size_t)x * 4) = f4;
        }
    }
    float v[NCOL][CP][8];
    int qi[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const int cc = w * CPW + sub + c * FP_WAVES * CPW;
        qi[c] = cc < 32 ? P.colmap[(size_t)cb * 32 + cc] : -1;
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const float* row = P.q + (size_t)(qi[c] < 0 ? 0 : qi[c]) * d;
#pragma unroll
        for (int i = 0; i < CP; ++i) {
            const int j = gl + GS * i;
            fr_load8<VEC>(row, d, 8 * j, qi[c] >= 0 && j < nchunk_row, v[c][i]);
        }
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const int cc = w * CPW + sub + c * FP_WAVES * CPW;
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < CP; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) mx = fmaxf(mx, fabsf(v[c][i][t]));
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float s = scale_of_max(__float_as_uint(mx));   // (an idle column: 1)
        float acc = 0.0f, dl = 0.0f;
#pragma unroll
        for (int i = 0; i < CP; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float vs = v[c][i][t] * s;
                const float e = (float)(_Float16)vs - vs;
                acc += vs * vs;
                dl += e * e;
                v[c][i][t] = vs;
            }
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); dl += __shfl_xor(dl, o); }
        if (gl == 0 && cc < 32) {
            const size_t col = (size_t)cb * 32 + cc;
            if (qi[c] >= 0) {
                const float qn = sqrtf(acc) * guard, dq = sqrtf(dl) * guard;
                P.qnorm[qi[c]] = qn; P.qdelta[qi[c]] = dq; P.qscale[qi[c]] = s;
                // slot_bound_kernel's bound (lmi_prefilter.h): Cauchy-Schwarz on the measured norms + the two binary32 summations
                const float e = dq * (xn + dx) + qn * dx + 4.0f * (float)(KG16 * 16) * 5.96046448e-8f * (qn + dq) * (xn + dx);
                P.eps2[col] = 2.0f * e * 1.001f;
            } else {
                P.eps2[col] = 0.0f;   // idle column of the bucket's last col-block (never tested: its threshold is +inf)
            }
        }
    }
    for (int g0 = 0; g0 < KG16; g0 += FR_SLICE_G) {
        const int gs = min(FR_SLICE_G, KG16 - g0);
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int cc = w * CPW + sub + c * FP_WAVES * CPW;
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int j = gl + GS * i;
                const int g = j >> 1, hh = j & 1;
                if (cc < 32 && g >= g0 && g < g0 + gs) {
                    half8 h;
#pragma unroll
                    for (int t = 0; t < 8; ++t) h[t] = (_Float16)v[c][i][t];
                    int slot;
                    if (P.f16x16) {   // convert16_kernel's K > 128 shape; the LDS image is XOR-swizzled in 16-slot rows: a row's 64 chunks would
                                      // otherwise all fall into ONE 16-byte bank group (a 64-way conflict per ds_write_b128)
                        slot = (2 * (g >> 1) + (cc >> 4) - g0) * 64 + 16 * (2 * (g & 1) + hh) + (cc & 15);
                        slot ^= ((slot >> 4) & 3) | (((slot >> 7) & 3) << 2);
                    } else {
                        slot = (g - g0) * 64 + hh * 32 + cc;
                    }
                    *reinterpret_cast<uint4*>(stage + (size_t)slot * 16) = *reinterpret_cast<uint4*>(&h);
                }
            }
        }
        __syncthreads();
        uint4* dst = P.qfrag16 + ((size_t)cb * (size_t)KG16 + (size_t)g0) * 64;
        for (int i = tid; i < gs * 64; i += FP_THREADS) {
            const int src = P.f16x16 ? (i ^ (((i >> 4) & 3) | (((i >> 7) & 3) << 2))) : i;
            dst[i] = *reinterpret_cast<const uint4*>(stage + (size_t)src * 16);
        }
        __syncthreads();
    }
}
#undef FR_DBG

}  // namespace lmi
