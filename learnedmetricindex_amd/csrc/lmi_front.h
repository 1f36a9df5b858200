// lmi_front.h -- the per-batch preparation of the prefilter path as ONE launch (gfx950).
//
// Until round 4 a batch was prepared by eight launches on the critical path in front of pass 1 (fill_ranges, route_count,
// route_scan, route_group on a side stream, route_fill, query_norm, pack_queries16, slot_bound: ~60 us of runs and gaps at C5,
// ~100 us at C2, each kernel waiting for a grid-wide result of the one before).  front_kernel does all of it without any
// communication between workgroups -- nothing to publish, poll, re-initialise or order, and nothing a graph replay could freeze:
//
//   * EVERY block walks the whole bucket_order array (40 000 slots = 160 KB, from L2) and builds the full routing histogram in
//     LDS: m[b] / m0[b] (queries routed to bucket b; of them primary, RouteArrays) for every bucket.  From it a block knows the
//     col-block prefix cb_start of its own bucket -- the one grid-wide quantity routing needs -- by itself.
//   * block 0 writes the global copies (m, m0, cb_start, statistics), initialises the small per-call arrays (work-queue heads,
//     candidate counters, flags ..: the old fill_ranges list), marks unvisited slots and builds the XCD-affine work queues
//     (route_group_body: the same code as route_group_kernel, fed from the LDS histogram).
//   * block 1 + b * parts + j owns part j of bucket b's columns.  A second walk gives every slot routed to b its column: positions
//     are handed out in a FIXED order (wave's query range, 64-query step, rank, lane) from ballots, so every part of the bucket
//     (and any other launch) derives the same columns without talking to anyone; primary slots first, the others behind them.
//     Part 0 writes slot_col[].  Then, per owned col-block of 32 columns: each column's query row is read ONCE, coalesced (a group
//     of 8..64 lanes per row): max |q| -> the query's power-of-two scale -> fp16 image -> ||q'||, ||q^ - q'|| -> eps' of the slot
//     (the arithmetic of query_norm_kernel / slot_bound_kernel, lmi_prefilter.h) -- and the fp16 values go into an LDS image of
//     the col-block's fragments, written out as whole 1-KiB fragments.  The owner also sets its columns' pass-1 lists to -inf
//     (the one large fill of the old path: 1 KiB per column).
//
// The redundant walks cost each block ~5 us (LDS atomics); the launch replaces ~35 us of dependent launches at every shape and
// the query rows are read once per slot instead of twice.  Used when the fan-out and the batch are moderate (FR_MAX_L buckets,
// FR_MAX_SLOTS slots); beyond, scan_enqueue keeps the separate kernels.  Both routes give every slot a private column of its
// bucket -- which column is irrelevant to the results (tests/test_gpu_front.py runs both).
#pragma once
#include "lmi_prefilter.h"

namespace lmi {

constexpr int FR_THREADS = 1024, FR_WAVES = FR_THREADS / 64;
constexpr int FR_MAX_L = 2048;           // buckets: the histogram (8 L bytes), the bucket sizes (4 L) and the queue sort live in LDS
constexpr int FR_MAX_SLOTS = 1 << 17;    // slots per batch: every block walks all of them twice
constexpr int FR_CAPW = 1024;            // columns of a part handled per positioning walk (a larger part walks again per window)
constexpr int FR_SLICE_G = 48;           // k16-groups of a col-block staged at a time (48 KiB: d <= 768 in one slice)
constexpr int FR_MAX_PARTS = 8;
static_assert(FR_SLICE_G % 2 == 0, "the 16 x 32 fragment shape pairs k16-groups");

struct FrontParams {
    const int* bucket_order;   // [nq][nb]
    int nq, nb, L, parts;
    RouteArrays R;             // out: m, m0, cb_start, stats, grp_* (item_base / part_base: exact mode only, not written)
    FillRanges Z;              // the small per-call arrays block 0 initialises
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
    unsigned long long* ts;    // nullable: device time stamp of the launch's start (lmi_set_timing 2)
    unsigned long long* dbg;   // nullable (LMI_FR_DEBUG=1 in the environment): clock stamps of block 0 [0, 8), the first bucket block
                               // with queries [8, 16) and the last one [16, 24) at their phase boundaries (tools/front_phases.py)
};
#define FR_DBG(slot) do { if (P.dbg && tid == 0 && dbg_base >= 0) P.dbg[dbg_base + (slot)] = wall_clock64(); } while (0)

__host__ __device__ inline size_t fr_stage_bytes(int L, int KG16) {
    const size_t pack = (size_t)(KG16 < FR_SLICE_G ? KG16 : FR_SLICE_G) * 1024, grp = route_group_lds(L) + 16;
    return pack > grp ? pack : grp;
}
// dynamic LDS: bucket sizes [L] | histogram [2][L] | 64 words | column window [FR_CAPW] | staging
__host__ __device__ inline int fr_lpad(int L) { return (L + 3) & ~3; }   // (the staging area stays 16-byte aligned)
__host__ __device__ inline size_t fr_lds_bytes(int L, int KG16) { return (size_t)fr_lpad(L) * 12 + 256 + (size_t)FR_CAPW * 4 + fr_stage_bytes(L, KG16); }

// 8 consecutive floats of a query row starting at k0, zeros past the row's end or when !ok -- BRANCH-FREE (clamped addresses + selects): a
// wave's loads of several rows / chunks then issue back to back and are waited for once (with a branch per load hipcc waits per load:
// four serial round trips per col-block, round 5's first form: 8-12 us per col-block instead of ~3).  VEC: d % 8 == 0 (16-byte loads).
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

// One col-block (32 columns) of a bucket: rows -> scale, norms, eps', fp16 fragments (through the LDS image `stage`), pass-1 lists.
// GS lanes per query row (a power of two >= the row's 8-float chunks, capped at the wave); CP chunks per lane and row: a row is read
// ONCE, all of a wave's loads in flight together (a wave's NCOL rows x CP chunks), and stays in registers through the k-slices.
template <int GS, int CP, bool VEC>
__device__ __forceinline__ void fr_pack_colblock(const FrontParams& P, const int* __restrict__ colq32, int b, size_t cb_global,
                                                 char* stage, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int d = P.d, KG16 = P.KG16;
    const int nchunk_row = (d + 7) >> 3;               // chunks that hold data; the slab pads K to 16 KG16
    constexpr int CPW = 64 / GS;                       // rows per wave instruction
    constexpr int NCOL = (32 + FR_WAVES * CPW - 1) / (FR_WAVES * CPW);   // rows per wave and col-block (2 at GS = 64, else 1 or none)
    const int gl = lane % GS, sub = lane / GS;
    const float xn = __uint_as_float(P.bnorm[b]), dx = __uint_as_float(P.bdelta[b]);
    const float guard = norm_guard(d);
    float v[NCOL][CP][8];
    int qi[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const int cc = w * CPW + sub + c * FR_WAVES * CPW;
        qi[c] = cc < 32 ? colq32[cc] : -1;
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
        const int cc = w * CPW + sub + c * FR_WAVES * CPW;
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
            const size_t col = cb_global * 32 + cc;
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
            const int cc = w * CPW + sub + c * FR_WAVES * CPW;
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
        uint4* dst = P.qfrag16 + (cb_global * (size_t)KG16 + (size_t)g0) * 64;
        for (int i = tid; i < gs * 64; i += FR_THREADS) {
            const int src = P.f16x16 ? (i ^ (((i >> 4) & 3) | (((i >> 7) & 3) << 2))) : i;
            dst[i] = *reinterpret_cast<const uint4*>(stage + (size_t)src * 16);
        }
        __syncthreads();
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

template <int NB>
__global__ __launch_bounds__(FR_THREADS) void front_kernel(FrontParams P) {
    extern __shared__ __attribute__((aligned(16))) char fr_smem[];
    const int L = P.L, nq = P.nq;
    const int nb = NB > 0 ? NB : P.nb;
    int* nbr = reinterpret_cast<int*>(fr_smem);
    int* cnt = nbr + fr_lpad(L);    // [0, L): primary slots per bucket, [L, 2L): the others
    int* misc = cnt + 2 * fr_lpad(L);   // 64 words
    int* colq = misc + 64;          // [FR_CAPW] query of the window's columns (-1: idle)
    char* stage = reinterpret_cast<char*>(colq + FR_CAPW);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int b = bid == 0 ? -1 : (bid - 1) / P.parts, part = bid == 0 ? 0 : (bid - 1) % P.parts;
    if (bid == 0 && tid == 0 && P.ts) *P.ts = wall_clock64();
    // ---- the small per-call arrays (the old fill_ranges_kernel list): every block its share, before anything can make it leave ----
    for (int r = 0; r < P.Z.count; ++r)
        for (long long i = (long long)bid * FR_THREADS + tid; i < P.Z.n[r]; i += (long long)gridDim.x * FR_THREADS) P.Z.p[r][i] = P.Z.v[r];
    if (b >= 0 && P.R.nb_rows[b] == 0) return;   // an empty (or unowned) bucket: nothing is routed to it
    const int dbg_base = bid == 0 ? 0 : bid == 1 ? 8 : bid == (int)gridDim.x - 1 ? 16 : -1;
    FR_DBG(0);
    for (int i = tid; i < L; i += FR_THREADS) { nbr[i] = P.R.nb_rows[i]; cnt[i] = 0; cnt[L + i] = 0; }
    __syncthreads();
    // ---- walk A: the routing histogram of the whole batch; this bucket's slots per wave range ----
    const int Qw = (((nq + FR_WAVES - 1) / FR_WAVES) + 63) / 64 * 64;
    const int q0w = w * Qw, q1w = min(nq, q0w + Qw);
    const bool use_primary = P.R.primary_nb > 0;
    constexpr int NBC = NB > 0 ? NB : 1;
    constexpr int U = NB > 0 ? (NB <= 4 ? 4 : NB <= 8 ? 2 : 1) : 1;
    int own0 = 0, own1 = 0;   // wave-uniform
    if constexpr (NB > 0) {
        for (int qs = q0w; qs < q1w; qs += 64 * U) {
            FrChunk<NBC, U> C;
            C.load(P.bucket_order, qs, lane, q1w);
            C.sizes(nbr, L);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = qs + 64 * u + lane;
                bool big = false;   // a lower rank of this query holds a bucket of >= 64 rows: the slot is not primary (route_count_kernel)
#pragma unroll
                for (int r = 0; r < NBC; ++r) {
                    const int br = C.id[u][r], rows = C.rows[u][r];
                    const bool valid = rows > 0;
                    const bool other = use_primary && big;
                    if (valid) atomicAdd(&cnt[other ? L + br : br], 1);
                    const bool mine = valid && br == b;
                    own0 += (int)__popcll(__ballot(mine && !other));
                    own1 += (int)__popcll(__ballot(mine && other));
                    if (bid == 0 && q < q1w && !valid) P.slot_col[(size_t)q * NBC + r] = -1;   // unvisited (LearnedIndex.py:340-341)
                    big = big || rows >= 64;
                }
            }
        }
    } else {   // any rank count: one step at a time
        for (int qs = q0w; qs < q1w; qs += 64) {
            const int q = qs + lane;
            const bool live = q < q1w;
            bool big = false;
            for (int r = 0; r < nb; ++r) {
                const int br = live ? P.bucket_order[(size_t)q * nb + r] : -1;
                const bool inr = br >= 0 && br < L;
                const int rows = inr ? nbr[br] : 0;
                const bool valid = rows > 0;
                const bool other = use_primary && big;
                if (valid) atomicAdd(&cnt[other ? L + br : br], 1);
                const bool mine = valid && br == b;
                own0 += (int)__popcll(__ballot(mine && !other));
                own1 += (int)__popcll(__ballot(mine && other));
                if (bid == 0 && live && !valid) P.slot_col[(size_t)q * nb + r] = -1;
                big = big || rows >= 64;
            }
        }
    }
    if (lane == 0) { misc[w] = own0; misc[FR_WAVES + w] = own1; }
    __syncthreads();
    FR_DBG(1);

    if (bid == 0) {
        // ---- global copies of the histogram, the col-block prefix and the statistics ----
        static_assert(FR_MAX_L <= 2 * FR_THREADS, "two buckets per thread in the prefix");
        const int b0 = 2 * tid, b1 = 2 * tid + 1;
        const int m_0 = b0 < L ? cnt[b0] + cnt[L + b0] : 0, m_1 = b1 < L ? cnt[b1] + cnt[L + b1] : 0;
        const int c_0 = (m_0 + 31) >> 5, c_1 = (m_1 + 31) >> 5;
        long long pairs = 0, items = 0;
        if (b0 < L) { pairs += (long long)m_0 * nbr[b0]; items += (long long)query_tiles(m_0, P.R.tile_cb) * P.R.nch[b0]; }
        if (b1 < L) { pairs += (long long)m_1 * nbr[b1]; items += (long long)query_tiles(m_1, P.R.tile_cb) * P.R.nch[b1]; }
        int incl = c_0 + c_1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pairs += __shfl_xor(pairs, o); items += __shfl_xor(items, o); }
        long long* red = reinterpret_cast<long long*>(stage);   // [2][FR_WAVES]; the staging area is free until the queue sort
        if (lane == 63) misc[32 + w] = incl;
        if (lane == 0) { red[w] = pairs; red[FR_WAVES + w] = items; }
        __syncthreads();
        int wbase = 0, total = 0;
        for (int i = 0; i < FR_WAVES; ++i) { const int vv = misc[32 + i]; if (i < w) wbase += vv; total += vv; }
        const int ex = wbase + incl - (c_0 + c_1);
        if (b0 < L) { P.R.m[b0] = m_0; P.R.m0[b0] = cnt[b0]; P.R.cb_start[b0] = ex; }
        if (b1 < L) { P.R.m[b1] = m_1; P.R.m0[b1] = cnt[b1]; P.R.cb_start[b1] = ex + c_0; }
        if (tid == 0) {
            long long ps = 0, is = 0;
            for (int i = 0; i < FR_WAVES; ++i) { ps += red[i]; is += red[FR_WAVES + i]; }
            P.R.cb_start[L] = total;
            P.R.stats[0] = ps;
            P.R.stats[1] = is;
        }
        __syncthreads();
        FR_DBG(2);
        FR_DBG(3);
        // ---- the XCD-affine work queues from the LDS histogram: cnt[0, L) = m0 as it stands, cnt[L, 2L) <- m ----
        for (int i = tid; i < L; i += FR_THREADS) cnt[L + i] += cnt[i];
        __syncthreads();
        RouteArrays R2 = P.R;
        R2.m = cnt + L;
        R2.m0 = cnt;
        R2.nb_rows = nbr;
        route_group_body<false>(L, R2, stage, &misc[63]);
        FR_DBG(4);
        return;
    }

    // ---- a bucket's block: totals, this wave's bases, the bucket's first col-block ----
    int base0 = 0, base1 = 0, n0 = 0, n1 = 0;
    for (int i = 0; i < FR_WAVES; ++i) {
        const int a = misc[i], c = misc[FR_WAVES + i];
        if (i < w) { base0 += a; base1 += c; }
        n0 += a; n1 += c;
    }
    const int m_b = n0 + n1;
    if (m_b == 0) return;   // no query visits this bucket
    const int ncb = (m_b + 31) >> 5;
    const int per = (ncb + P.parts - 1) / P.parts;
    const int cb_lo = part * per, cb_hi = min(ncb, cb_lo + per);
    if (cb_lo >= cb_hi) return;
    int part_sum = 0;
    for (int i = tid; i < b; i += FR_THREADS) part_sum += (cnt[i] + cnt[L + i] + 31) >> 5;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part_sum += __shfl_xor(part_sum, o);
    if (lane == 0) misc[32 + w] = part_sum;
    __syncthreads();
    int cbs = 0;
    for (int i = 0; i < FR_WAVES; ++i) cbs += misc[32 + i];
    FR_DBG(2);
    // the pass-1 lists of the part's col-blocks: bound_rows rows of -inf, (cb_hi - cb_lo) * 128 contiguous bytes each (stores only:
    // they leave while the block walks on)
    {
        const float ninf = -INFINITY;
        const float4 f4 = make_float4(ninf, ninf, ninf, ninf);
        const int per_row = (cb_hi - cb_lo) * 8;   // float4 per row
        for (int i = tid; i < P.bound_rows * per_row; i += FR_THREADS) {
            const int row = i / per_row, x = i - row * per_row;
            *reinterpret_cast<float4*>(P.pf_bound + (size_t)row * (size_t)P.ncols + ((size_t)cbs + cb_lo) * 32 + (size_t)x * 4) = f4;
        }
    }

    for (int wlo = cb_lo * 32; wlo < cb_hi * 32; wlo += FR_CAPW) {
        const int whi = min(cb_hi * 32, wlo + FR_CAPW);
        for (int i = tid; i < FR_CAPW; i += FR_THREADS) colq[i] = -1;
        __syncthreads();
        // ---- walk B: the columns of this bucket's slots, in the fixed order (wave range, step, rank, lane) ----
        int c0 = base0, c1 = base1;
        const bool write_sc = part == 0 && wlo == 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        if constexpr (NB > 0) {
            for (int qs = q0w; qs < q1w; qs += 64 * U) {
                FrChunk<NBC, U> C;
                C.load(P.bucket_order, qs, lane, q1w);
                C.sizes(nbr, L);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = qs + 64 * u + lane;
                    bool big = false;
#pragma unroll
                    for (int r = 0; r < NBC; ++r) {
                        const int br = C.id[u][r], rows = C.rows[u][r];
                        const bool other = use_primary && big;
                        const bool mine = rows > 0 && br == b;
                        const unsigned long long bal0 = __ballot(mine && !other), bal1 = __ballot(mine && other);
                        if (mine) {
                            const int cpos = other ? n0 + c1 + (int)__popcll(bal1 & lt) : c0 + (int)__popcll(bal0 & lt);
                            if (write_sc) P.slot_col[(size_t)q * NBC + r] = cbs * 32 + cpos;
                            if (cpos >= wlo && cpos < whi) colq[cpos - wlo] = q;
                        }
                        c0 += (int)__popcll(bal0);
                        c1 += (int)__popcll(bal1);
                        big = big || rows >= 64;
                    }
                }
            }
        } else {
            for (int qs = q0w; qs < q1w; qs += 64) {
                const int q = qs + lane;
                const bool live = q < q1w;
                bool big = false;
                for (int r = 0; r < nb; ++r) {
                    const int br = live ? P.bucket_order[(size_t)q * nb + r] : -1;
                    const bool inr = br >= 0 && br < L;
                    const int rows = inr ? nbr[br] : 0;
                    const bool other = use_primary && big;
                    const bool mine = rows > 0 && br == b;
                    const unsigned long long bal0 = __ballot(mine && !other), bal1 = __ballot(mine && other);
                    if (mine) {
                        const int cpos = other ? n0 + c1 + (int)__popcll(bal1 & lt) : c0 + (int)__popcll(bal0 & lt);
                        if (write_sc) P.slot_col[(size_t)q * nb + r] = cbs * 32 + cpos;
                        if (cpos >= wlo && cpos < whi) colq[cpos - wlo] = q;
                    }
                    c0 += (int)__popcll(bal0);
                    c1 += (int)__popcll(bal1);
                    big = big || rows >= 64;
                }
            }
        }
        __syncthreads();
        FR_DBG(3);
        const int nchunk = (P.d + 7) >> 3;
        const bool vec = (P.d & 7) == 0;
        for (int cbi = wlo / 32; cbi * 32 < whi; ++cbi) {
            const int* colq32 = colq + (cbi * 32 - wlo);
            const size_t cbg = (size_t)cbs + (size_t)cbi;
#define FR_PACK(GSV, CPV) { if (vec) fr_pack_colblock<GSV, CPV, true>(P, colq32, b, cbg, stage, tid); else fr_pack_colblock<GSV, CPV, false>(P, colq32, b, cbg, stage, tid); }
            if (nchunk <= 8) FR_PACK(8, 1)
            else if (nchunk <= 16) FR_PACK(16, 1)
            else if (nchunk <= 32) FR_PACK(32, 1)
            else if (nchunk <= 64) FR_PACK(64, 1)
            else if (nchunk <= 128) FR_PACK(64, 2)
            else FR_PACK(64, 4)
#undef FR_PACK
            if (cbi == wlo / 32) FR_DBG(4);
        }
        __syncthreads();
    }
    FR_DBG(5);
}
#undef FR_DBG
constexpr int FR_MAX_D = 64 * 4 * 8;   // a row's chunks fit a wave's registers (4 per lane)

}  // namespace lmi
