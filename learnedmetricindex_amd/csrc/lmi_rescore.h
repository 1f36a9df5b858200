// lmi_rescore.h -- exact re-rank of the prefilter's survivors, memory-efficient form (gfx950).
//
// select_rescore_kernel (lmi_prefilter.h) lets every lane pull ITS survivor's row straight from global memory,
// 128 bytes per step.  rocprofv3 (profiles/r02_rescore_pmc.txt): 25 M L2 requests of 64 bytes per launch for
// 1.36 GB, 86 % misses -- the DRAM sees isolated 64-byte accesses to random 3-KiB rows and delivers 2.2 TB/s;
// packing more chains into a wave changed nothing (the access shape, not the lane count, was the limit).
// Here the work is split:
//   select_kernel    one wave per (query, rank) slot: That by bisection on the candidates' keys, the survivor
//                    rows -> surv_row[slot][..] (the first half of select_rescore_kernel, at full occupancy);
//   rescore_kernel   one wave per G slots of ONE query (G | n_buckets, <= 4), one wave per SIMD.  The wave's
//                    survivors are processed 64 rows at a time (one lane per chain); their rows are STREAMED through LDS in
//                    chunks of 32 floats: one LDS-DMA piece moves 1 KiB = 128 contiguous bytes of eight rows (lane -> (row,
//                    16-byte segment), rows 144 bytes apart in LDS so that the chains' ds_read_b128 hit distinct
//                    banks), chunks c+1, c+2 are in flight while lane r runs the canonical chain acc = fmaf(q[k], x[k], acc)
//                    over chunk c of row r (q broadcast from LDS).  The wave owns its buffers: no barriers, the
//                    ring is ordered by the wave's own counted `s_waitcnt vmcnt`.  Then, per slot, the 10 best by
//                    (score desc, row asc) become the slot's rank list exactly as before.
// Used for every row pitch (d rounded up to 4: the f32 rows are zero-padded to 16-byte multiples) whose per-wave buffers fit the LDS (the
// host launches 4, 2 or 1 waves per block, rc_waves_for: d <= 1 126 / 8 700 / 28 000; round 4 -- until then d > 1 024 kept select_rescore_kernel,
// 2.2 x slower at d = 1 536 / 2 048).  Results are bit-identical
// (same chain, same selection rule); tests/test_gpu_prefilter.py runs both.
#pragma once
#include <type_traits>

#include "lmi_prefilter.h"

namespace lmi {

// A/B on MI355X at C2 (rows, chunk, depth -> re-rank phase, RC_KEEP 64): (32, 128, 2) 0.574 ms, (64, 64, 2) 0.497, (64, 32, 3) 0.496,
// (32, 64, 3) 0.553, (32, 32, 4) 0.579: a wave's ~48 survivors are ONE batch of 64 chains instead of 32 + 16
#ifndef LMI_RC_ROWS
#define LMI_RC_ROWS 64
#endif
#ifndef LMI_RC_CHUNK
#define LMI_RC_CHUNK 32
#endif
#ifndef LMI_RC_DEPTH
#define LMI_RC_DEPTH 3
#endif
#ifndef LMI_RC_KEEP
#define LMI_RC_KEEP 256
#endif
constexpr int RC_ROWS = LMI_RC_ROWS;              // chains per wave and batch (32 or 64)
constexpr int RC_CHUNK = LMI_RC_CHUNK;            // floats of a row per chunk (a multiple of 32)
constexpr int RC_DEPTH = LMI_RC_DEPTH;            // chunk buffers per wave: RC_DEPTH - 1 chunks in flight while one is chained
// Survivors re-scored per slot on this path; more -> the exact fallback (one block brute-forcing the bucket per slot: the
// cliff of duplicate-heavy data, tools/dup_cliff.py).  256 instead of the simple kernel's 64 lets clusters of up to ~200
// copies of a vector through; (rows 64, chunk 32, depth 3) holds the 4 x 256 row / score lists in the same LDS and runs
// at the speed of (64, 64, 2).
constexpr int RC_KEEP = LMI_RC_KEEP;
constexpr int RC_PITCH = RC_CHUNK * 4 + 16;       // bytes between rows in a chunk buffer (bank spread)
constexpr int RC_PIECES = (RC_ROWS * RC_PITCH + 1023) / 1024;  // LDS-DMA pieces per chunk (9 for 64 rows x 144 bytes)
constexpr int RC_BUF = RC_PIECES * 1024;          // a chunk buffer holds WHOLE pieces: the last one writes 512 bytes past the rows
constexpr int RC_WAVES = 4;

struct SelectOut {
    unsigned* surv_row;   // [nslots][RC_KEEP]
    int G;                // slots per rescore wave
    int* grp_flag;        // [groups] zeroed per call: the group has a slot with survivors
    int* active;          // the groups to re-score, compacted (a rank of a sharded index owns 1/world of the slots: its
                          // rescore waves are packed into the first blocks) as RC_SUB sub-lists: [RC_SUB] counts, then
                          // [RC_SUB][sub_cap] group numbers; group g goes to sub-list g % RC_SUB (10 000 returning atomics on
                          // ONE counter were most of select_kernel's 139 us)
    int sub_cap;          // ceil(groups / RC_SUB)
    int* big;             // [1 + groups] groups with more survivors than the small-LDS launch holds: count, then group numbers
};
constexpr int RC_SUB = 64;

// The part of select_kernel after the candidate count is known, for slots of up to 64 PERV candidates.  The bisection is
// issue-bound (32 steps x one ballot per 64 candidates): a slot emits ~160, so PERV = 4 does a quarter of the work of the
// general PERV = 16 (select_kernel: 140 -> 45 us at C2).
template <int PERV, int SPEC>
__device__ __forceinline__ void select_tail(const RescoreParams& P, const SelectOut& O, int p, int col, int lane, unsigned cnt,
                                            const float* __restrict__ cs, const unsigned* __restrict__ cr,
                                            const float (&s_spec)[SPEC], const unsigned (&r_spec)[SPEC]) {
    const int nper = (int)((cnt + 63u) >> 6);
    // needed after the bisection, requested before it (two dependent loads: they would be a round trip of their own at the end)
    const float e2 = P.eps2[col];
    const int bkt = P.bucket_order[p];
    const unsigned row_base = (unsigned)P.rb_start[bkt] * 32u;
    unsigned key[PERV];  // monotone image of shat; 0 = no candidate
#pragma unroll
    for (int i = 0; i < PERV; ++i) {
        const int e = lane + 64 * i;
        key[i] = 0u;
        if (i < nper && e < (int)cnt) {
            const unsigned bits = __float_as_uint(i < SPEC ? s_spec[i < SPEC ? i : 0] : cs[e]);
            key[i] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
        }
    }
    float pv = -INFINITY;  // fewer than 10 candidates: everything survives
    if (cnt >= (unsigned)KPB) {
        unsigned T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned probe = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < PERV; ++i)
                if (PERV <= 4 || i < nper) c += (int)__popcll(__ballot(key[i] >= probe));  // empty entries (key 0) never count: probe >= 1
            if (c >= KPB) T = probe;
        }
        pv = __uint_as_float(T ^ ((T >> 31) ? 0x80000000u : 0xffffffffu));
    }
    const float cut = pv - e2;
    const unsigned cbits = __float_as_uint(cut);
    const unsigned kcut = cut != cut ? 1u : cbits ^ ((cbits >> 31) ? 0xffffffffu : 0x80000000u);
    unsigned* out = O.surv_row + (size_t)p * RC_KEEP;
    // survivors are stored as ABSOLUTE slab rows (bucket start + row): the re-rank wave then needs no bucket_order -> rb_start
    // round trips before it can request the rows (its waves run one per SIMD: every dependent load is exposed latency)
    unsigned nk = 0;
#pragma unroll
    for (int i = 0; i < PERV; ++i) {
        if (i < nper) {
            const bool keep = key[i] != 0u && key[i] >= kcut;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const unsigned k = nk + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                if (k < (unsigned)RC_KEEP) out[k] = row_base + (i < SPEC ? r_spec[i < SPEC ? i : 0] : cr[lane + 64 * i]);
            }
            nk += (unsigned)__popcll(bal);
        }
    }
    if (nk == 0u && lane < KPB) {
        // no candidate at all (a query-level bound above everything this bucket holds, query_bound_kernel): no rescore wave
        // may come by, so the slot's list is written here -- all padding (it sorts behind every real entry of the query)
        const int b = bkt;
        const int rb0 = P.rb_start[b], n_b = P.nb_rows[b];
        P.rank_d[(size_t)p * KPB + lane] = P.raw ? -3.402823466e+38f : pad_dist(P.qn2);
        P.rank_id[(size_t)p * KPB + lane] = P.raw ? NOROW : P.ids_slab[(size_t)rb0 * 32 + (n_b - 1)];
    }
    if (lane == 0) {
        if (nk > (unsigned)RC_KEEP) flag_fallback(P, p);
        else {
            P.nkeep[p] = (int)nk;
            if (nk > 0 && atomicExch(&O.grp_flag[p / O.G], 1) == 0) {
                const int g = p / O.G, sub = g % RC_SUB;
                O.active[RC_SUB + sub * O.sub_cap + atomicAdd(&O.active[sub], 1)] = g;
            }
        }
    }
}

// That + survivors of one slot; unvisited slots get their (inf, 0) rank list here.
__global__ __launch_bounds__(256) void select_kernel(RescoreParams P, SelectOut O) {
    select_stamps(P);
    pf_x_scatter(P, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wv;
    if (p >= P.nslots) return;
    const float FMAXV = 3.402823466e+38f;
    const int col = P.slot_col[p];
    if (lane == 0) { P.fallback[p] = 0; P.nkeep[p] = 0; }
    if (col < 0) {  // unvisited (LearnedIndex.py:340-341)
        if (lane < KPB) {
            P.rank_d[(size_t)p * KPB + lane] = P.raw ? -FMAXV : INFINITY;
            P.rank_id[(size_t)p * KPB + lane] = P.raw ? NOROW : 0u;
        }
        return;
    }
    const float* cs = P.cand_s + (size_t)col * PF_CAP;
    const unsigned* cr = P.cand_row + (size_t)col * PF_CAP;
    constexpr int PER = PF_CAP / 64;
    constexpr int SPEC = 4;  // the first 256 candidates (a slot emits ~160) are requested together with the count:
                             // three dependent memory round trips (column -> count -> scores -> rows) become two
    float s_spec[SPEC];
    unsigned r_spec[SPEC];
#pragma unroll
    for (int i = 0; i < SPEC; ++i) { s_spec[i] = cs[lane + 64 * i]; r_spec[i] = cr[lane + 64 * i]; }
    const unsigned cnt = P.cand_cnt[col];
    if (cnt > (unsigned)PF_CAP) {
        if (lane == 0) flag_fallback(P, p);
        return;
    }
    if (cnt <= 64u) select_tail<1, SPEC>(P, O, p, col, lane, cnt, cs, cr, s_spec, r_spec);   // (the columns behind a query's primary one: a handful each)
    else if (cnt <= 256u) select_tail<4, SPEC>(P, O, p, col, lane, cnt, cs, cr, s_spec, r_spec);
    else select_tail<PER, SPEC>(P, O, p, col, lane, cnt, cs, cr, s_spec, r_spec);
}

// Two launches share the code.  SMALL: every active group (query); a wave holds at most RC_SMALL_ROWS survivors -- since the bound is
// per query a wave has ~13 -- in 11.5 KiB of LDS (d = 768), three blocks per CU: the kernel is a chain of dependent loads at low
// occupancy, so more waves in flight is what it needs; a group with more survivors goes onto the `big` list and the second launch
// (!SMALL: one wave per SIMD, room for G x RC_KEEP survivors) takes those.
constexpr int RC_SMALL_ROWS = 28;                 // 4 LDS-DMA pieces of 7 rows
#ifndef LMI_RC_SMALL_RING
#define LMI_RC_SMALL_RING (8 * 1024)
#endif
constexpr int RC_SMALL_LDS_CAP = 64 * 1024 + RC_WAVES * (LMI_RC_SMALL_RING - 8 * 1024);   // dynamic LDS a block of the small-form kernels may ask for
constexpr int RC_SMALL_RING = LMI_RC_SMALL_RING;  // chunk buffers of the small form (4 x 2 pieces or 2 x 4 pieces)
// dynamic LDS per wave: chunk buffers | q [d] | rows [KEEPW] | scores [KEEPW]
__host__ __device__ inline int rc_wave_lds(int d, int G, bool small_form = false) {
    return small_form ? RC_SMALL_RING + d * 4 + 32 * 8 : RC_DEPTH * RC_BUF + d * 4 + G * RC_KEEP * 8;
}
// waves per block for rows of pitch d: as many of RC_WAVES as the big form's per-wave LDS allows (0: the shape does not fit at all)
__host__ inline int rc_waves_for(int d, int G) {
    for (int w = RC_WAVES; w >= 1; w >>= 1)
        if (w * rc_wave_lds(d, G) <= 160 * 1024 - 512) return w;
    return 0;
}
static_assert(RC_CHUNK % 32 == 0 && (RC_ROWS == 32 || RC_ROWS == 64) && RC_DEPTH >= 2 && RC_DEPTH <= 4, "rescore_kernel shapes");

// LDS of one re-rank wave: chunk ring | q [d] | rows [KEEPW] | scores [KEEPW]
template <int G, bool SMALL>
struct RcWave {
    static constexpr int RINGB = SMALL ? RC_SMALL_RING : RC_DEPTH * RC_BUF;
    static constexpr int KEEPW = SMALL ? 32 : G * RC_KEEP;
    unsigned char* mine;
    float* qs;
    unsigned* krow;
    float* ksc;
    __device__ __forceinline__ RcWave(unsigned char* base, int d) : mine(base), qs(reinterpret_cast<float*>(base + RINGB)),
        krow(reinterpret_cast<unsigned*>(base + RINGB + d * 4)), ksc(reinterpret_cast<float*>(base + RINGB + d * 4) + KEEPW) {}
    // the wave's query -> LDS (rows of pitch dp: the tail behind d is zero)
    __device__ __forceinline__ void stage_query(const RescoreParams& P, int q, int lane) const {
        const int d = P.dp;
        const float* qg = P.q + (size_t)q * P.d;
        if (P.d == d) {
            for (int k = lane * 4; k < d; k += 256) *reinterpret_cast<float4*>(qs + k) = *reinterpret_cast<const float4*>(qg + k);
        } else {   // d not a multiple of 4: the query's rows are not 16-byte aligned, its tail is padded with zeros here
            for (int k = lane; k < d; k += 64) qs[k] = k < P.d ? qg[k] : 0.0f;
        }
    }
};

// One batch of <= RC_ROWS chains: rows krow[base ..] (absolute slab rows, in LDS) streamed through the wave's chunk ring `mine`, lane r runs the
// canonical chain of row base + r against the query at LDS byte address `qaddr`; scores -> ksc[base ..].  NP = LDS-DMA pieces per chunk that
// hold rows of THIS batch (a wave re-scores ~13 rows since the bound is per query: 2 pieces instead of the 9 that cover 64 rows -- issuing the
// unused ones was most of the kernel's time).  RINGB: bytes of the ring.
template <int NP, int RINGB>
__device__ __forceinline__ void rc_run_batch(const RescoreParams& P, unsigned char* mine, const unsigned* krow, float* ksc, unsigned qaddr,
                                             int base, int total, int lane) {
    const int d = P.dp;
    const int nchunks = (d + RC_CHUNK - 1) / RC_CHUNK;
        // the wave's RC_DEPTH x RC_BUF bytes of chunk buffers, cut into buffers of NP pieces: fewer rows -> MORE chunks in flight
        // (13 instead of 3 at NP = 2).  A wave of ~13 rows waited 24 x for a 2-deep pipeline of 128-byte row segments: latency,
        // not bandwidth (profiles/r03_pass2_experiments.txt, section 12).
        constexpr int BUFB = NP * 1024;
        constexpr int DEPTH_RAW = RINGB / BUFB;
        constexpr int DEPTH = DEPTH_RAW > 16 ? 16 : ((DEPTH_RAW - 1) * NP > 60 ? 60 / NP + 1 : DEPTH_RAW);
        static_assert(DEPTH >= 2 && (DEPTH - 1) * NP <= 63, "vmcnt literal");
        const int nrows = min(RC_ROWS, total - base);
        // DMA plan: piece pc, lane i writes LDS bytes [pc*1024 + 16 i, +16) of the chunk buffer = row r, byte col of the
        // pitch; its source is row r's chunk + col.  (r, col) do not depend on the chunk: one source
        // pointer per piece, advanced by the chunk's bytes per chunk.  Lanes in the pad column or past the batch read the
        // first row's segment again (an L2 hit nobody uses).
        const float* src[NP];
        unsigned colb[NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            const int o = pc * 1024 + lane * 16;
            const int r = o / RC_PITCH;
            const int cb = o - r * RC_PITCH;
            const bool okl = r < nrows && cb < RC_CHUNK * 4;
            const int i = base + (okl ? r : 0);
            src[pc] = P.rows + (size_t)krow[i] * d + (okl ? cb / 4 : 0);
            colb[pc] = okl ? (unsigned)cb : 0u;
        }
        auto issue = [&](int c, unsigned char* buf) {
            const int cbytes = min(RC_CHUNK, d - c * RC_CHUNK) * 4;  // the last chunk of a row may be short
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                // a lane past the row's end re-reads the row's first bytes of this chunk (kept inside the row)
                const float* s_ = src[pc] + c * RC_CHUNK - ((int)colb[pc] < cbytes ? 0 : (int)colb[pc] / 4);
                glds16(reinterpret_cast<const float4*>(s_), reinterpret_cast<float4*>(buf + pc * 1024));
            }
        };
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < DEPTH - 1; ++c)
            if (c < nchunks) issue(c, mine + c * BUFB);
        int slot = 0;  // c % DEPTH
        for (int c = 0; c < nchunks; ++c) {
            unsigned char* cur = mine + slot * BUFB;
            if (c + DEPTH - 1 < nchunks) {   // steady state: chunk c has landed once only the DEPTH - 1 younger chunks are outstanding
                issue(c + DEPTH - 1, mine + (slot == 0 ? DEPTH - 1 : slot - 1) * BUFB);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * NP) : "memory");
            } else {                          // the row's tail: everything still in flight (one more latency, then the waits are free)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            slot = slot + 1 == DEPTH ? 0 : slot + 1;
            if (lane < nrows) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const unsigned xa = (unsigned)reinterpret_cast<uintptr_t>(cur) + (unsigned)lane * RC_PITCH;
                const unsigned qa = qaddr + (unsigned)c * RC_CHUNK * 4u;
                const int nsteps = (min(RC_CHUNK, d - c * RC_CHUNK) + 31) / 32;   // 32 floats per step
                const int tail = min(RC_CHUNK, d - c * RC_CHUNK) - (nsteps - 1) * 32;  // floats of the last step (multiple of 4)
                for (int st = 0; st < nsteps; ++st) {
                    f32x4 xv[8], qq[8];
                    const unsigned xo = xa + st * 128u, qo = qa + st * 128u;
                    // asm reads: hipcc would order a visible ds_read behind every pending LDS-DMA (vmcnt(0))
                    asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:16\n\tds_read_b128 %2, %16 offset:32\n\t"
                                 "ds_read_b128 %3, %16 offset:48\n\tds_read_b128 %4, %16 offset:64\n\tds_read_b128 %5, %16 offset:80\n\t"
                                 "ds_read_b128 %6, %16 offset:96\n\tds_read_b128 %7, %16 offset:112\n\t"
                                 "ds_read_b128 %8, %17\n\tds_read_b128 %9, %17 offset:16\n\tds_read_b128 %10, %17 offset:32\n\t"
                                 "ds_read_b128 %11, %17 offset:48\n\tds_read_b128 %12, %17 offset:64\n\tds_read_b128 %13, %17 offset:80\n\t"
                                 "ds_read_b128 %14, %17 offset:96\n\tds_read_b128 %15, %17 offset:112\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(xv[4]), "=&v"(xv[5]), "=&v"(xv[6]), "=&v"(xv[7]),
                                   "=&v"(qq[0]), "=&v"(qq[1]), "=&v"(qq[2]), "=&v"(qq[3]), "=&v"(qq[4]), "=&v"(qq[5]), "=&v"(qq[6]), "=&v"(qq[7])
                                 : "v"(xo), "v"(qo) : "memory");
                    const int nv = st + 1 < nsteps ? 8 : tail / 4;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (t < nv) {
                            acc = __builtin_fmaf(qq[t].x, xv[t].x, acc); acc = __builtin_fmaf(qq[t].y, xv[t].y, acc);
                            acc = __builtin_fmaf(qq[t].z, xv[t].z, acc); acc = __builtin_fmaf(qq[t].w, xv[t].w, acc);
                        }
                    }
                }
            }
        }
        if (lane < nrows) ksc[base + lane] = acc;
}
// the batch at `base`, with as few pieces per chunk as its rows need (wave-uniform choice)
template <bool SMALL>
__device__ __forceinline__ void rc_stream_batch(const RescoreParams& P, unsigned char* mine, const unsigned* krow, float* ksc, unsigned qaddr,
                                                int base, int total, int lane) {
    constexpr int RINGB = SMALL ? RC_SMALL_RING : RC_DEPTH * RC_BUF;
    const int np = (min(RC_ROWS, total - base) * RC_PITCH + 1023) / 1024;   // wave-uniform
    if (np <= 2) rc_run_batch<2, RINGB>(P, mine, krow, ksc, qaddr, base, total, lane);
    else if (SMALL || (np <= 4 && RC_PIECES > 4)) rc_run_batch<(RC_PIECES > 4 ? 4 : RC_PIECES), RINGB>(P, mine, krow, ksc, qaddr, base, total, lane);
    else { if constexpr (!SMALL) rc_run_batch<RC_PIECES, RINGB>(P, mine, krow, ksc, qaddr, base, total, lane); }
}

// The re-rank proper, shared by rescore_kernel and tail_kernel (lmi_tail.h): the wave's survivors krow[0, off[G]) (absolute slab rows, slot after
// slot) and its query are in LDS; rows streamed, canonical chains, then every slot's rank list.  LCOPY: the lists also go to the
// wave's LDS copy rl_d / rl_i [G][KPB] (the caller merges them in the wave).  colv / fbv: the slots' columns and fallback flags.
// PRE (tail_kernel, round 5): what the rank lists need from global memory was requested by the caller BEFORE the rows were streamed -- the id of
// survivor `lane` (my_id), and per slot the bucket's first slab row, its row count and the id of its last row (the padding) -- instead of
// three more dependent round trips at the wave's end (a tail wave is parked on memory two thirds of its life: profiles/r05_tail_ring.txt).
template <int G, bool SMALL, bool LCOPY, bool PRE = false>
__device__ __forceinline__ void rescore_core(const RescoreParams& P, const RcWave<G, SMALL>& W, const int (&off)[G + 1], int p0, int lane,
                                             const int (&colv)[G], const int (&fbv)[G], float* rl_d, unsigned* rl_i,
                                             unsigned my_id = 0u, const unsigned* row_base = nullptr, const int* nbr = nullptr, const unsigned* pad_id = nullptr) {
    unsigned char* mine = W.mine;   // (rows of pitch P.dp: the chain runs over the zero padding too -- +0 * +0 added to the sum changes nothing)
    unsigned* krow = W.krow;
    float* ksc = W.ksc;
    const float FMAXV = 3.402823466e+38f;
    const int total = off[G];
    const unsigned qaddr = (unsigned)reinterpret_cast<uintptr_t>(W.qs);
    for (int base = 0; base < total; base += RC_ROWS) rc_stream_batch<SMALL>(P, mine, krow, ksc, qaddr, base, total, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank lists: a survivor's position = the number of survivors of ITS slot that beat it by (score desc, row asc)
    for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        if (i < total) {
            int sl = 0;
#pragma unroll
            for (int t = 1; t < G; ++t) sl += i >= off[t] ? 1 : 0;
            int lo = off[0], hi = off[1];
#pragma unroll
            for (int t = 1; t < G; ++t) { lo = sl == t ? off[t] : lo; hi = sl == t ? off[t + 1] : hi; }
            const float sc = ksc[i];
            const unsigned row = krow[i];
            int pos = 0;
            for (int j = lo; j < hi; ++j) pos += better(ksc[j], krow[j], sc, row) ? 1 : 0;
            if (pos < KPB) {
                const int p = p0 + sl;
                const float dv = P.raw ? sc : sim_to_dist(sc, P.qn2, p / P.nb);
                // (krow holds absolute slab rows; the raw form -- lmi_knn_ip -- returns the row inside its bucket)
                unsigned iv;
                if constexpr (PRE) {
                    unsigned rbase = row_base[0];
#pragma unroll
                    for (int t = 1; t < G; ++t) rbase = sl == t ? row_base[t] : rbase;
                    iv = P.raw ? row - rbase : my_id;   // (PRE: total <= 64, survivor i sits in lane i)
                } else {
                    iv = P.raw ? row - (unsigned)P.rb_start[P.bucket_order[p]] * 32u : P.ids_slab[row];
                }
                P.rank_d[(size_t)p * KPB + pos] = dv;
                P.rank_id[(size_t)p * KPB + pos] = iv;
                if (LCOPY) { rl_d[sl * KPB + pos] = dv; rl_i[sl * KPB + pos] = iv; }
            }
        }
    }
    // faiss padding (Q4) of short lists: -FLT_MAX similarity, last id of the bucket
    if (lane < KPB) {
#pragma unroll
        for (int sl = 0; sl < G; ++sl) {
            const int p = p0 + sl;
            if (colv[sl] < 0 || fbv[sl]) continue;  // written by the selection / recomputed by fallback_kernel
            int n_b;
            unsigned pad_iv;
            if constexpr (PRE) {
                n_b = nbr[sl];
                pad_iv = pad_id[sl];
            } else {
                const int b = P.bucket_order[p];
                const int rb0 = P.rb_start[b];
                n_b = P.nb_rows[b];
                pad_iv = P.raw ? NOROW : P.ids_slab[(size_t)rb0 * 32 + (n_b - 1)];
            }
            const int nreal = min(min(off[sl + 1] - off[sl], KPB), n_b);
            if (lane >= nreal) {
                const float dv = P.raw ? -FMAXV : pad_dist(P.qn2);
                const unsigned iv = P.raw ? NOROW : pad_iv;
                P.rank_d[(size_t)p * KPB + lane] = dv;
                P.rank_id[(size_t)p * KPB + lane] = iv;
                if (LCOPY) { rl_d[sl * KPB + lane] = dv; rl_i[sl * KPB + lane] = iv; }
            }
        }
    }
}

template <int G, bool SMALL>
__device__ __forceinline__ void rescore_group(const RescoreParams& P, const SelectOut& O, unsigned char* rc_smem, int wv, int lane, int p0) {
    const int d = P.dp;
    const RcWave<G, SMALL> W(rc_smem + (size_t)wv * rc_wave_lds(d, G, SMALL), d);
    // the survivor lists of the wave's slots, slot after slot, and the wave's query (G divides nb: one query per wave)
    int off[G + 1];
    off[0] = 0;
    int colv[G], fbv[G], nkv[G];   // all 3 G loads in flight together (a short-circuit `&&` chained them: three round trips per slot)
#pragma unroll
    for (int sl = 0; sl < G; ++sl) { colv[sl] = P.slot_col[p0 + sl]; fbv[sl] = P.fallback[p0 + sl]; nkv[sl] = P.nkeep[p0 + sl]; }
#pragma unroll
    for (int sl = 0; sl < G; ++sl) off[sl + 1] = off[sl] + ((colv[sl] >= 0 && !fbv[sl]) ? nkv[sl] : 0);   // wave-uniform
    if (SMALL && off[G] > RC_SMALL_ROWS) {   // more than this launch holds: the second launch takes the group
        if (lane == 0) O.big[1 + atomicAdd(&O.big[0], 1)] = p0 / G;
        return;
    }
#pragma unroll
    for (int sl = 0; sl < G; ++sl) {
        const int nk = off[sl + 1] - off[sl];
        for (int i = lane; i < nk; i += 64) W.krow[off[sl] + i] = O.surv_row[(size_t)(p0 + sl) * RC_KEEP + i];
    }
    W.stage_query(P, p0 / P.nb, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rescore_core<G, SMALL, false>(P, W, off, p0, lane, colv, fbv, nullptr, nullptr);
}

template <int G, bool SMALL>
__global__ __launch_bounds__(64 * RC_WAVES, SMALL ? 3 : 1) void rescore_kernel(RescoreParams P, SelectOut O) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nwv = (int)(blockDim.x >> 6);   // waves of the block: RC_WAVES, fewer for rows so wide that four waves' buffers pass 160 KiB
    const int wid = blockIdx.x * nwv + wv;
    if (SMALL) {
        // wave wid -> the wid-th entry of the concatenated sub-lists (inclusive prefix of the 64 counts across the lanes)
        int incl = O.active[lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const int sub = (int)__popcll(__ballot(incl <= wid));  // sub-lists that end at or before entry wid
        if (sub >= RC_SUB) return;                             // wid >= total
        const int before = sub ? __shfl(incl, sub - 1, 64) : 0;
        rescore_group<G, true>(P, O, rc_smem, wv, lane, O.active[RC_SUB + sub * O.sub_cap + (wid - before)] * G);
    } else {
        // the passed-on groups are few (none on most batches): a grid of one block per CU walks the list, so that an empty list
        // costs one launch of 256 blocks and not one block (148 KiB of LDS each, one per CU at a time) per four groups
        const int nbig = __builtin_amdgcn_readfirstlane(O.big[0]);
        for (int i = wid; i < nbig; i += (int)gridDim.x * nwv) {
            rescore_group<G, false>(P, O, rc_smem, wv, lane, O.big[1 + i] * G);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the wave's LDS lists are rewritten by its next group
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace lmi
