// fp16 prefilter, low-dimensional form (d <= 128: at most PS_MAXKG k16-groups per row-block).
//
// pass2_kernel (lmi_pass2.h) streams K through a ring of LDS stages and keeps a 256 x 384 tile of scores in registers.  At
// d = 45 (BASELINE.json configs[4], the AlphaFind protein-embedding shape) a tile has two stages: its pipeline never warms up
// (stamps, profiles/r03_pass2_experiments.txt section 14: the loader waves wait for every stage, 15 k cycles per tile of which
// the matrix pipe works 0.6 k) and the 5 k-cycle epilogue over 192 accumulator registers is 8 x the K loop's arithmetic.
//
// Here K fits registers and LDS whole, so the roles turn round:
//   * the query tile's fragments (<= 12 col-blocks x KG KiB) are staged in LDS ONCE per work item and stay;
//   * a wave streams ROW-BLOCKS: it loads the KG fragments of two row-blocks straight from global memory into registers (they
//     are private to the wave: no LDS, no barrier) one pair ahead, and walks the col-blocks: KG MFMAs per (row-block,
//     col-block) chained on one 16-register accumulator, tested (pass 2) or reduced to its maximum (pass 1) at once;
//   * no K loop, no ring, no stage barrier; four waves per block, two blocks per CU (three fit the LDS at KG <= 4 but leave pass 2 too few registers);
//     K > 64: ONE row-block at a time instead of a pair (registers), query tiles of 8 / 7 / 6 col-blocks at 5-6 / 7 / 8 k16-groups (LDS);
//   * pass 2's candidates (round 4, K <= 64): the kernel is bound by VALU ISSUE, not by bytes or the matrix pipe -- PMC at C5: 16 vector
//     instructions per MFMA, the matrix pipe 11 % busy -- and two thirds of the 32 x 32 blocks hold a candidate (~42 per column and
//     39 k-row bucket), so the per-register search for it WAS the kernel.  Now a block's test is one 8-instruction max tree + one
//     compare; a LANE whose 16 scores hold a candidate spills the 16 scores (4 ds_write_b128) + a tag into a wave-private ring in
//     LDS, and whenever the ring holds 64 entries the wave drains them in parallel -- lane i takes entry i, finds its hits (almost
//     always one), issues their position atomics, and stores row / score at the NEXT drain (nobody waits for an atomic's round trip).
//     Every K <= 128 takes this path since round 4 (until then K > 64 kept round 3's per-register path: 1 270 / 1 436 spilled registers at
//     5 / 6 k16-groups, one wave per SIMD at 7 / 8: 0.13 / 0.13 / 0.24 / 0.30 of the HBM roofline at d = 80 / 96 / 112 / 128, now 0.43 / 0.43 / 0.41 / 0.46).
// Work items, queues, bounds, candidate buffers and the accumulation order (k-groups in order from 0, so shat is bit-identical
// to pass2_kernel's) are those of lmi_pass2.h: the host picks the kernel by KG16 alone.
#pragma once

#include <type_traits>

#include "lmi_pass2.h"

namespace lmi {

constexpr int PS_MAXKG = 8;
#ifndef LMI_PS_WAVES
#define LMI_PS_WAVES 4
#endif
constexpr int PS_WAVES = LMI_PS_WAVES;   // waves per block (they share the query tile in LDS) ...
// ... and twice as many in the WIDE form (K > 64 only): ONE block of eight waves per CU, whose LDS then holds a 12-col-block query tile.  The
// host takes it when the buckets receive more queries than the narrow form's tile holds (ps_use_wide): the rows are then streamed once
// instead of twice (345 queries per bucket, 4M rows: pass 2 at d = 80 / 96 / 112 / 128 0.377 / 0.445 / 0.494 / 0.552 -> 0.345 / 0.415 /
// 0.467 / 0.488 ms); with few queries per bucket (C5's 156) two independent 4-wave blocks are 1-7 % ahead and stay.
#ifndef LMI_PS_WIDE_KG
#define LMI_PS_WIDE_KG 5
#endif
__host__ __device__ constexpr bool ps_has_wide(int kg) { return kg >= LMI_PS_WIDE_KG; }
__host__ __device__ constexpr int ps_waves(int kg, bool wide) { return wide && ps_has_wide(kg) ? 2 * PS_WAVES : PS_WAVES; }
#ifndef LMI_PS_SPILL
#define LMI_PS_SPILL 96
#endif
constexpr int PS_SPILL = LMI_PS_SPILL;        // entries of a wave's spill ring (pass 2): 64 scores + tag each

// col-blocks per query tile: K > 64 takes smaller tiles, so that two blocks per CU hold the tile AND the spill ring
__host__ __device__ constexpr int ps_tile_cb(int kg, bool wide) { return kg <= 4 || (wide && ps_has_wide(kg)) ? P2_MAXCB : kg <= 6 ? 8 : kg == 7 ? 7 : 6; }
__host__ __device__ constexpr int ps_spill_bytes(int kg, bool wide) { return ps_waves(kg, wide) * PS_SPILL * (64 + 4); }
__host__ __device__ constexpr int ps_lds_bytes(int kg, bool wide) { return ps_tile_cb(kg, wide) * kg * 1024 + ps_tile_cb(kg, wide) * 32 * 4 + ps_spill_bytes(kg, wide); }
// the wide form pays when the mean number of queries per visited bucket exceeds what the narrow form's tile holds
__host__ inline bool ps_use_wide(int kg, double queries_per_bucket) { return ps_has_wide(kg) && queries_per_bucket > 0.8 * 32.0 * ps_tile_cb(kg, false); }
constexpr int PS_PREFIX_CAP = 257;   // buckets + 1 of a queue group held in LDS (more: the global prefix is searched)
#ifndef LMI_PS_BLOCKS4
#define LMI_PS_BLOCKS4 2   // blocks per CU at KG <= 4 (LDS allows 3, but pass 2 then has 168 registers and spills 175: 0.47 -> 1.06 ms)
#endif
__host__ __device__ constexpr int ps_blocks_per_cu(int kg, bool wide) { return wide && ps_has_wide(kg) ? 1 : kg <= 4 ? LMI_PS_BLOCKS4 : 2; }   // LDS: 160 KiB per CU
constexpr bool ps_lds_fits() {
    for (int kg = 1; kg <= PS_MAXKG; ++kg)
        for (int wide = 0; wide < 2; ++wide)
            if (ps_blocks_per_cu(kg, wide) * (ps_lds_bytes(kg, wide) + 4096) > 160 * 1024) return false;
    return true;
}
static_assert(PS_SPILL >= 128 - 32 && ps_lds_fits(), "LDS budget (dynamic + ~3 KiB static) for every k16-group count");

template <int KG, bool SAMPLE, bool WIDE = false>
__global__ __launch_bounds__(64 * ps_waves(KG, WIDE), ps_blocks_per_cu(KG, WIDE) * ps_waves(KG, WIDE) / 4) void pass2_small_kernel(PrefilterParams P) {
    static_assert(!WIDE || ps_has_wide(KG), "the wide form exists for K > 64 only");
    constexpr int WV = ps_waves(KG, WIDE);   // waves of the block
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    constexpr int TCB = ps_tile_cb(KG, WIDE);                                 // col-blocks per query tile
    uint4* sB = reinterpret_cast<uint4*>(ps_smem);                            // [col-blocks of the tile][KG][64 lanes]
    float* sThr = reinterpret_cast<float*>(ps_smem + TCB * KG * 1024);        // [TCB * 32] emission thresholds (pass 2)
    constexpr bool SPILL = !SAMPLE;   // pass 2's lane-granular spill ring (file header)
#ifndef LMI_PS_PAIR_KG
#define LMI_PS_PAIR_KG 4
#endif
    // A wave's unit of work: a PAIR of row-blocks at K <= 64 (a query fragment feeds two MFMAs), ONE row-block beyond -- two sets of a pair's
    // fragments are 16 KG registers (K = 96: with pairs 42 spilled registers and one query-fragment set; single row-blocks: none, two sets;
    // pass 2 at d = 80 / 96: 0.580 / 0.660 -> 0.472 / 0.562 ms; at d = 45 / 64 pairs stay 1-4 % ahead; profiles/r04_pass2_experiments.txt section 10)
    constexpr bool PAIR = KG <= LMI_PS_PAIR_KG;
    constexpr bool B2 = KG <= 4 || SAMPLE || !PAIR;        // two query-fragment sets taking turns
    constexpr int KG1 = PAIR ? KG : 1;                     // (the second row-block's fragment arrays)
    float4* sSpill = reinterpret_cast<float4*>(ps_smem + TCB * KG * 1024 + TCB * 32 * 4);   // [waves][PS_SPILL][4] the 16 scores of an entry
    unsigned* sTag = reinterpret_cast<unsigned*>(sSpill + WV * PS_SPILL * 4);                    // [waves][PS_SPILL] column in the tile | row base << 9
    __shared__ int s_item[2];
    __shared__ int s_prefix[PS_PREFIX_CAP];
    if (!SAMPLE && P.redo_count && *P.redo_count == 0u) return;  // the redo launch of a batch without overflowed columns
    ts_first(P.ts_start);
    unsigned long long clk_w0 = 0, clk_c0 = 0;
    if (!SAMPLE) clk_begin(P.ts_end_cell ? P.ts_start : nullptr, clk_w0, clk_c0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // pending candidates of the lane (pass 2): their position atomics were issued at the previous pair end / drain
    unsigned pend_pos = 0xffffffffu, pend_row = 0u, pend2_pos = 0xffffffffu, pend2_row = 0u;
    size_t pend_col = 0, pend2_col = 0;
    float pend_s = 0.0f, pend2_s = 0.0f;
    auto flush_pending = [&]() __attribute__((always_inline)) {
        if (pend_pos != 0xffffffffu) cand_store(P, pend_col, pend_pos, pend_row, pend_s);
        pend_pos = 0xffffffffu;
        if (SPILL) {
            if (pend2_pos != 0xffffffffu) cand_store(P, pend2_col, pend2_pos, pend2_row, pend2_s);
            pend2_pos = 0xffffffffu;
        }
    };
#ifdef LMI_P2_STAMPS   // developer builds (tools/p2_stamps.py --small): cycles per wave in 0 item start (queue + tile staging + barrier), 1 wait
                       // for the pair's vectors, 2 col-block loop, 3 drains, 4 item end (barrier); [7] = pairs
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#define PS_STAMP(PH) if (!SAMPLE) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[PH] += t_ - st_last; st_last = t_; }
#else
#define PS_STAMP(PH)
#endif
    // an item is a few microseconds of work: the queue takes the next ticket ahead (P2Queue, lmi_pass2.h)
    P2Queue<SAMPLE, 1, 64 * WV, PS_PREFIX_CAP> queue{P, s_item, s_prefix, nullptr};
    queue.init();
    P2Item item;
    while (queue.next(item)) {
        // the item came through LDS: tell the compiler it is wave-uniform (loop control and addresses become scalar)
        const int b = __builtin_amdgcn_readfirstlane(item.b), ncb = __builtin_amdgcn_readfirstlane(item.ncb_tile);
        const int item_ch = __builtin_amdgcn_readfirstlane(item.ch), item_cbt0 = __builtin_amdgcn_readfirstlane(item.cbt0);
        const int item_m_use = __builtin_amdgcn_readfirstlane(item.m_use), item_nt = __builtin_amdgcn_readfirstlane(item.nt);
        const int n_b = P.nb_rows[b];
        const int nrb_b = (n_b + 31) >> 5, rb_last = nrb_b - 1;
        const int stride = SAMPLE ? sample_stride(n_b, P.sample_max) : 1;
        const int crb = (!SAMPLE && P.chunk_rb_b) ? P.chunk_rb_b[b] : P.chunk_rb;
        const int rb0 = SAMPLE ? item_ch * stride * P2_TILE_RB : item_ch * crb;
        const int nrb = SAMPLE ? P2_TILE_RB : min(crb, nrb_b - rb0);
        // pass 1: the item's units run through its item_nt sampled tiles (UPT units each, tiles 2 stride apart; units past the bucket's end are skipped)
        constexpr int UPT = P2_TILE_RB / (PAIR ? 2 : 1);
        const int tile_step = 2 * stride * P2_TILE_RB;
        const int rb_end = SAMPLE ? rb_last : rb0 + nrb - 1;   // the last row-block a unit may touch
        auto unit_rb = [&](int pp) __attribute__((always_inline)) {
            return SAMPLE ? rb0 + (pp / UPT) * tile_step + (PAIR ? 2 : 1) * (pp % UPT) : rb0 + (PAIR ? 2 : 1) * pp;
        };
        int cur_rb0 = rb0, cur_list = item_ch % P2_NSL;       // pass 1: first row-block and list of the tile the current unit belongs to
        const int cb_tile = P.cb_start[b] + item_cbt0;
        const int m_left = item_m_use - item_cbt0 * 32;   // live columns of the tile from its first one
        const size_t col0 = (size_t)cb_tile * 32;
        const uint4* aslab = P.slab16 + (size_t)P.rb_start[b] * KG * 64 + lane;
        const int npairs = SAMPLE ? item_nt * UPT : PAIR ? (nrb + 1) >> 1 : nrb;   // units of the item
        const bool use_atomic = SAMPLE && p2_sample_tiles(n_b, P.sample_max) > P2_NSL;   // more sampled tiles than lists: every tile folds with the atomic
        half8 a0[KG], a1[KG1];
        auto load_pair = [&](int pp, half8 (&x0)[KG], half8 (&x1)[KG1]) __attribute__((always_inline)) {
            const int rbu = unit_rb(pp);
            const uint4* pa = aslab + (size_t)min(rbu, rb_last) * (KG * 64);
            const uint4* pb = aslab + (size_t)min(rbu + 1, rb_last) * (KG * 64);
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                x0[g] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(pa + g * 64));
                if constexpr (PAIR) x1[g] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(pb + g * 64));
            }
        };
        half8 b0[KG], b1[KG1];
        {
            // the query tile -> LDS: every load of a thread in flight before its first LDS store (one memory round trip, not one per step)
            const uint4* bsrc = P.qfrag16 + (size_t)cb_tile * KG * 64;   // the tile's col-blocks are consecutive
            constexpr int NV = (TCB * KG * 64 + 64 * WV - 1) / (64 * WV);
            const int nfrag = ncb * KG * 64;
            if constexpr (NV <= 12) {
                uint4 tmp[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) tmp[j] = bsrc[min(tid + j * 64 * WV, nfrag - 1)];   // (clamped: the tail re-reads the last fragment)
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (tid + j * 64 * WV < nfrag) sB[tid + j * 64 * WV] = tmp[j];
            } else {
                for (int i = tid; i < nfrag; i += 64 * WV) sB[i] = bsrc[i];
            }
            if (!SAMPLE) {
                for (int i = tid; i < ncb * 32; i += 64 * WV) {
                    const bool wanted = i < m_left && (!P.redo_col || P.redo_col[col0 + i]);
                    sThr[i] = wanted ? P.bound1[col0 + i] - P.eps2[col0 + i] : INFINITY;
                }
            }
        }
        __syncthreads();
        PS_STAMP(0)
        float4* my_spill = sSpill + w * (PS_SPILL * 4);
        unsigned* my_tag = sTag + w * PS_SPILL;
        int tot = 0;    // wave-uniform: entries in the wave's spill ring, the oldest at `head`
        int head = 0;
        // One (row-block, col-block) block of 32 x 32 scores: lane (h, c) holds column c, rows 4 h + (r & 3) + 8 (r >> 2).
        // the block's maximum by a flat v_max3 tree (8 instructions)
        auto flat_max = [&](const f32x16& acc) __attribute__((always_inline)) -> float {
            float m0, m1, m2, m3, m4;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(acc[3]), "v"(acc[4]), "v"(acc[5]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(acc[6]), "v"(acc[7]), "v"(acc[8]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(acc[9]), "v"(acc[10]), "v"(acc[11]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m4) : "v"(acc[12]), "v"(acc[13]), "v"(acc[14]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(m0), "v"(m1), "v"(acc[15]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(m2), "v"(m3), "v"(m4));
            asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(m0), "v"(m2));
            return m0;
        };
        // a candidate of column `col` that cannot wait (ring full / a lane's third hit): position atomic and stores at once
        auto emit_now = [&](size_t col, unsigned row, float sc) __attribute__((always_inline)) {
            const unsigned pos = atomicAdd(P.cand_cnt + col, 1u);
            cand_store(P, col, pos, row, sc);
        };
        // ... and the wave drains up to 64 entries at a time, one per lane: the lane's hits (one, rarely two, hardly ever more) get their
        // position atomics now and their stores at the NEXT drain (pend / pend2)
        auto drain = [&]() __attribute__((always_inline)) {
            flush_pending();
            const int nbatch = min(tot, 64);
            if (lane < nbatch) {
                int slot = head + lane;
                if (slot >= PS_SPILL) slot -= PS_SPILL;
                const float4* e = my_spill + slot * 4;
                const unsigned tag = my_tag[slot];
                const unsigned colt = tag & 511u, rowb = (unsigned)(rb0 * 32) + (tag >> 9);
                const float thr = sThr[colt];
                const float4 v0 = e[0], v1 = e[1], v2 = e[2], v3 = e[3];
                const float sv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
                unsigned m = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= sv[r] >= thr ? (1u << r) : 0u;
                const size_t col = col0 + colt;
                const float* es = reinterpret_cast<const float*>(e);
                if (m) {
                    const int r0 = __ffs((int)m) - 1;
                    m &= m - 1u;
                    pend_col = col;
                    pend_row = rowb + (unsigned)((r0 & 3) + 8 * (r0 >> 2));
                    pend_s = es[r0];
                    pend_pos = atomicAdd(P.cand_cnt + col, 1u);
                    if (m) {
                        const int r1 = __ffs((int)m) - 1;
                        m &= m - 1u;
                        pend2_col = col;
                        pend2_row = rowb + (unsigned)((r1 & 3) + 8 * (r1 >> 2));
                        pend2_s = es[r1];
                        pend2_pos = atomicAdd(P.cand_cnt + col, 1u);
                        while (m) {
                            const int r2 = __ffs((int)m) - 1;
                            m &= m - 1u;
                            emit_now(col, rowb + (unsigned)((r2 & 3) + 8 * (r2 >> 2)), es[r2]);
                        }
                    }
                }
            }
            head += nbatch;
            if (head >= PS_SPILL) head -= PS_SPILL;
            tot -= nbatch;
        };
        // K <= 64, pass 2: a lane whose block column holds a score >= thr parks its 16 scores in the wave's ring (a block adds at most
        // 64 entries: the ring is drained first whenever fewer than 64 are free)
        auto spill = [&](const f32x16& acc, float mx, float thr, int rb, int n) __attribute__((always_inline)) {
            bool any = mx >= thr;   // thr = +inf for idle columns, NaN (masked rows) never passes
#ifdef LMI_ABL_NOEMIT
            any = any && thr == 12345.678f;
#endif
            const unsigned long long mask = __ballot(any);
            if (__builtin_expect(mask != 0ull, 0)) {
                if (tot > PS_SPILL - 64) drain();
                const int my = tot + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (any) {
                    int slot = head + my;
                    if (slot >= PS_SPILL) slot -= PS_SPILL;
                    float4* e = my_spill + slot * 4;
                    e[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    e[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
                    e[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
                    e[3] = make_float4(acc[12], acc[13], acc[14], acc[15]);
                    my_tag[slot] = (unsigned)(n * 32 + c) | ((unsigned)((rb - rb0) * 32 + 4 * h) << 9);   // column in the tile | the lane's first row in the chunk
                }
                tot += (int)__popcll(mask);
            }
        };
        // zero-padded rows past the bucket's end are not scores.  pass 1: -inf, the maximum ignores them; pass 2: NaN -- a bucket
        // of fewer than ten rows has the bound -inf, which -inf would pass, and NaN >= x is false
        auto mask_rows = [&](f32x16& acc, int rb) __attribute__((always_inline)) {
            if (rb == rb_last && (n_b & 31)) {   // wave-uniform
                const int lim = n_b - rb * 32 - 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= lim) acc[r] = SAMPLE ? -INFINITY : __builtin_nanf("");
            }
        };
        // the block's test
        auto finish = [&](f32x16& acc, int rb, int n, float thr_n) __attribute__((always_inline)) {
            mask_rows(acc, rb);
            if constexpr (!SAMPLE) {
                spill(acc, flat_max(acc), thr_n, rb, n);
            } else {   // pass 1: the lane's maximum -> its slot of the list
                const float mx = flat_max(acc);
                if (n * 32 + c < m_left) {
                    // lists are COLUMN-minor: [list][slot = 2 (row-block in the tile) + h][column]
                    float* dst = P.bound + ((size_t)(cur_list * 16 + (rb - cur_rb0) * 2 + h)) * (size_t)P.ncols + (col0 + n * 32 + c);
                    if (!use_atomic) *dst = mx;
                    else {  // monotone float max through the order-preserving integer image
                        if (mx >= 0.0f) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(mx));
                        else atomicMin(reinterpret_cast<unsigned*>(dst), __float_as_uint(mx));
                    }
                }
            }
        };
        // a unit's end: full batches of the ring go out (the rest waits for company)
        auto pair_done = [&]() __attribute__((always_inline)) {
            if constexpr (!SAMPLE) {
                while (tot >= 64) drain();
            }
        };
        const half8* sBh = reinterpret_cast<const half8*>(sB) + lane;
        // A pair of row-blocks against every col-block of the tile.  The fragments of col-block n + 1 are requested before the
        // MFMAs of col-block n (two fragment sets taking turns): the LDS latency is off the MFMA chain.
        auto block = [&](int n, int rbA, bool second, const half8 (&x0)[KG], const half8 (&x1)[KG1], const half8 (&bfc)[KG],
                         auto& bfn) __attribute__((always_inline)) {
            const int nn = min(n + 1, ncb - 1);
            const float thr_n = SAMPLE ? 0.0f : sThr[n * 32 + c];   // requested ahead of the MFMAs: no LDS round trip in front of the test
            if constexpr (B2) {
#pragma unroll
                for (int g = 0; g < KG; ++g) bfn[g] = sBh[(nn * KG + g) * 64];
            }
            if constexpr (PAIR) {
                f32x16 c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[0], bfc[0], zero, 0, 0, 0);
                f32x16 c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[0], bfc[0], zero, 0, 0, 0);
#pragma unroll
                for (int g = 1; g < KG; ++g) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[g], bfc[g], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[g], bfc[g], c1, 0, 0, 0);
                }
                finish(c0, rbA, n, thr_n);
                if (second) finish(c1, rbA + 1, n, thr_n);   // (the chunk's odd last row-block has no partner: a clamped re-read)
            } else {
                f32x16 c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[0], bfc[0], zero, 0, 0, 0);
#pragma unroll
                for (int g = 1; g < KG; ++g) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[g], bfc[g], c0, 0, 0, 0);
                finish(c0, rbA, n, thr_n);
            }
        };
        auto do_pair = [&](int p, const half8 (&x0)[KG], const half8 (&x1)[KG1]) __attribute__((always_inline)) {
            const int rbA = unit_rb(p);
            if (SAMPLE) {
                if (rbA > rb_last) return;   // (wave-uniform: the bucket's last sampled tile is short)
                cur_rb0 = rb0 + (p / UPT) * tile_step;
                cur_list = (item_ch + 2 * (p / UPT)) % P2_NSL;
            }
            const bool second = PAIR && rbA + 1 <= rb_end;   // wave-uniform
#ifdef LMI_P2_STAMPS
            if (!SAMPLE) { asm volatile("s_waitcnt vmcnt(12)" : "+v"(const_cast<half8&>(x0[0])) :: "memory"); st_acc[7] += 1; }
            PS_STAMP(1)
#endif
            half8 bfa[KG], bfb[B2 ? KG : 1];
            if constexpr (B2) {
#pragma unroll
                for (int g = 0; g < KG; ++g) bfa[g] = sBh[g * 64];
            }
            // (Tried and dropped, round 4, profiles/r04_pass2_experiments.txt: the col-block loop software-pipelined on top of the spill
            // ring -- MFMAs of col-block n + 1 issued before the test of n, two accumulator pairs: 256 registers + 34 spilled, +10 % time.)
            if constexpr (B2) {
                for (int n = 0; n < ncb; n += 2) {
                    block(n, rbA, second, x0, x1, bfa, bfb);
                    if (n + 1 < ncb) block(n + 1, rbA, second, x0, x1, bfb, bfa);
                }
            } else {   // one fragment set, read when the col-block starts (the SIMD's other wave covers the LDS round trip)
                for (int n = 0; n < ncb; ++n) {
#pragma unroll
                    for (int g = 0; g < KG; ++g) bfa[g] = sBh[(n * KG + g) * 64];
                    block(n, rbA, second, x0, x1, bfa, bfa);
                }
            }
            PS_STAMP(2)
            pair_done();
            PS_STAMP(3)
        };
        {
            // (Tried and dropped, round 4, profiles/r04_pass2_experiments.txt: THREE register sets -- a pair's vectors requested two pairs
            // ahead -- with the pairs handed out by an LDS counter and the first two requested before the tile is staged: 9 spilled
            // registers and 3 % slower at C5; 6 waves per block / 3 blocks per CU for a third wave per SIMD: 76-92 spills, 45 % slower.)
            // the wave's pairs p = w, w + waves, ..: the next pair's fragments are requested before the current one is computed, two
            // register sets taking turns
            const int last = npairs - 1;
            int p = w;
#ifdef LMI_ABL_NOLOAD   // timing-only ablation: the row-blocks are not streamed (wrong results)
            load_pair(min(p, last), a0, a1);
            for (; p < npairs; p += WV) do_pair(p, a0, a1);
#else
            if (p < npairs) load_pair(p, a0, a1);
            while (p < npairs) {
                load_pair(min(p + WV, last), b0, b1);
                do_pair(p, a0, a1);
                p += WV;
                if (p >= npairs) break;
                load_pair(min(p + WV, last), a0, a1);
                do_pair(p, b0, b1);
                p += WV;
            }
#endif
        }
        if constexpr (SPILL) {
            while (tot > 0) drain();   // (the thresholds in LDS are the next item's after the barrier)
        }
        if (!SAMPLE) flush_pending();
        __syncthreads();   // the query fragments are replaced by the next item's
        PS_STAMP(4)
    }
#ifdef LMI_P2_STAMPS
    if (!SAMPLE && lane == 0) {
        unsigned long long* g = P.stamps + w * 12;
        for (int i = 0; i < 8; ++i) atomicAdd(g + i, st_acc[i]);
    }
#endif
    if (!SAMPLE) clk_end(P.ts_end_cell ? P.ts_start : nullptr, ST_P2, clk_w0, clk_c0);
    ts_max(P.ts_end_cell);
}

}  // namespace lmi
