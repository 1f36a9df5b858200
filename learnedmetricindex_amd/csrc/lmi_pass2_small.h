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
//   * pass 2's candidates of a row-block pair are compacted through a wave-private LDS list and go out one pair late (below).
// Work items, queues, bounds, candidate buffers and the accumulation order (k-groups in order from 0, so shat is bit-identical
// to pass2_kernel's) are those of lmi_pass2.h: the host picks the kernel by KG16 alone.
#pragma once

#include <type_traits>

#include "lmi_pass2.h"

namespace lmi {

constexpr int PS_MAXKG = 8;
constexpr int PS_WAVES = 4;
constexpr int PS_ROW_BITS = 23;   // list entry = column in the tile (9 bits) << 23 | row in the item's chunk (a chunk is at most 2^31 / 1024 rows)

__host__ __device__ constexpr int ps_lds_bytes(int kg) { return P2_MAXCB * kg * 1024 + P2_MAXCB * 32 * 4; }
#ifndef LMI_PS_BLOCKS4
#define LMI_PS_BLOCKS4 2   // blocks per CU at KG <= 4 (LDS allows 3, but pass 2 then has 168 registers and spills 175: 0.47 -> 1.06 ms)
#endif
__host__ __device__ constexpr int ps_blocks_per_cu(int kg) { return kg <= 4 ? LMI_PS_BLOCKS4 : kg <= 6 ? 2 : 1; }   // LDS: 160 KiB per CU
static_assert(2 * ps_lds_bytes(6) <= 160 * 1024 && ps_lds_bytes(PS_MAXKG) <= 160 * 1024 && ps_lds_bytes(5) <= 64 * 1024, "LDS budget");

template <int KG, bool SAMPLE>
__global__ __launch_bounds__(64 * PS_WAVES, ps_blocks_per_cu(KG)) void pass2_small_kernel(PrefilterParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    uint4* sB = reinterpret_cast<uint4*>(ps_smem);                            // [col-blocks of the tile][KG][64 lanes]
    float* sThr = reinterpret_cast<float*>(ps_smem + P2_MAXCB * KG * 1024);   // [P2_MAXCB * 32] emission thresholds (pass 2)
    __shared__ int s_item[2];
    __shared__ int s_prefix[P2_PREFIX_CAP];
    __shared__ uint2 s_list[SAMPLE ? 1 : PS_WAVES * 64];   // pass 2: a wave's candidates of one row-block pair (key, score bits)
    if (!SAMPLE && P.redo_count && *P.redo_count == 0u) return;  // the redo launch of a batch without overflowed columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // pending candidate of the lane (pass 2): its position atomic was issued at the end of the previous pair
    unsigned pend_pos = 0xffffffffu, pend_row = 0u;
    size_t pend_col = 0;
    float pend_s = 0.0f;
    auto flush_pending = [&]() __attribute__((always_inline)) {
        if (pend_pos < (unsigned)PF_CAP) {
            P.cand_row[pend_col * PF_CAP + pend_pos] = pend_row;
            P.cand_s[pend_col * PF_CAP + pend_pos] = pend_s;
        }
        pend_pos = 0xffffffffu;
    };
    // an item is a few microseconds of work: the queue takes the next ticket ahead (P2Queue, lmi_pass2.h)
    P2Queue<SAMPLE, true, 64 * PS_WAVES> queue{P, s_item, s_prefix};
    queue.init();
    P2Item item;
    while (queue.next(item)) {
        const int b = item.b, ncb = item.ncb_tile;
        const int n_b = P.nb_rows[b];
        const int nrb_b = (n_b + 31) >> 5, rb_last = nrb_b - 1;
        const int stride = SAMPLE ? sample_stride(n_b) : 1;
        const int rb0 = SAMPLE ? item.ch * stride * P2_TILE_RB : item.ch * P.chunk_rb;
        const int nrb = SAMPLE ? min(P2_TILE_RB, nrb_b - rb0) : min(P.chunk_rb, nrb_b - rb0);
        const int cb_tile = P.cb_start[b] + item.cbt0;
        const int m_left = item.m_use - item.cbt0 * 32;   // live columns of the tile from its first one
        const size_t col0 = (size_t)cb_tile * 32;
        {
            const uint4* bsrc = P.qfrag16 + (size_t)cb_tile * KG * 64;   // the tile's col-blocks are consecutive
            for (int i = tid; i < ncb * KG * 64; i += 64 * PS_WAVES) sB[i] = bsrc[i];
            if (!SAMPLE) {
                for (int i = tid; i < ncb * 32; i += 64 * PS_WAVES) {
                    const bool wanted = i < m_left && (!P.redo_col || P.redo_col[col0 + i]);
                    sThr[i] = wanted ? P.bound1[col0 + i] - P.eps2[col0 + i] : INFINITY;
                }
            }
        }
        __syncthreads();
        const uint4* aslab = P.slab16 + (size_t)P.rb_start[b] * KG * 64 + lane;
        const int npairs = (nrb + 1) >> 1;
        const int list_j = item.ch % P2_NSL;
        const bool use_atomic = SAMPLE && p2_sample_tiles(n_b) > P2_NSL;   // more sampled tiles than lists: every tile folds with the atomic
        half8 a0[KG], a1[KG];
        auto load_pair = [&](int pp, half8 (&x0)[KG], half8 (&x1)[KG]) __attribute__((always_inline)) {
            const uint4* pa = aslab + (size_t)min(rb0 + 2 * pp, rb_last) * (KG * 64);
            const uint4* pb = aslab + (size_t)min(rb0 + 2 * pp + 1, rb_last) * (KG * 64);
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                x0[g] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(pa + g * 64));
                x1[g] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(pb + g * 64));
            }
        };
        uint2* my_list = s_list + (SAMPLE ? 0 : w * 64);
        int tot = 0;   // wave-uniform: candidates of the current pair in my_list (entries past 63 went out directly)
        // One (row-block, col-block) block of 32 x 32 scores: lane (h, c) holds column c, rows 4 h + (r & 3) + 8 (r >> 2).
        //   group_max  maximum of register group j (rows 8 j + 4 h + 0..3); v_max3 as asm: fmaxf() costs a canonicalising
        //              v_max per operand, and v_max3 returns the other operands for a NaN (the masked rows of pass 2)
        //   settle     pass 1: the lane's maximum -> its slot of the list; pass 2: candidates -> the wave's list
        auto group_max = [&](const f32x16& acc, int j) __attribute__((always_inline)) -> float {
            float t, m;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(acc[4 * j]), "v"(acc[4 * j + 1]), "v"(acc[4 * j + 2]));
            asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(t), "v"(acc[4 * j + 3]));
            return m;
        };
        auto settle = [&](const f32x16& acc, const float (&gm)[4], float thr, int rb, int n) __attribute__((always_inline)) {
            float t, mx;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(gm[0]), "v"(gm[1]), "v"(gm[2]));
            asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(t), "v"(gm[3]));
            if (SAMPLE) {
                if (n * 32 + c < m_left) {
                    // lists are COLUMN-minor: [list][slot = 2 (row-block in the tile) + h][column]
                    float* dst = P.bound + ((size_t)(list_j * 16 + (rb - rb0) * 2 + h)) * (size_t)P.ncols + (col0 + n * 32 + c);
                    if (!use_atomic) *dst = mx;
                    else {  // monotone float max through the order-preserving integer image
                        if (mx >= 0.0f) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(mx));
                        else atomicMin(reinterpret_cast<unsigned*>(dst), __float_as_uint(mx));
                    }
                }
            } else {
                bool any = mx >= thr;   // thr = +inf for idle columns
#ifdef LMI_ABL_NOEMIT   // timing-only ablation: no candidate is ever emitted (wrong results)
                any = any && thr == 12345.678f;
#endif
                // About one score in a thousand passes (~40 candidates per column and bucket), i.e. about every second block
                // has one: the path below is not rare.  Group maxima first: a block with one candidate tests 4 + 4 values, not 16.
                if (__builtin_expect(__ballot(any) != 0ull, 0)) {
                    const unsigned rowh = (unsigned)((rb - rb0) * 32 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (__ballot(gm[j] >= thr) == 0ull) continue;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool pass = acc[4 * j + i] >= thr;
                            const unsigned long long mask = __ballot(pass);
                            if (mask == 0ull) continue;
                            const unsigned row = rowh + (unsigned)(i + 8 * j);   // in the chunk
                            if (pass) {
                                const int my = tot + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                if (my < 64) {
                                    my_list[my] = make_uint2(((unsigned)(n * 32 + c) << PS_ROW_BITS) | row, __float_as_uint(acc[4 * j + i]));
                                } else {   // a pair with more than 64 candidates: the rest goes out at once
                                    const size_t col = col0 + n * 32 + c;
                                    const unsigned pos = atomicAdd(P.cand_cnt + col, 1u);
                                    if (pos < (unsigned)PF_CAP) {
                                        P.cand_row[col * PF_CAP + pos] = (unsigned)(rb0 * 32) + row;
                                        P.cand_s[col * PF_CAP + pos] = acc[4 * j + i];
                                    }
                                }
                            }
                            tot += (int)__popcll(mask);
                        }
                    }
                }
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
        auto finish = [&](f32x16& acc, int rb, int n) __attribute__((always_inline)) {
            mask_rows(acc, rb);
            float gm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gm[j] = group_max(acc, j);
            settle(acc, gm, SAMPLE ? 0.0f : sThr[n * 32 + c], rb, n);
        };
        // the pair's candidates: lane i takes list entry i -- its position atomic goes out now, its two stores when the NEXT
        // pair is done (the atomic's round trip is about one pair's time: nobody waits for it)
        auto pair_done = [&]() __attribute__((always_inline)) {
            if (SAMPLE) return;
            flush_pending();
            if (tot > 0) {
                if (lane < min(tot, 64)) {
                    const uint2 e = my_list[lane];
                    pend_col = col0 + (e.x >> PS_ROW_BITS);
                    pend_row = (unsigned)(rb0 * 32) + (e.x & ((1u << PS_ROW_BITS) - 1u));
                    pend_s = __uint_as_float(e.y);
                    pend_pos = atomicAdd(P.cand_cnt + pend_col, 1u);
                }
                tot = 0;
            }
        };
        const half8* sBh = reinterpret_cast<const half8*>(sB) + lane;
        // A pair of row-blocks against every col-block of the tile.  The fragments of col-block n + 1 are requested before the
        // MFMAs of col-block n (two fragment sets taking turns): the LDS latency is off the MFMA chain.
        // (Tried and dropped, profiles/r03_pass2_experiments.txt section 14: the col-block loop software-pipelined by hand -- test of
        // col-block n - 1 between the MFMAs of n, two accumulator sets -- needs 256 registers + 73 spilled and is 20 % slower.)
        auto block = [&](int n, int rbA, bool second, const half8 (&x0)[KG], const half8 (&x1)[KG], const half8 (&bfc)[KG],
                         half8 (&bfn)[KG]) __attribute__((always_inline)) {
            const int nn = min(n + 1, ncb - 1);
#pragma unroll
            for (int g = 0; g < KG; ++g) bfn[g] = sBh[(nn * KG + g) * 64];
            f32x16 c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[0], bfc[0], zero, 0, 0, 0);
            f32x16 c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[0], bfc[0], zero, 0, 0, 0);
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0[g], bfc[g], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1[g], bfc[g], c1, 0, 0, 0);
            }
            finish(c0, rbA, n);
            if (second) finish(c1, rbA + 1, n);   // (the chunk's odd last row-block has no partner: a clamped re-read)
        };
        auto do_pair = [&](int p, const half8 (&x0)[KG], const half8 (&x1)[KG]) __attribute__((always_inline)) {
            const int rbA = rb0 + 2 * p;
            const bool second = 2 * p + 1 < nrb;   // wave-uniform
            half8 bfa[KG], bfb[KG];
#pragma unroll
            for (int g = 0; g < KG; ++g) bfa[g] = sBh[g * 64];
            for (int n = 0; n < ncb; n += 2) {
                block(n, rbA, second, x0, x1, bfa, bfb);
                if (n + 1 < ncb) block(n + 1, rbA, second, x0, x1, bfb, bfa);
            }
            pair_done();
        };
        // the wave's pairs p = w, w + 4, ..: the next pair's fragments are requested before the current one is computed, two register
        // sets taking turns (two pairs ahead with three sets measured equal: profiles/r03_pass2_experiments.txt section 14)
        half8 b0[KG], b1[KG];
        const int last = npairs - 1;
        int p = w;
#ifdef LMI_ABL_NOLOAD   // timing-only ablation: the row-blocks are not streamed (wrong results)
        load_pair(min(p, last), a0, a1);
        for (; p < npairs; p += PS_WAVES) do_pair(p, a0, a1);
#else
        if (p < npairs) load_pair(p, a0, a1);
        while (p < npairs) {
            load_pair(min(p + PS_WAVES, last), b0, b1);
            do_pair(p, a0, a1);
            p += PS_WAVES;
            if (p >= npairs) break;
            load_pair(min(p + PS_WAVES, last), a0, a1);
            do_pair(p, b0, b1);
            p += PS_WAVES;
        }
#endif
        if (!SAMPLE) flush_pending();
        __syncthreads();   // the query fragments are replaced by the next item's
    }
}

}  // namespace lmi
