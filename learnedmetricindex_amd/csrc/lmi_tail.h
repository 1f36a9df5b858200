// lmi_tail.h -- selection, exact re-rank and rank merge of a query in ONE wave (gfx950; n_buckets <= 4).
//
// Until round 4 the end of a prefilter search was five launches (select_kernel: a wave per slot -> survivor rows in global memory +
// compacted lists of active groups; rescore_kernel<G, small>: a wave per query re-reads them, streams the rows, writes the rank
// lists; rescore_kernel<G, big>; fallback_kernel; merge_ranks_kernel: a thread per query re-reads the rank lists) -- 119 us of a
// 630-us step at C5, 245 us at C2, every one a chain of dependent global round trips at low occupancy.  Here the wave that owns a
// query keeps everything it produces on the way in LDS:
//   tail_kernel<G>      a wave per query: That + survivors of its G = n_buckets slots (select_kernel's arithmetic; the loads of all
//                       slots in flight together), survivors -> the wave's LDS list (and, for the rare hand-overs below, global
//                       memory), rows streamed through the wave's LDS ring and re-scored by the canonical chain (rescore_core,
//                       lmi_rescore.h), the slots' rank lists -> global memory AND the wave's LDS, merged there by (distance, rank,
//                       position) = the reference's hstack + stable argsort (LearnedIndex.py:125-146) -> the caller's output rows.
//                       A query with more survivors than the ring holds at once re-scores them in batches and folds each batch
//                       into its slots' running ten best (no second launch).
//   fallback_kernel     (lmi_prefilter.h) the flagged slots; the LAST flagged slot of a query to finish merges the query from the
//                       rank lists in global memory (a per-query counter set by tail_kernel; agent-scope release / acquire around it).
// Same arithmetic, same selection rule, same tie order: results are bit-identical to the five-launch form (LMI_TAIL=0 in the
// environment keeps it; tests/test_gpu_tail.py runs both).
#pragma once
#include "lmi_rescore.h"

namespace lmi {

struct TailParams {
    int ngroups;         // waves of work: groups of G slots of one query (G = n_buckets: a query per wave, merged in the wave; G a proper divisor of
                         // n_buckets -- 8 buckets: two groups of 4 -- the groups' rank lists go to global memory and merge_ranks_kernel merges)
    int merge;           // G == n_buckets
    int kout;
    float* out_d;        // [nq][kout]
    unsigned* out_id;
    unsigned* out_key;   // nullable
    int* pending;        // [nq] flagged slots of the query still to be re-scored by fallback_kernel (written by tail_kernel for EVERY query)
};

// the wave's LDS behind the small / big re-rank layout: the rank lists' copy [G][KPB] distances, then ids
// (+ the running best lists of the many-survivor path: [G][KPB] scores, then rows)
__host__ __device__ inline int tail_wave_lds(int d, int G, bool small_form) { return (rc_wave_lds(d, G, small_form) + 15) / 16 * 16 + 4 * G * KPB * 4; }

// That + survivors of ONE slot (select_tail's arithmetic, lmi_rescore.h); survivors -> out_g[k] (global, k < RC_KEEP) and krow_l[k]
// (LDS, k < cap_l) as absolute slab rows.  Returns the survivor count (wave-uniform; may exceed RC_KEEP: the caller flags the slot).
template <int PERV, int SPEC>
__device__ __forceinline__ unsigned tail_select(int lane, unsigned cnt, float e2, unsigned row_base, const float* __restrict__ cs,
                                                const unsigned* __restrict__ cr, const float (&s_spec)[SPEC], const unsigned (&r_spec)[SPEC],
                                                unsigned* __restrict__ out_g, unsigned* krow_l, int cap_l) {
    const int nper = (int)((cnt + 63u) >> 6);
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
                if (PERV <= 4 || i < nper) c += (int)__popcll(__ballot(key[i] >= probe));
            if (c >= KPB) T = probe;
        }
        pv = __uint_as_float(T ^ ((T >> 31) ? 0x80000000u : 0xffffffffu));
    }
    const float cut = pv - e2;
    const unsigned cbits = __float_as_uint(cut);
    const unsigned kcut = cut != cut ? 1u : cbits ^ ((cbits >> 31) ? 0xffffffffu : 0x80000000u);
    unsigned nk = 0;
#pragma unroll
    for (int i = 0; i < PERV; ++i) {
        if (i < nper) {
            const bool keep = key[i] != 0u && key[i] >= kcut;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const unsigned k = nk + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                const unsigned row = row_base + (i < SPEC ? r_spec[i < SPEC ? i : 0] : cr[lane + 64 * i]);
                if (k < (unsigned)RC_KEEP) out_g[k] = row;
                if ((int)k < cap_l) krow_l[k] = row;
            }
            nk += (unsigned)__popcll(bal);
        }
    }
    return nk;
}

template <int G>
__global__ __launch_bounds__(64 * RC_WAVES, 3) void tail_kernel(RescoreParams P, SelectOut O, TailParams T) {
    select_stamps(P);
    pf_x_scatter(P, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nwv = (int)(blockDim.x >> 6);
    const int grp = blockIdx.x * nwv + wv;
    if (grp >= T.ngroups) return;
    const int q = (grp * G) / P.nb;   // the group's query
    const int dp = P.dp;
    unsigned char* base = tl_smem + (size_t)wv * tail_wave_lds(dp, G, true);
    const RcWave<G, true> W(base, dp);
    float* rl_d = reinterpret_cast<float*>(base + (rc_wave_lds(dp, G, true) + 15) / 16 * 16);
    unsigned* rl_i = reinterpret_cast<unsigned*>(rl_d + G * KPB);
    const int p0 = grp * G;
    const float FMAXV = 3.402823466e+38f;
    constexpr int SPEC = 4;
    constexpr int PER = PF_CAP / 64;
    // ---- round trip 1: the slots' columns and buckets; 2: counts, the first 256 candidates, eps', the buckets' first rows and sizes; 3 (not waited
    //      for until the rank lists): the id of each bucket's last row (faiss padding, Q4) ----
    int colv[G], fbv[G], off[G + 1], bkt[G], nbr[G];
    unsigned cnt[G], row_base[G], pad_id[G];
    float e2[G];
    float s_spec[G][SPEC];
    unsigned r_spec[G][SPEC];
#pragma unroll
    for (int sl = 0; sl < G; ++sl) { colv[sl] = P.slot_col[p0 + sl]; bkt[sl] = P.bucket_order[p0 + sl]; }
    W.stage_query(P, q, lane);   // the query's row travels with round trip 1 (it used to be a round trip of its own behind the selection)
#pragma unroll
    for (int sl = 0; sl < G; ++sl) {
        const size_t c = (size_t)(colv[sl] < 0 ? 0 : colv[sl]);
        if (colv[sl] < 0) bkt[sl] = 0;   // (a visited slot's bucket id is valid and the bucket has rows: the routing kernels said so)
        cnt[sl] = colv[sl] < 0 ? 0u : P.cand_cnt[c];
        e2[sl] = colv[sl] < 0 ? 0.0f : P.eps2[c];
        row_base[sl] = colv[sl] < 0 ? 0u : (unsigned)P.rb_start[bkt[sl]] * 32u;
        nbr[sl] = colv[sl] < 0 ? 0 : P.nb_rows[bkt[sl]];
#pragma unroll
        for (int i = 0; i < SPEC; ++i) {
            s_spec[sl][i] = colv[sl] < 0 ? 0.0f : P.cand_s[c * PF_CAP + lane + 64 * i];
            r_spec[sl][i] = colv[sl] < 0 ? 0u : P.cand_row[c * PF_CAP + lane + 64 * i];
        }
    }
#pragma unroll
    for (int sl = 0; sl < G; ++sl) pad_id[sl] = (colv[sl] < 0 || P.raw || nbr[sl] <= 0) ? NOROW : P.ids_slab[(size_t)row_base[sl] + (unsigned)(nbr[sl] - 1)];
    off[0] = 0;
    int nflag = 0;
#pragma unroll
    for (int sl = 0; sl < G; ++sl) {
        const int p = p0 + sl;
        off[sl + 1] = off[sl];
        fbv[sl] = 0;
        int nkeep = 0;
        if (colv[sl] < 0) {  // unvisited (LearnedIndex.py:340-341)
            if (lane < KPB) {
                const float dv = P.raw ? -FMAXV : INFINITY;
                const unsigned iv = P.raw ? NOROW : 0u;
                P.rank_d[(size_t)p * KPB + lane] = dv;
                P.rank_id[(size_t)p * KPB + lane] = iv;
                rl_d[sl * KPB + lane] = dv;
                rl_i[sl * KPB + lane] = iv;
            }
        } else if (cnt[sl] > (unsigned)PF_CAP) {
            fbv[sl] = 1;
        } else {
            const float* cs = P.cand_s + (size_t)colv[sl] * PF_CAP;
            const unsigned* cr = P.cand_row + (size_t)colv[sl] * PF_CAP;
            unsigned* out_g = O.surv_row + (size_t)p * RC_KEEP;
            unsigned* krow_l = W.krow + off[sl];
            const int cap_l = RcWave<G, true>::KEEPW - off[sl];
            unsigned nk;
            if (cnt[sl] <= 64u) nk = tail_select<1, SPEC>(lane, cnt[sl], e2[sl], row_base[sl], cs, cr, s_spec[sl], r_spec[sl], out_g, krow_l, cap_l);
            else if (cnt[sl] <= 256u) nk = tail_select<4, SPEC>(lane, cnt[sl], e2[sl], row_base[sl], cs, cr, s_spec[sl], r_spec[sl], out_g, krow_l, cap_l);
            else nk = tail_select<PER, SPEC>(lane, cnt[sl], e2[sl], row_base[sl], cs, cr, s_spec[sl], r_spec[sl], out_g, krow_l, cap_l);
            if (nk > (unsigned)RC_KEEP) fbv[sl] = 1;
            else { nkeep = (int)nk; off[sl + 1] = off[sl] + (int)nk; }
        }
        if (lane == 0) {
            P.nkeep[p] = nkeep;
            if (fbv[sl]) flag_fallback(P, p); else P.fallback[p] = 0;
        }
        nflag += fbv[sl];
    }
    if (lane == 0 && T.merge) T.pending[q] = nflag;
    const int total = off[G];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (total <= RC_SMALL_ROWS) {
        // the survivors' ids: requested now, used behind the chains
        const unsigned my_id = (lane < total && !P.raw) ? P.ids_slab[W.krow[lane]] : 0u;
        rescore_core<G, true, true, true>(P, W, off, p0, lane, colv, fbv, rl_d, rl_i, my_id, row_base, nbr, pad_id);
    } else {
        // More survivors than the small ring holds at once (a few per cent of the queries at C2, none at most shapes; round 4 and the
        // first form of this file passed them on to a second launch with a big ring -- 16-35 us per search, most of it the launch's
        // own turn-around): batches of RC_SMALL_ROWS rows, re-read from the slots' survivor lists in global memory (written above by
        // this wave; nobody else touches those lines), each batch's scores folded into the slots' running ten best.
        float* best_s = reinterpret_cast<float*>(rl_i + G * KPB);
        unsigned* best_r = reinterpret_cast<unsigned*>(best_s + G * KPB);
        int nbest[G];
#pragma unroll
        for (int sl = 0; sl < G; ++sl) nbest[sl] = 0;
        const unsigned qaddr = (unsigned)reinterpret_cast<uintptr_t>(W.qs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the survivor stores above have left
        for (int base = 0; base < total; base += RC_SMALL_ROWS) {
            const int nrows = min(RC_SMALL_ROWS, total - base);
            if (lane < nrows) {
                const int e = base + lane;
                int sl = 0;
#pragma unroll
                for (int t = 1; t < G; ++t) sl += e >= off[t] ? 1 : 0;
                int lo = off[0];
#pragma unroll
                for (int t = 1; t < G; ++t) lo = sl == t ? off[t] : lo;
                W.krow[lane] = __builtin_nontemporal_load(O.surv_row + (size_t)(p0 + sl) * RC_KEEP + (e - lo));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            rc_stream_batch<true>(P, W.mine, W.krow, W.ksc, qaddr, 0, nrows, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // fold: per slot, its old best (<= 10) and its rows of this batch (<= 28) ranked by (score desc, row asc); the ten best stay
#pragma unroll
            for (int sl = 0; sl < G; ++sl) {
                const int b_lo = max(off[sl], base) - base, b_hi = min(off[sl + 1], base + nrows) - base;   // the slot's rows of the batch
                const int nbat = max(0, b_hi - b_lo), nold = nbest[sl], n = nold + nbat;                        // wave-uniform
                if (nbat > 0) {
                    float sc = -INFINITY;
                    unsigned row = NOROW;
                    if (lane < nold) { sc = best_s[sl * KPB + lane]; row = best_r[sl * KPB + lane]; }
                    else if (lane < n) { sc = W.ksc[b_lo + lane - nold]; row = W.krow[b_lo + lane - nold]; }
                    int pos = 0;
                    for (int o = 0; o < n; ++o) {
                        const float os = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), o));
                        const unsigned orr = (unsigned)__builtin_amdgcn_readlane((int)row, o);
                        pos += better(os, orr, sc, row) ? 1 : 0;
                    }
                    if (lane < n && pos < KPB) { best_s[sl * KPB + pos] = sc; best_r[sl * KPB + pos] = row; }
                    nbest[sl] = min(n, KPB);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // the slots' rank lists from their ten best (rescore_core's last step)
        if (lane < KPB) {
#pragma unroll
            for (int sl = 0; sl < G; ++sl) {
                const int p = p0 + sl;
                if (colv[sl] < 0 || fbv[sl]) continue;   // written above / recomputed by fallback_kernel
                const int b = P.bucket_order[p];
                const int rb0 = P.rb_start[b], n_b = P.nb_rows[b];
                const int nreal = min(nbest[sl], n_b);
                float dv;
                unsigned iv;
                if (lane < nreal) {
                    const float sc = best_s[sl * KPB + lane];
                    const unsigned row = best_r[sl * KPB + lane];
                    dv = P.raw ? sc : sim_to_dist(sc, P.qn2, q);
                    iv = P.raw ? row - (unsigned)rb0 * 32u : P.ids_slab[row];
                } else {   // faiss padding (Q4)
                    dv = P.raw ? -FMAXV : pad_dist(P.qn2);
                    iv = P.raw ? NOROW : P.ids_slab[(size_t)rb0 * 32 + (n_b - 1)];
                }
                P.rank_d[(size_t)p * KPB + lane] = dv;
                P.rank_id[(size_t)p * KPB + lane] = iv;
                rl_d[sl * KPB + lane] = dv;
                rl_i[sl * KPB + lane] = iv;
            }
        }
    }
    if (nflag || !T.merge) return;   // fallback_kernel re-scores the flagged slot(s) and merges the query / merge_ranks_kernel merges the groups
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float dv = lane < G * KPB ? rl_d[lane] : 0.0f;
    const unsigned iv = lane < G * KPB ? rl_i[lane] : 0u;
    merge_entries<G>(dv, iv, lane, P.raw, T.kout, (size_t)q, T.out_d, T.out_id, T.out_key);
}

}  // namespace lmi
