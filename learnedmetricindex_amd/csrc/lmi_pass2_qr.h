// lmi_pass2_qr.h -- pass 2 of the fp16 prefilter, "query-resident" form (gfx950).
//
// Same contract as prefilter_kernel<false, 2> (lmi_prefilter.h): for every (bucket, query tile, chunk) item
// every row with shat >= bound1[col] - 2 eps'[col] is appended to the column's candidate buffer.  What differs
// is where the operands live.  The streamed form re-reads a tile's 256 queries (393 KiB of fp16 fragments at
// d = 768) from L2 for every 256 vectors: half of its LDS-DMA pieces, one barrier per 12-16 MFMAs.  Here
//   * a wave keeps ONE col-block (32 queries) x the whole K (<= 768: 48 fragments = 192 VGPRs) in registers for the
//     whole item, 8 waves = up to 8 col-blocks per block, one block per CU (2 waves per SIMD);
//   * only the vectors stream: a row-block (32 vectors x K, one contiguous 48-KiB run of the fragment slab) per ring
//     slot, 3 slots, filled by LDS-DMA (6 pieces per wave and row-block), ONE barrier per row-block = per 48 MFMAs;
//   * every wave reads every fragment of the row-block from LDS (1 ds_read_b128 per MFMA, 4 in flight) and owns a
//     single 32 x 32 accumulator tile, so the epilogue is 16 compares against ONE per-lane threshold.
// Tiles with <= 4 col-blocks run "split": waves w and w+4 (the two waves of a SIMD) hold the same col-block and
// take the even / odd row-blocks, so that one of them computes while the other runs its epilogue and DMA issue.
// A bucket's col-blocks are cut into tiles of 8 + a remainder (cost ~ 4 ceil(c/4) per tile, so 8 + 3 beats 6 + 5).
// d > 768 (more fragments than a wave can hold) keeps the streamed kernel.
#pragma once
#include "lmi_prefilter.h"

namespace lmi {

constexpr int QR_WAVES = 8;
constexpr int QR_KMAX = 48;   // k16-groups resident per wave
constexpr int QR_RING = 3;    // row-block slots in LDS
#ifndef LMI_QR_AHEAD
#define LMI_QR_AHEAD 3        // fragment reads in flight per wave
#endif
constexpr int QR_AHEAD = LMI_QR_AHEAD;
#ifndef LMI_QR_STAGGER
#define LMI_QR_STAGGER 1      // SIMD partners half a row-block apart (0: in phase; A/B on MI355X in DESIGN.md)
#endif
constexpr int QR_NA = QR_AHEAD + 1;  // fragment registers: the one an MFMA has just been issued on is not a read target

// dynamic LDS of pass2_qr_kernel: ring + 8 compaction lists + the item broadcast
constexpr int QR_SLOT = QR_KMAX * 1024;  // a slot is 48 KiB whatever KG is (fragments at its end)
constexpr size_t QR_LDS = (size_t)QR_RING * QR_SLOT + QR_WAVES * 64 * 8 + 16;

#define QR_WAITVM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// In-kernel phase timing (developer builds, -DLMI_QR_STAMPS): shader-clock cycles per wave and phase, summed into
// P.bound (free once bound_merge_kernel has run) as u64 [2 modes: full / split][8 waves][12 phases]; read with lmi_debug_peek.
#ifdef LMI_QR_STAMPS
#define QR_STAMP(PH) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[PH] += t_ - st_last; st_last = t_; }
#define QR_STAMP_MEMBERS unsigned long long st_acc[12], st_last;
#else
#define QR_STAMP(PH)
#define QR_STAMP_MEMBERS
#endif

template <bool K48>
struct QrItem {
    const PrefilterParams& P;
    char* smem;
    int KG, K0;  // fragments per row-block (<= 48) and 48 - KG
    int lane, w;
    half8 B[QR_KMAX];
    float thr;
    unsigned pend_pos;   // position returned for list entry `lane` of the previous epilogue (the entry itself stays in the LDS list)
    unsigned pend_row0;  // wave-uniform: first row of that epilogue's row-block
    int npc;  // DMA pieces this wave issues per row-block (pieces w, w + 8, ..)
    // vmcnt bookkeeping (wave-uniform): vector-memory operations complete in issue order, so "the pieces of row-block t
    // have landed" / "the position atomic has returned" are counted waits that leave every YOUNGER operation in flight
    int npend;   // lanes holding a candidate of the previous epilogue (their position atomic is in flight)
    int since;   // VM operations issued after that atomic
    int e1, e2;  // VM operations issued by the epilogues of the previous / the one before the previous iteration
    QR_STAMP_MEMBERS

    // The epilogue derives everything per-lane (column, row offsets, list and buffer addresses) from a lane id formed HERE,
    // opaque to the optimiser: hoisted out of the row-block loop those values held ~25 VGPRs across the K loop, where the
    // resident fragments leave none to spare.
    static __device__ __forceinline__ unsigned lane_now() {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }

    template <int OFF>
    static __device__ __forceinline__ void lds_rd(half8& r, unsigned addr) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    }
    template <int N>
    static __device__ __forceinline__ void lds_wait(half8& a) {
        static_assert(N >= 0 && N <= 7, "lgkmcnt literal");
        if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory");
        if (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a) :: "memory");
        if (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a) :: "memory");
        if (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a) :: "memory");
        if (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a) :: "memory");
        if (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a) :: "memory");
        if (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a) :: "memory");
        if (N == 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a) :: "memory");
    }

    static __device__ __forceinline__ void wait_vm(int n) {  // s_waitcnt vmcnt(min(n, 12)), n wave-uniform
#ifndef LMI_ABL_NOWAIT
        switch (n) {
            case 0: QR_WAITVM(0); break;
            case 1: QR_WAITVM(1); break;
            case 2: QR_WAITVM(2); break;
            case 3: QR_WAITVM(3); break;
            case 4: QR_WAITVM(4); break;
            case 5: QR_WAITVM(5); break;
            case 6: QR_WAITVM(6); break;
            case 7: QR_WAITVM(7); break;
            case 8: QR_WAITVM(8); break;
            case 9: QR_WAITVM(9); break;
            case 10: QR_WAITVM(10); break;
            case 11: QR_WAITVM(11); break;
            default: QR_WAITVM(12); break;
        }
#endif
    }

    // a wave-uniform pointer, told to the compiler: the DMA then takes the SGPR-base + 32-bit lane offset form
    template <typename T>
    static __device__ __forceinline__ T* uniform_ptr(T* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
    }

    // piece i (0..5) of row-block `src` (this wave's share: fragments w + 8 i) -> ring slot at `dst`.  A row-block of
    // KG < 48 fragments sits at the END of its 48-KiB slot (fragment j at (K0 + j) KiB, K0 = 48 - KG): the K loop below is
    // straight-line code for fragments 0..47 that is entered at K0, so every LDS offset is an immediate.
    template <int I>
    __device__ __forceinline__ void dma_piece(const uint4* src, uint4* dst) {
#ifndef LMI_ABL_NOLOAD
        if (K48 || I < npc)
            glds16(reinterpret_cast<const float4*>(uniform_ptr(src + (size_t)(w + 8 * I) * 64) + lane),
                   reinterpret_cast<float4*>(dst + (K0 + w + 8 * I) * 64));
#endif
    }
    __device__ __forceinline__ void dma_all(const uint4* src, uint4* dst) {
        dma_piece<0>(src, dst); dma_piece<1>(src, dst); dma_piece<2>(src, dst);
        dma_piece<3>(src, dst); dma_piece<4>(src, dst); dma_piece<5>(src, dst);
    }

    // one k16 step: fragment K has landed -> MFMA; read K + AHEAD goes to the registers MFMA K-1 was issued on
    template <int K>
    __device__ __forceinline__ void kstep(half8 (&a)[QR_NA], f32x16& acc, unsigned aA, int k0, bool dma, const uint4* src, uint4* dst) {
        if (!K48 && K < k0) return;
        if (K % 2 == 0 && (K48 ? K == 0 : K == k0)) {  // entry: the first AHEAD reads
            lds_rd<K * 1024>(a[K % QR_NA], aA);
            lds_rd<(K + 1) * 1024>(a[(K + 1) % QR_NA], aA);
            if (QR_AHEAD > 2 && K + 2 < QR_KMAX) lds_rd<(K + 2 < QR_KMAX ? K + 2 : 0) * 1024>(a[(K + 2) % QR_NA], aA);
            if (QR_AHEAD > 3 && K + 3 < QR_KMAX) lds_rd<(K + 3 < QR_KMAX ? K + 3 : 0) * 1024>(a[(K + 3) % QR_NA], aA);
            static_assert(QR_AHEAD >= 2 && QR_AHEAD <= 4, "entry reads written out for 2..4");
        }
        if (K48 && K % 4 == 0 && dma) dma_piece<(K % 24) / 4>(src, dst);
        lds_wait<(QR_KMAX - 1 - K < QR_AHEAD - 1 ? QR_KMAX - 1 - K : QR_AHEAD - 1)>(a[K % QR_NA]);
        if (K48 && K == 0) {
            const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[K % QR_NA], B[K], z, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[K % QR_NA], B[K], acc, 0, 0, 0);
        }
        if (K + QR_AHEAD < QR_KMAX) lds_rd<(K + QR_AHEAD < QR_KMAX ? K + QR_AHEAD : 0) * 1024>(a[(K + QR_AHEAD) % QR_NA], aA);
    }

    // half a row-block: fragments [24 HALF, 24 HALF + 24).  The read pipeline runs through from the first half into the
    // second (a[] and acc live across the barrier between them); a half-step that carries the DMA of a later row-block
    // issues its 6 pieces one before every 4th MFMA.
    template <int HALF>
    __device__ __forceinline__ void compute_half(half8 (&a)[QR_NA], f32x16& acc, unsigned aA, bool dma, const uint4* src, uint4* dst) {
        int k0 = K0;
        if (!K48) {
            asm volatile("" : "+s"(k0));  // opaque: the entry tests are redone per half, not kept in 48 SGPR pairs
            if (dma) dma_all(src, dst);
            if (HALF == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            }
        }
#define QR_K1(K) kstep<K>(a, acc, aA, k0, dma, src, dst);
#define QR_K8(K) QR_K1(K) QR_K1(K + 1) QR_K1(K + 2) QR_K1(K + 3) QR_K1(K + 4) QR_K1(K + 5) QR_K1(K + 6) QR_K1(K + 7)
        QR_K8(24 * HALF) QR_K8(24 * HALF + 8) QR_K8(24 * HALF + 16)
#undef QR_K8
#undef QR_K1
    }

    // The previous epilogue's candidates: lane i < npend holds entry i and the position its atomic returned.  A
    // position past the buffer is clamped (the slot's count says "overflow" and the exact fallback redoes the slot),
    // so the two stores are issued exactly when npend > 0 (the op counts above rely on it).
    __device__ __forceinline__ unsigned list_lds() const {
        return (unsigned)reinterpret_cast<uintptr_t>(smem) + (unsigned)(QR_RING * QR_SLOT) + (unsigned)w * 512u;
    }
    __device__ __forceinline__ int flush_pending(unsigned colb) {  // colb: first column of the wave's col-block (uniform)
        const unsigned col = colb;
        if (npend == 0) return 0;
        const unsigned l = lane_now();
        if ((int)l < npend) {
            uint2 e;
            asm volatile("ds_read_b64 %0, %1" : "=v"(e) : "v"(list_lds() + l * 8u) : "memory");
            wait_vm(since);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pend_pos), "+v"(e) :: "memory");  // both defined from here on (the atomic is
                                                                                        // inline asm: hipcc inserts no wait of its own)
            const unsigned pos = min(pend_pos, (unsigned)PF_CAP - 1u);
            const size_t o = (size_t)(col + (e.x >> 8)) * PF_CAP + pos;
            P.cand_row[o] = pend_row0 + (e.x & 255u);
            P.cand_s[o] = __uint_as_float(e.y);
        }
        npend = 0;
        return 2;
    }

    // the wave's 32 x 32 scores of row-block `row0 / 32` against its col-block (column = lane & 31: one threshold per lane);
    // returns the number of VM operations it issued
    __device__ __forceinline__ int epilogue(f32x16& acc, unsigned row0, int n_b, unsigned colb) {
        int ops = flush_pending(colb);
        QR_STAMP(8)
        const unsigned l = lane_now(), c = l & 31u, h = l >> 5;
        const unsigned col = colb + c;
        if (row0 + 32u > (unsigned)n_b) {  // the bucket's ragged end (zero-padded rows)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (row0 + acc_row(r, h) >= (unsigned)n_b) acc[r] = __builtin_nanf("");  // fails every >= (a bound may be -inf)
        }
        float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
        mx = fmaxf(mx, acc[15]);
        bool any = mx >= thr;
#ifdef LMI_ABL_NOEMIT
        any = any && thr == 12345.678f;
#endif
        if (__ballot(any) == 0ull) return ops;
        // compaction list of the wave (LDS, inline asm: hipcc would order a visible LDS access behind ALL pending LDS-DMA)
        int tot = 0;  // wave-uniform
        unsigned kb = (c << 8) | (4u * h);
        asm volatile("" : "+v"(kb));  // opaque: the 16 keys are formed where they are used, not held in 16 registers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pass = acc[r] >= thr;
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                if (pass) {
                    const int my = tot + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (my < 64) {
                        const uint2 e = make_uint2(kb + (unsigned)((r & 3) + 8 * (r >> 2)), __float_as_uint(acc[r]));
                        asm volatile("ds_write_b64 %0, %1" :: "v"(list_lds() + (unsigned)my * 8u), "v"(e) : "memory");
                    }
                }
                tot += (int)__popcll(mask);
            }
        }
        QR_STAMP(9)
        if (tot <= 64) {
            if ((int)l < tot) {
                unsigned ex;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ex) : "v"(list_lds() + l * 8u) : "memory");
                const unsigned* cnt = P.cand_cnt + (colb + (ex >> 8));
                const unsigned one = 1u;
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(pend_pos) : "v"(cnt), "v"(one) : "memory");
            }
            npend = tot;
            pend_row0 = row0;
            since = 0;
            ops += 1;
        } else {  // dense tile (tiny buckets, emit-all test hook): one returning atomic per candidate
            float t3 = thr;
            asm volatile("" : "+v"(t3));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] >= t3) {
                    const unsigned pos = atomicAdd(P.cand_cnt + col, 1u);
                    if (pos < (unsigned)PF_CAP) {
                        P.cand_row[(size_t)col * PF_CAP + pos] = row0 + acc_row(r, h);
                        P.cand_s[(size_t)col * PF_CAP + pos] = acc[r];
                    }
                }
            QR_WAITVM(0);  // op counts restart from a drained queue
            ops = 0;
            e1 = 0;
        }
        return ops;
    }

    // tile = col-blocks [cbt0, cbt0 + ncb_tile) of bucket b, chunk ch
    __device__ __forceinline__ void run(int b, int cbt0, int ncb_tile, int ch) {
#ifdef LMI_QR_STAMPS
        for (int i = 0; i < 12; ++i) st_acc[i] = 0;
        st_last = __builtin_readcyclecounter();
#endif
        const int tid = threadIdx.x;
        lane = tid & 63;
        const int c = lane & 31;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);
        npc = w < KG ? (KG - 1 - w) / 8 + 1 : 0;
        const bool split = ncb_tile <= 4;
        const int mycb = split ? (w & 3) : w;
        const int par = split ? (w >> 2) : 0;
        const bool active = mycb < ncb_tile;
        // item parameters: loaded by vector loads (uniform values in VGPRs) -> SGPRs, so that everything derived from them
        // (stream bases, trip counts) is scalar
        const int n_b = __builtin_amdgcn_readfirstlane(P.nb_rows[b]);
        const int rbs0 = __builtin_amdgcn_readfirstlane(P.rb_start[b]);
        const int cbs0 = __builtin_amdgcn_readfirstlane(P.cb_start[b]);
        const int m_b = __builtin_amdgcn_readfirstlane(P.m[b]);
        const int nrb_b = (n_b + 31) >> 5;
        const int rb0 = ch * P.chunk_rb;
        const int nrb = min(P.chunk_rb, nrb_b - rb0);
        const int cbg = cbs0 + cbt0 + (active ? mycb : 0);  // this wave's col-block (global)
        const unsigned colb = (unsigned)cbg * 32u;
        thr = INFINITY;
        if (active) {
            const uint4* qsrc = P.qfrag16 + ((size_t)cbg * KG) * 64 + lane;
#pragma unroll
            for (int k = 0; k < QR_KMAX; ++k) {
                if (K48 || k >= K0) {  // fragment j of the col-block lives in B[K0 + j]
                    const uint4 t = qsrc[(size_t)(k - K0) * 64];
                    B[k] = *reinterpret_cast<const half8*>(&t);
                }
            }
            if ((cbt0 + mycb) * 32 + c < m_b) thr = P.bound1[colb + c] - P.eps2[colb + c];
        }
        // every resident fragment is "used" here, once: hipcc's wait for these loads then sits here and not -- as
        // `s_waitcnt vmcnt(0)`, which also drains the whole LDS-DMA look-ahead -- before their first MFMA inside the loop
#pragma unroll
        for (int k = 0; k < QR_KMAX; ++k) asm volatile("" : "+v"(B[k]));
        asm volatile("" : "+v"(thr));
        pend_pos = 0u; pend_row0 = 0u;
        npend = 0; since = 0; e1 = 0; e2 = 0;
        const uint4* aslab = P.slab16 + ((size_t)(rbs0 + rb0) * KG) * 64;
        const size_t rbs = (size_t)KG * 64;       // uint4 per row-block in the slab
        constexpr size_t sls = QR_SLOT / 16;      // uint4 per ring slot
        uint4* ring = reinterpret_cast<uint4*>(smem);
        const unsigned ring_lds = (unsigned)reinterpret_cast<uintptr_t>(smem) + (unsigned)lane * 16u;  // generic LDS address: low 32 bits = offset
        dma_all(aslab, ring);
        if (nrb > 1) dma_all(aslab + rbs, ring + sls);
        // Half-steps h = 0 .. 2 nrb, one barrier each.  Waves 0-3 run row-block t in half-steps 2t, 2t+1, waves 4-7 (their
        // SIMD partners) in 2t+1, 2t+2: the partners are half a row-block apart, so one's epilogue, first fragment reads and
        // DMA issue fall into the other's MFMA stream instead of both pipes idling together.  A split tile's partners take
        // alternate row-blocks and need no offset.  The epilogue of a row-block runs at the start of the wave's NEXT
        // half-step, after the barrier.
        //   slot (t-1) % 3 is free once waves 4-7 have finished row-block t-1 (half-step 2t), so the pieces of t+2 go out in
        //   half-step 2t+1 and are waited for before the barrier of half-step 2t+4.
        const int o = (split || !LMI_QR_STAGGER) ? 0 : (w >> 2);
        f32x16 acc;
        half8 a[QR_NA];
        bool pend = false;        // acc holds a finished row-block whose epilogue has not run
        unsigned pend_r0 = 0u;
        QR_STAMP(5)
        for (int hs = 0; hs <= 2 * nrb; ++hs) {
            const int tt = hs >> 1;
            // before an even half-step 2t: the pieces of row-block t have landed.  Younger, in issue order: the epilogues of
            // the two previous half-steps and the pieces of t+1
            if (!(hs & 1) && tt < nrb) wait_vm(e2 + (tt + 1 < nrb ? npc : 0) + e1);
            QR_STAMP(0)
#ifndef LMI_ABL_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            QR_STAMP(1)
            const bool dma = (hs & 1) && tt + 2 < nrb;
            const uint4* src = aslab + (size_t)(tt + 2) * rbs;
            uint4* dst = ring + (size_t)((tt + 2) % QR_RING) * sls;
            e2 = e1;
            e1 = 0;
            if (dma) since += npc;
            if (pend) {
                e1 = epilogue(acc, pend_r0, n_b, colb);
                pend = false;
            }
            QR_STAMP(2)
            const int s = hs - o;
            const int t = s >> 1;
            if (active && s >= 0 && t < nrb && (!split || (t & 1) == par)) {
                const unsigned aA = ring_lds + (unsigned)(t % QR_RING) * (unsigned)QR_SLOT;
                if (s & 1) {
                    compute_half<1>(a, acc, aA, dma, src, dst);
                    pend = true;
                    pend_r0 = (unsigned)(rb0 + t) * 32u;
                } else {
                    compute_half<0>(a, acc, aA, dma, src, dst);
                }
                QR_STAMP(3)
            } else {
                if (dma) dma_all(src, dst);
                QR_STAMP(4)
            }
        }
        if (pend) epilogue(acc, pend_r0, n_b, colb);
        QR_WAITVM(0);
        flush_pending(colb);
        QR_WAITVM(0);
        __syncthreads();  // the ring and the lists are reused by the next item
#ifdef LMI_QR_STAMPS
        QR_STAMP(6)
        st_acc[7] = (unsigned long long)nrb;
        if (lane == 0) {
            unsigned long long* g = P.stamps + ((split ? 1 : 0) * 8 + w) * 12;
            for (int i = 0; i < 12; ++i) atomicAdd(g + i, st_acc[i]);
        }
#endif
    }
};

template <bool K48>
__global__ __launch_bounds__(64 * QR_WAVES, 1) void pass2_qr_kernel(PrefilterParams P) {
    extern __shared__ __attribute__((aligned(16))) char qr_smem[];
    const int KG = K48 ? QR_KMAX : P.KG16;
    int* s_item = reinterpret_cast<int*>(qr_smem + (size_t)QR_RING * QR_SLOT + QR_WAVES * 64 * 8);
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
                        int lo = 0, hi = P.grp_n[grp];
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (base[mid] <= it) lo = mid; else hi = mid;
                        }
                        b = P.grp_bucket[grp * P.L + lo];
                        local = it - base[lo];
                        break;
                    }
                }
                grp = (grp + 1) & (NGRP - 1);
            }
            s_item[0] = b;
            s_item[1] = local;
        }
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(s_item[0]), local = __builtin_amdgcn_readfirstlane(s_item[1]);
        __syncthreads();
        if (b < 0) return;
        // query tiles of the bucket: 8 col-blocks each, the last one takes the remainder (the item count
        // ceil(col-blocks / 8) x chunks is the one route_scan_kernel / route_group_kernel queued)
        const int ncb_b = (__builtin_amdgcn_readfirstlane(P.m[b]) + 31) >> 5;
        const int nqt = (ncb_b + QR_WAVES - 1) / QR_WAVES;
        const int qt = local % nqt, ch = local / nqt;
        const int cbt0 = qt * QR_WAVES;
        QrItem<K48> it{P, qr_smem, KG, QR_KMAX - KG};
        it.run(b, cbt0, min(QR_WAVES, ncb_b - cbt0), ch);
    }
}

}  // namespace lmi
