// lmi_pass2.h -- the prefilter's scan kernel (gfx950): ONE query tile per bucket chunk, vectors straight into registers.
//
// Tile (round 3): a block tile is 256 vectors x <= 384 queries (12 col-blocks) and a WAVE owns one 32-vector row-block x ALL
// the tile's col-blocks (12 x 16 = 192 accumulator registers): a bucket chunk is one tile whenever <= 384 queries are routed to
// the bucket, every wave does the same work for any number of col-blocks, and a 32-deep stage carries 2 NCB MFMAs per wave.
//
// MFMA shape (round 4): v_mfma_f32_16x16x32_f16 -- the 32 x 32 block of a wave and a col-block is four 16 x 16 tiles, a stage (32 k) is
// one k-step; the index slab and the query fragments are packed as 16-row x 32-k fragments for it (convert16_kernel /
// pack_queries16_kernel, lmi_prefilter.h).  Same flops, bytes and registers as round 3's 32x32x16 form, half the accumulator
// read-modify-write per MAC: pass 2 is power-limited and the chip holds a ~12 % higher clock on this shape; with the fragment
// stream as one asm statement per query fragment (wait, next read, MFMA pair) the net is -4 % time at C2.
//
// Operand paths (round 4).  A wave's vector fragments are PRIVATE to it (its row-block x the stage's two row halves = 2 KiB), so
// they need neither LDS nor a barrier: every wave loads them with two global_load_dwordx4 straight into registers, two stages
// ahead, into one of three 8-register sets, and waits for them with a counted vmcnt.  hipcc cannot express a load that stays
// in flight into registers it allocates (round 3, profiles/r03_pass2_experiments.txt section 21: tied waits are preceded by
// copies, untied ones let the allocator reuse the registers, visible loads are waited for with vmcnt(0)), so the sets live in
// v[232:255], which the kernel RESERVES from the compiler: __attribute__((amdgpu_num_vgpr(116))) -- the attribute counts the
// VGPR and AGPR halves of gfx90a+'s unified file, 116 -> hipcc allocates v0..v231 only (it spills rather than touch the rest;
// round 3 passed the full count, which the backend rejects as over the occupancy limit and silently ignores).  The loads and
// the MFMAs that read the sets are inline asm naming those registers; everything else is ordinary HIP.
// The query fragments are shared by the 8 waves and re-streamed (from L2: the XCD-affine queues keep a bucket's tiles on one
// XCD) for every 256 vectors: LDS ring of three 24-KiB stages filled by LDS-DMA from the younger half of the block, read with
// ds_read_b128 D fragments ahead of the MFMAs as ONE stream that continues across the stage barrier (the last D MFMAs of a
// stage are issued after the next stage's barrier, interleaved with its first reads).
// (Why not everything through LDS-DMA as in round 3: tools/micro/stream_paths.hip / profiles/r04_pass2_experiments.txt -- the
// two streams overlap either way once the query tiles are L2-resident; what the register path removes is 16 of the loader
// waves' 16 + 2 NCB pieces per stage, the vector fragments' LDS reads, and 48 KiB of ring that a deeper query ring can use.)
// Pass 1 (SAMPLE) = the same tile on sampled tiles only; one item per (bucket, query tile, sampled tile); per lane and
// column the MAXIMUM of its 8 scores is all it keeps: the slot maxima of a column come from disjoint rows, so the 10th
// largest of them is the 10th best of a subset of the bucket = a valid lower bound of That (lmi_prefilter.h, header).
#pragma once
#include <type_traits>

#include "lmi_prefilter.h"

namespace lmi {

constexpr int P2_G = 2;                                                  // k16-groups per stage (32 k)
constexpr int P2_MAXCB = 12;                                             // col-blocks per tile
// Waves per block = row-blocks per BLOCK tile.  8: one block per CU (both waves of a SIMD belong to it and meet at its stage barrier);
// 4: two blocks per CU, a SIMD hosts one wave of each -- what one block loses at its stage barrier or in its epilogue the other can
// use, at the price of every query fragment staged once per block, i.e. twice per CU (profiles/r04_pass2_experiments.txt).
#ifndef LMI_P2_WAVES
#define LMI_P2_WAVES 8
#endif
constexpr int P2_WAVES = LMI_P2_WAVES;
static_assert(P2_WAVES == 4 || P2_WAVES == 8, "");
constexpr int P2_BLOCKS_PER_CU = 8 / P2_WAVES;
constexpr int P2_TILE_RB = 8;                                            // row-blocks per sampled tile / chunk granule (256 vectors)
constexpr int P2_TILE_ROWS = 32 * P2_TILE_RB;
constexpr int P2_SLOT_BYTES = P2_MAXCB * P2_G * 1024;                    // 24 KiB per ring slot: the stage's query fragments
constexpr int P2_RING = 3;
constexpr int P2_NSL = 16;   // pass 1, low-dimensional kernels: lists per column (sampled tile j -> list j % P2_NSL), each 16 slot maxima
constexpr int P2_NSL_BIG = 8;   // pass 1, pass2_kernel: 8 lists of 32 slot maxima (the 16x16 MFMA tile gives a lane 8 rows of a column, not 16)
static_assert(P2_NSL * 16 == P2_NSL_BIG * 32, "bound_merge2_kernel reads 256 values per column either way");
constexpr int P2_LIST = 64 + 1;
// The vector-fragment register sets: v[P2_AREG0 + 8 s + 4 h .. + 3] = row half h of the stage in ring slot s.  hipcc is kept
// below P2_AREG0 by the kernels' amdgpu_num_vgpr attribute (= P2_AREG0 / 2: the attribute is per register-file half).
constexpr int P2_AREG0 = 232;
#define LMI_P2_NUM_VGPR_ATTR 116
static_assert(2 * LMI_P2_NUM_VGPR_ATTR == P2_AREG0 && P2_AREG0 + 8 * P2_RING == 256, "three 8-register sets at the top of the file");

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));   // asm operands must be vector types, not HIP's uint2 / uint4 structs
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int p2_sample_tiles(int n_b, int smax) { return sample_tiles256(n_b, smax); }
static_assert(P2_TILE_RB == 8, "sample_tiles256 counts 256-row tiles");

// In-kernel phase timing (developer builds, -DLMI_P2_STAMPS; tools/p2_stamps.py): shader-clock cycles per wave and phase,
// summed into P.stamps as u64 [8 waves][12]: 0 landed-wait (vmcnt), 1 barrier, 2 stage (reads + MFMAs + DMA issue),
// 3 epilogue, 4 item start, 5 item end, 7 = tiles.
#ifdef LMI_P2_STAMPS
#ifndef LMI_P2_STAMPS_SAMPLE
#define LMI_P2_STAMPS_SAMPLE 0
#endif
#define P2_STAMP(PH) if (SAMPLE == (LMI_P2_STAMPS_SAMPLE != 0)) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[PH] += t_ - st_last; st_last = t_; }
#else
#define P2_STAMP(PH)
#endif
// a stamp is s_memtime + s_waitcnt lgkmcnt(0): three per stage drain the fragment read-ahead and cost a third of the kernel's
// time (measured); the default stamped build stamps once per tile phase, -DLMI_P2_STAMPS_FINE adds the per-stage ones
#if defined(LMI_P2_STAMPS) && defined(LMI_P2_STAMPS_FINE)
#define P2_STAMP_FINE(PH) P2_STAMP(PH)
#else
#define P2_STAMP_FINE(PH)
#endif

template <int NCB, bool SAMPLE>
struct Tile2 {
    static constexpr int G = P2_G;
    static constexpr int Q = G * NCB;                       // B fragments = MFMAs per stage and wave
#ifndef LMI_P2_BR
#define LMI_P2_BR 3
#endif
    static constexpr int BR = LMI_P2_BR;                    // B-fragment register ring: fragment q of the stage in ring slot s sits in
                                                            // register (s Q + q) mod BR -- 3 Q = 6 NCB is a multiple of 3 and of 6, so the
                                                            // numbering runs on through the three unrolled stages and closes at the back edge
    static_assert(BR == 3 || BR == 6, "3 Q must be a multiple of BR for every NCB");
    static constexpr int D = NCB < BR - 1 ? NCB : BR - 1;   // read-ahead (fragments); the last D MFMAs of a stage are deferred
    // LDS-DMA (query fragments only) is issued by the YOUNGER half of the block (waves 4..7: "loaders"): the two waves of a SIMD
    // run a stage one after the other (the matrix pipe goes to the older wave until it has issued all of its MFMAs: stamps,
    // profiles/r03_*), so the younger wave issues the stage's pieces at its start, in the shadow of its partner's MFMAs.
    // 2 NCB one-KiB pieces per stage; loader lw takes k-group lw & 1 of col-blocks (lw >> 1), + 2, + 4, ..: PBL = ceil(NCB / 2)
    // pieces each (a loader past the tile's last col-block repeats it: 4 PBL - 2 NCB is 0 or 2 pieces).
    static constexpr int PBL = (NCB + 1) / 2;
    static constexpr int PA = G;                            // vector-fragment loads per wave and stage (one per k-group)
    static_assert(G == 2, "written out for two k-groups per stage");

    const PrefilterParams& P;
    unsigned ring;       // LDS byte address of the ring
    uint4* ring_p;       // the same as a pointer (DMA destinations)
    uint2* sList;        // [8 waves][P2_LIST] candidate compaction lists (pass 2)
    float* sThr;         // [P2_MAXCB * 32] emission thresholds of the tile's columns (pass 2)
    uint4* sPend;        // [8 waves][64] (column, row, score) of the candidate whose position atomic is in flight (pass 2)
    int lane, w;
    unsigned lds_lane;   // ring + lane * 16: the lane's LDS read address AND its byte offset in the vector loads (whose SGPR base has
                         // `ring` subtracted): one register for both
    // acc[n][rh][ch]: the 16 x 16 block (row half rh, column half ch) of col-block n; a lane holds column 16 ch + (lane & 15) and rows
    // 16 rh + 4 (lane >> 4) + 0..3 of it (the C/D map of v_mfma_f32_16x16x32_f16)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v acc[NCB][2][2];
    half8 b[BR];         // query-fragment ring
    unsigned pend_pos;
#ifdef LMI_P2_STAMPS
    unsigned long long st_acc[12], st_last;
#endif

    static constexpr int breg(int slot, int q) { return (slot * Q + q) % BR; }
    static constexpr int areg(int set, int g) { return P2_AREG0 + 8 * set + 4 * g; }
    // LDS reads issued after B(q)'s and before MFMA q waits for it: the younger B fragments
    static constexpr int younger(int q) {
#if defined(LMI_ABL_NOLDSB) || defined(LMI_ABL_NOLGKM) || defined(LMI_ABL_NOLDS)
        return 15;   // timing-only ablations: never wait for a fragment
#endif
        return (q + D - 1 < Q - 1 ? q + D - 1 : Q - 1) - q;
    }

    template <int OFF>
    static __device__ __forceinline__ void lds_rd(half8& r, unsigned addr) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    }
    template <int N>
    static __device__ __forceinline__ void lgkm_wait(half8& y) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(y) : "n"(N) : "memory");
    }

    // Shape (round 4): v_mfma_f32_16x16x32_f16 -- a stage (32 k) is ONE k-step; the wave's 32 x 32 block of col-block n is four
    // 16 x 16 MFMA tiles, its vector fragments are the two row halves (set registers +0: rows 0..15, +4: rows 16..31), a query
    // fragment is one column half of a col-block (16 columns x 32 k) and feeds two MFMAs.  Same flops, bytes and registers as the
    // 32x32x16 form of round 3, half the accumulator read-modify-write per MAC: the kernel is POWER-limited
    // (profiles/r04_pass2_experiments.txt) and the chip holds a ~15 % higher clock on this shape.
    // The MFMAs read their vector fragments from the reserved set by NAME (hipcc knows nothing of v[232:255]); being volatile asm
    // they keep their program order against the fragment reads, the waits and the loads.  Neighbouring MFMAs go to different
    // accumulators: no wait states needed between them; the tile's epilogue is preceded by the states an MFMA result needs
    // before a VALU read (end of run()'s K loop).
    template <int SLOT, int QI>
    __device__ __forceinline__ void mfma_q() {   // query fragment QI = (column half QI / NCB, col-block QI % NCB)
        constexpr int ch = QI / NCB, n = QI % NCB, R = areg(SLOT, 0);
        asm volatile("v_mfma_f32_16x16x32_f16 %0, v[%c3:%c4], %2, %0\n\tv_mfma_f32_16x16x32_f16 %1, v[%c5:%c6], %2, %1"
                     : "+v"(acc[n][0][ch]), "+v"(acc[n][1][ch]) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7));
    }
    // the tile's first stage starts the accumulators at 0 (srcC = the inline constant: no clearing pass, no zero registers)
    template <int SLOT, int QI>
    __device__ __forceinline__ void mfma_q0() {
        constexpr int ch = QI / NCB, n = QI % NCB, R = areg(SLOT, 0);
        asm volatile("v_mfma_f32_16x16x32_f16 %0, v[%c3:%c4], %2, 0\n\tv_mfma_f32_16x16x32_f16 %1, v[%c5:%c6], %2, 0"
                     : "=&v"(acc[n][0][ch]), "=&v"(acc[n][1][ch]) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7));
    }
    template <int SLOT, int QI>
    __device__ __forceinline__ void read_b() {  // query fragment QI of the stage in ring slot SLOT
        constexpr int g = QI / NCB, n = QI % NCB;
#ifdef LMI_ABL_NOLDSB   // timing-only ablation: a quarter of the query-fragment reads (garbage results)
        if constexpr (QI % 4 != 0) { asm volatile("" : "+v"(b[breg(SLOT, QI)])); return; }
#endif
#ifdef LMI_ABL_NOLDS    // timing-only ablation: no fragment reads at all
        asm volatile("" : "+v"(b[breg(SLOT, QI)])); return;
#endif
        lds_rd<(n * G + g) * 1024>(b[breg(SLOT, QI)], lds_lane + SLOT * P2_SLOT_BYTES);
    }

    // One step of the fragment stream as ONE asm statement: [wait for fragment QI] [request fragment RQ of slot RSLOT] [the two MFMAs
    // of fragment QI].  The read goes out BEFORE the MFMAs: the second MFMA waits ~16 cycles for the matrix pipe, and an in-order
    // wave cannot issue the read behind it until then; and hipcc's one-state pad between a wait statement and an MFMA statement
    // is gone (PMC, round 4: the 16x16x32 stream needed 13 % more cycles than the 32x32x16 one for the same flops).
    //   WAIT < 0: no wait (the previous stage's deferred fragments: the stage barrier's lgkmcnt(0) covered them)
    //   RQ < 0: no read (the stage's last D fragments)
    template <int SLOT, int QI, int WAIT, bool ZERO, int RSLOT, int RQ>
    __device__ __forceinline__ void step() {
        constexpr int ch = QI / NCB, n = QI % NCB, R = areg(SLOT, 0);
#if defined(LMI_ABL_NOLDS) || defined(LMI_ABL_NOLDSB)
        if constexpr (WAIT >= 0) lgkm_wait<WAIT>(b[breg(SLOT, QI)]);
        if constexpr (ZERO) mfma_q0<SLOT, QI>(); else mfma_q<SLOT, QI>();
        if constexpr (RQ >= 0) read_b<RSLOT, RQ>();
        return;
#endif
#define P2_MFMA_ACC "v_mfma_f32_16x16x32_f16 %0, v[%c4:%c5], %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, v[%c6:%c7], %3, %1"
#define P2_MFMA_ZERO "v_mfma_f32_16x16x32_f16 %0, v[%c4:%c5], %3, 0\n\tv_mfma_f32_16x16x32_f16 %1, v[%c6:%c7], %3, 0"
        if constexpr (RQ >= 0) {
            constexpr int rg = RQ / NCB, rn = RQ % NCB;
            constexpr int OFF = (rn * G + rg) * 1024;
            const unsigned addr = lds_lane + RSLOT * P2_SLOT_BYTES;
            half8& bn = b[breg(RSLOT, RQ)];
            static_assert(breg(RSLOT, RQ) != breg(SLOT, QI), "the read's destination is the MFMAs' operand");
            if constexpr (WAIT >= 0) {
                if constexpr (ZERO)
                    asm volatile("s_waitcnt lgkmcnt(%c10)\n\tds_read_b128 %2, %8 offset:%c9\n\t" P2_MFMA_ZERO
                                 : "=&v"(acc[n][0][ch]), "=&v"(acc[n][1][ch]), "=&v"(bn) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "v"(addr), "n"(OFF), "n"(WAIT) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(%c10)\n\tds_read_b128 %2, %8 offset:%c9\n\t" P2_MFMA_ACC
                                 : "+v"(acc[n][0][ch]), "+v"(acc[n][1][ch]), "=&v"(bn) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "v"(addr), "n"(OFF), "n"(WAIT) : "memory");
            } else {
                if constexpr (ZERO)
                    asm volatile("ds_read_b128 %2, %8 offset:%c9\n\t" P2_MFMA_ZERO
                                 : "=&v"(acc[n][0][ch]), "=&v"(acc[n][1][ch]), "=&v"(bn) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "v"(addr), "n"(OFF) : "memory");
                else
                    asm volatile("ds_read_b128 %2, %8 offset:%c9\n\t" P2_MFMA_ACC
                                 : "+v"(acc[n][0][ch]), "+v"(acc[n][1][ch]), "=&v"(bn) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "v"(addr), "n"(OFF) : "memory");
            }
        } else {
            static_assert(WAIT >= 0, "");
            half8 dummy;   // (operand numbering shared with the forms above)
            if constexpr (ZERO)
                asm volatile("s_waitcnt lgkmcnt(%c8)\n\t" P2_MFMA_ZERO
                             : "=&v"(acc[n][0][ch]), "=&v"(acc[n][1][ch]), "=&v"(dummy) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "n"(WAIT) : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(%c8)\n\t" P2_MFMA_ACC
                             : "+v"(acc[n][0][ch]), "+v"(acc[n][1][ch]), "=&v"(dummy) : "v"(b[breg(SLOT, QI)]), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7), "n"(WAIT) : "memory");
        }
#undef P2_MFMA_ACC
#undef P2_MFMA_ZERO
    }

    __device__ __forceinline__ bool is_loader() const { return P2_WAVES == 4 ? true : w >= 4; }
    struct Stream {        // wave-uniform source pointers of the NEXT stage to load
        const char* a;     // the wave's row-block of the tile, k-group pair t, MINUS `ring` (the lane offset register is lds_lane)
        const uint4* b0;   // the tile's first col-block, k-group pair t (loaders)
        int bo[PBL];       // offsets (uint4) of this loader's pieces from b0: k-group lw & 1 of col-blocks (lw >> 1) + 2 j (clamped)
    };
    // this wave's vector fragments of the stage S points at -> register set SET (both k-groups; 1 KiB apart in the slab).
    // The leading s_nop: the SGPR base may come fresh from a v_readlane (an SGPR spill reload) -- VALU-written SGPR -> VMEM
    // address needs 5 wait states and hipcc pads nothing in front of an asm statement's contents.
    template <int SET>
    __device__ __forceinline__ void load_a(const Stream& S) {
#ifdef LMI_ABL_NOLOAD
        return;
#endif
        constexpr int R = areg(SET, 0);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c3], %0, %1\n\tglobal_load_dwordx4 v[%c4:%c5], %0, %1 offset:1024"
                     :: "v"(lds_lane), "s"(S.a), "n"(R), "n"(R + 3), "n"(R + 4), "n"(R + 7) : "memory");
    }
    template <int DST>
    __device__ __forceinline__ void dma_b(const Stream& S) {   // loaders: their pieces of the stage S points at -> ring slot DST
#ifdef LMI_ABL_NOLOAD
        return;
#endif
        uint4* slot = ring_p + DST * (P2_SLOT_BYTES / 16);
        const int lw = w & 3;
        static_for<0, PBL>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int cb = min((lw >> 1) + 2 * j, NCB - 1);
            glds16(reinterpret_cast<const float4*>(S.b0 + S.bo[j] + lane), reinterpret_cast<float4*>(slot + (cb * G + (lw & 1)) * 64));
        });
    }

    // One stage: ring slot / register set SLOT computes, stage (+2) is requested into slot / set DST.  `pend`: the previous
    // stage left its last D MFMAs (they read set DST's k-group 1 and the b[] ring) to be issued here -- the vector loads into
    // DST follow them; `last`: this stage issues all of its own (tile end / dead stage next).
    template <int SLOT, int DST>
    __device__ __forceinline__ void stage(const Stream& S, bool loader, bool pend, bool last, bool first, bool prev_first) {
        constexpr int PS = (SLOT + P2_RING - 1) % P2_RING;   // the previous stage's slot: its fragments' register numbering
        static_assert(PS == DST, "ring of three: the stage two ahead reuses the previous stage's slot");
        if (loader) dma_b<DST>(S);
        static_for<0, D>([&](auto i) {
            constexpr int I = decltype(i)::value;
            if (pend) {   // (every accumulator gets ONE MFMA per stage: all of the tile's first stage starts from 0, its deferred ones too)
                if constexpr (PS == 0) {
                    if (prev_first) step<PS, Q - D + I, -1, true, SLOT, I>(); else step<PS, Q - D + I, -1, false, SLOT, I>();
                } else {
                    step<PS, Q - D + I, -1, false, SLOT, I>();
                }
            } else {
                read_b<SLOT, I>();
            }
        });
        load_a<DST>(S);
        static_for<0, Q - D>([&](auto qi) {
            constexpr int q = decltype(qi)::value;
            if constexpr (SLOT == 0) {
                if (first) step<SLOT, q, younger(q), true, SLOT, q + D>(); else step<SLOT, q, younger(q), false, SLOT, q + D>();
            } else {
                step<SLOT, q, younger(q), false, SLOT, q + D>();
            }
        });
        if (last) {
            static_for<Q - D, Q>([&](auto qi) {
                constexpr int q = decltype(qi)::value;
                if constexpr (SLOT == 0) {   // (one-stage tiles)
                    if (first) step<SLOT, q, 0, true, -1, -1>(); else step<SLOT, q, 0, false, -1, -1>();
                } else {
                    step<SLOT, q, 0, false, -1, -1>();
                }
            });
        }
    }

    // Nothing derived from the lane number stays live across the K loop (192 accumulators + 28 operand registers + 3
    // addresses fill the file; a spilled value is a scratch reload = vector memory = `s_waitcnt vmcnt(0)` = the ring's
    // look-ahead drained): the epilogues recompute it behind an opaque zero.
    static __device__ __forceinline__ int lane_id() {
        unsigned z = 0u;
        asm volatile("" : "+v"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    }
    // A candidate's position atomic (previous tile) is OLDER than every load of the tile that has run since (24 stages x
    // (PA [+ PBL]) requests): once at most the two stages in flight are outstanding it has returned.  Written as asm so that
    // hipcc does not put `s_waitcnt vmcnt(0)` in front of the first use.  Column, row and score of that candidate wait in
    // the wave's LDS table (entry = lane), not in registers; LDS is touched by asm only (hipcc orders every LDS access it
    // sees behind ALL pending LDS-DMA).
    __device__ __forceinline__ void flush_pending(int ln) {
        // (two stages of the next tile are in flight: 2 PA vector loads, the loaders' 2 PBL pieces on top)
        if (is_loader()) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pend_pos) : "n"(2 * (PA + PBL)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pend_pos) : "n"(2 * PA) : "memory");
        if (pend_pos != 0xffffffffu) {
            u32x4 e;
            const unsigned pa = (unsigned)reinterpret_cast<uintptr_t>(sPend + w * 64) + (unsigned)ln * 16u;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(pa) : "memory");
            cand_store(P, (size_t)e.x, pend_pos, e.y, __uint_as_float(e.z));   // (past the column's buffer: the shared overflow log)
        }
        pend_pos = 0xffffffffu;
    }

    // pass 2: rows with shat >= threshold -> the slot's candidate buffer (compaction through a wave-private LDS list, one
    // position atomic per candidate whose stores go out at the NEXT tile end; > 64 candidates in the tile: direct path).
    // Lane (quad = lane >> 4, c16 = lane & 15) holds, per col-block n and column half ch, column 16 ch + c16 and the 8 rows
    // 16 rh + 4 quad + r (rh = 0 / 1, r = 0..3).
    __device__ __forceinline__ void epilogue_emit(int rb_tile0, int n_b, size_t col0) {
        const int ln = lane_id();
        const int quad = ln >> 4, c16 = ln & 15;
#ifdef LMI_ABL_NOEPI   // timing-only ablation: the tile's scores are never tested (no candidates: wrong results)
#pragma unroll
        for (int n = 0; n < NCB; ++n) asm volatile("" :: "v"(acc[n][0][0]), "v"(acc[n][1][0]), "v"(acc[n][0][1]), "v"(acc[n][1][1]));
        return;
#endif
        const unsigned row0 = (unsigned)((rb_tile0 + w) * 32);
        if (row0 + 32u > (unsigned)n_b) {  // wave-uniform: the bucket's ragged end (zero-padded / clamped rows never pass)
            int lim = n_b - (int)row0 - 4 * quad;   // lane's rows 16 rh + r at or past `lim` are beyond the bucket
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                asm volatile("" : "+v"(lim));    // opaque per col-block: hoisted lane masks would cost SGPRs in the K loop
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * rh + r >= lim) { acc[n][rh][0][r] = __builtin_nanf(""); acc[n][rh][1][r] = __builtin_nanf(""); }
            }
        }
        const unsigned list_addr = (unsigned)reinterpret_cast<uintptr_t>(sList + w * P2_LIST);   // wave-uniform
        int tot = 0;  // wave-uniform
        const unsigned kb = (unsigned)((c16 << 8) | (4 * quad));   // key = column in the tile << 8 | row in the row-block
        const unsigned thr_addr = (unsigned)reinterpret_cast<uintptr_t>(sThr) + (unsigned)c16 * 4u;
        // thresholds in batches of TB col-blocks (2 TB registers; all 2 NCB at once do not fit beside 192 accumulators): one LDS round
        // trip per batch, the next batch requested before the current one is tested
        constexpr int TB = 2;
        float thr[2][TB][2];
        auto thr_load = [&](int nb0, float (&t)[TB][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const int n = min(nb0 + i, NCB - 1);
                asm volatile("ds_read_b32 %0, %1" : "=v"(t[i][0]) : "v"(thr_addr + (unsigned)(n * 128)) : "memory");
                asm volatile("ds_read_b32 %0, %1" : "=v"(t[i][1]) : "v"(thr_addr + (unsigned)(n * 128 + 64)) : "memory");
            }
        };
        thr_load(0, thr[0]);
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            constexpr int dummy = 0; (void)dummy;
            if (n % TB == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TB; ++i) asm volatile("" : "+v"(thr[(n / TB) & 1][i][0]), "+v"(thr[(n / TB) & 1][i][1]));   // defined from here on
                if (n + TB < NCB) thr_load(n + TB, thr[((n / TB) + 1) & 1]);
            }
            // About one score in a thousand passes (~40 candidates per column and bucket): four col-blocks in ten hold one.  Maxima of the
            // four (row half, column half) groups of 4 values first, ONE branch per col-block: a col-block without a candidate costs
            // 11 max + 2 compares.  v_max3 returns the other operands for a NaN.
            float gm[2][2], t0, mh[2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(acc[n][rh][ch][0]), "v"(acc[n][rh][ch][1]), "v"(acc[n][rh][ch][2]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(gm[ch][rh]) : "v"(t0), "v"(acc[n][rh][ch][3]));
                }
                asm("v_max_f32 %0, %1, %2" : "=v"(mh[ch]) : "v"(gm[ch][0]), "v"(gm[ch][1]));
            }
            const float th0 = thr[(n / TB) & 1][n % TB][0], th1 = thr[(n / TB) & 1][n % TB][1];
            bool any = (mh[0] >= th0) | (mh[1] >= th1);
#ifdef LMI_ABL_NOEMIT
            any = any && th0 == 12345.678f;
#endif
            if (__builtin_expect(__ballot(any) == 0ull, 1)) continue;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const float th = ch ? th1 : th0;
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    if (__ballot(gm[ch][rh] >= th) == 0ull) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool pass = acc[n][rh][ch][r] >= th;  // thr = +inf for idle columns, NaN scores never pass
                        const unsigned long long mask = __ballot(pass);
                        if (__builtin_expect(mask != 0ull, 0)) {
                            if (pass) {
                                const int my = tot + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                const unsigned key = kb + (unsigned)(((n * 32 + 16 * ch) << 8) | (16 * rh + r));
                                const unsigned la = list_addr + (unsigned)min(my, 64) * 8u;
                                const unsigned sb = __float_as_uint(acc[n][rh][ch][r]);
                                const u32x2 ent = {key, sb};
                                asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(ent) : "memory");
                            }
                            tot += (int)__popcll(mask);
                        }
                    }
                }
            }
        }
        // the PREVIOUS tile's candidates (their position atomics have long returned) leave here, behind the tests: the accumulators are
        // dead now, the stores' / the overflow log's temporaries cost the epilogue no spill; the wave's sPend entries are free after it
        flush_pending(ln);
        if (tot > 0 && tot <= 64) {
            if (ln < tot) {
                u32x2 e;
                const unsigned la = list_addr + (unsigned)ln * 8u;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(la) : "memory");
                const unsigned pcol = (unsigned)(col0 + (e.x >> 8));
                const u32x4 pe = {pcol, row0 + (e.x & 255u), e.y, 0u};
                const unsigned pa = (unsigned)reinterpret_cast<uintptr_t>(sPend + w * 64) + (unsigned)ln * 16u;
                asm volatile("ds_write_b128 %0, %1" :: "v"(pa), "v"(pe) : "memory");
#ifdef LMI_ABL_NOATOMIC
                pend_pos = (unsigned)ln;
#else
                const unsigned* cnt_addr = P.cand_cnt + pcol;
                const unsigned one = 1u;
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(pend_pos) : "v"(cnt_addr), "v"(one) : "memory");
#endif
            }
        } else if (tot > 64) {
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float th = sThr[n * 32 + 16 * ch + c16];
                    unsigned rowq = row0 + 4u * (unsigned)quad;
                    asm volatile("" : "+v"(rowq));   // the row numbers are formed where they are used, not kept across the K loop
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (acc[n][rh][ch][r] >= th) {
                                const size_t col = col0 + n * 32 + 16 * ch + c16;
                                const unsigned pos = atomicAdd(P.cand_cnt + col, 1u);
                                cand_store(P, col, pos, rowq + (unsigned)(16 * rh + r), acc[n][rh][ch][r]);
                            }
                }
            }
        }
    }

    // pass 1: per lane, col-block and column half the maximum of its 8 scores -> bound[list][slot = 4 (row-block in the tile) + quad][column]
    // (32 slots per sampled tile, each the best of 8 rows; P2_NSL_BIG lists of them = the same 256 values per column that the
    // low-dimensional kernels' 16 lists x 16 slots give bound_merge2_kernel)
    __device__ __forceinline__ void epilogue_sample(int rb_tile0, int n_b, size_t col0, int m_left, int list_j, bool use_atomic, int rb_in_tile) {
        const int ln = lane_id();
        const int quad = ln >> 4, c16 = ln & 15;
        const unsigned row0 = (unsigned)((rb_tile0 + w) * 32);
        const bool ragged = row0 + 32u > (unsigned)n_b;
        int lim = ragged ? n_b - (int)row0 - 4 * quad : 64;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            if (ragged) {
                asm volatile("" : "+v"(lim));
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * rh + r >= lim) { acc[n][rh][0][r] = -INFINITY; acc[n][rh][1][r] = -INFINITY; }
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float m0, m1, m2;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(acc[n][0][ch][0]), "v"(acc[n][0][ch][1]), "v"(acc[n][0][ch][2]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(acc[n][1][ch][0]), "v"(acc[n][1][ch][1]), "v"(acc[n][1][ch][2]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(m0), "v"(m1), "v"(acc[n][0][ch][3]));
                asm("v_max_f32 %0, %1, %2" : "=v"(m2) : "v"(m2), "v"(acc[n][1][ch][3]));
                const float mx = m2;
                if (n * 32 + 16 * ch + c16 < m_left) {
                    // lists are COLUMN-minor: a wave's store is four runs of 16 consecutive floats
                    float* dst = P.bound + ((size_t)(list_j * 32 + (rb_in_tile + w) * 4 + quad)) * (size_t)P.ncols + (col0 + n * 32 + 16 * ch + c16);
                    if (!use_atomic) *dst = mx;
                    else {  // more sampled tiles than lists: monotone float max through the order-preserving integer image
                        if (mx >= 0.0f) atomicMax(reinterpret_cast<int*>(dst), __float_as_int(mx));
                        else atomicMin(reinterpret_cast<unsigned*>(dst), __float_as_uint(mx));
                    }
                }
            }
        }
    }

    // The tile = col-blocks [cbt0, cbt0 + NCB) of bucket b.  !SAMPLE: `ch` = chunk of the bucket, every tile of it;
    // SAMPLE: `ch` = sampled tile j of the bucket (tile j * stride), one tile.
    template <typename Q>
    __device__ __forceinline__ void run(int b_, int cbt0_, int ch_, int m_use_, int nt_, Q& queue) {
        // the item came through LDS (s_item): tell the compiler it is wave-uniform, so that every address derived from it is
        // scalar (the vector loads take their base as an "s" operand)
        const int b = __builtin_amdgcn_readfirstlane(b_), cbt0 = __builtin_amdgcn_readfirstlane(cbt0_);
        const int ch = __builtin_amdgcn_readfirstlane(ch_), m_use = __builtin_amdgcn_readfirstlane(m_use_);
        const int nt = __builtin_amdgcn_readfirstlane(nt_);
#ifdef LMI_P2_STAMPS
        for (int i = 0; i < 12; ++i) st_acc[i] = 0;
        st_last = __builtin_readcyclecounter();
#endif
        const int tid = threadIdx.x;
        lane = tid & 63;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);
        lds_lane = ring + (unsigned)lane * 16u;
        const int KG = P.KG16, NS = KG / G;
        const int n_b = P.nb_rows[b];
        const int nrb_b = (n_b + 31) >> 5;
        const int stride = SAMPLE ? sample_stride(n_b, P.sample_max) : 1;
        const int crb = (!SAMPLE && P.chunk_rb_b) ? P.chunk_rb_b[b] : P.chunk_rb;
        const int rb0 = SAMPLE ? ch * stride * P2_TILE_RB : ch * crb;
        const int nrb_all = SAMPLE ? P2_TILE_RB : min(crb, nrb_b - rb0);
        // block tiles of P2_WAVES row-blocks.  pass 1: the item's nt sampled tiles, each VPT block tiles, 2 stride tiles apart (row-blocks
        // past the bucket's end are clamped re-reads, masked in the epilogue)
        constexpr int VPT = P2_TILE_RB / P2_WAVES;
        const int nvt = SAMPLE ? nt * VPT : (nrb_all + P2_WAVES - 1) / P2_WAVES;
        const int tile_step = 2 * stride * P2_TILE_RB;
        auto vt_rb = [&](int v) __attribute__((always_inline)) { return SAMPLE ? rb0 + (v / VPT) * tile_step + (v % VPT) * P2_WAVES : rb0 + v * P2_WAVES; };
        const int cb_tile = P.cb_start[b] + cbt0;
        const int m_left = m_use - cbt0 * 32;   // live columns of the tile from its first one (pass 1: m or m0, see the kernel)
        const size_t col0 = (size_t)cb_tile * 32;
        const uint4* aslab = P.slab16 + ((size_t)P.rb_start[b] * KG) * 64;
        const size_t rb_stride = (size_t)KG * 64;
        const int rb_last = nrb_b - 1;
        const bool loader = is_loader();
        const int lw = w & 3;
        const uint4* bbase0 = P.qfrag16 + ((size_t)cb_tile * KG) * 64;
        pend_pos = 0xffffffffu;
        int vt_n = 0, t_n = 0;
        Stream S;
        // the wave's own row-block of the tile (rows past the bucket's end: the last row-block again, masked in the epilogues)
        const char* abase = reinterpret_cast<const char*>(aslab) - ring;
        S.a = abase + ((size_t)min(rb0 + w, rb_last) * rb_stride) * 16;
        S.b0 = bbase0;
#pragma unroll
        for (int j = 0; j < PBL; ++j) S.bo[j] = (min((lw >> 1) + 2 * j, NCB - 1) * KG + (lw & 1)) * 64;
        const int NSR = (NS + P2_RING - 1) / P2_RING * P2_RING;
#define P2_ADVANCE                                                                               \
        if (++t_n < NS) { S.a += G * 1024; S.b0 += G * 64; }                                     \
        else if (t_n == NSR) {                                                                   \
            t_n = 0;                                                                             \
            if (vt_n + 1 < nvt) {                                                                \
                ++vt_n; S.b0 = bbase0;                                                           \
                S.a = abase + ((size_t)min(vt_rb(vt_n) + w, rb_last) * rb_stride) * 16;          \
            }                                                                                    \
        }
        // stage u's requests are older than stage u + 1's PA (+ PBL) requests: all but those have landed
#ifdef LMI_ABL_NOWAIT
#define P2_WAIT_LANDED
#elif defined(LMI_DBG_VM0)
#define P2_WAIT_LANDED asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define P2_WAIT_LANDED if (loader) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PBL) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA) : "memory");
#endif
#if defined(LMI_ABL_NOBAR) && defined(LMI_ABL_NOBARLGKM)   // timing-only ablations (garbage results)
#define P2_BARRIER
#elif defined(LMI_ABL_NOBAR)
#define P2_BARRIER asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(LMI_ABL_NOBARLGKM)
#define P2_BARRIER __builtin_amdgcn_s_barrier();
#else
#define P2_BARRIER asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
#endif
#define P2_STEP(SLOT, LIVE, LAST)                                                                \
        P2_WAIT_LANDED                                                                           \
        P2_STAMP_FINE(0)                                                                         \
        P2_BARRIER                                                                               \
        P2_STAMP_FINE(1)                                                                         \
        if (LIVE) {                                                                              \
            stage<SLOT, (SLOT + P2_RING - 1) % P2_RING>(S, loader, pend, (LAST), SLOT == 0 && t == 0, SLOT == 1 && t == 0); \
            pend = !(LAST);                                                                      \
        } else {                                                                                 \
            if (loader) dma_b<(SLOT + P2_RING - 1) % P2_RING>(S);                                \
            load_a<(SLOT + P2_RING - 1) % P2_RING>(S);                                           \
        }                                                                                        \
        P2_ADVANCE                                                                               \
        P2_STAMP_FINE(2)
        P2_STAMP(4)
        if (nvt > 0) {
            if (loader) dma_b<0>(S);
            load_a<0>(S);
            P2_ADVANCE
            if (loader) dma_b<1>(S);
            load_a<1>(S);
            P2_ADVANCE
        }
        if (!SAMPLE) {
            // thresholds of the tile's columns -> LDS (the caller's barrier made sThr free; the first stage's barrier publishes it).  BEHIND
            // the first two stages' requests (round 5): their wait covers both round trips instead of one after the other
            for (int i = tid; i < NCB * 32; i += 64 * P2_WAVES) {
                const bool wanted = i < m_left && (!P.redo_col || P.redo_col[col0 + i]);
                sThr[i] = wanted ? P.bound1[col0 + i] - P.eps2[col0 + i] : INFINITY;
            }
        }
        for (int vt = 0; vt < nvt; ++vt) {
            bool pend = false;
            for (int t = 0; t < NSR; t += P2_RING) {
                P2_STEP(0, true, t + 1 >= NS)
                P2_STEP(1, t + 1 < NS, t + 2 >= NS)
                P2_STEP(2, t + 2 < NS, t + 3 >= NS)
            }
            // an MFMA's result needs 18 wait states before a VALU reads it (16-pass worst case); hipcc pads nothing behind asm
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            P2_STAMP(2)   // (coarse builds: the whole K loop of the tile, waits and barriers included)
            // (a bucket with more sampled tiles than lists: EVERY tile that shares a list folds with the monotone atomic -- a plain
            // store from tile j could land behind tile j + 16's atomic and discard it: still a valid bound, but a different one from
            // run to run)
            if (SAMPLE) epilogue_sample(vt_rb(vt), n_b, col0, m_left, (ch + 2 * (vt / VPT)) % P2_NSL_BIG, p2_sample_tiles(n_b, P.sample_max) > P2_NSL_BIG, (vt % VPT) * P2_WAVES);
            else epilogue_emit(rb0 + vt * P2_WAVES, n_b, col0);
            P2_STAMP(3)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead before the LDS is reused
        queue.prefetch();   // the next ticket: its round trip beside the barrier and the flush below
        __syncthreads();
        P2_STAMP(5)
#ifdef LMI_P2_STAMPS
        if (SAMPLE == (LMI_P2_STAMPS_SAMPLE != 0)) {
            st_acc[7] = (unsigned long long)nvt;
            if (lane == 0) {
                unsigned long long* g = P.stamps + w * 12;
                for (int i = 0; i < 12; ++i) atomicAdd(g + i, st_acc[i]);
            }
        }
#endif
#undef P2_STEP
#undef P2_ADVANCE
#undef P2_WAIT_LANDED
#undef P2_BARRIER
        if (!SAMPLE) flush_pending(lane_id());
    }
};

// 10th largest of a column's P2_NSL x 16 slot maxima (every one the best score of a disjoint set of rows; -inf where no tile
// wrote).  A block takes 64 columns; thread (quarter qd, column c) keeps the ten best of lists 4 qd .. 4 qd + 3 (64 values; the
// lists are column-minor: the 64 threads of a quarter read 64 neighbouring floats), the four partial lists of a column meet in
// LDS.  Values only; fewer than 10 finite values: -inf (no bound).
// (epoch_bump: nullable -- the tag word of route_kernel's granules, lmi_front.h: every consumer of this call's tag has finished by now)
__device__ __forceinline__ void front_epoch_bump(unsigned* epoch_bump) {
    if (epoch_bump && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned e = *epoch_bump + 1u; *epoch_bump = e ? e : 1u; }
}
__global__ void front_epoch_bump_kernel(unsigned* epoch_bump) { front_epoch_bump(epoch_bump); }
__global__ __launch_bounds__(256) void bound_merge2_kernel(const float* __restrict__ lists, long long ncols, float* __restrict__ bound1, unsigned* epoch_bump = nullptr) {
    __shared__ float part[4][KPB][64];
    front_epoch_bump(epoch_bump);
    const int c = threadIdx.x & 63, qd = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + c;
    float v[KPB];
#pragma unroll
    for (int j = 0; j < KPB; ++j) v[j] = -INFINITY;
    auto insert = [&](float s) {
        if (s > v[KPB - 1]) {
#pragma unroll
            for (int t = 0; t < KPB; ++t) {  // sorted insert (descending), values only
                const float hi = fmaxf(v[t], s);
                s = fminf(v[t], s);
                v[t] = hi;
            }
        }
    };
    if (col < ncols) {
        // all 64 values requested before the first insert: one memory round trip instead of eight
        const float* src = lists + (size_t)(qd * (P2_NSL / 4) * 16) * ncols + col;
        float in[(P2_NSL / 4) * 16];
#pragma unroll
        for (int i = 0; i < (P2_NSL / 4) * 16; ++i) in[i] = __builtin_nontemporal_load(src + (size_t)i * ncols);
#pragma unroll
        for (int i = 0; i < (P2_NSL / 4) * 16; ++i) insert(in[i]);
    }
#pragma unroll
    for (int j = 0; j < KPB; ++j) part[qd][j][c] = v[j];
    __syncthreads();
    if (qd == 0 && col < ncols) {
#pragma unroll
        for (int o = 1; o < 4; ++o)
#pragma unroll
            for (int j = 0; j < KPB; ++j) insert(part[o][j][c]);
        bound1[col] = v[KPB - 1];
    }
}

// Query-level bound (k <= 10: the caller merges the ranks to the k <= 10 best of ALL visited buckets, LearnedIndex.py:125-146,
// so a row below the query's 10th best canonical score T_q over its buckets cannot be returned, whatever its bucket).
// bound1[col] <= That of the column's bucket, so That - eps' <= the bucket's canonical 10th best <= T_q: L_q = max over the
// query's columns of (bound1 - eps') is a lower bound of T_q, and a row of bucket b with canonical score >= T_q has
// shat >= L_q - eps'_b.  The kernel raises every column's bound1 to L_q + eps'_b, so that pass 2's threshold
// bound1 - 2 eps'_b becomes L_q - eps'_b.  Scores of one query share its scale (and the index's), so they compare across buckets.
__global__ void query_bound_kernel(const int* __restrict__ slot_col, int nq, int nb, const float* __restrict__ eps2, float* __restrict__ bound1) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float lq = -INFINITY;
    for (int r = 0; r < nb; ++r) {
        const int col = slot_col[q * nb + r];
        if (col >= 0) lq = fmaxf(lq, bound1[col] - 0.5f * eps2[col]);
    }
    if (!(lq > -INFINITY)) return;
    lq -= fabsf(lq) * 4.8e-7f;  // the few roundings above, generously
    for (int r = 0; r < nb; ++r) {
        const int col = slot_col[q * nb + r];
        if (col >= 0) bound1[col] = fmaxf(bound1[col], lq + 0.5f * eps2[col] * 0.9999f);
    }
}

// One work item for the block (thread 0 pops, the block learns it through s_item): bucket b, its query tile = col-blocks
// [cbt0, cbt0 + ncb_tile), chunk / sampled tile `ch`, and the live columns m_use.  Returns false when the queues are empty.
// The XCD-affine queues (route_group_kernel): pass 1 has its own prefixes (items = query tiles x SAMPLED tiles) and heads; a
// bucket's items go to the same XCD in both passes: its queries' fragments stay in that L2.
struct P2Item { int b, cbt0, ncb_tile, ch, m_use, nt; };   // (nt: pass 1 -- sampled tiles of the item: ch, ch + 2, ..)
// Item `local` of bucket b.  Query tiles of the bucket: its col-blocks split evenly over nqt = ceil(col-blocks / P.tile_cb) tiles.
// pass 2: local = chunk * nqt + tile, all columns.  pass 1: pass1_decode -> sampled tiles j0, j0 + 2, .. (up to P1_TPI), query tile; the even
// sampled tiles run over all the columns, the odd ones over the primary ones (the bucket's first m0).
template <bool SAMPLE>
__device__ __forceinline__ void p2_decode_item(const PrefilterParams& P, int b, int local, P2Item& it) {
    int qt = 0, ch = 0, nt = 1;
    bool all_cols = true;
    if (SAMPLE) {
        const int nqa = query_tiles(P.m[b], P.tile_cb), nqp = query_tiles(P.m0[b], P.tile_cb);
        all_cols = pass1_decode(local, p2_sample_tiles(P.nb_rows[b], P.sample_max), nqa, nqp, &ch, &nt, &qt);
    }
    it.nt = nt;
    it.m_use = all_cols ? P.m[b] : P.m0[b];
    const int ncb_b = (it.m_use + 31) >> 5;
    const int nqt = (ncb_b + P.tile_cb - 1) / P.tile_cb;
    const int per = (ncb_b + nqt - 1) / nqt;
    if (!SAMPLE) { qt = local % nqt; ch = local / nqt; }
    it.b = b;
    it.ch = ch;
    it.cbt0 = qt * per;
    it.ncb_tile = min(per, ncb_b - it.cbt0);
}
template <bool SAMPLE>
__device__ __forceinline__ bool p2_pop_item(const PrefilterParams& P, int& grp, int* s_item, P2Item& it) {
    if (threadIdx.x == 0) {
        int b = -1, local = 0;
        do {
            b = -1;
            unsigned* heads = P.head + (SAMPLE ? 24 : 0);
            const int* totals = SAMPLE ? P.grp_total1 : P.grp_total;
            const int* bases = SAMPLE ? P.grp_base1 : P.grp_base;
            for (int tries = 0; tries < NGRP; ++tries) {
                const int tot = totals[grp];
                if (__hip_atomic_load(&heads[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tot) {
                    const int itn = (int)atomicAdd(&heads[grp], 1u);
                    if (itn < tot) {
                        const int* base = bases + grp * (P.L + 1);
                        int lo = 0, hi = P.grp_n[grp];
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (base[mid] <= itn) lo = mid; else hi = mid;
                        }
                        b = P.grp_bucket[grp * P.L + lo];
                        local = itn - base[lo];
                        break;
                    }
                }
                grp = (grp + 1) & (NGRP - 1);
            }
        } while (!SAMPLE && P.redo_bucket && b >= 0 && !P.redo_bucket[b]);
        s_item[0] = b;
        s_item[1] = local;
    }
    __syncthreads();
    const int b = s_item[0], local = s_item[1];
    __syncthreads();
    if (b < 0) return false;
    p2_decode_item<SAMPLE>(P, b, local, it);
    return true;
}

// The block's OWN queue (its XCD's group of buckets) with the group's item prefix in LDS: the binary search is ~10 LDS reads
// instead of as many L2 round trips (~3 us per item: a tenth of a pass-1 item at d = 768, as much as a whole item at d = 45).
// PREFETCH (lmi_pass2_small.h, where an item is a few microseconds): thread 0 also takes the NEXT ticket when an item starts,
// the atomic's round trip hides behind the item (costs a live register across the item: not for pass2_kernel's K loop).
// Once the own group is used up the block falls back to p2_pop_item's walk over the other groups (the tail of the launch).
constexpr int P2_PREFIX_CAP = 1025;   // buckets + 1 of a group held in LDS (more: the global prefix is searched)
constexpr int P2_PREFIX_CAP_K = P2_WAVES == 4 ? 116 : P2_PREFIX_CAP;   // pass2_kernel at two blocks per CU: 80 KiB of LDS per block, to the byte
// PREFETCH 2 (pass2_kernel, round 5): the ticket is taken by prefetch(), which the tile loop calls once the item's look-ahead is drained --
// the atomic's round trip runs beside the item's last barrier and candidate flush, and no register is live across the K loop; with the
// group's bucket ids in LDS too (s_gb) the pop makes no global round trip of its own.
template <bool SAMPLE, int PREFETCH, int NTHREADS, int PREFIX_CAP = P2_PREFIX_CAP>
struct P2Queue {
    const PrefilterParams& P;
    int* s_item;     // [2] LDS
    int* s_prefix;   // [PREFIX_CAP] LDS
    int* s_gb;       // nullable [PREFIX_CAP] LDS: the group's bucket ids
    int grp, own, own_tot, own_n, ticket;
    const int* own_base;
    unsigned* own_head;
    bool prefix_lds, own_live;

    __device__ __forceinline__ void init() {
        own = grp = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (NGRP - 1));
        own_head = P.head + (SAMPLE ? 24 : 0) + own;
        own_tot = (SAMPLE ? P.grp_total1 : P.grp_total)[own];
        own_n = P.grp_n[own];
        own_base = (SAMPLE ? P.grp_base1 : P.grp_base) + own * (P.L + 1);
        prefix_lds = own_n + 1 <= PREFIX_CAP;
        if (prefix_lds)
            for (int i = threadIdx.x; i <= own_n; i += NTHREADS) {
                s_prefix[i] = own_base[i];
                if (s_gb && i < own_n) s_gb[i] = P.grp_bucket[own * P.L + i];
            }
        own_live = own_tot > 0 && !(!SAMPLE && P.redo_bucket);   // (the redo launch skips buckets: it keeps to the plain pop)
        ticket = -1;
        if (PREFETCH == 1 && own_live && threadIdx.x == 0) ticket = (int)atomicAdd(own_head, 1u);
        __syncthreads();
    }
    __device__ __forceinline__ void prefetch() {
        if (PREFETCH == 2 && own_live && threadIdx.x == 0) ticket = (int)atomicAdd(own_head, 1u);
    }
    __device__ __forceinline__ bool next(P2Item& it) {
        if (own_live) {
            if (threadIdx.x == 0) {
                int b = -1, local = 0;
                if (PREFETCH == 0 || (PREFETCH == 2 && ticket < 0)) ticket = (int)atomicAdd(own_head, 1u);
                if (ticket < own_tot) {
                    int lo = 0, hi = own_n;
                    if (prefix_lds) {
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_prefix[mid] <= ticket) lo = mid; else hi = mid; }
                        local = ticket - s_prefix[lo];
                    } else {
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (own_base[mid] <= ticket) lo = mid; else hi = mid; }
                        local = ticket - own_base[lo];
                    }
                    b = (s_gb && prefix_lds) ? s_gb[lo] : P.grp_bucket[own * P.L + lo];
                    if (PREFETCH == 1) ticket = (int)atomicAdd(own_head, 1u);   // the next one: consumed when this item is done
                }
                if (PREFETCH == 2) ticket = -1;   // consumed
                s_item[0] = b;
                s_item[1] = local;
            }
            __syncthreads();
            const int b = s_item[0], local = s_item[1];
            __syncthreads();
            if (b >= 0) { p2_decode_item<SAMPLE>(P, b, local, it); return true; }
            own_live = false;
            grp = (own + 1) & (NGRP - 1);
        }
        return p2_pop_item<SAMPLE>(P, grp, s_item, it);
    }
};

template <bool SAMPLE>
__global__ __launch_bounds__(64 * P2_WAVES, 2) __attribute__((amdgpu_num_vgpr(LMI_P2_NUM_VGPR_ATTR))) void pass2_kernel(PrefilterParams P) {
    // v[232:255] are reserved from the compiler (file header); naming the last one makes the kernel descriptor allocate all 256
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("" ::: "v255");
#pragma clang diagnostic pop
    __shared__ __attribute__((aligned(16))) uint4 ring[P2_RING * P2_SLOT_BYTES / 16];
    __shared__ uint2 sList[SAMPLE ? 1 : P2_WAVES * P2_LIST];
    __shared__ float sThr[SAMPLE ? 1 : P2_MAXCB * 32];
    __shared__ __attribute__((aligned(16))) uint4 sPend[SAMPLE ? 1 : P2_WAVES * 64];
    __shared__ int s_item[2];
    __shared__ int s_prefix[P2_PREFIX_CAP_K];
    __shared__ int s_gb[P2_PREFIX_CAP_K];
    if (!SAMPLE && P.redo_count && *P.redo_count == 0u) return;  // the redo launch of a batch without overflowed columns
    ts_first(P.ts_start);
    unsigned long long clk_w0 = 0, clk_c0 = 0;
    if (!SAMPLE) clk_begin(P.ts_end_cell ? P.ts_start : nullptr, clk_w0, clk_c0);   // (pass 2 proper: the launch that carries an end cell)
    P2Queue<SAMPLE, 2, 64 * P2_WAVES, P2_PREFIX_CAP_K> queue{P, s_item, s_prefix, s_gb};
    queue.init();
    P2Item item;
    while (queue.next(item)) {
        const int b = item.b, cbt0 = item.cbt0, ch = item.ch, m_use = item.m_use, nt = item.nt;
#define P2_CASE(N) case N: { Tile2<N, SAMPLE> it{P, (unsigned)reinterpret_cast<uintptr_t>(ring), ring, sList, sThr, sPend}; it.run(b, cbt0, ch, m_use, nt, queue); break; }
        switch (item.ncb_tile) {
            P2_CASE(1) P2_CASE(2) P2_CASE(3) P2_CASE(4) P2_CASE(5) P2_CASE(6)
            P2_CASE(7) P2_CASE(8) P2_CASE(9) P2_CASE(10) P2_CASE(11)
            default: { Tile2<12, SAMPLE> it{P, (unsigned)reinterpret_cast<uintptr_t>(ring), ring, sList, sThr, sPend}; it.run(b, cbt0, ch, m_use, nt, queue); break; }
        }
#undef P2_CASE
    }
    if (!SAMPLE) clk_end(P.ts_end_cell ? P.ts_start : nullptr, ST_P2, clk_w0, clk_c0);
#ifdef LMI_P2_ENDS   // developer builds (bench.py LMI_P2_ENDS=1): when each workgroup of pass 2 ran out of items -- the launch's ragged end
    if (!SAMPLE && P.ts_end_cell && threadIdx.x == 0 && blockIdx.x < 256) {
        P.stamps[192 + blockIdx.x] = wall_clock64();
        if (blockIdx.x == 0) P.stamps[191] = clk_w0;
    }
#endif
    ts_max(P.ts_end_cell);
}

}  // namespace lmi
