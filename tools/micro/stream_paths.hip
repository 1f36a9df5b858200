// micro-benchmark (developer aid, round 4): which PATH should pass 2's two streams take?
// profiles/r03_pass2_experiments.txt section 20 found that a CU's LDS-DMA path moves one 1-KiB piece per ~60 cycles whatever its
// source, and that pass 2 puts both the index vectors (HBM, 16 pieces per stage) and the re-streamed query fragments (L2,
// 2 x col-blocks pieces per stage) on it.  Here the vectors go global_load_dwordx4 -> VGPR instead (a wave's row-block is private
// to it: no LDS needed), into three register sets two stages ahead with counted vmcnt waits -- registers v[232:255], which
// hipcc is kept out of by amdgpu_num_vgpr (the attribute counts VGPR + AGPR halves: 116 -> the compiler gets v0..v231) --
// and the query fragments either by LDS-DMA (MODE 1) or global_load -> VGPR -> ds_write_b128 (MODE 2), or not at all (MODE 0).
// No MFMA, no fragment reads: stream times only, same ring / waits / stage barrier as lmi_pass2.h.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_paths.hip -o /tmp/stream_paths && /tmp/stream_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int KG = 48;           // k16-groups per row-block (d = 768)
constexpr int SLOT = 24 * 1024;  // ring slot: the query fragments of <= 12 col-blocks x 2 k-groups

#define A_LOAD(R0, R1)                                                                              \
    asm volatile("global_load_dwordx4 v[" #R0 ":" #R1 "], %0, %1" :: "v"(voff), "s"(ap) : "memory")
#define A_LOAD1K(R0, R1)                                                                            \
    asm volatile("global_load_dwordx4 v[" #R0 ":" #R1 "], %0, %1 offset:1024" :: "v"(voff), "s"(ap) : "memory")
#define A_USE(R)  asm volatile("v_xor_b32 %0, %0, v" #R : "+v"(accx))

// NCB col-blocks per tile; MODE 0: vectors only; 1: + query fragments by LDS-DMA (loader waves 4..7); 2: + query fragments through
// registers (every wave PQ pieces per stage, written to LDS one stage later)
template <int NCB, int MODE, int QSRC>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(100)))
void k(const uint4* __restrict__ slab, const uint4* __restrict__ qfrag, long long n_tiles, unsigned* __restrict__ head, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) uint4 ring[3 * SLOT / 16];
    __shared__ long long s_tile;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool loader = w >= 4;
    const int lw = w & 3;
    constexpr int PBL = (NCB + 1) / 2;          // MODE 1: pieces per loader wave and stage
    constexpr int PQ = (2 * NCB + 7) / 8;       // MODE 2: pieces per wave and stage (<= 3)
    unsigned accx = 0;
    asm volatile("" ::: "v255");   // the kernel descriptor must allocate all 256 registers (the compiler itself stays below v200)
    const unsigned voff = (unsigned)lane * 16u;
    for (;;) {
        if (threadIdx.x == 0) s_tile = (long long)atomicAdd(head, 8u);   // a chunk = 8 tiles
        __syncthreads();
        const long long t0 = (long long)__builtin_amdgcn_readfirstlane((int)s_tile);   // wave-uniform for the "s" operands
        __syncthreads();
        if (t0 >= n_tiles) break;
        const long long t1 = t0 + 8 < n_tiles ? t0 + 8 : n_tiles;
        constexpr int NS = KG / 2;   // 24 stages per tile
        const long long total = (t1 - t0) * NS;   // a multiple of 3
        // the chunk's query tile.  QSRC 0: 64 tiles in rotation by chunk (r03's micro-benchmark: 37 MB in use by the 256 CUs at any
        // time, i.e. NOT L2-resident, only Infinity-Cache-resident at best); QSRC 1: one tile per XCD (0.6 MB per 4-MiB L2: real L2
        // hits, which is what pass 2's XCD-affine queues arrange: the blocks of an XCD work on the same few buckets)
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);
        const uint4* qb = qfrag + (QSRC ? (long long)xcc * 8 + ((t0 / 8 / 256) % 8) : (t0 / 8) % 64) * (12 * KG * 64);
        long long is = 0;            // next stage to issue (tile = is / NS, stage = is % NS)
        const uint4* ap = nullptr;   // wave-uniform source of the next A load
        auto a_src = [&](long long i) -> const uint4* {
            const long long i2 = i < total ? i : total - 1;   // past the end: re-load the last stage (constant vmcnt literals)
            return slab + (((t0 + i2 / NS) * 8 + w) * KG + 2 * (i2 % NS)) * 64;
        };
        auto q_dma = [&](long long i, int slot) {   // MODE 1, loaders
            const long long i2 = i < total ? i : total - 1;
            const int s = (int)(i2 % NS);
            uint4* dst = ring + slot * (SLOT / 16);
#pragma unroll
            for (int j = 0; j < PBL; ++j) {
                int cb = (lw >> 1) + 2 * j; if (cb > NCB - 1) cb = NCB - 1;
                const uint4* b = qb + (cb * KG + 2 * s + (lw & 1)) * 64 + lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b,
                                                 (__attribute__((address_space(3))) void*)(dst + (cb * 2 + (lw & 1)) * 64), 16, 0, 0);
            }
        };
        // MODE 2: this wave's PQ query pieces of stage i -> registers v[208:219] (even stages) / v[220:231] (odd stages)
#define Q_LOAD(R0, R1, J)                                                                                    \
        { int pc = w * PQ + J; if (pc > 2 * NCB - 1) pc = 2 * NCB - 1;                                        \
          const uint4* qp = qb + ((pc >> 1) * KG + 2 * s_ + (pc & 1)) * 64;                                   \
          asm volatile("global_load_dwordx4 v[" #R0 ":" #R1 "], %0, %1" :: "v"(voff), "s"(qp) : "memory"); }
#define Q_WRITE(R0, R1, J)                                                                                   \
        { int pc = w * PQ + J; if (pc > 2 * NCB - 1) pc = 2 * NCB - 1;                                        \
          const unsigned la = (unsigned)reinterpret_cast<uintptr_t>(ring) + (unsigned)(slot_ * SLOT + pc * 1024) + voff; \
          asm volatile("ds_write_b128 %0, v[" #R0 ":" #R1 "]" :: "v"(la) : "memory"); }
        auto q_load_even = [&](long long i) {
            const long long i2 = i < total ? i : total - 1; const int s_ = (int)(i2 % NS);
            Q_LOAD(208, 211, 0) if (PQ > 1) Q_LOAD(212, 215, 1) if (PQ > 2) Q_LOAD(216, 219, 2)
        };
        auto q_load_odd = [&](long long i) {
            const long long i2 = i < total ? i : total - 1; const int s_ = (int)(i2 % NS);
            Q_LOAD(220, 223, 0) if (PQ > 1) Q_LOAD(224, 227, 1) if (PQ > 2) Q_LOAD(228, 231, 2)
        };
        auto q_write_even = [&](int slot_) { Q_WRITE(208, 211, 0) if (PQ > 1) Q_WRITE(212, 215, 1) if (PQ > 2) Q_WRITE(216, 219, 2) };
        auto q_write_odd = [&](int slot_) { Q_WRITE(220, 223, 0) if (PQ > 1) Q_WRITE(224, 227, 1) if (PQ > 2) Q_WRITE(228, 231, 2) };

        // prologue: stages 0 and 1
        ap = a_src(0); A_LOAD(232, 235); A_LOAD1K(236, 239);
        if (MODE == 1 && loader) q_dma(0, 0);
        if (MODE == 2) { q_load_even(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q_write_even(0); }
        ap = a_src(1); A_LOAD(240, 243); A_LOAD1K(244, 247);
        if (MODE == 1 && loader) q_dma(1, 1);
        if (MODE == 2) q_load_odd(1);
        is = 2;
        // stage i computes from set i % 3 / ring slot i % 3; issues stage i + 2 into set / slot (i + 2) % 3
        // outstanding at the top of stage i, oldest first: [stage i: A x 2, Q] [stage i + 1: A x 2, Q]
#define STAGE(SET_A0, SET_A1, SET_A2, SET_A3, USE0, USE1, SLOT_NEXT2, SLOT_NEXT1, PARITY_EVEN)               \
        {                                                                                                    \
            if (MODE == 1) { if (loader) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + PBL) : "memory");     \
                             else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }                         \
            else if (MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + PQ) : "memory");               \
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
            __builtin_amdgcn_s_barrier();                                                                    \
            A_USE(USE0); A_USE(USE1);                                                                        \
            if (MODE == 1 && loader) q_dma(is, SLOT_NEXT2);                                                  \
            ap = a_src(is); A_LOAD(SET_A0, SET_A1); A_LOAD1K(SET_A2, SET_A3);                                \
            if (MODE == 2) {                                                                                 \
                /* the pieces of stage i + 1 (loaded one stage ago) -> LDS; then request those of stage i + 2 */ \
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                             \
                if (PARITY_EVEN) { q_write_odd(SLOT_NEXT1); q_load_even(is); }                               \
                else { q_write_even(SLOT_NEXT1); q_load_odd(is); }                                           \
            }                                                                                                \
            unsigned vx;                                                                                     \
            const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(ring) + (unsigned)((int)((SLOT_NEXT2 + 1) % 3) * SLOT + (w * 128 + lane) * 16); \
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(vx) : "v"(addr) : "memory");   \
            accx ^= vx;                                                                                      \
            ++is;                                                                                            \
        }
        // (MODE 2 keeps two register sets for the query pieces by the stage's parity, so the loop is unrolled by 6)
        for (long long i = 0; i < total; i += 6) {
            STAGE(248, 251, 252, 255, 232, 236, 2, 1, true)
            STAGE(232, 235, 236, 239, 240, 244, 0, 2, false)
            STAGE(240, 243, 244, 247, 248, 252, 1, 0, true)
            STAGE(248, 251, 252, 255, 232, 236, 2, 1, false)
            STAGE(232, 235, 236, 239, 240, 244, 0, 2, true)
            STAGE(240, 243, 244, 247, 248, 252, 1, 0, false)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (accx == 0x12345u) sink[0] = accx;
}

template <int NCB, int MODE, int QSRC>
static void run(const char* name, const uint4* slab, const uint4* qfrag, long long n_tiles, unsigned* head, unsigned* sink, size_t bytes) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(head, 0, 4);
        (void)hipEventRecord(e0);
        k<NCB, MODE, QSRC><<<256, 512>>>(slab, qfrag, n_tiles, head, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    printf("%-112s %.3f ms -> %.2f TB/s of vectors%s\n", name, best, bytes / 1e9 / best, err == hipSuccess ? "" : "  (ERROR)");
    fflush(stdout);
}

int main() {
    const long long rows = 10'000'000 / 32 / 64 * 64;          // row-blocks, whole 8-tile chunks
    const long long n_tiles = rows / 8;
    const size_t bytes = (size_t)rows * KG * 1024;
    uint4* slab; uint4* qfrag; unsigned* head; unsigned* sink;
    if (hipMalloc(&slab, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const size_t qbytes = (size_t)64 * 12 * KG * 1024;
    (void)hipMalloc(&qfrag, qbytes); (void)hipMalloc(&head, 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(slab, 0x11, bytes); (void)hipMemset(qfrag, 0x22, qbytes);
    run<12, 0, 0>("vectors -> VGPR (global_load_dwordx4, 3 sets, 2 stages ahead), nothing else", slab, qfrag, n_tiles, head, sink, bytes);
    run<4, 1, 0>("vectors -> VGPR + query fragments of  4 col-blocks by LDS-DMA (waves 4-7) [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<4, 1, 1>("vectors -> VGPR + query fragments of  4 col-blocks by LDS-DMA (waves 4-7) [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    run<8, 1, 0>("vectors -> VGPR + query fragments of  8 col-blocks by LDS-DMA [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<8, 1, 1>("vectors -> VGPR + query fragments of  8 col-blocks by LDS-DMA [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    run<12, 1, 0>("vectors -> VGPR + query fragments of 12 col-blocks by LDS-DMA [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<12, 1, 1>("vectors -> VGPR + query fragments of 12 col-blocks by LDS-DMA [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    run<4, 2, 0>("vectors -> VGPR + query fragments of  4 col-blocks global_load -> VGPR -> ds_write_b128 [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<4, 2, 1>("vectors -> VGPR + query fragments of  4 col-blocks global_load -> VGPR -> ds_write_b128 [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    run<8, 2, 0>("vectors -> VGPR + query fragments of  8 col-blocks global_load -> VGPR -> ds_write_b128 [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<8, 2, 1>("vectors -> VGPR + query fragments of  8 col-blocks global_load -> VGPR -> ds_write_b128 [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    run<12, 2, 0>("vectors -> VGPR + query fragments of 12 col-blocks global_load -> VGPR -> ds_write_b128 [37 MB of tiles]", slab, qfrag, n_tiles, head, sink, bytes);
    run<12, 2, 1>("vectors -> VGPR + query fragments of 12 col-blocks global_load -> VGPR -> ds_write_b128 [one tile per XCD]", slab, qfrag, n_tiles, head, sink, bytes);
    return 0;
}
