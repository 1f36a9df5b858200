// micro-benchmark (developer aid): HBM -> LDS by LDS-DMA in pass 2's access pattern against a contiguous one.
//   pattern 0 (the slab as it is): a block's stage = 8 row-blocks x 2 KiB, the row-blocks KG KiB apart (48 KiB at d = 768),
//              the next stage 2 KiB further inside every row-block; a tile = 24 stages, then the next 8 row-blocks
//   pattern 1 (stage-major): a stage = 16 KiB contiguous, stage after stage
// Same bytes, same number of 1-KiB pieces per loader wave, same ring of 3 slots and waits as lmi_pass2.h (4 loader waves of 8,
// no MFMA work: the consumers only meet at the stage barrier).  Prints TB/s per pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldsdma_pattern.hip -o /tmp/ldsdma && /tmp/ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int KG = 48;          // k16-groups per row-block (d = 768)
constexpr int SLOT = 48 * 1024; // ring slot (vectors at 0 .. 32 KiB, query fragments behind)
// query-tile source (round 4): 0 = 64 tiles in rotation by chunk (37 MB in use at any time: NOT L2-resident, as round 3 assumed);
// 1 = one tile per XCD (real L2 hits, what pass 2's XCD-affine queues arrange)
__device__ int g_qsrc = 0;

__device__ __forceinline__ void glds16(const uint4* g, uint4* lds_wave_base, int off) {
    // LDS destination = wave-uniform base + lane * 16
    if (off == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 1024, 0);
}

// PBL: query-fragment pieces per loader wave and stage (0, 2, 4, 6 = col-blocks lw, lw + 4, lw + 8 x two k-groups), read from a
// 576-KiB region per 8-tile chunk that every tile of the chunk re-reads (L2 hits, as in pass 2)
template <int PATTERN, int PBL, int SPLIT = 0, int WIDE = 0>
__global__ __launch_bounds__(512, 1) void k(const uint4* __restrict__ slab, const uint4* __restrict__ qfrag, long long n_tiles, unsigned* __restrict__ head, float* sink) {
    __shared__ __attribute__((aligned(16))) uint4 ring[3 * SLOT / 16];
    __shared__ long long s_tile;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = SPLIT ? true : w >= 4;
    const int lw = w & 3;
    float acc = 0.f;
    for (;;) {
        if (threadIdx.x == 0) s_tile = (long long)atomicAdd(head, 8u);   // a chunk = 8 tiles
        __syncthreads();
        const long long t0 = s_tile;
        __syncthreads();
        if (t0 >= n_tiles) break;
        const long long t1 = t0 + 8 < n_tiles ? t0 + 8 : n_tiles;
        const int NS = KG / 2;   // 24 stages per tile
        // stage s of tile t: source of loader lw's row-blocks lw and lw + 4 (two k-groups = 2 KiB each)
        auto src = [&](long long t, int s, int rb) -> const uint4* {
            if (PATTERN == 0) return slab + (((t * 8 + rb) * KG) + 2 * s) * 64 + lane;
            return slab + (((t * NS + s) * 8 + rb) * 2) * 64 + lane;
        };
        auto issue = [&](long long t, int s, int slot) {
            uint4* dst = ring + slot * (SLOT / 16);
            if (!SPLIT || w >= 4) {
                const uint4* a0 = src(t, s, lw);
                const uint4* a1 = src(t, s, lw + 4);
                glds16(a0, dst + lw * 128, 0);
                glds16(a0, dst + lw * 128, 1);
                glds16(a1, dst + (lw + 4) * 128, 0);
                glds16(a1, dst + (lw + 4) * 128, 1);
                if (WIDE) {   // the second 256 vectors of a 512-vector tile (the next tile's row-blocks), same stage
                    const uint4* a2 = src(t + 1, s, lw);
                    const uint4* a3 = src(t + 1, s, lw + 4);
                    glds16(a2, dst + (8 + lw) * 128, 0);
                    glds16(a2, dst + (8 + lw) * 128, 1);
                    glds16(a3, dst + (12 + lw) * 128, 0);
                    glds16(a3, dst + (12 + lw) * 128, 1);
                }
            }
            if (SPLIT && w >= 4) return;
            const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);
            const uint4* qb = qfrag + (g_qsrc ? (long long)xcc * 8 + ((t0 / 8 / 256) % 8) : (t0 / 8) % 64) * (12 * KG * 64) + lane;   // the chunk's query tile
#pragma unroll
            for (int j = 0; j < PBL / 2; ++j) {
                const uint4* b = qb + ((lw + 4 * j) * KG + 2 * s) * 64;
                glds16(b, dst + 2048 + (lw + 4 * j) * 128, 0);
                glds16(b, dst + 2048 + (lw + 4 * j) * 128, 1);
            }
        };
        const long long total = (t1 - t0) * NS / (WIDE ? 2 : 1);   // WIDE: two tiles per turn
        long long issued = 0;
        if (loader) { issue(t0, 0, 0); issue(t0 + 1 / NS, 1 % NS, 1); }
        issued = 2;
        for (long long i = 0; i < total; ++i) {
            if (SPLIT && w < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PBL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SPLIT ? 4 : (WIDE ? 8 : 4) + PBL) : "memory");
            __builtin_amdgcn_s_barrier();
            if (issued < total) {
                if (loader) issue(t0 + (issued / NS) * (WIDE ? 2 : 1), (int)(issued % NS), (int)(issued % 3));
                ++issued;
            } else if (loader) {   // keep the count of outstanding pieces constant for the vmcnt literal
                issue(t0, 0, (int)((i + 2) % 3));
            }
            // consumers: touch the stage (one ds_read per wave) so that the data is really needed
            // (asm: hipcc would put `s_waitcnt vmcnt(0)` in front of an LDS read it can see, draining the look-ahead)
            unsigned vx;
            const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(ring) + (unsigned)((int)(i % 3) * SLOT + (w * 128 + lane) * 16);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(vx) : "v"(addr) : "memory");
            acc += __uint_as_float(vx & 0x3f800000u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main() {
    const long long rows = 10'000'000 / 32 / 8 * 8;          // row-blocks, whole tiles
    const long long n_tiles = rows / 8;
    const size_t bytes = (size_t)rows * KG * 1024;
    uint4* slab; uint4* qfrag; unsigned* head; float* sink;
    if (hipMalloc(&slab, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const size_t qbytes = (size_t)64 * 12 * KG * 1024;
    (void)hipMalloc(&qfrag, qbytes); (void)hipMalloc(&head, 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(slab, 0, bytes); (void)hipMemset(qfrag, 0, qbytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[9] = {"slab as it is (8 x 2 KiB at 48 KiB stride), vectors only", "stage-major (16 KiB contiguous per stage), vectors only",
                            "slab as it is + query fragments of 4 col-blocks", "slab as it is + query fragments of 8 col-blocks", "slab as it is + query fragments of 12 col-blocks",
                            "the same, 8 col-blocks, query pieces issued by waves 0-3 and vector pieces by waves 4-7", "the same, 12 col-blocks",
                            "512-vector tiles (8 vector pieces per loader and stage) + query fragments of 4 col-blocks", "512-vector tiles + 8 col-blocks"};
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 9; ++pat) {
            const int qsrc = rep;   // first round: 64 tiles in rotation; second round: one tile per XCD
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qsrc), &qsrc, sizeof(int));
            (void)hipMemset(head, 0, 4);
            (void)hipEventRecord(e0);
            switch (pat) {
                case 0: k<0, 0><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 1: k<1, 0><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 2: k<0, 2><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 3: k<0, 4><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 4: k<0, 6><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 5: k<0, 4, 1><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 6: k<0, 6, 1><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                case 7: k<0, 2, 0, 1><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
                default: k<0, 4, 0, 1><<<256, 512>>>(slab, qfrag, n_tiles, head, sink); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-82s [%s] %.3f ms for %.2f GB of vectors -> %.2f TB/s\n", names[pat], qsrc ? "one query tile per XCD" : "37 MB of query tiles", ms, bytes / 1e9, bytes / 1e9 / ms);
        }
    return 0;
}
