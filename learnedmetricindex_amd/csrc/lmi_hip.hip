// lmi_hip.hip -- host side of liblmi_hip.so: the C ABI declared in include/lmi_hip.h.
// Owns the device-resident index (fragment-major slab, ids, CSR of buckets), the packed MLP
// weights and the per-call workspaces; enqueues the kernels of lmi_kernels.h on one HIP stream.
#include "lmi_kernels.h"
#include "lmi_prefilter.h"
#include "lmi_pass2.h"
#include "lmi_pass2_small.h"
#include "lmi_mlp_fused.h"
#include "lmi_rescore.h"
#include "lmi_front.h"
#include "lmi_tail.h"

#include <algorithm>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "lmi_hip.h"
#include <rccl/rccl.h>  // types only: the functions are resolved at run time (lmi_comm_*), the library does not link RCCL

using namespace lmi;

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIPCHK(expr)                                                                                \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHK(expr)              \
    do {                       \
        int r_ = (expr);       \
        if (r_ != 0) return r_; \
    } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline long long rup(long long a, long long b) { return (a + b - 1) / b * b; }

// device buffer that only grows
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;  // lmi_clone_view: the memory belongs to the handle this one was cloned from
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (borrowed) return fail("internal: a buffer shared with the parent handle would have to grow");
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
    }
    void borrow() { borrowed = p != nullptr; }   // keep the pointer, never free it
    void forget() { p = nullptr; cap = 0; borrowed = false; }  // a copied struct's workspace: start empty
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

struct lmi_index {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;           // library-owned: the per-layer MLP of a batch's tail beside the fused kernel (mlp_enqueue)
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    int num_cus = 256;
    int scan_blocks_per_cu = 2;

    // ---- MLP ----
    int n_layers = 0;
    std::vector<int> dims;    // dims[0..n_layers]
    std::vector<int> n_rb;    // per layer: output row-blocks
    std::vector<int> KG;      // per layer: k-groups of the layer's input
    std::vector<DevBuf> Wf;   // packed weights
    std::vector<DevBuf> bias; // padded bias

    // ---- fused MLP / multi-level navigation (lmi_mlp_fused.h) ----
    struct NodeModel {            // an internal node's model (model id >= 1; the root is the fields above)
        int n_layers = 0;
        std::vector<int> dims, n_rb, KG;
        std::vector<DevBuf> Wf, bias;
    };
    std::vector<NodeModel> node_models;   // index = model id - 1
    int fused_mlp = 1;                    // lmi_set_fused_mlp: 0 never, 1 when the batch fills the chip, 2 always
    bool desc_dirty = true;
    DevBuf d_models;                      // ModelDesc[1 + node_models.size()]
    int fm_s0 = 0, fm_s1 = 0, fm_act0 = 0, fm_lds = 0, fm_logits_lds = 0;  // LDS plan of the current model set
    bool fm_ok = false;                   // every model fits the fused kernel
    // the tree: flat child index = child_offset[model] + class
    std::vector<int> h_child_offset, h_child_model, h_child_bucket;
    DevBuf d_child_offset, d_child_model, d_child_bucket;
    bool tree_set = false;
    DevBuf gather_send, gather_recv;      // lmi_allgather_merge
    DevBuf pq_prob, pq_ent, pq_len, nav_len, nav_slab, nav_ent, nav_count, nav_colq, nav_active;

    // ---- buckets ----
    bool building = false, built = false;
    int64_t N = 0;
    int d = 0, L = 0, KGs = 0;     // d: dims of the STORED vectors (L2 metric: user dims + the norm column, padded to 4)
    int metric = 0, d_user = 0;    // lmi_set_metric; dims of the caller's vectors
    DevBuf aug_rows, q_aug, qn2;   // L2: augmented ingest pieces / queries, |q|^2
    int chunk_rows = 2048;
    bool chunk_rows_auto = true;  // until lmi_set_chunk_rows: lmi_buckets_begin picks 256..2048 by the index size
    int64_t n_rb_total = 0;
    std::vector<int> h_nb_rows, h_rb_start, h_nch;
    DevBuf slab, ids_slab, pos, d_nb_rows, d_rb_start, d_nch;
    int64_t rows_added = 0, owned_total = 0;
    bool indexed_ingest = false;  // lmi_buckets_add_owned_rows: only the owned objects are passed in
    DevBuf stage;  // H2D staging for add_rows / host query uploads
    // ---- fp16 prefilter (lmi_prefilter.h) ----
    bool prefilter = true;   // lmi_set_prefilter
    bool pf_hw_ok = false;   // fp16 subnormal self-test passed on this device
    bool have16 = false;     // slab16 built by lmi_buckets_end
    int KG16 = 0;
    int dp = 0;   // row pitch (floats) of `rowmajor`
    DevBuf slab16, rowmajor, xscale, xmaxbits, bnorm, bdelta, qdelta;
    DevBuf qnorm, qscale, qfrag16, eps2, cand_cnt, cand_row, cand_s, fallback, pf_bound, nkeep, surv_row, rs_flag, rs_active;
    DevBuf grp_scratch;      // route_group_kernel<true>: the bucket sort of fan-outs past ROUTE_MAX_BUCKETS
    bool ps_wide = false;    // this call's form of the low-dimensional kernels (lmi_pass2_small.h)
    int ps_force_wide = -1;  // LMI_PS_WIDE=0/1 pins it (developer aid)
    int n_nonempty = 1;      // buckets with rows, on any rank (lmi_buckets_begin)
    DevBuf x_log, x_ext, x_off, fb_list;  // the candidates' overflow log, its by-column sorted form and offsets (lmi_prefilter.h, OverflowLog);
                                          // the fallback list: [count, fail0, fail1, log head, sorted total, pad x 3 | nslots slots]
    unsigned x_cap = 0;                   // entries of the log (0: not allocated yet)
    DevBuf redo;   // [1] count | [L] bucket flags | [columns] column flags (bytes): overflow_rebound_kernel
    size_t stamps_off = 0;         // developer builds: byte offset of the phase stamps inside pf_bound
    bool pf_small = true;          // d <= 128: pass2_small_kernel (LMI_PF_SMALL=0 in the environment: pass2_kernel for every d)
    bool pf_redo = true;           // overflow_rebound_kernel + pass 2's redo launch (LMI_PF_NO_REDO=1 in the environment: off)
    bool rescore_streamed = true;  // lmi_rescore.h (LMI_RESCORE_SIMPLE=1 in the environment: select_rescore_kernel)
    int last_nslots = 0, last_nb = 0;
    long long last_ncols = 0;
    bool last_fast = false;
    bool pf_qbound = true;        // LMI_PF_QBOUND=0: per-bucket bounds only (query_bound_kernel off)
    bool pf_primary = true;       // LMI_PF_PRIMARY=0: pass 1 samples every column although one bound per query is used
    bool debug_emit_all = false;  // lmi_debug_emit_all

    // ---- per-call workspaces ----
    DevBuf act[2], xfrag, logits, order, q_nav, q_srch;
    DevBuf m, cb_start, item_base, part_base, stats, head, slot_local, slot_col, colmap, qfrag, grp, col_thr;
    DevBuf part_score, part_row, rank_d, rank_id, out_d, out_id, out_key;
    // hipEvents of the last EV_RING calls: lmi_timings reads the newest set, lmi_timings_mean averages all
    // sets since lmi_timings_reset with ONE stream synchronisation (no per-call sync in a timed loop)
    static constexpr int EV_RING = 128;
    hipEvent_t ev_ring[EV_RING][10] = {};
    bool valid_ring[EV_RING][10] = {};
    int ev_cur = 0;
    int timing_level = 2;  // lmi_set_timing
    long long ev_calls = 0;  // calls since lmi_timings_reset
    hipEvent_t* ev = ev_ring[0];
    bool* ev_valid = valid_ring[0];
    long long h_stats[4] = {0, 0, 0, 0};
    bool stats_pending = false;
    // device-side phase stamps (timing level 2; lmi_kernels.h): a ring of EV_RING sets of ST_COUNT words, the set of the current
    // call, which of its stamps a kernel of the call was given (host-side mask), the chip's constant clock in kHz
    DevBuf ts_ring;
    unsigned ts_mask[EV_RING] = {};
    unsigned long long* ts_set = nullptr;
    double wall_khz = 100000.0;
    DevBuf fr_dbg;                // LMI_FR_DEBUG=1: route_kernel's / pack_kernel's phase stamps (lmi_debug_peek "fr_dbg")
    DevBuf cb_alloc, cb_bucket;   // lmi_front.h: the call-tagged granules of route_kernel (zero at allocation) and the col-blocks' buckets
    unsigned* h_oflag = nullptr;  // a word of pinned host memory (device-visible): "the last batch used the overflow log" (RescoreParams::host_oflag)
    int overflow_armed = 0;       // calls for which overflow_rebound_kernel + pass 2's redo launch stay in the sequence (re-armed by h_oflag)
    bool fr_bump_pending = false; // route_kernel was launched and the launch that bumps the granules' tag (bound_merge2_kernel) not yet: a call that
                                  // failed in between is repaired by a bump launch of its own at the next call
    int use_tail = 1;             // tail_kernel (lmi_tail.h): selection + re-rank + rank merge in one wave per query (LMI_TAIL=0: the five launches of round 4;
                                  // 2: also group-wise for n_buckets > 4)
    bool graded_chunks = true;    // pass 2's items: chunk length per bucket and call (LMI_P2_GRADED=0: the index's static chunk everywhere)
    int chunk_lvl_rows[3] = {0, 0, 0};   // LMI_P2_CHUNKS=a,b,c (rows; 0 = 1, 1/2, 1/4 of the static chunk)
    float chunk_frac[2] = {0.16f, 0.05f};   // LMI_P2_CHUNK_FRAC=f0,f1: the last f0 of the work in chunks of b rows, the last f1 in chunks of c
    bool use_front = true;        // route_kernel + pack_kernel (lmi_front.h) instead of the eight preparation launches (LMI_FRONT=0 in the environment: off)
};

// which fp16 fragment shape the index and the queries are packed in: 16 x 32 for pass2_kernel, 32 x 16 for the low-dimensional kernels
static int frag16x16(const lmi_index* h) { return (h->pf_small && h->KG16 <= PS_MAXKG) ? 0 : 1; }


static int set_dev(lmi_index* h) {
    HIPCHK(hipSetDevice(h->device));
    return 0;
}

extern "C" LMI_API int lmi_abi_version(void) { return LMI_ABI_VERSION; }
#ifndef LMI_SOURCE_SHA16
#define LMI_SOURCE_SHA16 "unknown"
#endif
#define LMI_STR2(x) #x
#define LMI_STR(x) LMI_STR2(x)
extern "C" LMI_API const char* lmi_build_info(void) {
    return "src=" LMI_SOURCE_SHA16 " p2_waves=" LMI_STR(LMI_P2_WAVES) " p2_bring=" LMI_STR(LMI_P2_BR) " pf_cap=" LMI_STR(LMI_PF_CAP);
}
extern "C" LMI_API const char* lmi_last_error(void) { return g_err.c_str(); }

extern "C" LMI_API int lmi_create(int device, lmi_index** out) {
    if (!out) return fail("lmi_create: out is NULL");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("lmi_create: device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail("lmi_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    lmi_index* h = new lmi_index();
    h->device = device;
    h->num_cus = prop.multiProcessorCount;
    // route_group_kernel sorts the buckets in dynamic LDS (route_group_lds: fan-outs up to ROUTE_MAX_BUCKETS)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&route_group_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));  // its static LDS: 4 bytes
    int occ = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_kernel, 256, SCAN_LDS));
    h->scan_blocks_per_cu = std::max(1, std::min(occ, RB == 1 ? 2 : 1));
    // the timing events are created on first use (record): a handle that lives for one lmi_knn_ip call
    // touches 4 of the ring's 1 280
    {   // fp16 subnormal self-test (lmi_prefilter.h): the error bound of the prefilter relies on it
        static std::mutex mu;    // per process; every MI355X behaves the same
        static int cached = -1;
        std::lock_guard<std::mutex> lock(mu);
        if (cached < 0) {
            int* d_ok = nullptr;
            int ok = 0;
            HIPCHK(hipMalloc(&d_ok, sizeof(int)));
            pf_selftest_kernel<<<1, 64>>>(d_ok);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost));
            HIPCHK(hipFree(d_ok));
            cached = ok;
        }
        h->pf_hw_ok = cached == 1;
        if (!h->pf_hw_ok) h->prefilter = false;
    }
    if (const char* e = getenv("LMI_RESCORE_SIMPLE")) h->rescore_streamed = !(e[0] && e[0] != '0');
    if (const char* e = getenv("LMI_PF_NO_REDO")) h->pf_redo = !(e[0] && e[0] != '0');
    if (const char* e = getenv("LMI_PF_SMALL")) h->pf_small = !(e[0] == '0');
    if (const char* e = getenv("LMI_PF_QBOUND")) h->pf_qbound = e[0] && e[0] != '0';
    if (const char* e = getenv("LMI_PF_PRIMARY")) h->pf_primary = e[0] && e[0] != '0';
    if (const char* e = getenv("LMI_PS_WIDE")) h->ps_force_wide = e[0] == '1' ? 1 : e[0] == '0' ? 0 : -1;
    if (const char* e = getenv("LMI_FRONT")) h->use_front = !(e[0] == '0');
    if (const char* e = getenv("LMI_P2_GRADED")) h->graded_chunks = !(e[0] == '0');
    if (const char* e = getenv("LMI_P2_CHUNKS")) (void)sscanf(e, "%d,%d,%d", &h->chunk_lvl_rows[0], &h->chunk_lvl_rows[1], &h->chunk_lvl_rows[2]);
    if (const char* e = getenv("LMI_P2_CHUNK_FRAC")) (void)sscanf(e, "%f,%f", &h->chunk_frac[0], &h->chunk_frac[1]);
    if (const char* e = getenv("LMI_TAIL")) h->use_tail = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
    if (const char* e = getenv("LMI_FR_DEBUG")) { if (e[0] == '1') { CHK(h->fr_dbg.reserve(256)); HIPCHK(hipMemset(h->fr_dbg.p, 0, 256)); } }
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->wall_khz = (double)khz;
    }
    // per handle = per device (a process may hold handles on several devices; the attribute is per device)
#define LMI_PS_ATTR(K) \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass2_small_kernel<K, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(K, false))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass2_small_kernel<K, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(K, false))); \
    if constexpr (ps_has_wide(K)) { \
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass2_small_kernel<K, false, ps_has_wide(K)>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(K, true))); \
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass2_small_kernel<K, true, ps_has_wide(K)>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(K, true))); \
    }
    // every K: the dynamic part alone stays under 64 KiB up to K = 5, but the static arrays beside it (queue prefix, candidate
    // list, item) put the block's total above it from K = 5 on
    LMI_PS_ATTR(1) LMI_PS_ATTR(2) LMI_PS_ATTR(3) LMI_PS_ATTR(4) LMI_PS_ATTR(5) LMI_PS_ATTR(6) LMI_PS_ATTR(7) LMI_PS_ATTR(8)
#undef LMI_PS_ATTR
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_SMALL_LDS_CAP));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_SMALL_LDS_CAP));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_SMALL_LDS_CAP));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rescore_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_SMALL_LDS_CAP));
#define LMI_TL_ATTR(GV) \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<GV>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_SMALL_LDS_CAP));
    LMI_TL_ATTR(1) LMI_TL_ATTR(2) LMI_TL_ATTR(3) LMI_TL_ATTR(4)
#undef LMI_TL_ATTR
    *out = h;
    return 0;
}

extern "C" LMI_API int lmi_destroy(lmi_index* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamDestroy(h->side);
        (void)hipEventDestroy(h->side_fork);
        (void)hipEventDestroy(h->side_join);
    }
    for (auto& b : h->Wf) b.release();
    for (auto& b : h->bias) b.release();
    DevBuf* bufs[] = {&h->slab, &h->ids_slab, &h->pos, &h->d_nb_rows, &h->d_rb_start, &h->d_nch, &h->stage,
                      &h->act[0], &h->act[1], &h->xfrag, &h->logits, &h->order, &h->q_nav, &h->q_srch, &h->m,
                      &h->cb_start, &h->item_base, &h->part_base, &h->stats, &h->head, &h->slot_local,
                      &h->slot_col, &h->colmap, &h->qfrag, &h->grp, &h->col_thr, &h->slab16, &h->rowmajor, &h->xscale, &h->xmaxbits, &h->bnorm, &h->bdelta, &h->qdelta, &h->qnorm, &h->qscale, &h->qfrag16, &h->eps2, &h->cand_cnt, &h->cand_row, &h->cand_s, &h->fallback, &h->pf_bound, &h->nkeep, &h->redo, &h->part_score, &h->part_row, &h->rank_d,
                      &h->rank_id, &h->out_d, &h->out_id, &h->out_key, &h->x_log, &h->x_ext, &h->x_off, &h->fb_list, &h->grp_scratch, &h->ts_ring, &h->fr_dbg, &h->cb_alloc, &h->cb_bucket};
    for (DevBuf* b : bufs) b->release();
    if (h->h_oflag) (void)hipHostFree(h->h_oflag);
    for (auto& m : h->node_models) { for (auto& b : m.Wf) b.release(); for (auto& b : m.bias) b.release(); }
    DevBuf* nav[] = {&h->d_models, &h->d_child_offset, &h->d_child_model, &h->d_child_bucket, &h->pq_prob, &h->pq_ent, &h->pq_len, &h->surv_row, &h->rs_flag, &h->rs_active, &h->gather_send, &h->gather_recv, &h->aug_rows, &h->q_aug, &h->qn2,
                     &h->nav_len, &h->nav_slab, &h->nav_ent, &h->nav_count, &h->nav_colq, &h->nav_active};
    for (DevBuf* b : nav) b->release();
    for (int r = 0; r < lmi_index::EV_RING; ++r)
        for (int i = 0; i < 10; ++i)
            if (h->ev_ring[r][i]) (void)hipEventDestroy(h->ev_ring[r][i]);
    delete h;
    return 0;
}

static int build_descs(lmi_index* h);

// A second handle on the SAME index: its MLP weights, tree and bucket slabs are the parent's memory (borrowed), its
// per-call workspaces, stream, side stream and timing events are its own.  Two searches can then be in flight on one
// index (one per handle, each on its own stream): the pipeline alternates handles so that a batch's kernels start in
// the tails of the previous batch's.  The clone must be destroyed before the parent and the parent's index must not be
// rebuilt while it lives.
extern "C" LMI_API int lmi_clone_view(lmi_index* h, lmi_index** out) {
    if (!h || !out) return fail("lmi_clone_view: NULL argument");
    if (h->building) return fail("lmi_clone_view: the parent's index is being built");
    CHK(set_dev(h));
    CHK(build_descs(h));                       // the device copy of the model descriptors is shared as it stands
    HIPCHK(hipStreamSynchronize(h->stream));
    lmi_index* c = new lmi_index(*h);          // vectors are copied, DevBufs are pointer copies: sorted out below
    c->stream = nullptr; c->side = nullptr; c->side_fork = nullptr; c->side_join = nullptr;
    for (auto& b : c->Wf) b.borrow();
    for (auto& b : c->bias) b.borrow();
    for (auto& m : c->node_models) { for (auto& b : m.Wf) b.borrow(); for (auto& b : m.bias) b.borrow(); }
    DevBuf* shared[] = {&c->d_models, &c->d_child_offset, &c->d_child_model, &c->d_child_bucket, &c->slab, &c->ids_slab, &c->pos,
                        &c->d_nb_rows, &c->d_rb_start, &c->d_nch, &c->slab16, &c->rowmajor, &c->xscale, &c->xmaxbits, &c->bnorm, &c->bdelta};
    for (DevBuf* b : shared) b->borrow();
    DevBuf* own[] = {&c->gather_send, &c->gather_recv, &c->pq_prob, &c->pq_ent, &c->pq_len, &c->nav_len, &c->nav_slab, &c->nav_ent,
                     &c->nav_count, &c->nav_colq, &c->nav_active, &c->aug_rows, &c->q_aug, &c->qn2, &c->stage, &c->qdelta, &c->qnorm,
                     &c->qscale, &c->qfrag16, &c->eps2, &c->cand_cnt, &c->cand_row, &c->cand_s, &c->fallback, &c->pf_bound, &c->nkeep, &c->redo,
                     &c->surv_row, &c->rs_flag, &c->rs_active, &c->act[0], &c->act[1], &c->xfrag, &c->logits, &c->order, &c->q_nav,
                     &c->q_srch, &c->m, &c->cb_start, &c->item_base, &c->part_base, &c->stats, &c->head, &c->slot_local, &c->slot_col,
                     &c->colmap, &c->qfrag, &c->grp, &c->col_thr, &c->part_score, &c->part_row, &c->rank_d, &c->rank_id, &c->out_d,
                     &c->out_id, &c->out_key, &c->x_log, &c->x_ext, &c->x_off, &c->fb_list, &c->grp_scratch, &c->ts_ring, &c->fr_dbg, &c->cb_alloc, &c->cb_bucket};
    for (DevBuf* b : own) b->forget();
    c->ts_set = nullptr;
    memset(c->ts_mask, 0, sizeof(c->ts_mask));
    c->h_oflag = nullptr;
    c->overflow_armed = 0;
    c->x_cap = 0;
    memset(c->ev_ring, 0, sizeof(c->ev_ring));
    memset(c->valid_ring, 0, sizeof(c->valid_ring));
    c->ev_cur = 0; c->ev_calls = 0;
    c->ev = c->ev_ring[0]; c->ev_valid = c->valid_ring[0];
    c->stats_pending = false;
    c->last_nslots = 0; c->last_fast = false;
    *out = c;
    return 0;
}

extern "C" LMI_API int lmi_set_stream(lmi_index* h, void* s) {
    if (!h) return fail("lmi_set_stream: NULL handle");
    h->stream = reinterpret_cast<hipStream_t>(s);
    return 0;
}

extern "C" LMI_API int lmi_set_chunk_rows(lmi_index* h, int rows) {
    if (!h) return fail("lmi_set_chunk_rows: NULL handle");
    if (rows < P2_TILE_ROWS || rows % P2_TILE_ROWS) return fail("lmi_set_chunk_rows: rows must be a positive multiple of %d", P2_TILE_ROWS);
    if (h->building || h->built) return fail("lmi_set_chunk_rows: must be called before lmi_buckets_begin");
    h->chunk_rows = rows;
    h->chunk_rows_auto = false;
    return 0;
}

extern "C" LMI_API int lmi_set_metric(lmi_index* h, int metric) {
    if (!h) return fail("lmi_set_metric: NULL handle");
    if (metric != LMI_METRIC_IP && metric != LMI_METRIC_L2) return fail("lmi_set_metric: unknown metric %d", metric);
    if ((h->built || h->building) && metric != h->metric) return fail("lmi_set_metric: the metric is fixed once lmi_buckets_begin has run");
    h->metric = metric;
    return 0;
}

extern "C" LMI_API int lmi_set_prefilter(lmi_index* h, int on) {
    if (!h) return fail("lmi_set_prefilter: NULL handle");
    if ((h->built || h->building) && (on != 0) != h->prefilter)
        return fail("lmi_set_prefilter: the mode is fixed once lmi_buckets_begin has run (the index is stored differently)");
    if (on && !h->pf_hw_ok) return fail("lmi_set_prefilter: fp16 subnormal self-test failed on this device; the prefilter's error bound does not hold");
    if (on != 0 && on != 1) return fail("lmi_set_prefilter: mode %d unknown (0: all-f32 scan, 1: fp16 prefilter + exact re-rank)", on);
    h->prefilter = on != 0;
    return 0;
}

// upload a row-major host matrix and pack it fragment-major (rows padded to 32, K to 8*KG)
static int pack_from_host(lmi_index* h, const float* src, int rows, int cols, int n_rb, int KG, DevBuf& dst) {
    CHK(h->stage.reserve((size_t)rows * cols * sizeof(float)));
    HIPCHK(hipMemcpyAsync(h->stage.p, src, (size_t)rows * cols * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CHK(dst.reserve((size_t)n_rb * KG * 1024));
    long long total = (long long)n_rb * 32 * KG;
    pack_gather_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(h->stage.as<float>(), cols, nullptr, rows,
                                                               (long long)n_rb * 32, KG, dst.as<float4>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));  // staging buffer is reused by the next call
    return 0;
}

// packs one Linear stack (torch layout W[out][in]) fragment-major: shared by lmi_set_mlp and lmi_nav_set_model
static int pack_model(lmi_index* h, const char* who, int n_layers, const int* dims, const float* const* W, const float* const* b,
                      std::vector<int>& o_dims, std::vector<int>& o_nrb, std::vector<int>& o_KG, std::vector<DevBuf>& o_W,
                      std::vector<DevBuf>& o_b) {
    if (n_layers < 1 || n_layers > LMI_MAX_LAYERS) return fail("%s: n_layers %d out of range", who, n_layers);
    for (int i = 0; i <= n_layers; ++i)
        if (dims[i] < 1) return fail("%s: dims[%d] = %d", who, i, dims[i]);
    for (auto& x : o_W) x.release();
    for (auto& x : o_b) x.release();
    o_dims.assign(dims, dims + n_layers + 1);
    o_nrb.assign(n_layers, 0);
    o_KG.assign(n_layers, 0);
    o_W.assign(n_layers, DevBuf());
    o_b.assign(n_layers, DevBuf());
    for (int i = 0; i < n_layers; ++i) {
        o_nrb[i] = cdiv(dims[i + 1], 32);
        o_KG[i] = (i == 0) ? cdiv(dims[0], 8) : o_nrb[i - 1] * 4;  // hidden K = padded features
        if (!W[i] || !b[i]) return fail("%s: NULL weight/bias for layer %d", who, i);
        CHK(pack_from_host(h, W[i], dims[i + 1], dims[i], o_nrb[i], o_KG[i], o_W[i]));
        std::vector<float> bp((size_t)o_nrb[i] * 32, 0.0f);
        std::copy(b[i], b[i] + dims[i + 1], bp.begin());
        CHK(o_b[i].reserve(bp.size() * sizeof(float)));
        HIPCHK(hipMemcpy(o_b[i].p, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" LMI_API int lmi_set_mlp(lmi_index* h, int n_layers, const int* dims, const float* const* W,
                           const float* const* b) {
    if (!h) return fail("lmi_set_mlp: NULL handle");
    CHK(set_dev(h));
    h->desc_dirty = true;
    h->n_layers = 0;
    CHK(pack_model(h, "lmi_set_mlp", n_layers, dims, W, b, h->dims, h->n_rb, h->KG, h->Wf, h->bias));
    h->n_layers = n_layers;
    return 0;
}

extern "C" LMI_API int lmi_set_fused_mlp(lmi_index* h, int on) {
    if (!h) return fail("lmi_set_fused_mlp: NULL handle");
    if (on < 0 || on > 2) return fail("lmi_set_fused_mlp: mode %d outside 0..2", on);
    h->fused_mlp = on;
    return 0;
}

// ---- multi-level index: the internal nodes' models and the tree (lmi_mlp_fused.h) ----
extern "C" LMI_API int lmi_nav_set_model(lmi_index* h, int model_id, int n_layers, const int* dims, const float* const* W,
                                 const float* const* b) {
    if (!h) return fail("lmi_nav_set_model: NULL handle");
    if (model_id < 1 || model_id > 1 << 20) return fail("lmi_nav_set_model: model_id %d (the root, model 0, is lmi_set_mlp)", model_id);
    CHK(set_dev(h));
    if ((size_t)model_id > h->node_models.size()) h->node_models.resize(model_id);
    auto& m = h->node_models[model_id - 1];
    h->desc_dirty = true;
    h->tree_set = false;
    m.n_layers = 0;
    CHK(pack_model(h, "lmi_nav_set_model", n_layers, dims, W, b, m.dims, m.n_rb, m.KG, m.Wf, m.bias));
    m.n_layers = n_layers;
    return 0;
}

extern "C" LMI_API int lmi_nav_set_tree(lmi_index* h, int n_models, const int32_t* child_offset, const int32_t* child_model,
                                const int32_t* child_bucket) {
    if (!h) return fail("lmi_nav_set_tree: NULL handle");
    if (h->n_layers == 0) return fail("lmi_nav_set_tree: no root model (lmi_set_mlp)");
    if (n_models != 1 + (int)h->node_models.size()) return fail("lmi_nav_set_tree: %d models, %d set (root + lmi_nav_set_model)", n_models, 1 + (int)h->node_models.size());
    if (!child_offset || !child_model || !child_bucket || child_offset[0] != 0) return fail("lmi_nav_set_tree: bad arguments");
    for (int m = 0; m < n_models; ++m) {
        const int classes = m == 0 ? h->dims[h->n_layers] : (h->node_models[m - 1].n_layers ? h->node_models[m - 1].dims.back() : -1);
        if (classes < 0) return fail("lmi_nav_set_tree: model %d has no weights (lmi_nav_set_model)", m);
        if (child_offset[m + 1] - child_offset[m] != classes) return fail("lmi_nav_set_tree: model %d has %d classes, %d children listed", m, classes, child_offset[m + 1] - child_offset[m]);
    }
    const int total = child_offset[n_models];
    for (int e = 0; e < total; ++e) {
        if (child_model[e] < -1 || child_model[e] == 0 || child_model[e] >= n_models) return fail("lmi_nav_set_tree: child_model[%d] = %d", e, child_model[e]);
        if (child_bucket[e] < -2) return fail("lmi_nav_set_tree: child_bucket[%d] = %d", e, child_bucket[e]);
    }
    CHK(set_dev(h));
    h->h_child_offset.assign(child_offset, child_offset + n_models + 1);
    h->h_child_model.assign(child_model, child_model + total);
    h->h_child_bucket.assign(child_bucket, child_bucket + total);
    CHK(h->d_child_offset.reserve((size_t)(n_models + 1) * 4));
    CHK(h->d_child_model.reserve((size_t)std::max(total, 1) * 4));
    CHK(h->d_child_bucket.reserve((size_t)std::max(total, 1) * 4));
    HIPCHK(hipMemcpy(h->d_child_offset.p, child_offset, (size_t)(n_models + 1) * 4, hipMemcpyHostToDevice));
    if (total) {
        HIPCHK(hipMemcpy(h->d_child_model.p, child_model, (size_t)total * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_child_bucket.p, child_bucket, (size_t)total * 4, hipMemcpyHostToDevice));
    }
    h->tree_set = true;
    return 0;
}

// device descriptors of every model + the LDS plan of mlp_fused_kernel for the current model set
static int build_descs(lmi_index* h) {
    if (!h->desc_dirty) return 0;
    const int nm = 1 + (int)h->node_models.size();
    std::vector<ModelDesc> D(nm);
    bool ok = h->n_layers > 0, logits_lds = true;
    int w0 = 0, w1 = 0;
    auto add = [&](ModelDesc& d, int n_layers, const std::vector<int>& dims, const std::vector<int>& n_rb, const std::vector<int>& KG,
                   const std::vector<DevBuf>& Wf, const std::vector<DevBuf>& bias) {
        memset(&d, 0, sizeof(d));
        d.n_layers = n_layers;
        if (n_layers == 0) { ok = false; return; }
        for (int i = 0; i <= n_layers; ++i) d.dims[i] = dims[i];
        for (int i = 0; i < n_layers; ++i) {
            d.KG[i] = KG[i];
            d.W[i] = Wf[i].as<float4>();
            d.b[i] = bias[i].as<float>();
            const int padded = n_rb[i] * 32;
            const bool last = i + 1 == n_layers;
            if (padded > FM_MAXH) { if (last) logits_lds = false; else ok = false; continue; }
            int& wref = (i & 1) ? w1 : w0;
            wref = std::max(wref, padded);
        }
    };
    add(D[0], h->n_layers, h->dims, h->n_rb, h->KG, h->Wf, h->bias);
    for (int m = 1; m < nm; ++m) {
        const auto& M = h->node_models[m - 1];
        add(D[m], M.n_layers, M.dims, M.n_rb, M.KG, M.Wf, M.bias);
    }
    h->fm_s0 = w0 + 1;
    h->fm_s1 = w1 + 1;
    h->fm_act0 = FM_COLS * h->fm_s0;
    h->fm_lds = (2 * FM_COLS * FM_CHUNK_S + FM_COLS * h->fm_s0 + FM_COLS * h->fm_s1) * 4;
    if (h->fm_lds > 160 * 1024 - 1024) ok = false;
    h->fm_ok = ok;
    h->fm_logits_lds = logits_lds ? 1 : 0;
    CHK(h->d_models.reserve(sizeof(ModelDesc) * nm));
    HIPCHK(hipMemcpy(h->d_models.p, D.data(), sizeof(ModelDesc) * nm, hipMemcpyHostToDevice));
    if (ok) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<FM_TOPK>), hipFuncAttributeMaxDynamicSharedMemorySize, h->fm_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<FM_PROBA>), hipFuncAttributeMaxDynamicSharedMemorySize, h->fm_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<FM_NAV>), hipFuncAttributeMaxDynamicSharedMemorySize, h->fm_lds));
    }
    h->desc_dirty = false;
    return 0;
}

static void fused_base(lmi_index* h, const float* d_q, int nq, FusedParams& P) {
    memset(&P, 0, sizeof(P));
    P.models = h->d_models.as<ModelDesc>();
    P.n_models = 1 + (int)h->node_models.size();
    P.x = d_q;
    P.d = h->dims[0];
    P.nq = nq;
    P.s0 = h->fm_s0;
    P.s1 = h->fm_s1;
    P.act0_floats = h->fm_act0;
    P.logits_in_lds = h->fm_logits_lds;
}

extern "C" LMI_API int lmi_buckets_begin(lmi_index* h, int64_t N, int d, int L, const int64_t* labels,
                                 const uint32_t* ids, const uint8_t* owned) {
    if (!h) return fail("lmi_buckets_begin: NULL handle");
    if (N < 0 || d < 1 || L < 1 || (N > 0 && !labels)) return fail("lmi_buckets_begin: bad arguments");
    if (N >= (1ll << 31) - 64ll * L) return fail("lmi_buckets_begin: N too large for 32-bit positions");
    if (L >= (1 << ROUTE_ID_BITS)) return fail("lmi_buckets_begin: %d buckets, the routing kernels take fewer than %d", L, 1 << ROUTE_ID_BITS);
    CHK(set_dev(h));
    h->N = N;
    h->d_user = d;
    h->d = h->metric == LMI_METRIC_L2 ? (int)rup(d + 1, 4) : d;  // L2: + the -|x|^2/2 column (sim_to_dist, lmi_kernels.h)
    d = h->d;
    h->dp = (int)rup(d, 4);   // floats per row of the row-major f32 copy: 16-byte rows for the streamed re-rank (d = 45: 48, zero-filled)
    h->L = L;
    h->KGs = (int)rup(cdiv(d, 8), STAGE_G);
    h->built = false;
    h->h_nb_rows.assign(L, 0);
    std::vector<unsigned char> seen(owned ? L : 0, 0);   // (a sharded rank: which buckets hold rows on ANY rank)
    for (int64_t i = 0; i < N; ++i) {
        int64_t b = labels[i];
        if (b < 0 || b >= L) return fail("lmi_buckets_begin: labels[%lld] = %lld outside [0,%d)", (long long)i, (long long)b, L);
        if (!owned || owned[b]) h->h_nb_rows[b]++;
        if (owned) seen[b] = 1;
    }
    h->h_rb_start.assign(L + 1, 0);
    h->h_nch.assign(L, 0);
    // Chunk rows not set by the caller: small indexes (or small shards) get smaller chunks so that a scan has
    // many more work items than the 256 blocks that share them (100 000 rows x 1 000 queries: pass 2 0.169 ms
    // with 2048-row chunks, 0.089 ms with 256; the 1.25M-row shard of an 8-way split that holds the largest
    // bucket: 0.90 ms with 2048, 0.69 ms with 512); 10M rows keep 2048.
    // A bucket is scanned in at most 1024 chunks: very large buckets get larger chunks than that.
    {
        int max_rows = 0;
        long long owned_rows = 0;
        // buckets with rows on any rank: the queries of a batch spread over all of them, whoever owns them
        int nonempty = 0;
        for (int b = 0; b < L; ++b) { max_rows = std::max(max_rows, h->h_nb_rows[b]); owned_rows += h->h_nb_rows[b]; nonempty += owned ? seen[b] : h->h_nb_rows[b] > 0; }
        h->n_nonempty = std::max(1, nonempty);
        if (h->chunk_rows_auto) {
            h->chunk_rows = (int)std::min<long long>(2048, std::max<long long>(P2_TILE_ROWS, rup(owned_rows / 4096, P2_TILE_ROWS)));
            // d <= 128 (lmi_pass2_small.h): a 2048-row item is ~5 us of work there, about what taking it from the queue and
            // staging its query fragments costs, while 8192-row items are too few to share out evenly (10M x 45, pass 2 at
            // 1024 / 2048 / 4096 / 8192 rows per item: 0.590 / 0.441 / 0.385 / 0.412 ms): up to 4096
            if (h->pf_small && cdiv(d, 16) <= PS_MAXKG)
                h->chunk_rows = (int)std::min<long long>(4096, std::max<long long>(P2_TILE_ROWS, rup(owned_rows / 1024, P2_TILE_ROWS)));
            // all-f32 scan (scan_kernel: 128-query tiles, so a bucket's chunk is read by several items): the chunk's 4 d-byte rows should
            // stay in an XCD's 4-MiB L2 until the bucket's last query tile has come by -- 10M x 768: 2 048-row chunks (6 MB) 35.28 ms,
            // 1 024-row chunks 34.76 (profiles/r05_exact_chunks.txt)
            if (!h->prefilter)
                h->chunk_rows = (int)std::max<long long>(P2_TILE_ROWS, std::min<long long>(h->chunk_rows, (3ll << 20) / (4ll * d) / P2_TILE_ROWS * P2_TILE_ROWS));
        }
        const int need = (int)rup(cdiv(max_rows, 1024), 256);
        if (need > h->chunk_rows) h->chunk_rows = need;
    }
    const int chunk_rb = h->chunk_rows / 32;
    for (int b = 0; b < L; ++b) {
        int nrb = cdiv(h->h_nb_rows[b], 32);
        h->h_rb_start[b + 1] = h->h_rb_start[b] + nrb;
        h->h_nch[b] = cdiv(nrb, chunk_rb);
    }
    h->n_rb_total = h->h_rb_start[L];
    // bucket-contiguous position of every object (stable: ascending original row inside a bucket,
    // the order pandas groupby yields) and the id of every slab row
    std::vector<int> pos((size_t)N);
    std::vector<uint32_t> ids_slab((size_t)std::max<int64_t>(h->n_rb_total, 1) * 32, 0u);
    std::vector<int> fill(L, 0);
    for (int64_t i = 0; i < N; ++i) {
        int b = (int)labels[i];
        if (owned && !owned[b]) { pos[i] = -1; continue; }
        int p = h->h_rb_start[b] * 32 + fill[b]++;
        pos[i] = p;
        ids_slab[p] = ids ? ids[i] : (uint32_t)(i + 1);  // search.py:190-191: 1-based labels
    }
    const size_t slab_bytes = (size_t)std::max<int64_t>(h->n_rb_total, 1) * h->KGs * 1024;
    if (h->prefilter) {  // row-major f32 (exact re-rank / fallback / read-back); fp16 fragments at buckets_end
        const size_t rm_bytes = (size_t)std::max<int64_t>(h->n_rb_total, 1) * 32 * h->dp * 4;
        h->slab.release();
        CHK(h->rowmajor.reserve(rm_bytes));
        HIPCHK(hipMemsetAsync(h->rowmajor.p, 0, rm_bytes, h->stream));
    } else {             // f32 fragments for the all-f32 scan
        h->rowmajor.release();
        h->slab16.release();
        CHK(h->slab.reserve(slab_bytes));
        HIPCHK(hipMemsetAsync(h->slab.p, 0, slab_bytes, h->stream));
    }
    CHK(h->ids_slab.reserve(ids_slab.size() * 4));
    HIPCHK(hipMemcpy(h->ids_slab.p, ids_slab.data(), ids_slab.size() * 4, hipMemcpyHostToDevice));
    CHK(h->pos.reserve(std::max<size_t>(pos.size(), 1) * 4));
    if (N) HIPCHK(hipMemcpy(h->pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    CHK(h->d_nb_rows.reserve(L * 4));
    CHK(h->d_rb_start.reserve((L + 1) * 4));
    CHK(h->d_nch.reserve(L * 4));
    HIPCHK(hipMemcpy(h->d_nb_rows.p, h->h_nb_rows.data(), L * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_rb_start.p, h->h_rb_start.data(), (L + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_nch.p, h->h_nch.data(), L * 4, hipMemcpyHostToDevice));
    h->rows_added = 0;
    h->indexed_ingest = false;
    h->owned_total = 0;
    for (int b = 0; b < L; ++b) h->owned_total += h->h_nb_rows[b];
    h->building = true;
    return 0;
}

// rows [nrows][d] are objects row0.. (index == NULL) or objects index[0..nrows) (host or device like `rows`)
static int add_rows_impl(lmi_index* h, const float* rows, int64_t row0, const int64_t* index, int64_t nrows, int on_device) {
    CHK(set_dev(h));
    const int64_t piece = std::max<int64_t>(1, (256ll << 20) / ((int64_t)h->d * 4));
    for (int64_t off = 0; off < nrows; off += piece) {
        const int64_t n = std::min(piece, nrows - off);
        const float* src = rows + off * h->d_user;
        const long long* idx = index ? reinterpret_cast<const long long*>(index + off) : nullptr;
        if (!on_device) {
            const size_t row_bytes = (size_t)n * h->d_user * 4;
            CHK(h->stage.reserve(row_bytes + (index ? (size_t)n * 8 : 0)));
            HIPCHK(hipMemcpyAsync(h->stage.p, src, row_bytes, hipMemcpyHostToDevice, h->stream));
            src = h->stage.as<float>();
            if (index) {
                HIPCHK(hipMemcpyAsync(h->stage.as<char>() + row_bytes, index + off, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
                idx = reinterpret_cast<const long long*>(h->stage.as<char>() + row_bytes);
            }
        }
        if (h->metric == LMI_METRIC_L2) {  // the piece with its norm column, then ingested like any d-column piece
            CHK(h->aug_rows.reserve((size_t)n * h->d * 4));
            augment_copy_kernel<<<cdiv((long long)n * h->d, 256), 256, 0, h->stream>>>(src, h->d_user, h->d, n, h->aug_rows.as<float>());
            HIPCHK(hipGetLastError());
            augment_norm_kernel<<<cdiv(n, 256), 256, 0, h->stream>>>(src, h->d_user, h->d, n, h->aug_rows.as<float>(), nullptr);
            HIPCHK(hipGetLastError());
            src = h->aug_rows.as<float>();
        }
        if (h->prefilter) {
            long long total = (long long)n * h->d;
            scatter_rows_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(src, h->d, h->pos.as<int>(), row0 + off, idx, (long long)h->N, n,
                                                                        h->rowmajor.as<float>(), h->dp);
        } else {
            long long total = (long long)n * h->KGs;
            pack_scatter_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(src, h->d, h->pos.as<int>(), row0 + off, idx, (long long)h->N, n,
                                                                        h->KGs, h->slab.as<float4>());
        }
        HIPCHK(hipGetLastError());
        if (!on_device) HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" LMI_API int lmi_buckets_add_rows(lmi_index* h, const float* rows, int64_t row0, int64_t nrows, int on_device) {
    if (!h || !h->building) return fail("lmi_buckets_add_rows: call lmi_buckets_begin first");
    if (h->indexed_ingest) return fail("lmi_buckets_add_rows: this build already uses lmi_buckets_add_owned_rows");
    if (row0 < 0 || nrows < 0 || row0 + nrows > h->N) return fail("lmi_buckets_add_rows: rows [%lld,%lld) outside [0,%lld)", (long long)row0, (long long)(row0 + nrows), (long long)h->N);
    if (nrows == 0) return 0;
    CHK(add_rows_impl(h, rows, row0, nullptr, nrows, on_device));
    h->rows_added += nrows;
    return 0;
}

extern "C" LMI_API int lmi_buckets_add_owned_rows(lmi_index* h, const float* rows, const int64_t* index, int64_t nrows,
                                          int on_device) {
    if (!h || !h->building) return fail("lmi_buckets_add_owned_rows: call lmi_buckets_begin first");
    if (h->rows_added > 0 && !h->indexed_ingest) return fail("lmi_buckets_add_owned_rows: this build already uses lmi_buckets_add_rows");
    if (nrows < 0 || (nrows > 0 && (!rows || !index))) return fail("lmi_buckets_add_owned_rows: bad arguments");
    if (!on_device)
        for (int64_t i = 0; i < nrows; ++i)
            if (index[i] < 0 || index[i] >= h->N) return fail("lmi_buckets_add_owned_rows: index[%lld] = %lld outside [0,%lld)", (long long)i, (long long)index[i], (long long)h->N);
    h->indexed_ingest = true;
    if (nrows == 0) return 0;
    CHK(add_rows_impl(h, rows, 0, index, nrows, on_device));
    h->rows_added += nrows;
    return 0;
}

extern "C" LMI_API int lmi_buckets_end(lmi_index* h) {
    if (!h || !h->building) return fail("lmi_buckets_end: call lmi_buckets_begin first");
    const int64_t expect = h->indexed_ingest ? h->owned_total : h->N;
    if (h->rows_added != expect) return fail("lmi_buckets_end: %lld of %lld rows were added", (long long)h->rows_added, (long long)expect);
    CHK(set_dev(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->pos.release();
    h->have16 = false;
    if (h->prefilter && h->n_rb_total > 0) {
        // fp16 copy of the slab for the prefilter: one power-of-two scale for the whole index
        // (pass2_kernel's stages hold two k16-groups; the low-dimensional form has no stages: d = 45 is 48 wide, not 64)
        h->KG16 = (h->pf_small && cdiv(h->d, 16) <= PS_MAXKG) ? (int)cdiv(h->d, 16) : (int)rup(cdiv(h->d, 16), PF_STAGE_G);
        const long long n_rows = (long long)h->n_rb_total * 32;
        CHK(h->xmaxbits.reserve(16));
        CHK(h->xscale.reserve(16));
        CHK(h->bnorm.reserve((size_t)h->L * 4));
        CHK(h->bdelta.reserve((size_t)h->L * 4));
        // (+ 8 KiB: pass2_kernel's look-ahead requests up to two stages = 4 KiB past the last row-block's fragments before it learns that
        // the item is over; the data is never used, the addresses must be the allocation's)
        CHK(h->slab16.reserve((size_t)h->n_rb_total * h->KG16 * 1024 + 8192));
        HIPCHK(hipMemsetAsync(h->xmaxbits.p, 0, 16, h->stream));
        HIPCHK(hipMemsetAsync(h->bnorm.p, 0, (size_t)h->L * 4, h->stream));
        HIPCHK(hipMemsetAsync(h->bdelta.p, 0, (size_t)h->L * 4, h->stream));
        absmax_kernel<<<h->num_cus * 8, 256, 0, h->stream>>>(h->rowmajor.as<float>(), n_rows * h->dp, h->xmaxbits.as<unsigned>());
        HIPCHK(hipGetLastError());
        make_scale_kernel<<<1, 1, 0, h->stream>>>(h->xmaxbits.as<unsigned>(), h->xscale.as<float>());
        HIPCHK(hipGetLastError());
        const long long total = n_rows * h->KG16 * 2;
        convert16_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(h->rowmajor.as<float>(), h->d, h->dp, n_rows, h->KG16,
                                                                 h->xscale.as<float>(), h->slab16.as<uint4>(), frag16x16(h));
        HIPCHK(hipGetLastError());
        dim3 g(64, h->L);
        bucket_norm_kernel<<<g, 256, 0, h->stream>>>(h->rowmajor.as<float>(), h->d, h->dp, h->d_rb_start.as<int>(),
                                                    h->d_nb_rows.as<int>(), h->xscale.as<float>(), h->bnorm.as<unsigned>(),
                                                    h->bdelta.as<unsigned>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        h->have16 = true;
    }
    h->building = false;
    h->built = true;
    return 0;
}

extern "C" LMI_API int lmi_bucket_sizes(lmi_index* h, int64_t* sizes) {
    if (!h || !(h->built || h->building)) return fail("lmi_bucket_sizes: no buckets");
    for (int b = 0; b < h->L; ++b) sizes[b] = h->h_nb_rows[b];
    return 0;
}

extern "C" LMI_API int lmi_bucket_read(lmi_index* h, int bucket, float* rows, uint32_t* ids) {
    if (!h || !h->built) return fail("lmi_bucket_read: the bucket index is not built");
    if (bucket < 0 || bucket >= h->L) return fail("lmi_bucket_read: bucket %d outside [0,%d)", bucket, h->L);
    const int64_t n = h->h_nb_rows[bucket];
    if (n == 0) return 0;
    CHK(set_dev(h));
    const int64_t p0 = (int64_t)h->h_rb_start[bucket] * 32;
    const int du = h->d_user;  // the caller's columns (the L2 norm column is not returned)
    if (rows && h->prefilter) {
        HIPCHK(hipMemcpy2DAsync(rows, (size_t)du * 4, h->rowmajor.as<float>() + (size_t)p0 * h->dp, (size_t)h->dp * 4, (size_t)du * 4, (size_t)n,
                                hipMemcpyDeviceToHost, h->stream));
    } else if (rows) {
        CHK(h->stage.reserve((size_t)n * du * 4));
        long long total = n * cdiv(du, 8);
        unpack_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(h->slab.as<float4>(), h->KGs, p0, n, du, h->stage.as<float>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(rows, h->stage.p, (size_t)n * du * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (ids) HIPCHK(hipMemcpyAsync(ids, h->ids_slab.as<uint32_t>() + p0, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
static void begin_call(lmi_index* h) {
    h->ev_cur = (h->ev_cur + 1) % lmi_index::EV_RING;
    h->ev = h->ev_ring[h->ev_cur];
    h->ev_valid = h->valid_ring[h->ev_cur];
    for (int i = 0; i < 10; ++i) h->ev_valid[i] = false;
    ++h->ev_calls;
    h->ts_mask[h->ev_cur] = 0u;
    h->ts_set = nullptr;
    if (h->timing_level == 2) {   // device stamps: the call's set of the ring (allocated with the first timed call; a failed allocation: no stamps)
        if (!h->ts_ring.p && h->ts_ring.reserve((size_t)lmi_index::EV_RING * ST_COUNT * 8) != 0) return;
        h->ts_set = h->ts_ring.as<unsigned long long>() + (size_t)h->ev_cur * ST_COUNT;
    }
}
// the device word a kernel of this call writes stamp `idx` to (nullptr: stamps are off)
static unsigned long long* tsp(lmi_index* h, int idx) {
    if (!h->ts_set) return nullptr;
    h->ts_mask[h->ev_cur] |= 1u << idx;
    return h->ts_set + idx;
}
// the end of a call whose last kernel carries no stamp (lmi_mlp_topk, lmi_nav_order ..): one single-thread launch behind it
static int stamp_end(lmi_index* h, int idx) {
    if (unsigned long long* p = tsp(h, idx)) {
        stamp_kernel<<<1, 1, 0, h->stream>>>(p);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static int record(lmi_index* h, int i) {
    // every recorded event is a ~5 us bubble between two kernels: level 3 records every phase boundary, level 1 the call's first
    // and last event (LMI_T_TOTAL), levels 0 and 2 none (2: the kernels stamp the chip's clock themselves, tsp)
    if (h->timing_level == 0 || h->timing_level == 2 || (h->timing_level == 1 && i != 0 && i != 1 && i != 4)) return 0;
    if (!h->ev[i]) HIPCHK(hipEventCreateWithFlags(&h->ev[i], hipEventDisableSystemFence));   // timing only: no system-scope release at the record
    HIPCHK(hipEventRecord(h->ev[i], h->stream));
    h->ev_valid[i] = true;
    return 0;
}

// device pointer to the caller's input (uploads host data into `buf`)
static int input_ptr(lmi_index* h, const void* src, size_t bytes, int on_device, DevBuf& buf, const void** out) {
    if (on_device) { *out = src; return 0; }
    CHK(buf.reserve(bytes));
    HIPCHK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    *out = buf.p;
    return 0;
}

// MLP forward + class ranking (+ softmax when d_probs: then nb == L and d_order receives the full class order)
// Side stream (library-owned): work that does not depend on what the handle's stream runs next is forked onto it and
// joined before its results are needed.  side_fork: the side stream waits for everything enqueued on h->stream so far.
static int side_ensure(lmi_index* h) {
    if (!h->side) {
        HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->side_join, hipEventDisableTiming));
    }
    return 0;
}
static int side_fork(lmi_index* h) {
    CHK(side_ensure(h));
    HIPCHK(hipEventRecord(h->side_fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->side, h->side_fork, 0));
    return 0;
}
static int side_join(lmi_index* h) {
    HIPCHK(hipEventRecord(h->side_join, h->side));
    HIPCHK(hipStreamWaitEvent(h->stream, h->side_join, 0));
    return 0;
}

// the per-layer form: one mlp_layer_kernel launch per Linear, then the ranking (and softmax) kernels, on stream `st`
static int mlp_layers_enqueue(lmi_index* h, hipStream_t st, const float* d_q, int nq, int nb, int* d_order, float* d_logits_out, float* d_probs) {
    const int L = h->dims[h->n_layers];
    const int ncb = cdiv(nq, 32);
    // pack the queries as the B operand of layer 0
    CHK(h->xfrag.reserve((size_t)ncb * h->KG[0] * 1024));
    {
        long long total = (long long)ncb * 32 * h->KG[0];
        pack_gather_kernel<<<cdiv(total, 256), 256, 0, st>>>(d_q, h->dims[0], nullptr, nq, (long long)ncb * 32,
                                                            h->KG[0], h->xfrag.as<float4>(), st == h->stream ? tsp(h, ST_MLP0) : nullptr);
        HIPCHK(hipGetLastError());
    }
    int maxrb = 0;
    for (int i = 0; i + 1 < h->n_layers; ++i) maxrb = std::max(maxrb, h->n_rb[i]);
    for (int i = 0; i < 2; ++i) CHK(h->act[i].reserve((size_t)std::max(1, ncb) * std::max(1, maxrb) * 4 * 1024));
    float* d_logits = d_logits_out;
    if (!d_logits) {
        CHK(h->logits.reserve((size_t)nq * L * 4));
        d_logits = h->logits.as<float>();
    }
    const float4* in = h->xfrag.as<float4>();
    for (int i = 0; i < h->n_layers; ++i) {
        const bool last = i + 1 == h->n_layers;
        // col-blocks per wave: 4 when that already gives every CU two blocks, else 2, else 1 (e.g. the
        // 120-class output layer has 4 feature blocks = one block row; 768->512 on 10 000 queries had 316
        // blocks of CBW 4 on 256 CUs)
        const int rows = cdiv(h->n_rb[i], 4);
        const int cbw = rows * cdiv(ncb, 4) >= 2 * h->num_cus ? 4 : rows * cdiv(ncb, 2) >= 2 * h->num_cus ? 2 : 1;
        dim3 grid(cdiv(ncb, cbw), rows);
        float* o = last ? d_logits : h->act[i & 1].as<float>();
        const int KGn = last ? 0 : h->n_rb[i] * 4;
#define LMI_MLP_LAUNCH(LASTV, CBWV)                                                                        \
        mlp_layer_kernel<LASTV, CBWV><<<grid, 256, 0, st>>>(h->Wf[i].as<float4>(), h->bias[i].as<float>(), in, \
                                                           h->KG[i], h->n_rb[i], ncb, o, KGn, nq, L)
        if (last) {
            if (cbw == 4) LMI_MLP_LAUNCH(true, 4); else if (cbw == 2) LMI_MLP_LAUNCH(true, 2); else LMI_MLP_LAUNCH(true, 1);
        } else {
            if (cbw == 4) LMI_MLP_LAUNCH(false, 4); else if (cbw == 2) LMI_MLP_LAUNCH(false, 2); else LMI_MLP_LAUNCH(false, 1);
            in = reinterpret_cast<const float4*>(o);
        }
#undef LMI_MLP_LAUNCH
        HIPCHK(hipGetLastError());
    }
    rank_classes_kernel<<<nq, 64, 0, st>>>(d_logits, nq, L, nb, d_order);
    HIPCHK(hipGetLastError());
    if (d_probs) {
        softmax_ranked_kernel<<<cdiv(nq, 64), 64, 0, st>>>(d_logits, d_order, nq, L, d_probs);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static int mlp_enqueue(lmi_index* h, const float* d_q, int nq, int nb, int* d_order, float* d_logits_out, float* d_probs = nullptr) {
    if (h->n_layers == 0) return fail("lmi_mlp_topk: no MLP set (lmi_set_mlp)");
    const int L = h->dims[h->n_layers];
    if (nb < 1 || nb > L) return fail("lmi_mlp_topk: n_buckets %d outside [1,%d]", nb, L);
    CHK(build_descs(h));
    // One launch for every layer + ranking when the batch fills the chip (a block = 32 queries, one per CU for the wide
    // models: 8 192 queries 100 us against 138 us for the per-layer kernels); small batches (a rank's slice of a
    // sharded batch, single queries) have too few 32-query blocks for that and take the per-layer kernels, whose grids
    // also split the features (2 048 queries: 79 us against 88 us).  predict_proba always takes the fused kernel.
    const bool fill = cdiv(nq, FM_COLS) * 2 >= h->num_cus || d_probs != nullptr || h->fused_mlp == 2;
    if (h->fused_mlp && h->fm_ok && fill) {
        // every layer, the ranking and the softmax in ONE launch (lmi_mlp_fused.h)
        // A batch whose last round of 32-query blocks would fill under 30 % of the CUs (10 000 queries: 313 blocks = 256 + 57)
        // pays a whole second round for it.  The tail's queries go through the per-layer kernels on a side stream instead,
        // beside the fused kernel's one full round (the fused blocks leave wave slots and half of the MFMA pipe): 182 ->
        // ~125 us at 10 000 queries; identical results (both forms are the canonical chain).
        int grid = cdiv(nq, FM_COLS);
        int nq_head = nq;
        const int rem = grid % h->num_cus;
        if (h->fused_mlp == 1 && !d_probs && !d_logits_out && h->fm_logits_lds && grid > h->num_cus && rem > 0 && rem * 10 < h->num_cus * 3) {
            nq_head = (grid - rem) * FM_COLS;
            grid -= rem;
        }
        FusedParams P;
        fused_base(h, d_q, nq_head, P);
        float* d_logits = d_logits_out;
        if (!h->fm_logits_lds && !d_logits) {  // wide output layer: logits through global memory, ranked below
            CHK(h->logits.reserve((size_t)nq * L * 4));
            d_logits = h->logits.as<float>();
        }
        P.logits_out = d_logits;
        P.nb = nb;
        P.order = d_order;
        P.probs = d_probs;
        P.classes = d_order;
        P.ts = tsp(h, ST_MLP0);
        if (nq_head < nq) CHK(side_fork(h));
        if (d_probs) mlp_fused_kernel<FM_PROBA><<<grid, 256, h->fm_lds, h->stream>>>(P);
        else mlp_fused_kernel<FM_TOPK><<<grid, 256, h->fm_lds, h->stream>>>(P);
        HIPCHK(hipGetLastError());
        if (nq_head < nq) {
            CHK(mlp_layers_enqueue(h, h->side, d_q + (size_t)nq_head * h->dims[0], nq - nq_head, nb, d_order + (size_t)nq_head * nb, nullptr, nullptr));
            CHK(side_join(h));
        }
        if (!h->fm_logits_lds) {
            rank_classes_kernel<<<nq, 64, 0, h->stream>>>(d_logits, nq, L, nb, d_order);
            HIPCHK(hipGetLastError());
            if (d_probs) {
                softmax_ranked_kernel<<<cdiv(nq, 64), 64, 0, h->stream>>>(d_logits, d_order, nq, L, d_probs);
                HIPCHK(hipGetLastError());
            }
        }
        return 0;
    }
    return mlp_layers_enqueue(h, h->stream, d_q, nq, nb, d_order, d_logits_out, d_probs);
}

extern "C" LMI_API int lmi_mlp_topk(lmi_index* h, const float* queries_nav, int nq, int nb, int32_t* bucket_order,
                            float* logits, int on_device) {
    if (!h) return fail("lmi_mlp_topk: NULL handle");
    if (nq < 0) return fail("lmi_mlp_topk: nq < 0");
    if (nq == 0) return 0;
    CHK(set_dev(h));
    if (h->n_layers == 0) return fail("lmi_mlp_topk: no MLP set (lmi_set_mlp)");
    const int L = h->dims[h->n_layers];
    const void* d_q = nullptr;
    CHK(input_ptr(h, queries_nav, (size_t)nq * h->dims[0] * 4, on_device, h->q_nav, &d_q));
    int* d_order = bucket_order;
    float* d_logits = logits;
    if (!on_device) {
        CHK(h->order.reserve((size_t)nq * nb * 4));
        d_order = h->order.as<int>();
        if (logits) { CHK(h->logits.reserve((size_t)nq * L * 4)); d_logits = h->logits.as<float>(); }
    }
    begin_call(h);
    CHK(record(h, 0));
    CHK(mlp_enqueue(h, static_cast<const float*>(d_q), nq, nb, d_order, d_logits));
    CHK(record(h, 1));
    CHK(stamp_end(h, ST_MLP1));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(bucket_order, d_order, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        if (logits) HIPCHK(hipMemcpyAsync(logits, d_logits, (size_t)nq * L * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" LMI_API int lmi_mlp_proba(lmi_index* h, const float* queries_nav, int nq, float* probs, int32_t* classes,
                             int on_device) {
    if (!h) return fail("lmi_mlp_proba: NULL handle");
    if (nq < 0) return fail("lmi_mlp_proba: nq < 0");
    if (nq == 0) return 0;
    CHK(set_dev(h));
    if (h->n_layers == 0) return fail("lmi_mlp_proba: no MLP set (lmi_set_mlp)");
    const int L = h->dims[h->n_layers];
    const void* d_q = nullptr;
    CHK(input_ptr(h, queries_nav, (size_t)nq * h->dims[0] * 4, on_device, h->q_nav, &d_q));
    int* d_order = classes;
    float* d_probs = probs;
    if (!on_device) {
        CHK(h->order.reserve((size_t)nq * L * 4));
        CHK(h->out_d.reserve((size_t)nq * L * 4));
        d_order = h->order.as<int>();
        d_probs = h->out_d.as<float>();
    }
    begin_call(h);
    CHK(record(h, 0));
    CHK(mlp_enqueue(h, static_cast<const float*>(d_q), nq, L, d_order, nullptr, d_probs));
    CHK(record(h, 1));
    CHK(stamp_end(h, ST_MLP1));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(classes, d_order, (size_t)nq * L * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(probs, d_probs, (size_t)nq * L * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// the exact re-rank runs in its streamed form (select_kernel + rescore_kernel, lmi_rescore.h) for these shapes
static bool rescore_is_streamed(const lmi_index* h) {
    return h->rescore_streamed && rc_waves_for(h->dp, 4) > 0 && rc_wave_lds(h->dp, 4, true) <= RC_SMALL_LDS_CAP;
}
static int rescore_group_size(int nb) { return nb % 4 == 0 ? 4 : nb % 3 == 0 ? 3 : nb % 2 == 0 ? 2 : 1; }

// pass 1 (SAMPLE) / pass 2 of the fp16 prefilter: the low-dimensional form for d <= 128 (lmi_pass2_small.h), else lmi_pass2.h
template <bool SAMPLE>
static int launch_pass2(lmi_index* h, const PrefilterParams& F) {
    if (h->pf_small && F.KG16 <= PS_MAXKG) {
        const bool wide = h->ps_wide;   // (chosen per call with the routing's tile size, scan_enqueue)
        const int grid = h->num_cus * ps_blocks_per_cu(F.KG16, wide), lds = ps_lds_bytes(F.KG16, wide) - (SAMPLE ? ps_spill_bytes(F.KG16, wide) : 0);
#define LMI_PS_CASE(K) case K: \
            if (wide && ps_has_wide(K)) pass2_small_kernel<K, SAMPLE, ps_has_wide(K)><<<grid, 64 * ps_waves(K, true), lds, h->stream>>>(F); \
            else pass2_small_kernel<K, SAMPLE, false><<<grid, 64 * ps_waves(K, false), lds, h->stream>>>(F); \
            break;
        switch (F.KG16) {
            LMI_PS_CASE(1) LMI_PS_CASE(2) LMI_PS_CASE(3) LMI_PS_CASE(4) LMI_PS_CASE(5) LMI_PS_CASE(6) LMI_PS_CASE(7) LMI_PS_CASE(8)
            default: return fail("internal: KG16 = %d outside the low-dimensional form (%s:%d)", F.KG16, __FILE__, __LINE__);
        }
#undef LMI_PS_CASE
    } else {
        pass2_kernel<SAMPLE><<<h->num_cus * P2_BLOCKS_PER_CU, 64 * P2_WAVES, 0, h->stream>>>(F);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int scan_enqueue(lmi_index* h, const float* d_qs, int nq, const int* d_order, int nb, int kout, int raw,
                        float* d_dists, uint32_t* d_ids, uint32_t* d_keys) {
    const float* d_qn2 = nullptr;
    if (h->metric == LMI_METRIC_L2) {  // queries [nq][d_user] -> [q, 1, 0..] of the stored width, |q|^2 aside
        CHK(h->q_aug.reserve((size_t)nq * h->d * 4));
        CHK(h->qn2.reserve((size_t)nq * 4));
        augment_copy_kernel<<<cdiv((long long)nq * h->d, 256), 256, 0, h->stream>>>(d_qs, h->d_user, h->d, nq, h->q_aug.as<float>());
        HIPCHK(hipGetLastError());
        augment_norm_kernel<<<cdiv(nq, 256), 256, 0, h->stream>>>(d_qs, h->d_user, h->d, nq, h->q_aug.as<float>(), h->qn2.as<float>());
        HIPCHK(hipGetLastError());
        d_qs = h->q_aug.as<float>();
        d_qn2 = h->qn2.as<float>();
    }
    const int L = h->L;
    const int nslots = nq * nb;
    const long long ncb_bound = (long long)nslots / 32 + L + 4;
    // partial-list bound (exact mode only; the prefilter path writes rank lists, not chunk partials): a row of
    // bucket_order is caller data and may repeat a bucket, so a query owns at most nb x the largest chunk count
    const bool fast = h->prefilter && h->have16;
    bool use_tail = false, tail_merges = false;   // (set where the prefilter path picks its tail)
    int max_nch = 0;
    for (int b = 0; b < L; ++b) max_nch = std::max(max_nch, h->h_nch[b]);
    const long long part_lists = fast ? 1 : std::max<long long>(1, (long long)nb * max_nch * nq);

    CHK(h->m.reserve((size_t)L * 4 * 3));   // m [L] | m0 [L] | the non-primary slots' counter [L]
    CHK(h->cb_start.reserve((L + 1) * 4));
    CHK(h->item_base.reserve((L + 1) * 4));
    CHK(h->part_base.reserve((L + 1) * 8));
    CHK(h->stats.reserve(32));
    CHK(h->head.reserve(256));   // [0, 32): queue heads | [32, 36): the end-of-launch cells of the device stamps (pass 2, scan_kernel)
    const size_t grp_ints = (size_t)NGRP * L + 2 * (size_t)NGRP * (L + 1) + 3 * NGRP + (size_t)L;   // (+ the call's chunk length per bucket)
    CHK(h->grp.reserve(grp_ints * 4));
    CHK(h->slot_local.reserve((size_t)nslots * 4));
    CHK(h->slot_col.reserve((size_t)nslots * 4));
    CHK(h->colmap.reserve((size_t)ncb_bound * 32 * 4));
    CHK(h->col_thr.reserve((size_t)ncb_bound * 32 * 4));
    CHK(h->qfrag.reserve((size_t)ncb_bound * h->KGs * 1024));
    CHK(h->part_score.reserve((size_t)part_lists * KPB * 4));
    CHK(h->part_row.reserve((size_t)part_lists * KPB * 4));
    CHK(h->rank_d.reserve((size_t)nslots * KPB * 4));
    CHK(h->rank_id.reserve((size_t)nslots * KPB * 4));

    RouteArrays R;
    R.nb_rows = h->d_nb_rows.as<int>();
    R.nch = h->d_nch.as<int>();
    R.m = h->m.as<int>();
    R.m0 = R.m + L;
    R.cb_start = h->cb_start.as<int>();
    R.item_base = h->item_base.as<int>();
    R.part_base = h->part_base.as<long long>();
    R.stats = h->stats.as<long long>();
    R.grp_bucket = h->grp.as<int>();
    R.grp_base = R.grp_bucket + (size_t)NGRP * L;
    R.grp_n = R.grp_base + (size_t)NGRP * (L + 1);
    R.grp_total = R.grp_n + NGRP;
    R.grp_base1 = R.grp_total + NGRP;
    R.grp_total1 = R.grp_base1 + (size_t)NGRP * (L + 1);
    R.dbg = nullptr;
    const bool v2 = h->prefilter && h->have16;   // the prefilter's kernels: lmi_pass2.h (tiles of up to 12 col-blocks)
    // graded pass-2 items (lmi_kernels.h RouteArrays): long chunks for the buckets a queue serves first, short ones for the last
    R.chunk_rb = h->chunk_rows / 32;
    R.chunk_rb_b = (v2 && h->graded_chunks) ? R.grp_total1 + NGRP : nullptr;
    {
        const int base = h->chunk_rows;
        int rows[3] = {base, base / 2, base / 4};   // (longer than the static chunk: no gain at C2, and a 4 096-row chunk of 768-d rows no longer fits an L2 beside a second query tile: hard leg +3.5 %)
        // d <= 128 (lmi_pass2_small.h): an item's start and end are a fifth of its time there and the rows are short -- twice the static chunk for the
        // buckets served first (10M x 45: pass 2 0.270 -> 0.256-0.262 ms; four times: 0.38, too few items for 512 workgroups)
        if (h->pf_small && h->KG16 <= PS_MAXKG) rows[0] = 2 * base;
        for (int i = 0; i < 3; ++i) {
            if (h->chunk_lvl_rows[i] > 0) rows[i] = h->chunk_lvl_rows[i];
            rows[i] = std::max(P2_TILE_ROWS, rows[i] / P2_TILE_ROWS * P2_TILE_ROWS);
            R.chunk_lvl[i] = rows[i] / 32;
        }
        R.chunk_frac[0] = h->chunk_frac[0];
        R.chunk_frac[1] = h->chunk_frac[1];
    }
    // low-dimensional kernels: the wide form (one 8-wave block per CU, 12-col-block tiles) when the visited buckets receive more
    // queries than the narrow form's tile holds -- decided from the call's shape alone (no device round trip)
    h->ps_wide = false;
    if (v2 && h->pf_small && h->KG16 <= PS_MAXKG) {
        const double per_bucket = (double)nq * nb / std::max(1, std::min(h->n_nonempty, (int)std::min<long long>((long long)nq * nb, 1 << 30)));
        h->ps_wide = h->ps_force_wide >= 0 ? h->ps_force_wide != 0 : ps_use_wide(h->KG16, per_bucket);
    }
    R.tile_cb = !v2 ? 4 : (h->pf_small && h->KG16 <= PS_MAXKG) ? ps_tile_cb(h->KG16, h->ps_wide) : P2_MAXCB;
    R.sample_max = (v2 && h->pf_small && h->KG16 <= PF_SAMPLE_LOWD_KG) ? PF_SAMPLE_LOWD : PF_SAMPLE;
    R.sample_items = v2 ? 1 : 0;
    // one bound per QUERY is enough when the caller keeps the k <= 10 best over all ranks (query_bound_kernel, lmi_pass2.h): pass 1
    // then samples only each query's primary slot(s) -- a quarter of the columns at n_buckets = 4
    const bool qbound = v2 && h->pf_qbound && nb > 1 && kout <= KPB;
    R.primary_nb = (qbound && h->pf_primary) ? nb : 0;

    const size_t ncols = (size_t)ncb_bound * 32;
    FillRanges Z;
    Z.count = 0;
    bool fill_ok = true;
    auto fill = [&](void* ptr, long long words, unsigned value) { fill_ok = Z.add(ptr, words, value) && fill_ok; };
    // front_kernel (lmi_front.h): routing, query norms / packing / bounds and these fills in ONE launch -- for moderate fan-outs and
    // batches; its blocks compute m / m0 and set their own columns' pass-1 lists, so those fills are not queued
    const bool use_front = fast && h->use_front && L <= FR_MAX_L && nslots <= FR_MAX_SLOTS && h->d <= FR_MAX_D;
    if (!use_front) fill(h->m.p, 3ll * L, 0u);
    fill(h->head.p, 64, 0u);   // [0..8] pass-2 queue heads + the pass-1 head, [16..24) the heads of pass 2's redo launch, [32..36) stamp cells
    if (!use_front) fill(h->colmap.p, (long long)ncols, 0xFFFFFFFFu);
    fill(h->col_thr.p, (long long)ncols, 0xFF800000u /* -inf */);
    if (fast) {
        CHK(h->qnorm.reserve((size_t)nq * 4));
        CHK(h->qdelta.reserve((size_t)nq * 4));
        CHK(h->qscale.reserve((size_t)nq * 4));
        CHK(h->qfrag16.reserve((size_t)ncb_bound * h->KG16 * 1024 + 8192));   // (+ 8 KiB: the same look-ahead on the query fragments)
        CHK(h->eps2.reserve(ncols * 4));
        CHK(h->cand_cnt.reserve(ncols * 4));
        CHK(h->cand_row.reserve(ncols * PF_CAP * 4));
        CHK(h->cand_s.reserve(ncols * PF_CAP * 4));
        CHK(h->fallback.reserve((size_t)nslots * 4));
        CHK(h->nkeep.reserve((size_t)nslots * 4));
        const size_t bound_words = ncols * P2_NSL * 16;   // pass 1: [P2_NSL lists][16 slots][columns]
        CHK(h->pf_bound.reserve(bound_words * 4 + 4096));  // + room for the developer builds' phase stamps
        if (!use_front) fill(h->pf_bound.p, (long long)bound_words, 0xFF800000u /* -inf */);
        fill(h->cand_cnt.p, (long long)ncols, 0u);
        fill(h->stats.as<long long>() + 2, 4, 0u);
        CHK(h->redo.reserve((size_t)(1 + L) * 4 + ncols));
        fill(h->redo.p, (long long)(1 + L) + (long long)((ncols + 3) / 4), 0u);
        CHK(h->fb_list.reserve((size_t)(8 + nslots) * 4));
        fill(h->fb_list.p, 8, 0u);   // fallback count, fail flags of the two pass-2 launches, log head, sorted total
        if (h->x_cap == 0) {         // the overflow log (16 B an entry) and its sorted form (8 B): allocated with the first prefilter batch
            const size_t cap = (size_t)1 << LMI_PF_X_LOG2;
            CHK(h->x_log.reserve(cap * 16));
            CHK(h->x_ext.reserve(cap * 8));
            h->x_cap = (unsigned)cap;
        }
        CHK(h->x_off.reserve(ncols * 4));
        if (rescore_is_streamed(h)) {   // the streamed re-rank's flags and list counters (lmi_rescore.h): zeroed here, not by a launch of their own
            const int groups = nslots / rescore_group_size(nb), sub_cap = cdiv(groups, RC_SUB);
            CHK(h->rs_flag.reserve((size_t)groups * 4));
            CHK(h->rs_active.reserve((size_t)(RC_SUB + RC_SUB * sub_cap) * 4 + (size_t)(1 + groups) * 4));
            fill(h->rs_flag.p, groups, 0u);
            fill(h->rs_active.p, RC_SUB, 0u);
            fill(h->rs_active.as<int>() + RC_SUB + RC_SUB * sub_cap, 1, 0u);
        }
    }
    if (!fill_ok) return fail("internal: more than %d fill ranges queued (%s:%d)", FillRanges::MAXR, __FILE__, __LINE__);
    Z.ts = nullptr;
    if (!use_front) {
    Z.ts = tsp(h, ST_FRONT);
    fill_ranges_kernel<<<h->num_cus * 4, 256, 0, h->stream>>>(Z);
    HIPCHK(hipGetLastError());
    route_count_kernel<<<cdiv(nslots, 256), 256, 0, h->stream>>>(d_order, nslots, L, R, h->slot_local.as<int>());
    HIPCHK(hipGetLastError());
    route_scan_kernel<<<1, 256, 0, h->stream>>>(L, R);
    HIPCHK(hipGetLastError());
    // the work queues (one 1 024-thread block, ~20 us) are only read by the scan kernels: built on the side stream while
    // this one packs the queries
    CHK(side_fork(h));
    if (L <= ROUTE_MAX_BUCKETS) {
        route_group_kernel<false><<<1, 1024, route_group_lds(L), h->side>>>(L, R, nullptr);
    } else {   // huge fan-outs: the same sort in a global scratch buffer
        CHK(h->grp_scratch.reserve(route_group_lds(L) + (size_t)L * 4));
        route_group_kernel<true><<<1, 1024, 0, h->side>>>(L, R, h->grp_scratch.as<char>());
    }
    HIPCHK(hipGetLastError());
    route_fill_kernel<<<cdiv(nslots, 256), 256, 0, h->stream>>>(d_order, h->slot_local.as<int>(), nslots, nb,
                                                               R.cb_start, R.m0, h->colmap.as<int>(), h->slot_col.as<int>());
    HIPCHK(hipGetLastError());
    }
    ScanParams S;
    S.slab = h->slab.as<float4>();
    S.qfrag = h->qfrag.as<float4>();
    S.KG = h->KGs;
    S.L = L;
    S.chunk_rb = h->chunk_rows / 32;
    S.rb_start = h->d_rb_start.as<int>();
    S.nb_rows = R.nb_rows;
    S.nch = R.nch;
    S.m = R.m;
    S.cb_start = R.cb_start;
    S.grp_bucket = R.grp_bucket;
    S.grp_base = R.grp_base;
    S.grp_n = R.grp_n;
    S.grp_total = R.grp_total;
    S.part_base = R.part_base;
    S.head = h->head.as<unsigned>();
    S.col_thr = h->col_thr.as<float>();
    S.part_score = h->part_score.as<float>();
    S.part_row = h->part_row.as<unsigned>();
    S.ts_start = nullptr;
    S.ts_end_cell = nullptr;
    unsigned long long* const p2_end_cell = reinterpret_cast<unsigned long long*>(h->head.as<unsigned>() + 32);
    unsigned long long* const scan_end_cell = reinterpret_cast<unsigned long long*>(h->head.as<unsigned>() + 34);
    if (!fast) {
        long long total = ncb_bound * 32 * h->KGs;
        pack_gather_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(d_qs, h->d, h->colmap.as<int>(), nq, ncb_bound * 32,
                                                                   h->KGs, h->qfrag.as<float4>());
        HIPCHK(hipGetLastError());
        CHK(side_join(h));   // the work queues
        CHK(record(h, 2));
        S.ts_start = tsp(h, ST_SCAN0);
        if (S.ts_start) { S.ts_end_cell = scan_end_cell; (void)tsp(h, ST_SCAN1); }
        scan_kernel<<<h->num_cus * h->scan_blocks_per_cu, 256, SCAN_LDS, h->stream>>>(S);
        HIPCHK(hipGetLastError());
        CHK(record(h, 3));
    } else {
        // fp16 prefilter + exact re-rank (lmi_prefilter.h)
        if (use_front) {
            FrontParams A;
            A.bucket_order = d_order;
            A.nq = nq; A.nb = nb; A.L = L;
            // granules [FR_MAX_L + 1] | the tag word (device-resident: a kernel argument would be frozen by a graph replay of this call)
            if (!h->cb_alloc.p) {
                CHK(h->cb_alloc.reserve((size_t)(FR_MAX_L + 2) * 8));
                HIPCHK(hipMemsetAsync(h->cb_alloc.p, 0, (size_t)(FR_MAX_L + 2) * 8, h->stream));
                front_epoch_bump_kernel<<<1, 1, 0, h->stream>>>(reinterpret_cast<unsigned*>(h->cb_alloc.as<unsigned long long>() + FR_MAX_L + 1));   // 0 -> 1
                HIPCHK(hipGetLastError());
            }
            CHK(h->cb_bucket.reserve((size_t)ncb_bound * 4));
            unsigned* const epoch_dev = reinterpret_cast<unsigned*>(h->cb_alloc.as<unsigned long long>() + FR_MAX_L + 1);
            if (h->fr_bump_pending) {   // an earlier call left after route_kernel and before its bump: bump now
                front_epoch_bump_kernel<<<1, 1, 0, h->stream>>>(epoch_dev);
                HIPCHK(hipGetLastError());
            }
            h->fr_bump_pending = true;
            A.epoch_dev = epoch_dev;
            A.gran = h->cb_alloc.as<unsigned long long>();
            A.cb_bucket = h->cb_bucket.as<int>();
            A.colmap = h->colmap.as<int>();
            A.R = R;
            A.R.dbg = h->fr_dbg.as<unsigned long long>();
            A.Z = Z;
            A.q = d_qs;
            A.d = h->d; A.KG16 = h->KG16; A.f16x16 = frag16x16(h);
            A.qnorm = h->qnorm.as<float>(); A.qdelta = h->qdelta.as<float>(); A.qscale = h->qscale.as<float>();
            A.qfrag16 = h->qfrag16.as<uint4>();
            A.slot_col = h->slot_col.as<int>();
            A.eps2 = h->eps2.as<float>();
            A.bnorm = h->bnorm.as<unsigned>(); A.bdelta = h->bdelta.as<unsigned>();
            A.pf_bound = h->pf_bound.as<float>();
            A.ncols = (long long)ncols;
            A.bound_rows = P2_NSL * 16;
            A.ts = tsp(h, ST_FRONT);
            A.dbg = h->fr_dbg.as<unsigned long long>();
            const size_t rlds = fr_route_lds(L);
            switch (nb) {   // the rank count as a compile-time constant: a wave's bucket ids of several steps are loaded at once (lmi_front.h, FrChunk)
#define LMI_FR_CASE(NBV) case NBV: route_kernel<NBV><<<L, FR_THREADS, rlds, h->stream>>>(A); break;
                LMI_FR_CASE(1) LMI_FR_CASE(2) LMI_FR_CASE(3) LMI_FR_CASE(4) LMI_FR_CASE(5) LMI_FR_CASE(6) LMI_FR_CASE(8) LMI_FR_CASE(10) LMI_FR_CASE(16)
#undef LMI_FR_CASE
                default: route_kernel<0><<<L, FR_THREADS, rlds, h->stream>>>(A); break;
            }
            HIPCHK(hipGetLastError());
            A.ts = nullptr;
            const int pgrid = 1 + (int)ncb_bound;
            const size_t plds = fr_pack_lds(L, h->KG16);
            const int nchunk = (h->d + 7) / 8;
            const bool vec = h->d % 8 == 0;
#define LMI_FP_LAUNCH(GSV, CPV) { if (vec) pack_kernel<GSV, CPV, true><<<pgrid, FP_THREADS, plds, h->stream>>>(A); \
                                  else pack_kernel<GSV, CPV, false><<<pgrid, FP_THREADS, plds, h->stream>>>(A); }
            if (nchunk <= 8) LMI_FP_LAUNCH(8, 1)
            else if (nchunk <= 16) LMI_FP_LAUNCH(16, 1)
            else if (nchunk <= 32) LMI_FP_LAUNCH(32, 1)
            else if (nchunk <= 64) LMI_FP_LAUNCH(64, 1)
            else if (nchunk <= 128) LMI_FP_LAUNCH(64, 2)
            else LMI_FP_LAUNCH(64, 4)
#undef LMI_FP_LAUNCH
            HIPCHK(hipGetLastError());
        } else {
        query_norm_kernel<<<cdiv(nq, 4), 256, 0, h->stream>>>(d_qs, nq, h->d, h->qnorm.as<float>(), h->qdelta.as<float>(),
                                                              h->qscale.as<float>());
        HIPCHK(hipGetLastError());
        {
            long long total = (long long)ncols * h->KG16 * 2;
            pack_queries16_kernel<<<cdiv(total, 256), 256, 0, h->stream>>>(d_qs, h->d, h->colmap.as<int>(), (long long)ncols,
                                                                          h->KG16, h->qscale.as<float>(), h->qfrag16.as<uint4>(), frag16x16(h));
            HIPCHK(hipGetLastError());
        }
        slot_bound_kernel<<<cdiv(nslots, 256), 256, 0, h->stream>>>(d_order, h->slot_col.as<int>(), nslots, nb, h->KG16 * 16,
                                                                   h->qnorm.as<float>(), h->qdelta.as<float>(),
                                                                   h->bnorm.as<unsigned>(), h->bdelta.as<unsigned>(), h->eps2.as<float>());
        HIPCHK(hipGetLastError());
        CHK(side_join(h));   // the work queues
        }
        CHK(record(h, 2));
        PrefilterParams F;
        F.slab16 = h->slab16.as<uint4>();
        F.qfrag16 = h->qfrag16.as<uint4>();
        F.KG16 = h->KG16;
        F.L = L;
        F.chunk_rb = S.chunk_rb;
        F.chunk_rb_b = R.chunk_rb_b;
        F.tile_cb = R.tile_cb;
        F.sample_max = R.sample_max;
        F.rb_start = S.rb_start;
        F.nb_rows = R.nb_rows;
        F.nch = R.nch;
        F.m = R.m;
        F.m0 = R.m0;
        F.cb_start = R.cb_start;
        F.grp_bucket = R.grp_bucket;
        F.grp_base = R.grp_base;
        F.grp_n = R.grp_n;
        F.grp_total = R.grp_total;
        F.grp_base1 = R.grp_base1;
        F.grp_total1 = R.grp_total1;
        F.ncols = (long long)ncols;
        F.head = S.head;
        F.bound = h->pf_bound.as<float>();
        F.bound1 = S.col_thr;
        F.eps2 = h->eps2.as<float>();
        F.cand_cnt = h->cand_cnt.as<unsigned>();
        F.cand_row = h->cand_row.as<unsigned>();
        F.cand_s = h->cand_s.as<float>();
        F.redo_count = nullptr; F.redo_bucket = nullptr; F.redo_col = nullptr;
        unsigned* fbw = h->fb_list.as<unsigned>();   // [0] fallback count, [1] / [2] fail flags, [3] log head, [4] sorted total
        F.x.log = h->x_log.as<uint4>();
        F.x.cap = h->x_cap;
        F.x.head = fbw + 3;
        F.x.fail = fbw + 1;
        F.x.launch = 0;
        h->stamps_off = (ncols * P2_NSL * 16 * 4 + 255) / 256 * 256;
        F.stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->pf_bound.p) + h->stamps_off);
        F.ts_start = tsp(h, ST_P1);
        F.ts_end_cell = nullptr;
#if defined(LMI_P2_STAMPS)
        HIPCHK(hipMemsetAsync(F.stamps, 0, 2 * 8 * 12 * 8, h->stream));
#endif
        {
            CHK(launch_pass2<true>(h, F));   // pass 1: slot maxima of the sampled tiles
            bound_merge2_kernel<<<cdiv((long long)ncols, 64), 256, 0, h->stream>>>(F.bound, (long long)ncols, F.bound1,
                use_front ? reinterpret_cast<unsigned*>(h->cb_alloc.as<unsigned long long>() + FR_MAX_L + 1) : nullptr);
            HIPCHK(hipGetLastError());
            h->fr_bump_pending = false;
            if (qbound) {   // the caller keeps the k <= 10 best over all ranks: one bound per query
                query_bound_kernel<<<cdiv(nq, 256), 256, 0, h->stream>>>(h->slot_col.as<int>(), nq, nb, F.eps2, F.bound1);
                HIPCHK(hipGetLastError());
            }
        }
        if (h->debug_emit_all) {  // test hook: bound = -inf, every row of the bucket is a candidate
            FillRanges D;
            D.count = 1; D.p[0] = reinterpret_cast<unsigned*>(F.bound1); D.n[0] = (long long)ncols; D.v[0] = 0xFF800000u; D.ts = nullptr;
            fill_ranges_kernel<<<h->num_cus * 4, 256, 0, h->stream>>>(D);
            HIPCHK(hipGetLastError());
        }
        CHK(record(h, 5));

        // pass 2: candidates
        F.ts_start = tsp(h, ST_P2);
        F.ts_end_cell = F.ts_start ? p2_end_cell : nullptr;
        if (F.ts_start) { (void)tsp(h, ST_CLK_WALL); (void)tsp(h, ST_CLK_CYC); }
        CHK(launch_pass2<false>(h, F));
        F.ts_start = nullptr;
        F.ts_end_cell = nullptr;
        CHK(record(h, 6));
        // the fused tail (lmi_tail.h): a wave per query selects, re-ranks and merges -- n_buckets <= 4 (one wave holds the query's slots)
        // n_buckets <= 4 (a query's slots in ONE wave).  tail_kernel also runs group-wise (8 buckets: two waves of 4 + merge_ranks_kernel; LMI_TAIL=2),
        // but there the five launches are faster -- 4M x 768, 16 buckets: re-rank 0.39 against 0.25 ms; 4M x 45, 2 000 leaves, 8 buckets: 0.23 against
        // 0.18: most of the 160 000+ slots have nothing to re-rank, which select_kernel's compacted lists skip and a wave per group does not
        use_tail = rescore_is_streamed(h) && h->use_tail && (rescore_group_size(nb) == nb || h->use_tail == 2) &&
                   RC_WAVES * tail_wave_lds(h->dp, rescore_group_size(nb), true) <= RC_SMALL_LDS_CAP;
        tail_merges = use_tail && rescore_group_size(nb) == nb;
        // The overflow machinery (overflow_rebound_kernel + pass 2's redo launch: two launches that return at once on ordinary batches,
        // 11 us of a 0.2-0.5 ms search) stays OUT of the fused-tail sequence until a batch needs it: fallback_kernel then picks a flagged
        // column's entries out of the unsorted log (or, log full, scans the bucket: always correct) and raises a flag in pinned host memory;
        // the next 1 000 calls run with the machinery in.  The five-launch tail keeps it always.
        if (use_tail && !h->h_oflag) {
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_oflag), 64, hipHostMallocMapped));
            *h->h_oflag = 0u;
        }
        if (use_tail && *reinterpret_cast<volatile unsigned*>(h->h_oflag) != 0u) { h->overflow_armed = 1000; *h->h_oflag = 0u; }
        const bool overflow_sorted = h->pf_redo && !h->debug_emit_all && (!use_tail || h->overflow_armed > 0);
        if (use_tail && h->overflow_armed > 0) --h->overflow_armed;
        if (overflow_sorted) {
            // columns whose candidate buffer overflowed get the 10th best stored score as their bound and one more run of pass 2
            // over their buckets (a launch that returns at once when there is none: ~15 us per batch; lmi_prefilter.h)
            unsigned* rc = h->redo.as<unsigned>();
            int* rb = reinterpret_cast<int*>(rc + 1);
            unsigned char* rcol = reinterpret_cast<unsigned char*>(rc + 1 + L);
            overflow_rebound_kernel<<<cdiv(nslots, 256), 256, 0, h->stream>>>(h->slot_col.as<int>(), d_order, nslots, F.cand_cnt, F.cand_s,
                                                                             F.bound1, rc, rb, rcol, F.x.fail, F.x.cap, h->x_off.as<unsigned>(), fbw + 4);
            HIPCHK(hipGetLastError());
            PrefilterParams F2 = F;
            F2.x.cap = 0;          // (only runs when the first launch filled the log: what overflows again takes the exact fallback)
            F2.x.fail = fbw + 2;
            F2.x.launch = 1;
            F2.head = F.head + 16;
            F2.redo_count = rc; F2.redo_bucket = rb; F2.redo_col = rcol;
            CHK(launch_pass2<false>(h, F2));
        }
        RescoreParams Q;
        Q.bucket_order = d_order;
        Q.slot_col = h->slot_col.as<int>();
        Q.nslots = nslots;
        Q.nb = nb;
        Q.d = h->d;
        Q.raw = raw;
        Q.rb_start = S.rb_start;
        Q.nb_rows = R.nb_rows;
        Q.cand_cnt = F.cand_cnt;
        Q.cand_row = F.cand_row;
        Q.cand_s = F.cand_s;
        Q.eps2 = F.eps2;
        Q.rows = h->rowmajor.as<float>();
        Q.dp = h->dp;
        Q.q = d_qs;
        Q.qn2 = d_qn2;
        Q.ids_slab = h->ids_slab.as<unsigned>();
        Q.rank_d = h->rank_d.as<float>();
        Q.rank_id = h->rank_id.as<unsigned>();
        Q.fallback = h->fallback.as<int>();
        Q.nkeep = h->nkeep.as<int>();
        Q.fb_count = reinterpret_cast<int*>(fbw);
        Q.fb_list = reinterpret_cast<int*>(fbw + 8);
        const bool sorted_overflow = overflow_sorted;   // (overflow_rebound_kernel hands out the ranges)
        Q.x_fail = fbw + 1;
        Q.x_off = sorted_overflow ? h->x_off.as<unsigned>() : nullptr;
        Q.x_ext = h->x_ext.as<uint2>();
        Q.x_log = F.x.log;
        Q.x_head = F.x.head;
        Q.x_cap = F.x.cap;
        Q.redo_col = sorted_overflow ? reinterpret_cast<const unsigned char*>(h->redo.as<unsigned>() + 1 + L) : nullptr;
        Q.ts = h->ts_set;
        Q.p2_end = p2_end_cell;
        if (Q.ts) { (void)tsp(h, ST_TAIL); (void)tsp(h, ST_P2END); (void)tsp(h, ST_FB); }
        Q.merge_pending = nullptr; Q.m_kout = 0; Q.m_out_d = nullptr; Q.m_out_id = nullptr; Q.m_out_key = nullptr;
        Q.host_oflag = use_tail ? h->h_oflag : nullptr;
#ifndef LMI_ABL_NOEMIT  // timing-only ablation builds emit nothing: no re-rank, no fallback
        if (use_tail) {
            const int G = rescore_group_size(nb), groups = nslots / G;
            CHK(h->surv_row.reserve((size_t)nslots * RC_KEEP * 4));
            const int sub_cap = cdiv(groups, RC_SUB);
            SelectOut O;
            O.surv_row = h->surv_row.as<unsigned>();
            O.G = G;
            O.grp_flag = nullptr;
            O.active = h->rs_active.as<int>();
            O.sub_cap = sub_cap;
            O.big = O.active + RC_SUB + RC_SUB * sub_cap;
            TailParams T;
            T.ngroups = groups; T.merge = tail_merges ? 1 : 0; T.kout = kout;
            T.out_d = d_dists; T.out_id = d_ids; T.out_key = d_keys;
            T.pending = h->rs_flag.as<int>();   // [groups] (used as [nq] when the tail merges: then groups == nq)
            if (tail_merges) {
                Q.merge_pending = T.pending; Q.m_kout = kout; Q.m_out_d = d_dists; Q.m_out_id = d_ids; Q.m_out_key = d_keys;
                if (Q.ts) (void)tsp(h, ST_END);
            }
            const int lds_s = RC_WAVES * tail_wave_lds(h->dp, G, true);
            const int blocks = cdiv(groups, RC_WAVES);
#define LMI_TL_LAUNCH(GV) { tail_kernel<GV><<<blocks, 64 * RC_WAVES, lds_s, h->stream>>>(Q, O, T); }
            if (G == 4) LMI_TL_LAUNCH(4) else if (G == 3) LMI_TL_LAUNCH(3) else if (G == 2) LMI_TL_LAUNCH(2) else LMI_TL_LAUNCH(1)
#undef LMI_TL_LAUNCH
            HIPCHK(hipGetLastError());
        } else if (rescore_is_streamed(h)) {
            // selection at full occupancy, then the survivors' rows streamed through LDS in coalesced pieces (lmi_rescore.h)
            const int G = rescore_group_size(nb);  // slots of one query per wave
            const int groups = nslots / G;
            CHK(h->surv_row.reserve((size_t)nslots * RC_KEEP * 4));
            const int sub_cap = cdiv(groups, RC_SUB);
            SelectOut O;
            O.surv_row = h->surv_row.as<unsigned>();
            O.G = G;
            O.grp_flag = h->rs_flag.as<int>();
            O.active = h->rs_active.as<int>();
            O.sub_cap = sub_cap;
            O.big = O.active + RC_SUB + RC_SUB * sub_cap;
            select_kernel<<<cdiv(nslots, 4), 256, 0, h->stream>>>(Q, O);
            HIPCHK(hipGetLastError());
            // (wide rows: fewer waves per block in the big form, whose per-wave buffers hold the query and 256 survivors per slot; the small
            // form keeps four waves as long as a block stays under 64 KiB)
            const int wb = rc_waves_for(h->dp, G), ws = RC_WAVES * rc_wave_lds(h->dp, G, true) <= RC_SMALL_LDS_CAP ? RC_WAVES : 1;
            const int blocks = cdiv(groups, ws);
            const int lds = wb * rc_wave_lds(h->dp, G), lds_s = ws * rc_wave_lds(h->dp, G, true);
            // first every group in the small-LDS form (three blocks per CU), then the groups it passed on (more survivors than it holds)
#define LMI_RC_LAUNCH(GV) { rescore_kernel<GV, true><<<blocks, 64 * ws, lds_s, h->stream>>>(Q, O); \
                            rescore_kernel<GV, false><<<std::min(cdiv(groups, wb), h->num_cus), 64 * wb, lds, h->stream>>>(Q, O); }
            if (G == 4) LMI_RC_LAUNCH(4) else if (G == 3) LMI_RC_LAUNCH(3) else if (G == 2) LMI_RC_LAUNCH(2) else LMI_RC_LAUNCH(1)
#undef LMI_RC_LAUNCH
            HIPCHK(hipGetLastError());
        } else {
            select_rescore_kernel<<<cdiv(nslots, RS_WAVES), 64 * RS_WAVES, 0, h->stream>>>(Q);
            HIPCHK(hipGetLastError());
        }
#endif
        CHK(record(h, 7));
#ifndef LMI_ABL_NOEMIT
        fallback_kernel<<<std::min(cdiv(nslots, 4), h->num_cus * 4), 256, 0, h->stream>>>(Q);
        HIPCHK(hipGetLastError());
#endif
        CHK(record(h, 3));
    }
    MergeParams M;
    M.bucket_order = d_order;
    M.slot_col = h->slot_col.as<int>();
    M.nq = nq;
    M.nb = nb;
    M.L = L;
    M.kout = kout;
    M.raw = raw;
    M.skip_a = fast ? 1 : 0;
    M.rb_start = S.rb_start;
    M.nb_rows = R.nb_rows;
    M.nch = R.nch;
    M.cb_start = R.cb_start;
    M.part_base = R.part_base;
    M.part_score = S.part_score;
    M.part_row = S.part_row;
    M.ids_slab = h->ids_slab.as<unsigned>();
    M.qn2 = d_qn2;
    M.rank_d = h->rank_d.as<float>();
    M.rank_id = h->rank_id.as<unsigned>();
    M.ts = h->ts_set;
    M.scan_end = (!fast && h->ts_set) ? scan_end_cell : nullptr;
    if (M.ts && !tail_merges) { (void)tsp(h, ST_MERGE); (void)tsp(h, ST_END); }
    M.out_d = d_dists;
    M.out_id = d_ids;
    M.out_key = d_keys;
    if (tail_merges) { /* merged by tail_kernel / fallback_kernel */ }
    else if (M.skip_a && nb <= 16) merge_ranks_kernel<<<cdiv(nq, 64), 64, 0, h->stream>>>(M);  // rank lists exist: a thread per query
    else merge_kernel<<<nq, 64, 0, h->stream>>>(M);
    HIPCHK(hipGetLastError());
    CHK(record(h, 4));
    h->stats_pending = true;
    h->last_nslots = nslots;
    h->last_nb = nb;
    h->last_ncols = (long long)ncols;
    h->last_fast = fast;
    return 0;
}

// Device memory one lmi_search / lmi_scan_topk call of nq queries x nb buckets needs for its per-call workspaces (the
// sizes scan_enqueue reserves, summed; host-pointer calls add the staged inputs and outputs).  A caller with a memory budget
// sizes its query chunks from this instead of a constant (li/LearnedIndex.py).
extern "C" LMI_API int lmi_workspace_bytes(lmi_index* h, int nq, int nb, int64_t* bytes) {
    if (!h || !bytes) return fail("lmi_workspace_bytes: NULL argument");
    if (nq < 0 || nb < 1) return fail("lmi_workspace_bytes: bad nq/n_buckets");
    if (!h->built) return fail("lmi_workspace_bytes: the bucket index is not built");
    const long long L = h->L, nslots = (long long)nq * nb;
    const long long ncb = nslots / 32 + L + 4, ncols = ncb * 32;
    const bool fast = h->prefilter && h->have16;
    int max_nch = 0;
    for (int b = 0; b < h->L; ++b) max_nch = std::max(max_nch, h->h_nch[b]);
    long long t = 0;
    t += nslots * (4 + 4 + 2 * KPB * 4);                         // slot_local, slot_col, rank lists
    t += ncols * (4 + 4) + ncb * h->KGs * 1024;                  // colmap, col_thr, f32 query fragments
    t += (long long)nq * h->d * 4 * 2 + nslots * 4 + (long long)nq * std::max(nb, KPB) * 12;   // staged queries, bucket order, outputs
    if (fast) {
        t += (long long)nq * 12 + ncb * h->KG16 * 1024;           // query norms / scales, fp16 query fragments
        t += ncols * (4 + 4 + 2ll * PF_CAP * 4 + 1);             // eps2, candidate counts + buffers, redo flags
        t += ncols * P2_NSL * 16 * 4 + 4096;                     // pass-1 lists
        t += nslots * (4 + 4 + (long long)RC_KEEP * 4) + nslots; // fallback, nkeep, survivor rows, re-rank lists
        t += nslots * 4 + 32 + ncols * 4;                        // fallback list, overflow offsets
        t += ((long long)1 << LMI_PF_X_LOG2) * (16 + 8);         // the handle's overflow log + its sorted form (96 MiB, allocated with the first prefilter batch:
                                                                 // part of what a caller's memory budget must leave room for, whatever nq is)
    } else {
        t += std::max<long long>(1, (long long)nb * max_nch * nq) * KPB * 8;   // chunk partial lists of the exact scan
    }
    t += L * (3 * NGRP + 16) * 4;                                // per-bucket routing arrays, the work queues, the call's chunk lengths
    if (h->metric == LMI_METRIC_L2) t += (long long)nq * (h->d + 1) * 4;
    *bytes = t;
    return 0;
}

static int check_scan_args(lmi_index* h, int nq, int nb, int k, int* kout) {
    if (!h->built) return fail("lmi_scan_topk: the bucket index is not built (lmi_buckets_begin/add_rows/end)");
    if (nq < 0 || nb < 1) return fail("lmi_scan_topk: bad nq/n_buckets");
    if (nb > 1024) return fail("lmi_scan_topk: n_buckets %d exceeds 1024 (the rank merge keeps 16 four-bit cursors per lane)", nb);
    if (k < 1 || k > LMI_MAX_K) return fail("lmi_scan_topk: k %d outside [1,%d]", k, LMI_MAX_K);
    *kout = nb == 1 ? KPB : k;  // LearnedIndex.py:122-124: a single rank is returned unmerged
    if ((long long)nb * KPB < *kout) return fail("lmi_scan_topk: k %d exceeds n_buckets*10 candidates", k);
    if ((long long)nq * nb >= (1ll << 31)) return fail("lmi_scan_topk: nq*n_buckets too large");
    return 0;
}

extern "C" LMI_API int lmi_scan_topk(lmi_index* h, const float* queries_search, int nq, const int32_t* bucket_order,
                             int nb, int k, float* dists, uint32_t* ids, uint32_t* keys, int on_device) {
    if (!h) return fail("lmi_scan_topk: NULL handle");
    int kout = 0;
    CHK(check_scan_args(h, nq, nb, k, &kout));
    if (nq == 0) return 0;
    CHK(set_dev(h));
    const void* d_qs = nullptr;
    const void* d_order = nullptr;
    CHK(input_ptr(h, queries_search, (size_t)nq * h->d_user * 4, on_device, h->q_srch, &d_qs));
    CHK(input_ptr(h, bucket_order, (size_t)nq * nb * 4, on_device, h->order, &d_order));
    float* d_d = dists;
    uint32_t* d_i = ids;
    uint32_t* d_k = keys;
    if (!on_device) {
        CHK(h->out_d.reserve((size_t)nq * kout * 4));
        CHK(h->out_id.reserve((size_t)nq * kout * 4));
        d_d = h->out_d.as<float>();
        d_i = h->out_id.as<uint32_t>();
        if (keys) { CHK(h->out_key.reserve((size_t)nq * kout * 4)); d_k = h->out_key.as<uint32_t>(); }
    }
    begin_call(h);
    CHK(record(h, 1));
    CHK(scan_enqueue(h, static_cast<const float*>(d_qs), nq, static_cast<const int*>(d_order), nb, kout, 0, d_d, d_i, d_k));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(dists, d_d, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(ids, d_i, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        if (keys) HIPCHK(hipMemcpyAsync(keys, d_k, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" LMI_API int lmi_search(lmi_index* h, const float* queries_nav, const float* queries_search, int nq, int nb,
                          int k, float* dists, uint32_t* ids, uint32_t* keys, int32_t* bucket_order, int on_device) {
    if (!h) return fail("lmi_search: NULL handle");
    int kout = 0;
    CHK(check_scan_args(h, nq, nb, k, &kout));
    if (h->n_layers == 0) return fail("lmi_search: no MLP set (lmi_set_mlp)");
    if (h->dims[h->n_layers] != h->L) return fail("lmi_search: MLP has %d classes, index has %d buckets", h->dims[h->n_layers], h->L);
    if (nq == 0) return 0;
    CHK(set_dev(h));
    const void* d_qn = nullptr;
    const void* d_qs = nullptr;
    CHK(input_ptr(h, queries_nav, (size_t)nq * h->dims[0] * 4, on_device, h->q_nav, &d_qn));
    if (queries_search == queries_nav && h->dims[0] == h->d_user) d_qs = d_qn;
    else CHK(input_ptr(h, queries_search, (size_t)nq * h->d_user * 4, on_device, h->q_srch, &d_qs));
    int* d_order = bucket_order;
    float* d_d = dists;
    uint32_t* d_i = ids;
    uint32_t* d_k = keys;
    if (!on_device || !bucket_order) { CHK(h->order.reserve((size_t)nq * nb * 4)); d_order = h->order.as<int>(); }
    if (!on_device) {
        CHK(h->out_d.reserve((size_t)nq * kout * 4));
        CHK(h->out_id.reserve((size_t)nq * kout * 4));
        d_d = h->out_d.as<float>();
        d_i = h->out_id.as<uint32_t>();
        if (keys) { CHK(h->out_key.reserve((size_t)nq * kout * 4)); d_k = h->out_key.as<uint32_t>(); }
    }
    begin_call(h);
    CHK(record(h, 0));
    CHK(mlp_enqueue(h, static_cast<const float*>(d_qn), nq, nb, d_order, nullptr));
    CHK(record(h, 1));
    CHK(scan_enqueue(h, static_cast<const float*>(d_qs), nq, d_order, nb, kout, 0, d_d, d_i, d_k));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(dists, d_d, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(ids, d_i, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        if (keys) HIPCHK(hipMemcpyAsync(keys, d_k, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        if (bucket_order) HIPCHK(hipMemcpyAsync(bucket_order, d_order, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" LMI_API int lmi_merge_gathered(lmi_index* h, const float* gd, const uint32_t* gi, const uint32_t* gk, int world,
                                  int64_t world_stride, int nq, int kout, float* dists, uint32_t* ids,
                                  int on_device) {
    if (!h) return fail("lmi_merge_gathered: NULL handle");
    if (world < 1 || world > 64) return fail("lmi_merge_gathered: world %d outside [1,64]", world);
    if (nq <= 0 || kout < 1) return nq == 0 ? 0 : fail("lmi_merge_gathered: bad nq/kout");
    CHK(set_dev(h));
    const size_t nin = (size_t)world * nq * kout * 4, nout = (size_t)nq * kout * 4;
    if (world_stride == 0) world_stride = (int64_t)nq * kout;
    if (on_device) {
        merge_gathered_kernel<<<nq, 64, 0, h->stream>>>(gd, gi, gk, world, world_stride, nq, kout, dists, ids);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (world_stride != (int64_t)nq * kout) return fail("lmi_merge_gathered: host buffers must be dense (world_stride 0)");
    DevBuf in, out;
    CHK(in.reserve(3 * nin));
    CHK(out.reserve(2 * nout));
    char* ip = in.as<char>();
    HIPCHK(hipMemcpyAsync(ip, gd, nin, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(ip + nin, gi, nin, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(ip + 2 * nin, gk, nin, hipMemcpyHostToDevice, h->stream));
    merge_gathered_kernel<<<nq, 64, 0, h->stream>>>(reinterpret_cast<float*>(ip), reinterpret_cast<unsigned*>(ip + nin),
                                                   reinterpret_cast<unsigned*>(ip + 2 * nin), world, world_stride, nq, kout,
                                                   out.as<float>(), reinterpret_cast<unsigned*>(out.as<char>() + nout));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dists, out.p, nout, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ids, out.as<char>() + nout, nout, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    in.release();
    out.release();
    return 0;
}

// ---- RCCL (resolved from the process image -- PyTorch-ROCm has it loaded -- or from librccl.so) ----------------
namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = RTLD_DEFAULT;
        if (!dlsym(lib, "ncclAllGather")) {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
                if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!lib) return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    });
    return &r;
}
int rccl_fail(const char* what, ncclResult_t e) {
    Rccl* r = rccl();
    return fail("%s failed: %s", what, r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
}
}  // namespace

extern "C" LMI_API int lmi_comm_unique_id(void* id128) {
    if (!id128) return fail("lmi_comm_unique_id: NULL buffer");
    if (!rccl()->ok) return fail("lmi_comm_unique_id: RCCL (librccl.so) is not available in this process");
    ncclUniqueId id;
    ncclResult_t e = rccl()->GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" LMI_API int lmi_comm_init(lmi_index* h, int rank, int world, const void* id128, void** comm) {
    if (!h || !id128 || !comm) return fail("lmi_comm_init: NULL argument");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail("lmi_comm_init: rank %d of %d", rank, world);
    if (!rccl()->ok) return fail("lmi_comm_init: RCCL (librccl.so) is not available in this process");
    CHK(set_dev(h));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t e = rccl()->CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
    *comm = c;
    return 0;
}

extern "C" LMI_API int lmi_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (!rccl()->ok) return fail("lmi_comm_destroy: RCCL is not available");
    ncclResult_t e = rccl()->CommDestroy(static_cast<ncclComm_t>(comm));
    return e == ncclSuccess ? 0 : rccl_fail("ncclCommDestroy", e);
}

// The exchange step of the bucket-sharded mode through the C ABI alone: this rank's lmi_scan_topk /
// lmi_search outputs (device pointers, [nq][kout] each) -> ONE ncclAllGather of the packed [dists | ids | keys] block
// over `comm` on the handle's stream -> merge_gathered_kernel -> dists / ids [nq][kout] (device), identical on
// every rank and to the single-GPU result.
extern "C" LMI_API int lmi_allgather_merge(lmi_index* h, void* comm, int rank, int world, const float* local_dists,
                                   const uint32_t* local_ids, const uint32_t* local_keys, int nq, int kout, float* dists,
                                   uint32_t* ids) {
    if (!h || !comm) return fail("lmi_allgather_merge: NULL handle/communicator");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail("lmi_allgather_merge: rank %d of %d", rank, world);
    if (nq <= 0 || kout < 1) return nq == 0 ? 0 : fail("lmi_allgather_merge: bad nq/kout");
    if (!rccl()->ok) return fail("lmi_allgather_merge: RCCL (librccl.so) is not available in this process");
    CHK(set_dev(h));
    const size_t plane = (size_t)nq * kout;
    CHK(h->gather_send.reserve(3 * plane * 4));
    CHK(h->gather_recv.reserve((size_t)world * 3 * plane * 4));
    char* snd = h->gather_send.as<char>();
    HIPCHK(hipMemcpyAsync(snd, local_dists, plane * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(snd + plane * 4, local_ids, plane * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(snd + 2 * plane * 4, local_keys, plane * 4, hipMemcpyDeviceToDevice, h->stream));
    ncclResult_t e = rccl()->AllGather(snd, h->gather_recv.p, 3 * plane, ncclInt32, static_cast<ncclComm_t>(comm), h->stream);
    if (e != ncclSuccess) return rccl_fail("ncclAllGather", e);
    const char* rcv = h->gather_recv.as<char>();
    merge_gathered_kernel<<<nq, 64, 0, h->stream>>>(reinterpret_cast<const float*>(rcv), reinterpret_cast<const unsigned*>(rcv + plane * 4),
                                                   reinterpret_cast<const unsigned*>(rcv + 2 * plane * 4), world, (long long)(3 * plane),
                                                   nq, kout, dists, ids);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" LMI_API int lmi_knn_ip(int device, const float* xq, int64_t nq, const float* xb, int64_t nb, int d, int k,
                          float* D, int64_t* I) {
    if (k < 1 || k > KPB) return fail("lmi_knn_ip: k %d outside [1,%d]", k, KPB);
    if (nq < 0 || nb < 0 || d < 1) return fail("lmi_knn_ip: bad sizes");
    if (nq >= (1ll << 31)) return fail("lmi_knn_ip: nq too large");
    for (int64_t i = 0; i < nq * k; ++i) { D[i] = -FLT_MAX; I[i] = -1; }
    if (nq == 0 || nb == 0) return 0;
    lmi_index* h = nullptr;
    CHK(lmi_create(device, &h));
    int rc = 0;
    std::vector<int64_t> labels((size_t)nb, 0);
    std::vector<int32_t> order((size_t)nq, 0);
    std::vector<float> dd((size_t)nq * KPB);
    std::vector<uint32_t> ii((size_t)nq * KPB);
    do {
        if ((rc = lmi_buckets_begin(h, nb, d, 1, labels.data(), nullptr, nullptr))) break;
        if ((rc = lmi_buckets_add_rows(h, xb, 0, nb, 0))) break;
        if ((rc = lmi_buckets_end(h))) break;
        if ((rc = hipSetDevice(device) == hipSuccess ? 0 : fail("hipSetDevice"))) break;
        const void *d_qs, *d_order;
        if ((rc = input_ptr(h, xq, (size_t)nq * d * 4, 0, h->q_srch, &d_qs))) break;
        if ((rc = input_ptr(h, order.data(), (size_t)nq * 4, 0, h->order, &d_order))) break;
        if ((rc = h->out_d.reserve((size_t)nq * KPB * 4))) break;
        if ((rc = h->out_id.reserve((size_t)nq * KPB * 4))) break;
        if ((rc = scan_enqueue(h, static_cast<const float*>(d_qs), (int)nq, static_cast<const int*>(d_order), 1, KPB, 1,
                               h->out_d.as<float>(), h->out_id.as<uint32_t>(), nullptr))) break;
        if (hipMemcpy(dd.data(), h->out_d.p, dd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ii.data(), h->out_id.p, ii.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail("lmi_knn_ip: result copy failed");
            break;
        }
        for (int64_t q = 0; q < nq; ++q)
            for (int j = 0; j < k; ++j) {
                D[q * k + j] = dd[q * KPB + j];
                I[q * k + j] = ii[q * KPB + j] == NOROW ? -1 : (int64_t)ii[q * KPB + j];
            }
    } while (0);
    std::string keep = g_err;
    lmi_destroy(h);
    if (rc) g_err = keep;
    return rc;
}

static int read_event_set(const hipEvent_t* ev, const bool* ev_valid, float* ms) {
    for (int i = 0; i < LMI_T_COUNT; ++i) ms[i] = 0.0f;
    auto span = [&](int a, int b, float* out) -> int {
        if (ev_valid[a] && ev_valid[b]) HIPCHK(hipEventElapsedTime(out, ev[a], ev[b]));
        return 0;
    };
    CHK(span(0, 1, &ms[LMI_T_INFERENCE]));
    CHK(span(1, 2, &ms[LMI_T_ROUTE]));
    CHK(span(2, 3, &ms[LMI_T_SCAN]));
    CHK(span(3, 4, &ms[LMI_T_MERGE]));
    CHK(span(2, 5, &ms[LMI_T_PF_SAMPLE]));
    CHK(span(5, 6, &ms[LMI_T_PF_EMIT]));
    CHK(span(6, 7, &ms[LMI_T_RESCORE]));
    CHK(span(7, 3, &ms[LMI_T_FALLBACK]));
    int first = ev_valid[0] ? 0 : 1;
    int last = ev_valid[4] ? 4 : 1;
    CHK(span(first, last, &ms[LMI_T_TOTAL]));
    return 0;
}

// the same phases from one set of device stamps (timing level 2): `v` the set's ST_COUNT words, `mask` the stamps this call's
// kernels were given; ticks of the chip's constant clock -> ms
static void read_stamp_set(const lmi_index* h, const unsigned long long* v, unsigned mask, float* ms) {
    for (int i = 0; i < LMI_T_COUNT; ++i) ms[i] = 0.0f;
    auto have = [&](int a) { return (mask >> a) & 1u; };
    auto span = [&](int a, int b, float* out) {
        if (have(a) && have(b) && v[b] >= v[a]) *out = (float)((double)(v[b] - v[a]) / h->wall_khz);
    };
    span(ST_MLP0, have(ST_MLP1) ? ST_MLP1 : ST_FRONT, &ms[LMI_T_INFERENCE]);
    span(ST_FRONT, have(ST_P1) ? ST_P1 : ST_SCAN0, &ms[LMI_T_ROUTE]);
    if (have(ST_P1)) {
        span(ST_P1, ST_P2, &ms[LMI_T_PF_SAMPLE]);
        span(ST_P2, ST_P2END, &ms[LMI_T_PF_EMIT]);
        span(ST_P2END, ST_FB, &ms[LMI_T_RESCORE]);
        const int after = have(ST_MERGE) ? ST_MERGE : ST_END;   // (the fused tail merges in its own kernels: no merge launch)
        span(ST_FB, after, &ms[LMI_T_FALLBACK]);
        span(ST_P1, after, &ms[LMI_T_SCAN]);
    } else {
        span(ST_SCAN0, ST_SCAN1, &ms[LMI_T_SCAN]);
    }
    span(ST_MERGE, ST_END, &ms[LMI_T_MERGE]);
    const int first = have(ST_MLP0) ? ST_MLP0 : ST_FRONT, last = have(ST_END) ? ST_END : ST_MLP1;
    span(first, last, &ms[LMI_T_TOTAL]);
    // the clock the chip held under the dominant kernel: block 0's life in shader cycles (s_memtime) over the same in 100 MHz ticks
    if (have(ST_CLK_WALL) && have(ST_CLK_CYC) && v[ST_CLK_WALL] > 0) ms[LMI_T_CLOCK_MHZ] = (float)((double)v[ST_CLK_CYC] / (double)v[ST_CLK_WALL] * (h->wall_khz / 1000.0));
}

extern "C" LMI_API int lmi_timings(lmi_index* h, float* ms) {
    if (!h || !ms) return fail("lmi_timings: NULL argument");
    CHK(set_dev(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->timing_level == 2) {
        unsigned long long v[ST_COUNT] = {};
        if (h->ts_set) HIPCHK(hipMemcpy(v, h->ts_set, sizeof(v), hipMemcpyDeviceToHost));
        read_stamp_set(h, v, h->ts_set ? h->ts_mask[h->ev_cur] : 0u, ms);
        return 0;
    }
    return read_event_set(h->ev, h->ev_valid, ms);
}

extern "C" LMI_API int lmi_set_timing(lmi_index* h, int level) {
    if (!h) return fail("lmi_set_timing: NULL handle");
    if (level < 0 || level > 3) return fail("lmi_set_timing: level %d outside 0..3", level);
    h->timing_level = level;
    return 0;
}

extern "C" LMI_API int lmi_timings_reset(lmi_index* h) {
    if (!h) return fail("lmi_timings_reset: NULL handle");
    h->ev_calls = 0;
    return 0;
}

extern "C" LMI_API int lmi_timings_mean(lmi_index* h, float* ms, int* n_calls) {
    if (!h || !ms) return fail("lmi_timings_mean: NULL argument");
    CHK(set_dev(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int n = (int)std::min<long long>(h->ev_calls, lmi_index::EV_RING);
    double sum[LMI_T_COUNT] = {};
    std::vector<unsigned long long> ring;
    if (h->timing_level == 2 && h->ts_ring.p) {
        ring.resize((size_t)lmi_index::EV_RING * ST_COUNT);
        HIPCHK(hipMemcpy(ring.data(), h->ts_ring.p, ring.size() * 8, hipMemcpyDeviceToHost));
    }
    for (int j = 0; j < n; ++j) {
        const int r = ((h->ev_cur - j) % lmi_index::EV_RING + lmi_index::EV_RING) % lmi_index::EV_RING;
        float one[LMI_T_COUNT];
        if (h->timing_level == 2) {
            if (ring.empty()) break;
            read_stamp_set(h, ring.data() + (size_t)r * ST_COUNT, h->ts_mask[r], one);
        } else
        CHK(read_event_set(h->ev_ring[r], h->valid_ring[r], one));
        for (int i = 0; i < LMI_T_COUNT; ++i) sum[i] += one[i];
    }
    for (int i = 0; i < LMI_T_COUNT; ++i) ms[i] = n ? (float)(sum[i] / n) : 0.0f;
    if (n_calls) *n_calls = n;
    return 0;
}

extern "C" LMI_API int lmi_prefilter_stats(lmi_index* h, int* active, int64_t* survivors, int64_t* fallbacks) {
    if (!h) return fail("lmi_prefilter_stats: NULL handle");
    CHK(set_dev(h));
    unsigned long long acc[2] = {0, 0};
    if (h->last_fast && h->last_nslots > 0) {
        unsigned long long* d_acc = reinterpret_cast<unsigned long long*>(h->stats.as<long long>() + 2);
        HIPCHK(hipMemsetAsync(d_acc, 0, 16, h->stream));
        prefilter_stats_kernel<<<64, 256, 0, h->stream>>>(h->nkeep.as<int>(), h->fallback.as<int>(), h->last_nslots, d_acc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(acc, d_acc, 16, hipMemcpyDeviceToHost));
    }
    if (active) *active = (h->prefilter && h->have16) ? 1 : 0;
    if (survivors) *survivors = (int64_t)acc[0];
    if (fallbacks) *fallbacks = (int64_t)acc[1];
    return 0;
}

extern "C" LMI_API int lmi_copy_out(lmi_index* h, void* dst, const void* src, int64_t bytes) {
    if (!h) return fail("lmi_copy_out: NULL handle");
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail("lmi_copy_out: bad arguments");
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) return fail("lmi_copy_out: pointers must be 16-byte aligned");
    if (bytes == 0) return 0;
    CHK(set_dev(h));
    const int blocks = (int)std::min<long long>(h->num_cus * 2, cdiv(cdiv(bytes, 16), 256));
    copy_bytes_kernel<<<std::max(1, blocks), 256, 0, h->stream>>>(static_cast<const unsigned char*>(src), static_cast<unsigned char*>(dst), bytes);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" LMI_API int lmi_copy_out_many(lmi_index* h, int n, void* const* dst, const void* const* src, const int64_t* bytes) {
    if (!h) return fail("lmi_copy_out_many: NULL handle");
    if (n < 0 || n > 4 || (n > 0 && (!dst || !src || !bytes))) return fail("lmi_copy_out_many: bad arguments (1..4 ranges)");
    CopyRanges C;
    long long most = 0;
    int used = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] < 0 || (bytes[i] > 0 && (!dst[i] || !src[i]))) return fail("lmi_copy_out_many: bad range %d", i);
        if ((reinterpret_cast<uintptr_t>(dst[i]) | reinterpret_cast<uintptr_t>(src[i])) & 15) return fail("lmi_copy_out_many: pointers must be 16-byte aligned");
        if (bytes[i] == 0) continue;
        C.src[used] = static_cast<const unsigned char*>(src[i]);
        C.dst[used] = static_cast<unsigned char*>(dst[i]);
        C.bytes[used] = bytes[i];
        most = std::max<long long>(most, bytes[i]);
        ++used;
    }
    if (used == 0) return 0;
    CHK(set_dev(h));
    const int bx = (int)std::max<long long>(1, std::min<long long>(h->num_cus, cdiv(cdiv(most, 16), 256)));
    copy_ranges_kernel<<<dim3(bx, used), 256, 0, h->stream>>>(C);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" LMI_API int lmi_pipeline_submit(lmi_index* h, void* s_in_, void* s_nav_, void* s_run_, void* ev_in_, void* ev_nav_, void* ev_out_,
                                          const float* qn_host, const float* qs_host, float* qn_dev, float* qs_dev, int nq, int nb, int k,
                                          float* dists_out, uint32_t* ids_out, int32_t* bo_dev, int32_t* bo_host, int overlap_nav) {
    if (!h) return fail("lmi_pipeline_submit: NULL handle");
    if (!s_in_ || !s_run_ || !ev_in_ || !ev_out_ || !qn_host || !qn_dev || !dists_out || !ids_out || !bo_dev) return fail("lmi_pipeline_submit: NULL argument");
    if (overlap_nav && (!s_nav_ || !ev_nav_)) return fail("lmi_pipeline_submit: overlap_nav needs a navigation stream and event");
    if ((qs_host == nullptr) != (qs_dev == nullptr)) return fail("lmi_pipeline_submit: qs_host and qs_dev go together");
    if (nq < 1 || h->n_layers == 0 || !h->built) return fail("lmi_pipeline_submit: empty batch, no MLP or no bucket index");
    CHK(set_dev(h));
    hipStream_t s_in = static_cast<hipStream_t>(s_in_), s_nav = static_cast<hipStream_t>(s_nav_), s_run = static_cast<hipStream_t>(s_run_);
    hipEvent_t ev_in = static_cast<hipEvent_t>(ev_in_), ev_nav = static_cast<hipEvent_t>(ev_nav_), ev_out = static_cast<hipEvent_t>(ev_out_);
    HIPCHK(hipMemcpyAsync(qn_dev, qn_host, (size_t)nq * h->dims[0] * 4, hipMemcpyHostToDevice, s_in));
    if (qs_host) HIPCHK(hipMemcpyAsync(qs_dev, qs_host, (size_t)nq * h->d_user * 4, hipMemcpyHostToDevice, s_in));
    HIPCHK(hipEventRecord(ev_in, s_in));
    const float* q_scan = qs_dev ? qs_dev : qn_dev;
    if (overlap_nav) {
        HIPCHK(hipStreamWaitEvent(s_nav, ev_in, 0));
        h->stream = s_nav;
        int rc = lmi_mlp_topk(h, qn_dev, nq, nb, bo_dev, nullptr, 1);
        h->stream = s_run;
        CHK(rc);
        HIPCHK(hipEventRecord(ev_nav, s_nav));
        HIPCHK(hipStreamWaitEvent(s_run, ev_nav, 0));
        CHK(lmi_scan_topk(h, q_scan, nq, bo_dev, nb, k, dists_out, ids_out, nullptr, 1));
    } else {
        h->stream = s_run;
        HIPCHK(hipStreamWaitEvent(s_run, ev_in, 0));
        CHK(lmi_search(h, qn_dev, q_scan, nq, nb, k, dists_out, ids_out, nullptr, bo_dev, 1));
    }
    if (bo_host) CHK(lmi_copy_out(h, bo_host, bo_dev, (int64_t)nq * nb * 4));
    HIPCHK(hipEventRecord(ev_out, s_run));
    return 0;
}

// Multi-level navigation on the device: LearnedIndex._precompute_bucket_order for len(n_categories) > 1
// (LearnedIndex.py:216-252) -- the batched priority-queue walk.  slab_ids[nq][nb] <- slab bucket id of the
// j-th visited bucket (-1: listed bucket without objects or queue exhausted), entries[nq][nb] <- its flat child
// index (child_offset[parent model] + class; -1: none) from which the caller rebuilds the path.
// The multi-level walk of one batch, enqueued on h->stream: d_slab / d_ent [nq][nb] receive the visited buckets in visiting order.
// Trees of up to NAV_ENQUEUE_ALL models: EVERY possible step is enqueued up front and a step whose predecessor left no query waiting
// returns at once (nav_pop_kernel: prev_active) -- no host round trip inside the walk, the call is asynchronous like every other
// enqueue.  Larger trees: steps in batches of 4 with the count read back after each (one small synchronisation).
constexpr int NAV_ENQUEUE_ALL = 16;
static int nav_check(lmi_index* h, int nq, int nb, const char* who) {
    if (!h->tree_set) return fail("%s: no tree (lmi_nav_set_model / lmi_nav_set_tree)", who);
    CHK(set_dev(h));
    CHK(build_descs(h));
    if (!h->fm_ok || !h->fm_logits_lds)
        return fail("%s: a model of the tree does not fit the fused kernel (layer outputs <= %d, LDS plan %d bytes)", who, FM_MAXH, h->fm_lds);
    const int nm = 1 + (int)h->node_models.size();
    const int cap = h->h_child_offset[nm];
    if ((long long)nq * cap >= (1ll << 31) || (long long)nq * nb >= (1ll << 31)) return fail("%s: nq too large for this tree", who);
    if (cap == 0) return fail("%s: empty tree", who);
    return 0;
}
static int nav_enqueue(lmi_index* h, const float* d_q, int nq, int nb, int* d_slab, int* d_ent) {
    const int nm = 1 + (int)h->node_models.size();
    const int cap = h->h_child_offset[nm];
    CHK(h->pq_prob.reserve((size_t)nq * cap * 4));
    CHK(h->pq_ent.reserve((size_t)nq * cap * 4));
    CHK(h->pq_len.reserve((size_t)nq * 4));
    CHK(h->nav_len.reserve((size_t)nq * 4));
    CHK(h->nav_count.reserve((size_t)2 * (nm + 1) * 4));  // [2][nm + 1]: per-model counters + the step's active-query count
    CHK(h->nav_colq.reserve((size_t)nm * nq * 4));
    FillRanges Z;
    Z.count = 0;
    bool fill_ok = true;
    auto fill = [&](void* ptr, long long words, unsigned value) { fill_ok = Z.add(ptr, words, value) && fill_ok; };
    fill(h->pq_len.p, nq, 0u);
    fill(h->nav_len.p, nq, 0u);
    fill(d_slab, (long long)nq * nb, 0xFFFFFFFFu);
    fill(d_ent, (long long)nq * nb, 0xFFFFFFFFu);
    fill(h->nav_count.p, 2 * (nm + 1), 0u);
    if (!fill_ok) return fail("internal: more than %d fill ranges queued (%s:%d)", FillRanges::MAXR, __FILE__, __LINE__);
    Z.ts = tsp(h, ST_MLP0);
    fill_ranges_kernel<<<h->num_cus * 2, 256, 0, h->stream>>>(Z);
    HIPCHK(hipGetLastError());
    FusedParams P;
    fused_base(h, d_q, nq, P);
    P.pq_prob = h->pq_prob.as<float>();
    P.pq_ent = h->pq_ent.as<int>();
    P.pq_len = h->pq_len.as<int>();
    P.cap = cap;
    P.child_offset = h->d_child_offset.as<int>();
    P.reverse = 1;  // root children: least probable first (LearnedIndex.py:220-227)
    mlp_fused_kernel<FM_NAV><<<cdiv(nq, FM_COLS), 256, h->fm_lds, h->stream>>>(P);
    HIPCHK(hipGetLastError());
    P.reverse = 0;
    P.col_query = h->nav_colq.as<int>();
    NavParams N;
    N.nq = nq; N.nb = nb; N.cap = cap;
    N.pq_prob = P.pq_prob; N.pq_ent = P.pq_ent; N.pq_len = P.pq_len;
    N.child_model = h->d_child_model.as<int>();
    N.child_bucket = h->d_child_bucket.as<int>();
    N.out_len = h->nav_len.as<int>();
    N.out_slab = d_slab;
    N.out_ent = d_ent;
    N.col_query = h->nav_colq.as<int>();
    int* counts = h->nav_count.as<int>();
    // A step pops entries until the query hits an internal node; a query expands each node at most once, so there are
    // at most (models) steps.
    const int max_steps = nm + 1;
    const bool all = nm <= NAV_ENQUEUE_ALL;
    const bool pop_lds = cap <= NAV_LDS_CAP;
    int h_active = 1;
    for (int it = 0; it < max_steps && h_active > 0;) {
        int last_par = 0;
        for (int k4 = 0; (all || k4 < 4) && it < max_steps; ++k4, ++it) {
            const int par = it & 1;
            N.node_count = counts + par * (nm + 1);
            N.active = counts + par * (nm + 1) + nm;
            N.prev_active = (all && it > 0) ? counts + (1 - par) * (nm + 1) + nm : nullptr;
            if (pop_lds) nav_pop_lds_kernel<<<cdiv(nq, 64), 64, (size_t)cap * 64 * 8, h->stream>>>(N);
            else nav_pop_kernel<<<cdiv(nq, 256), 256, 0, h->stream>>>(N);
            HIPCHK(hipGetLastError());
            P.node_count = N.node_count;
            P.zero_counts = counts + (1 - par) * (nm + 1);
            P.n_zero = nm + 1;
            mlp_fused_kernel<FM_NAV><<<cdiv(nq, FM_COLS) + nm, 256, h->fm_lds, h->stream>>>(P);
            HIPCHK(hipGetLastError());
            last_par = par;
        }
        if (all) break;
        HIPCHK(hipMemcpyAsync(&h_active, counts + last_par * (nm + 1) + nm, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" LMI_API int lmi_nav_order(lmi_index* h, const float* queries_nav, int nq, int nb, int32_t* slab_ids, int32_t* entries,
                             int on_device) {
    if (!h) return fail("lmi_nav_order: NULL handle");
    if (nq < 0 || nb < 1) return fail("lmi_nav_order: bad nq/n_buckets");
    if (nq == 0) return 0;
    CHK(nav_check(h, nq, nb, "lmi_nav_order"));
    const void* d_q = nullptr;
    CHK(input_ptr(h, queries_nav, (size_t)nq * h->dims[0] * 4, on_device, h->q_nav, &d_q));
    CHK(h->nav_slab.reserve((size_t)nq * nb * 4));
    CHK(h->nav_ent.reserve((size_t)nq * nb * 4));
    int* d_slab = on_device ? slab_ids : h->nav_slab.as<int>();
    int* d_ent = on_device ? entries : h->nav_ent.as<int>();
    begin_call(h);
    CHK(record(h, 0));
    CHK(nav_enqueue(h, static_cast<const float*>(d_q), nq, nb, d_slab, d_ent));
    CHK(record(h, 1));
    CHK(stamp_end(h, ST_MLP1));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(slab_ids, d_slab, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(entries, d_ent, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// LearnedIndex.search for a multi-level index in ONE call (LearnedIndex.py:216-325 the walk, :328-373 the bucket scans): lmi_nav_order +
// lmi_scan_topk without the host in between.  Host buffers: the scan vectors' upload (30 MB at 10 000 x 768: 1.2 ms from pageable memory)
// goes over the library's side stream WHILE the walk runs (0.4-1 ms); the walk's bucket order never leaves the device unless asked for.
extern "C" LMI_API int lmi_search_tree(lmi_index* h, const float* queries_nav, const float* queries_search, int nq, int nb, int k,
                               float* dists, uint32_t* ids, uint32_t* keys, int32_t* slab_ids, int32_t* entries, int on_device) {
    if (!h) return fail("lmi_search_tree: NULL handle");
    int kout = 0;
    CHK(check_scan_args(h, nq, nb, k, &kout));
    if (nq == 0) return 0;
    CHK(nav_check(h, nq, nb, "lmi_search_tree"));
    const void* d_qn = nullptr;
    const void* d_qs = queries_search;
    CHK(input_ptr(h, queries_nav, (size_t)nq * h->dims[0] * 4, on_device, h->q_nav, &d_qn));
    const bool same = queries_search == queries_nav && h->dims[0] == h->d_user;
    if (same) d_qs = d_qn;
    CHK(h->nav_slab.reserve((size_t)nq * nb * 4));
    CHK(h->nav_ent.reserve((size_t)nq * nb * 4));
    int* d_slab = (on_device && slab_ids) ? slab_ids : h->nav_slab.as<int>();
    int* d_ent = (on_device && entries) ? entries : h->nav_ent.as<int>();
    float* d_d = dists;
    uint32_t* d_i = ids;
    uint32_t* d_k = keys;
    if (!on_device) {
        CHK(h->out_d.reserve((size_t)nq * kout * 4));
        CHK(h->out_id.reserve((size_t)nq * kout * 4));
        d_d = h->out_d.as<float>();
        d_i = h->out_id.as<uint32_t>();
        if (keys) { CHK(h->out_key.reserve((size_t)nq * kout * 4)); d_k = h->out_key.as<uint32_t>(); }
        if (!same) CHK(h->q_srch.reserve((size_t)nq * h->d_user * 4));
    }
    begin_call(h);
    CHK(record(h, 0));
    CHK(nav_enqueue(h, static_cast<const float*>(d_qn), nq, nb, d_slab, d_ent));
    CHK(record(h, 1));
    CHK(stamp_end(h, ST_MLP1));
    if (!on_device && !same) {
        // the scan vectors: uploaded beside the walk (the copy's host side returns when the bytes are staged; the stream waits for its event)
        CHK(side_ensure(h));   // (no fork: the buffer's last reader was the previous call's scan, and a host-buffer call ends synchronised)
        HIPCHK(hipMemcpyAsync(h->q_srch.p, queries_search, (size_t)nq * h->d_user * 4, hipMemcpyHostToDevice, h->side));
        HIPCHK(hipEventRecord(h->side_join, h->side));
        HIPCHK(hipStreamWaitEvent(h->stream, h->side_join, 0));
        d_qs = h->q_srch.p;
    }
    CHK(scan_enqueue(h, static_cast<const float*>(d_qs), nq, d_slab, nb, kout, 0, d_d, d_i, d_k));
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(dists, d_d, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(ids, d_i, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        if (keys) HIPCHK(hipMemcpyAsync(keys, d_k, (size_t)nq * kout * 4, hipMemcpyDeviceToHost, h->stream));
        if (slab_ids) HIPCHK(hipMemcpyAsync(slab_ids, d_slab, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        if (entries) HIPCHK(hipMemcpyAsync(entries, d_ent, (size_t)nq * nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// ---- test hooks (tests/test_gpu_bound.py): the fp16 scores the pass-2 kernel really produced --------------
extern "C" LMI_API int lmi_debug_emit_all(lmi_index* h, int on) {
    if (!h) return fail("lmi_debug_emit_all: NULL handle");
    h->debug_emit_all = on != 0;
    return 0;
}

extern "C" LMI_API int lmi_debug_read_candidates(lmi_index* h, int64_t slot, int cap, uint32_t* rows, float* shat,
                                         int* count, float* eps2, float* qscale, float* xscale) {
    if (!h) return fail("lmi_debug_read_candidates: NULL handle");
    if (!h->last_fast) return fail("lmi_debug_read_candidates: the last scan did not use the prefilter");
    if (slot < 0 || slot >= h->last_nslots) return fail("lmi_debug_read_candidates: slot outside the last scan's %d", h->last_nslots);
    CHK(set_dev(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    int col = -1;
    HIPCHK(hipMemcpy(&col, h->slot_col.as<int>() + slot, 4, hipMemcpyDeviceToHost));
    unsigned cnt = 0;
    float e2 = 0.0f, qs = 1.0f, xs[2] = {1.0f, 1.0f};
    if (col >= 0) {
        HIPCHK(hipMemcpy(&cnt, h->cand_cnt.as<unsigned>() + col, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&e2, h->eps2.as<float>() + col, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&qs, h->qscale.as<float>() + slot / h->last_nb, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(xs, h->xscale.p, 8, hipMemcpyDeviceToHost));
        const unsigned n = std::min<unsigned>(std::min<unsigned>(cnt, (unsigned)PF_CAP), (unsigned)std::max(cap, 0));
        if (n && rows) HIPCHK(hipMemcpy(rows, h->cand_row.as<unsigned>() + (size_t)col * PF_CAP, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (n && shat) HIPCHK(hipMemcpy(shat, h->cand_s.as<float>() + (size_t)col * PF_CAP, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    if (count) *count = col >= 0 ? (int)cnt : -1;
    if (eps2) *eps2 = e2;
    if (qscale) *qscale = qs;
    if (xscale) *xscale = xs[0];
    return 0;
}

// developer aid: the first `bytes` of a named internal device buffer ("pf_bound": LMI_PF_STAMPS builds keep phase timings there)
extern "C" LMI_API int lmi_debug_peek(lmi_index* h, const char* name, void* dst, int64_t bytes) {
    if (!h || !name || !dst) return fail("lmi_debug_peek: NULL argument");
    CHK(set_dev(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    DevBuf* b = nullptr;
    size_t off = 0;
    if (!strcmp(name, "pf_bound")) b = &h->pf_bound;
    if (!strcmp(name, "fr_dbg")) b = &h->fr_dbg;
    if (!strcmp(name, "pf_stamps")) { b = &h->pf_bound; off = h->stamps_off; }
    if (!strcmp(name, "cand_total")) {   // 8 bytes: candidates pass 2 emitted in the last scan, summed over the columns (capped counts not: the counters run on)
        if (bytes != 8 || !h->last_fast) return fail("lmi_debug_peek: cand_total is 8 bytes after a prefilter scan");
        unsigned long long* d_acc = reinterpret_cast<unsigned long long*>(h->stats.as<long long>() + 2);
        HIPCHK(hipMemsetAsync(d_acc, 0, 8, h->stream));
        sum_u32_kernel<<<64, 256, 0, h->stream>>>(h->cand_cnt.as<unsigned>(), h->last_ncols, d_acc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(dst, d_acc, 8, hipMemcpyDeviceToHost));
        return 0;
    }
    if (!strcmp(name, "pf_redo")) b = &h->redo;   // [0]: columns whose candidate buffer overflowed in the last scan (second run of pass 2)
    // 32 bytes: [0] slots fallback_kernel handled, [1] / [2] fail flags of the overflow log (pass 2 / its redo launch), [3] entries
    // appended to the log, [4] entries sorted by column, [5] slots that scanned their WHOLE bucket (the rest re-scored candidates)
    if (!strcmp(name, "pf_fallback")) b = &h->fb_list;
    if (!b) return fail("lmi_debug_peek: unknown buffer '%s'", name);
    if (bytes < 0 || off + (size_t)bytes > b->cap) return fail("lmi_debug_peek: %lld bytes asked of a %zu-byte buffer", (long long)bytes, b->cap);
    if (bytes) HIPCHK(hipMemcpy(dst, static_cast<char*>(b->p) + off, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" LMI_API int lmi_scan_stats(lmi_index* h, double* flops, int64_t* pairs, int64_t* items) {
    if (!h) return fail("lmi_scan_stats: NULL handle");
    CHK(set_dev(h));
    if (h->stats_pending) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->h_stats, h->stats.p, 32, hipMemcpyDeviceToHost));
        h->stats_pending = false;
    }
    if (pairs) *pairs = h->h_stats[0];
    if (items) *items = h->h_stats[1];
    if (flops) *flops = 2.0 * h->d * (double)h->h_stats[0];
    return 0;
}
