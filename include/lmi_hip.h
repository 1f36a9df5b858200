/*
 * lmi_hip.h -- C ABI of liblmi_hip.so: the MI355X (gfx950) implementation of the
 * LearnedMetricIndex query hot path.
 *
 * The reference (Coda-Research-Group/LearnedMetricIndex, pure Python) has no FFI of its own; its
 * native arithmetic is reached through two third-party Python calls.  Each entry point below
 * names the reference interface it replaces (paths relative to /root/reference/search/li/):
 *
 *   lmi_set_mlp / lmi_mlp_topk     NeuralNetwork.predict_proba            model.py:226-241
 *                                  (Sequential(Linear,ReLU,..)(x), softmax, topk)  model.py:45-49,97-99
 *                                  + _precompute_bucket_order 1-level     LearnedIndex.py:197-214
 *   lmi_mlp_proba                  NeuralNetwork.predict_proba (probabilities + full class order)
 *   lmi_buckets_*                  data_navigation.groupby(category_L*) + data_search.loc[...]
 *                                                                         LearnedIndex.py:101-104,350,357
 *   lmi_scan_topk                  the `for rank` x `for bucket` loop: filter_path_idxs, faiss.knn,
 *                                  1 - sim, local->global ids, stable merge
 *                                                                         LearnedIndex.py:107-146,328-373, utils.py:61-65
 *   lmi_search                     LearnedIndex.search (1-level index)    LearnedIndex.py:41-161
 *   lmi_knn_ip                     faiss.knn(xq, xb, k, METRIC_INNER_PRODUCT)   call site LearnedIndex.py:360-365
 *   lmi_merge_gathered             (no reference counterpart) merge of per-GPU top-k after the one
 *                                  RCCL all-gather of the bucket-sharded multi-GPU mode
 *
 * Conventions
 *   - Every function returns 0 on success and a negative code on failure; lmi_last_error()
 *     returns a thread-local, NUL-terminated description of the last failure.  Nothing throws.
 *   - A handle is bound to one device and one HIP stream (lmi_set_stream; default: the NULL
 *     stream).  All device work is enqueued on that stream.  A handle is not thread-safe.
 *   - `on_device` != 0: the float/int buffers of that call are device pointers valid on the
 *     handle's device and the call is asynchronous on the handle's stream.  `on_device` == 0: they
 *     are host pointers; the call copies in/out and returns after the results have landed.
 *   - All matrices are dense row-major.  Weights use torch.nn.Linear layout W[out][in].
 *   - Arithmetic contract (identical to oracle/lmi_oracle.c): every inner product is the k-ordered
 *     binary32 chain acc = fmaf(a[k], b[k], acc); Linear layers start the chain at the bias, the
 *     scan at 0.  Ties: lower class index / lower in-bucket row first; ranks merge by
 *     (distance, bucket rank, position) exactly like the reference's stable argsort.
 */
#ifndef LMI_HIP_H
#define LMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMI_ABI_VERSION 1
#define LMI_API __attribute__((visibility("default")))
#define LMI_METRIC_IP 0 /* dist = 1 - <q,x> : the reference's only metric (LearnedIndex.py:360-368) */
#define LMI_METRIC_L2 1 /* dist = |q - x|^2, computed as |q|^2 - 2 (<q,x> - |x|^2/2); see lmi_set_metric */
#define LMI_K_PER_BUCKET 10 /* LearnedIndex.py:334: k is never forwarded to the bucket scan */
#define LMI_MAX_K 64
#define LMI_MAX_LAYERS 8

typedef struct lmi_index lmi_index;

/* Timing slots filled by lmi_timings (milliseconds; measured on the handle's stream by device-side clock stamps the kernels
 * write themselves -- lmi_set_timing 2, the default -- or by hipEvents between them -- levels 1 and 3). */
enum {
    LMI_T_INFERENCE = 0, /* MLP forward + class ranking   -> measured_time["inference"]            */
    LMI_T_ROUTE = 1,     /* routing (CSR of queries per bucket) + query packing                     */
    LMI_T_SCAN = 2,      /* the bucket scan (exact: scan_kernel; prefilter: slots 5+6+7) -> ["seq_search"] */
    LMI_T_MERGE = 3,     /* chunk/rank merge kernel       -> measured_time["sort"]                  */
    LMI_T_TOTAL = 4,     /* first to last event           -> measured_time["search"]                */
    LMI_T_PF_SAMPLE = 5, /* prefilter pass 1 (bounds from a sample); 0 in exact mode                */
    LMI_T_PF_EMIT = 6,   /* prefilter pass 2 (fp16 scan + candidate emission) -- the dominant kernel */
    LMI_T_RESCORE = 7,   /* select + exact re-rank of the survivors                                  */
    LMI_T_FALLBACK = 8,  /* exact brute-force fallback for overflowed slots (normally empty)         */
    LMI_T_CLOCK_MHZ = 9, /* NOT a time: the shader clock (MHz) the chip held under pass 2 -- block 0's life in s_memtime cycles over the
                            same in 100 MHz s_memrealtime ticks (timing level 2 only; 0 otherwise)             */
    LMI_T_COUNT = 12
};

LMI_API int lmi_abi_version(void);
LMI_API const char *lmi_last_error(void);
/* Provenance (no reference counterpart): "src=<sha256/16 of the library's sources this binary was built from> p2_waves=..
 * p2_bring=.. pf_cap=.." -- csrc/build.sh bakes the hash in (learnedmetricindex_amd/_srchash.py: every .h / .hip under csrc
 * and every .h under include); bench.py and profiles/summarize.py tie PMC summaries to the library that was really
 * LOADED with it (not to whatever sources lie in the working tree). */
LMI_API const char *lmi_build_info(void);

/* Lifetime. */
LMI_API int lmi_create(int device, lmi_index **out);
LMI_API int lmi_destroy(lmi_index *h);
LMI_API int lmi_set_stream(lmi_index *h, void *hip_stream);

/* MLP weights (host pointers).  dims[0] = input dim, dims[n_layers] = number of classes L;
 * W[i] is [dims[i+1]][dims[i]], b[i] is [dims[i+1]].  ReLU between layers, none after the last. */
LMI_API int lmi_set_mlp(lmi_index *h, int n_layers, const int *dims, const float *const *W,
                const float *const *b);

/* Which kernels run the navigation MLP: 2 = always the one-launch kernel of lmi_mlp_fused.h, 0 = always one kernel
 * per layer + ranking kernels, 1 (default) = the fused kernel when the batch has enough 32-query blocks to fill the
 * chip (and for predict_proba), the per-layer kernels for small batches.  All produce bit-identical logits, orders
 * and probabilities. */
LMI_API int lmi_set_fused_mlp(lmi_index *h, int mode);

/* Multi-level index (len(n_categories) > 1; LearnedIndex.py:216-325, PriorityQueue.py:18-94): the models of
 * the internal nodes and the tree.  Model 0 is the root (lmi_set_mlp); lmi_nav_set_model sets model_id >= 1
 * (arguments as lmi_set_mlp).  lmi_nav_set_tree: child e = child_offset[m] + c is class c of model m:
 * child_model[e] >= 1 -> that internal node's model, -1 -> a leaf; child_bucket[e] >= 0 -> slab bucket id
 * (the bucket ids of lmi_buckets_begin), -1 -> a bucket path that holds no object (LearnedIndex.bucket_paths
 * lists it: it is recorded and its slot stays unvisited), -2 -> not a bucket (the popped path is dropped, as
 * the reference's _visit_buckets does).
 * lmi_nav_order: the batched priority-queue walk on the device -- every query pops its most probable entry
 * (priority = the child's LOCAL softmax probability; ties: the entry pushed later), internal nodes
 * are expanded by their model for all queries that popped them (one grouped launch per step), until nb
 * buckets are recorded.  slab_ids[nq][nb], entries[nq][nb] (flat child index of each visited bucket; -1 where
 * the queue ran out). */
LMI_API int lmi_nav_set_model(lmi_index *h, int model_id, int n_layers, const int *dims, const float *const *W,
                      const float *const *b);
LMI_API int lmi_nav_set_tree(lmi_index *h, int n_models, const int32_t *child_offset, const int32_t *child_model,
                     const int32_t *child_bucket);
LMI_API int lmi_nav_order(lmi_index *h, const float *queries_nav, int nq, int nb, int32_t *slab_ids,
                  int32_t *entries, int on_device);

/* LearnedIndex.search for a multi-level index in ONE call (reference: search/li/LearnedIndex.py:41-83 the entry point, :216-325 the walk,
 * :328-373 the bucket scans): lmi_nav_order followed by lmi_scan_topk on the walk's buckets, without the host in between.  Arguments as
 * lmi_search (queries_nav [nq][model inputs], queries_search [nq][d]; dists / ids / keys [nq][kout]); slab_ids / entries: nullable
 * [nq][nb] outputs of the walk (as lmi_nav_order).  With host buffers the scan vectors are uploaded on a library-owned stream while the
 * walk runs.  Trees of up to 16 models run without any host round trip inside the call. */
LMI_API int lmi_search_tree(lmi_index *h, const float *queries_nav, const float *queries_search, int nq, int nb, int k,
                    float *dists, uint32_t *ids, uint32_t *keys, int32_t *slab_ids, int32_t *entries, int on_device);

/* Metric of the bucket scan (call before lmi_buckets_begin; default LMI_METRIC_IP).  The reference scans with
 * faiss.METRIC_INNER_PRODUCT only (LearnedIndex.py:364); LMI_METRIC_L2 is what faiss.knn(..., METRIC_L2) would be in
 * its place: squared Euclidean distances, ascending.  Canonical arithmetic (oracle/lmi_oracle.c:lmi_oracle_knn_l2):
 * with the k-ordered fmaf chains s = <q,x>, xn = <x,x>, qn = <q,q>:  key = s + (-xn/2)  (one rounding),
 * dist = fmaf(-2, key, qn);  neighbours by key descending, ties -> lower row; short buckets are padded with
 * FLT_MAX.  Internally every stored vector carries the column -xn/2 (lmi_bucket_read does not return it). */
LMI_API int lmi_set_metric(lmi_index *h, int metric);

/* Bucket-contiguous index in HBM.
 * begin: labels[N] = data_prediction[:,0] (bucket of every object, 0 <= label < L), ids[N] = the
 *        DataFrame index labels (NULL -> 1..N, search.py:190-191), owned[L] = which buckets this
 *        handle keeps (NULL -> all; used by the bucket-sharded multi-GPU mode).  Host pointers.
 *        L < 2^20 (any fan-out the reference's n_categories can name in practice, e.g. [100, 100]).
 * add_rows: rows [nrows][d] are the original objects row0 .. row0+nrows-1 (any order of calls, each
 *        object exactly once); they are scattered to their bucket-contiguous position on device.
 * end:   finishes the build; the index is immutable afterwards. */
LMI_API int lmi_buckets_begin(lmi_index *h, int64_t N, int d, int L, const int64_t *labels,
                      const uint32_t *ids, const uint8_t *owned);
LMI_API int lmi_buckets_add_rows(lmi_index *h, const float *rows, int64_t row0, int64_t nrows, int on_device);
/* Owned-only ingest (bucket-sharded ranks): rows [nrows][d] are the objects index[0..nrows) (original row
 * numbers, each OWNED object exactly once, any order); objects of buckets this handle does not own are never
 * passed in, so a rank of an 8-way shard reads 1/8 of the dataset.  Not to be mixed with lmi_buckets_add_rows
 * within one build.  `on_device` covers both pointers. */
LMI_API int lmi_buckets_add_owned_rows(lmi_index *h, const float *rows, const int64_t *index, int64_t nrows,
                               int on_device);
LMI_API int lmi_buckets_end(lmi_index *h);
/* sizes[L] <- number of objects per bucket (0 for buckets not owned). */
LMI_API int lmi_bucket_sizes(lmi_index *h, int64_t *sizes);
/* Reads one bucket back to the host in bucket order (what `data_search.loc[g.index].to_numpy()`
 * and `g.index.to_numpy()` return, LearnedIndex.py:351,357): rows[n_b][d], ids[n_b]; either may
 * be NULL. */
LMI_API int lmi_bucket_read(lmi_index *h, int bucket, float *rows, uint32_t *ids);

/* Navigation: bucket_order[nq][nb] <- the nb most probable classes per query, most probable first.
 * logits (nullable) [nq][L] <- raw outputs of the last Linear layer. */
LMI_API int lmi_mlp_topk(lmi_index *h, const float *queries_nav, int nq, int nb, int32_t *bucket_order,
                 float *logits, int on_device);

/* NeuralNetwork.predict_proba (model.py:226-241): probs[nq][L] = softmax of the outputs sorted
 * descending, classes[nq][L] = the matching class indices (int32; the reference returns int64). */
LMI_API int lmi_mlp_proba(lmi_index *h, const float *queries_nav, int nq, float *probs, int32_t *classes,
                  int on_device);

/* Scan: for every query, top-LMI_K_PER_BUCKET by inner product inside each of its nb buckets,
 * dist = 1 - ip (binary32), merged over the ranks to k results (k <= LMI_MAX_K).
 * dists[nq][kout], ids[nq][kout] with kout = (nb == 1 ? LMI_K_PER_BUCKET : k).
 * Unvisited slots: dist = +inf, id = 0.  keys (nullable) [nq][kout] <- rank*16 + position, the
 * tie-break key needed by lmi_merge_gathered. */
LMI_API int lmi_scan_topk(lmi_index *h, const float *queries_search, int nq, const int32_t *bucket_order,
                  int nb, int k, float *dists, uint32_t *ids, uint32_t *keys, int on_device);

/* lmi_mlp_topk followed by lmi_scan_topk, nothing leaves the device in between. */
LMI_API int lmi_search(lmi_index *h, const float *queries_nav, const float *queries_search, int nq, int nb,
               int k, float *dists, uint32_t *ids, uint32_t *keys, int32_t *bucket_order,
               int on_device);

/* Multi-GPU: gathered_{dists,ids,keys}[w] is rank w's lmi_scan_topk output [nq][kout], found
 * world_stride elements after rank w-1's (0 -> dense, nq*kout; a packed all-gather of
 * [dists|ids|keys] per rank uses 3*nq*kout); writes the merged dists/ids [nq][kout]. */
LMI_API int lmi_merge_gathered(lmi_index *h, const float *gathered_dists, const uint32_t *gathered_ids,
                       const uint32_t *gathered_keys, int world, int64_t world_stride, int nq,
                       int kout, float *dists, uint32_t *ids, int on_device);

/* Copies `bytes` from device memory to any device-accessible destination -- in particular PINNED host memory
 * (hipHostMalloc / torch pin_memory) -- by a kernel on the handle's stream: the result download of a pipelined
 * caller (learnedmetricindex_amd/pipeline.py) without hipMemcpyAsync, whose D2H form was seen to block the
 * submitting host thread for milliseconds behind queued kernels (ROCm 7.2).  Both pointers 16-byte aligned.
 * Replaces the `.cpu().numpy()` of model.py:240-241 / the numpy results of LearnedIndex.py:340-341. */
LMI_API int lmi_copy_out(lmi_index *h, void *dst, const void *src, int64_t bytes);
/* Up to 4 such copies as ONE launch (dists, ids and the bucket order of a batch: three launches were 25 us of a 0.6-ms search). */
LMI_API int lmi_copy_out_many(lmi_index *h, int n, void *const *dst, const void *const *src, const int64_t *bytes);
/* One batch of a host-in -> host-out pipeline as ONE call (learnedmetricindex_amd/pipeline.py: a dozen Python-level stream / event / ctypes
 * operations per batch are 0.2-0.3 ms of host time -- more than the GPU's 0.17 ms for a 1 000-query search).  Streams and events are the caller's
 * (hipStream_t / hipEvent_t as void*): upload of the pinned host queries on s_in -> ev_in; overlap_nav != 0: lmi_mlp_topk on s_nav behind ev_in
 * -> ev_nav, lmi_scan_topk on s_run behind ev_nav; else lmi_search on s_run behind ev_in; (dists, ids) are stored where dists_out / ids_out point
 * (device memory, or pinned host memory: then no download is needed); the bucket order goes to bo_dev and, bo_host != NULL, by one copy
 * kernel to pinned bo_host; ev_out is recorded on s_run behind everything.  qs_host == NULL: navigation and scan vectors are the same array.
 * The handle's stream is s_run on return.  Replaces the body of LearnedIndex.search for one batch (LearnedIndex.py:85-159) like lmi_search. */
LMI_API int lmi_pipeline_submit(lmi_index *h, void *s_in, void *s_nav, void *s_run, void *ev_in, void *ev_nav, void *ev_out,
                                const float *qn_host, const float *qs_host, float *qn_dev, float *qs_dev, int nq, int nb, int k,
                                float *dists_out, uint32_t *ids_out, int32_t *bo_dev, int32_t *bo_host, int overlap_nav);

/* The same exchange through RCCL inside the library (no reference counterpart; SURVEY 8b `lmi_allgather_merge(h,
 * ncclComm_t, ...)`): a C/C++ caller runs the bucket-sharded mode without torch.distributed.
 *   lmi_comm_unique_id   rank 0: 128 bytes (ncclUniqueId) to hand to every rank by any side channel
 *   lmi_comm_init        every rank: *comm <- ncclComm_t over `world` ranks on the handle's device (collective call)
 *   lmi_allgather_merge  every rank: its lmi_scan_topk / lmi_search outputs (DEVICE pointers [nq][kout], keys
 *                        included) -> ONE ncclAllGather of the packed [dists|ids|keys] block on the handle's stream
 *                        -> merge kernel -> dists/ids [nq][kout] (device), identical on every rank.  `comm` may be
 *                        any ncclComm_t of the process (e.g. PyTorch's).
 * RCCL is resolved at run time (the process image, else librccl.so); the calls fail cleanly when it is absent. */
LMI_API int lmi_comm_unique_id(void *id128);
LMI_API int lmi_comm_init(lmi_index *h, int rank, int world, const void *id128, void **comm);
LMI_API int lmi_comm_destroy(void *comm);
LMI_API int lmi_allgather_merge(lmi_index *h, void *comm, int rank, int world, const float *local_dists,
                        const uint32_t *local_ids, const uint32_t *local_keys, int nq, int kout, float *dists,
                        uint32_t *ids);

/* faiss.knn(xq, xb, k, metric=METRIC_INNER_PRODUCT) on host pointers: D[nq][k] similarities in
 * descending order, I[nq][k] row numbers; nb < k pads with D = -FLT_MAX, I = -1.  k <= 10. */
LMI_API int lmi_knn_ip(int device, const float *xq, int64_t nq, const float *xb, int64_t nb, int d, int k,
               float *D, int64_t *I);

/* Timings of the last lmi_mlp_topk / lmi_scan_topk / lmi_search call (synchronises the stream). */
LMI_API int lmi_timings(lmi_index *h, float *ms /* [LMI_T_COUNT] */);
/* Mean of the timing slots over the calls made since lmi_timings_reset (the newest 128 at most), read
 * with ONE stream synchronisation, so a timed loop needs no per-call sync; *n_calls = calls averaged. */
LMI_API int lmi_timings_reset(lmi_index *h);
/* How much is timed, and how.  2 (default): every phase, from stamps of the chip's constant 100 MHz clock that the first / last
 * workgroups of the search's kernels write into a per-handle ring -- no event, no bubble, nothing to wait for.  3: every phase
 * from hipEvents recorded between the kernels (each one is a ~5 us bubble on the stream: 8 of them are 6 % of a 10M x 45 search);
 * 1: hipEvents for LMI_T_TOTAL (and LMI_T_INFERENCE) only; 0: nothing. */
LMI_API int lmi_set_timing(lmi_index *h, int level);
LMI_API int lmi_timings_mean(lmi_index *h, float *ms /* [LMI_T_COUNT] */, int *n_calls /* nullable */);
/* Work done by the last scan: flops = 2 * d * sum over (query, rank) of the bucket size;
 * items = work items executed by the persistent scan kernel. */
LMI_API int lmi_scan_stats(lmi_index *h, double *flops, int64_t *pairs, int64_t *items);
/* Scan mode.  on (default): fp16-MFMA prefilter with a proven error bound + exact binary32
 * re-ranking of the survivors (lmi_prefilter.h); off: every similarity by f32 MFMA.  Both modes
 * return bit-identical results; call before lmi_buckets_begin (the index is stored differently:
 * row-major f32 + fp16 fragments vs f32 fragments).  lmi_prefilter_stats: whether the last scan used the prefilter, how many candidates were
 * re-scored exactly and how many (query, rank) slots fell back to the exact brute-force kernel.  Any other value of
 * `on` is an error. */
LMI_API int lmi_set_prefilter(lmi_index *h, int on);
/* A second handle on the SAME index (no reference counterpart: the reference is single-threaded Python).  The clone
 * borrows the parent's MLP weights, tree and bucket slabs and has per-call workspaces, a stream and timing events of its
 * own, so that two searches can be in flight on one index, one per handle and stream (learnedmetricindex_amd/pipeline.py
 * alternates handles: a batch's kernels start in the tails of the previous batch's).  Destroy the clone before the parent;
 * do not rebuild the parent's index while a clone lives. */
LMI_API int lmi_clone_view(lmi_index *h, lmi_index **out);
/* Developer aid: copies the first `bytes` of a named internal device buffer to host memory: "pf_bound" (pass 1's slot maxima),
 * "pf_stamps" (phase cycles of -DLMI_P2_STAMPS builds), "pf_redo" ([0]: columns whose candidate buffer overflowed in the last scan);
 * "cand_total" (8 bytes): the candidates pass 2 emitted in the last scan, summed over all columns. */
LMI_API int lmi_debug_peek(lmi_index *h, const char *name, void *dst, int64_t bytes);
LMI_API int lmi_prefilter_stats(lmi_index *h, int *active, int64_t *survivors, int64_t *fallbacks);
/* Tuning: rows per scan chunk (multiple of the 256-row block tile).  Not called: lmi_buckets_begin picks
 * 256..2048 by the size of the index (this rank's rows / 4096), and more for buckets beyond 1024 chunks. */
LMI_API int lmi_set_chunk_rows(lmi_index *h, int rows);
/* Device memory (bytes) the per-call workspaces of one lmi_search / lmi_scan_topk of nq queries x n_buckets need on a built
 * index -- the candidate buffers of the prefilter dominate (~10 KiB per (query, bucket) slot).  No reference counterpart (the
 * reference holds no device memory); li/LearnedIndex.py sizes its query chunks from it. */
LMI_API int lmi_workspace_bytes(lmi_index *h, int nq, int n_buckets, int64_t *bytes);

/* Test hooks for the prefilter's error bound (tests/test_gpu_bound.py; no reference counterpart).
 * lmi_debug_emit_all(1): the next scans drop the sampled bound, so pass 2 emits EVERY row of a visited
 * bucket (buckets of <= 1024 rows fit the candidate buffer) -- results are unchanged (overflowing slots take
 * the exact fallback).  lmi_debug_read_candidates: for (query, rank) slot = q*nb + r of the last scan, the
 * in-bucket rows and the fp16-MFMA scores shat pass 2 computed for them (scaled units: shat ~ xscale *
 * qscale * <q, x>), the emitted count (-1: unvisited), 2*eps' of the slot and the two power-of-two scales. */
LMI_API int lmi_debug_emit_all(lmi_index *h, int on);
LMI_API int lmi_debug_read_candidates(lmi_index *h, int64_t slot, int cap, uint32_t *rows, float *shat,
                              int *count, float *eps2, float *qscale, float *xscale);

#ifdef __cplusplus
}
#endif
#endif /* LMI_HIP_H */
