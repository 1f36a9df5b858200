"""ctypes binding of liblmi_hip.so (the C ABI in include/lmi_hip.h).

There is no CPU fallback: if the HIP library is missing or fails, this module raises.
torch is imported first so that the HIP runtime (libamdhip64.so.7) already loaded by PyTorch-ROCm
is the one the library binds to; torch device pointers and streams are then valid inside it.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMI_LIB") or os.path.join(_HERE, "liblmi_hip.so")  # LMI_LIB: A/B builds (tools/)
K_PER_BUCKET = 10
T_INFERENCE, T_ROUTE, T_SCAN, T_MERGE, T_TOTAL, T_PF_SAMPLE, T_PF_EMIT, T_RESCORE, T_FALLBACK, T_CLOCK_MHZ, T_COUNT = (
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12)   # (T_CLOCK_MHZ is not a time: the shader clock held under pass 2, from in-kernel counters)

_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p

#: every symbol include/lmi_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lmi_abi_version": (ctypes.c_int, []),
    "lmi_last_error": (ctypes.c_char_p, []),
    "lmi_build_info": (ctypes.c_char_p, []),
    "lmi_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "lmi_destroy": (ctypes.c_int, [_vp]),
    "lmi_set_stream": (ctypes.c_int, [_vp, _vp]),
    "lmi_set_mlp": (ctypes.c_int, [_vp, ctypes.c_int, _i32p, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "lmi_set_fused_mlp": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_set_metric": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_nav_set_model": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _i32p, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "lmi_nav_set_tree": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "lmi_nav_order": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    "lmi_search_tree": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "lmi_buckets_begin": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "lmi_buckets_add_rows": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "lmi_buckets_add_owned_rows": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int]),
    "lmi_buckets_end": (ctypes.c_int, [_vp]),
    "lmi_bucket_sizes": (ctypes.c_int, [_vp, _vp]),
    "lmi_mlp_topk": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    "lmi_mlp_proba": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int]),
    "lmi_scan_topk": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                     ctypes.c_int]),
    "lmi_search": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp,
                                  ctypes.c_int]),
    "lmi_merge_gathered": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, _vp, _vp, ctypes.c_int]),
    "lmi_comm_unique_id": (ctypes.c_int, [_vp]),
    "lmi_comm_init": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
    "lmi_comm_destroy": (ctypes.c_int, [_vp]),
    "lmi_allgather_merge": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                           _vp, _vp]),
    "lmi_bucket_read": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    "lmi_copy_out": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64]),
    "lmi_copy_out_many": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "lmi_pipeline_submit": (ctypes.c_int, [_vp] * 7 + [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "lmi_knn_ip": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int,
                                  ctypes.c_int, _vp, _vp]),
    "lmi_timings": (ctypes.c_int, [_vp, _vp]),
    "lmi_timings_reset": (ctypes.c_int, [_vp]),
    "lmi_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_timings_mean": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    "lmi_scan_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), _i64p, _i64p]),
    "lmi_set_chunk_rows": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_workspace_bytes": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _i64p]),
    "lmi_set_prefilter": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_debug_peek": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_int64]),
    "lmi_clone_view": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "lmi_prefilter_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), _i64p, _i64p]),
    "lmi_debug_emit_all": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lmi_debug_read_candidates": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp,
                                                 ctypes.POINTER(ctypes.c_int), _f32p, _f32p, _f32p]),
}


class LmiError(RuntimeError):
    """Raised when a C-ABI call returns non-zero (message from lmi_last_error)."""


def lib() -> ctypes.CDLL:
    """Loads liblmi_hip.so (after torch, so one HIP runtime serves both) and declares prototypes."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LmiError(
                f"{LIB_PATH} is missing: build it with learnedmetricindex_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU fallback")
        import torch  # noqa: F401  (loads PyTorch-ROCm's libamdhip64 first)

        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.lmi_abi_version() != 1:
            raise LmiError("liblmi_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise LmiError(lib().lmi_last_error().decode("utf-8", "replace"))


def _ptr(a) -> int:
    """Raw address of a numpy array or torch tensor (None -> NULL)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch.Tensor


def _np(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class Index:
    """One device-resident index: thin, typed wrapper over an `lmi_index*`."""

    METRICS = {"ip": 0, "l2": 1}

    def __init__(self, device: int = 0, chunk_rows: Optional[int] = None, prefilter: Optional[bool] = None,
                 metric: str = "ip"):
        self._h = _vp()
        _check(lib().lmi_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        self.n_classes = None
        self.d_nav = None
        self.d = None
        self.L = None
        if chunk_rows is not None:
            _check(lib().lmi_set_chunk_rows(self._h, int(chunk_rows)))
        if prefilter is None and os.environ.get("LMI_PREFILTER") is not None:
            prefilter = os.environ["LMI_PREFILTER"] not in ("0", "off", "false")
        if prefilter is not None:
            self.set_prefilter(prefilter)
        if metric != "ip":  # "l2": squared Euclidean distances instead of 1 - inner product
            _check(lib().lmi_set_metric(self._h, self.METRICS[metric]))
        self.metric = metric

    def close(self) -> None:
        for c in getattr(self, "_views", []):   # clones borrow this handle's memory: they go first
            c.close()
        self._views = []
        if getattr(self, "_h", None) is not None and self._h:
            lib().lmi_destroy(self._h)
            self._h = None

    def clone_view(self) -> "Index":
        """A second handle on the same index (`lmi_clone_view`): shares the weights and the bucket slabs, has its own
        workspaces, stream and timings.  Closed with (before) this one."""
        v = Index.__new__(Index)
        v._h = _vp()
        _check(lib().lmi_clone_view(self._h, ctypes.byref(v._h)))
        for a in ("device", "n_classes", "d_nav", "d", "L", "N", "metric"):
            setattr(v, a, getattr(self, a, None))
        v._views = []
        v._parent = self
        self.__dict__.setdefault("_views", []).append(v)
        return v

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_prefilter(self, on) -> None:
        """fp16 prefilter + exact re-rank (default) or f32 MFMA for every similarity; same results."""
        _check(lib().lmi_set_prefilter(self._h, int(on)))

    def debug_peek(self, name: str, nbytes: int) -> np.ndarray:
        """Developer aid: the first `nbytes` of a named internal device buffer as uint8."""
        out = np.empty(nbytes, np.uint8)
        _check(lib().lmi_debug_peek(self._h, name.encode(), out.ctypes.data_as(_vp), nbytes))
        return out

    def prefilter_stats(self):
        """(active, survivors re-scored exactly, slots that fell back to exact brute force)."""
        a = ctypes.c_int(0)
        sv = ctypes.c_int64(0)
        fb = ctypes.c_int64(0)
        _check(lib().lmi_prefilter_stats(self._h, ctypes.byref(a), ctypes.byref(sv), ctypes.byref(fb)))
        return bool(a.value), sv.value, fb.value

    def debug_emit_all(self, on: bool) -> None:
        """Test hook: pass 2 emits every row of a visited bucket (see include/lmi_hip.h)."""
        _check(lib().lmi_debug_emit_all(self._h, 1 if on else 0))

    def debug_read_candidates(self, slot: int, cap: int = 1024):
        """Test hook: (rows u32[n], shat f32[n], emitted count, 2*eps', qscale, xscale) of slot q*nb+r."""
        rows = np.empty(cap, dtype=np.uint32)
        shat = np.empty(cap, dtype=np.float32)
        cnt = ctypes.c_int(0)
        e2, qs, xs = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
        _check(lib().lmi_debug_read_candidates(self._h, int(slot), int(cap), _ptr(rows), _ptr(shat), ctypes.byref(cnt),
                                               ctypes.byref(e2), ctypes.byref(qs), ctypes.byref(xs)))
        n = max(0, min(cnt.value, cap))
        return rows[:n], shat[:n], cnt.value, e2.value, qs.value, xs.value

    def set_stream(self, stream_ptr: int) -> None:
        _check(lib().lmi_set_stream(self._h, _vp(stream_ptr)))

    # ---- MLP -------------------------------------------------------------------------------
    def set_mlp(self, layers: Sequence) -> None:
        """layers = [(W [out,in], b [out]), ...] in torch.nn.Linear layout."""
        Ws = [_np(W, np.float32) for W, _ in layers]
        bs = [_np(b, np.float32) for _, b in layers]
        dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
        for i, (W, b) in enumerate(zip(Ws, bs)):
            assert W.shape == (dims[i + 1], dims[i]) and b.shape == (dims[i + 1],)
        n = len(Ws)
        dims_c = (ctypes.c_int32 * (n + 1))(*dims)
        Wp = (_vp * n)(*[W.ctypes.data for W in Ws])
        bp = (_vp * n)(*[b.ctypes.data for b in bs])
        _check(lib().lmi_set_mlp(self._h, n, dims_c, Wp, bp))
        self.d_nav, self.n_classes = dims[0], dims[-1]

    def set_fused_mlp(self, mode: int) -> None:
        """2: always the one-launch MLP kernel, 0: always the per-layer kernels, 1 (default): by batch size; identical outputs."""
        _check(lib().lmi_set_fused_mlp(self._h, int(mode)))

    # ---- multi-level navigation ---------------------------------------------------------------
    @staticmethod
    def _pack_layers(layers):
        Ws = [_np(W, np.float32) for W, _ in layers]
        bs = [_np(b, np.float32) for _, b in layers]
        dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
        for i, (W, b) in enumerate(zip(Ws, bs)):
            assert W.shape == (dims[i + 1], dims[i]) and b.shape == (dims[i + 1],)
        n = len(Ws)
        return Ws, bs, n, (ctypes.c_int32 * (n + 1))(*dims), (_vp * n)(*[W.ctypes.data for W in Ws]), (_vp * n)(*[b.ctypes.data for b in bs])

    def nav_set_model(self, model_id: int, layers: Sequence) -> None:
        """Model of an internal node (model_id >= 1; the root is set_mlp)."""
        Ws, bs, n, dims_c, Wp, bp = self._pack_layers(layers)
        _check(lib().lmi_nav_set_model(self._h, int(model_id), n, dims_c, Wp, bp))

    def nav_set_tree(self, child_offset, child_model, child_bucket) -> None:
        co, cm, cbk = _np(child_offset, np.int32), _np(child_model, np.int32), _np(child_bucket, np.int32)
        assert cm.shape == cbk.shape == (int(co[-1]),)
        _check(lib().lmi_nav_set_tree(self._h, co.shape[0] - 1, _ptr(co), _ptr(cm), _ptr(cbk)))

    def nav_order(self, queries_nav, nb: int):
        """(slab bucket ids i32[nq,nb], flat child indices i32[nq,nb]) of the multi-level walk."""
        q = _np(queries_nav, np.float32)
        slab = np.empty((q.shape[0], nb), dtype=np.int32)
        ent = np.empty((q.shape[0], nb), dtype=np.int32)
        _check(lib().lmi_nav_order(self._h, _ptr(q), q.shape[0], int(nb), _ptr(slab), _ptr(ent), 0))
        return slab, ent

    def search_tree(self, queries_nav, queries_search, nb: int, k: int = 10, want_keys: bool = False, want_order: bool = False):
        """The multi-level walk + the scan of its buckets in one call (lmi_search_tree): (dists f32[nq,kout], ids u32[nq,kout]
        [, keys] [, slab bucket ids i32[nq,nb], flat child indices i32[nq,nb]])."""
        qn = _np(queries_nav, np.float32)
        qs = qn if queries_search is queries_nav else _np(queries_search, np.float32)
        nq = qn.shape[0]
        ko = self.kout(nb, k)
        d = np.empty((nq, ko), dtype=np.float32)
        i = np.empty((nq, ko), dtype=np.uint32)
        keys = np.empty((nq, ko), dtype=np.uint32) if want_keys else None
        slab = np.empty((nq, nb), dtype=np.int32) if want_order else None
        ent = np.empty((nq, nb), dtype=np.int32) if want_order else None
        _check(lib().lmi_search_tree(self._h, _ptr(qn), _ptr(qs), nq, int(nb), int(k), _ptr(d), _ptr(i), _ptr(keys), _ptr(slab), _ptr(ent), 0))
        out = (d, i) + ((keys,) if want_keys else ()) + ((slab, ent) if want_order else ())
        return out

    def search_tree_device(self, qn_t, qs_t, nb: int, k: int, d_t, i_t, keys_t=None, slab_t=None, ent_t=None) -> None:
        _check(lib().lmi_search_tree(self._h, _ptr(qn_t), _ptr(qs_t), int(qn_t.shape[0]), int(nb), int(k), _ptr(d_t), _ptr(i_t),
                                     _ptr(keys_t), _ptr(slab_t), _ptr(ent_t), 1))

    # ---- buckets ---------------------------------------------------------------------------
    def buckets_begin(self, labels, d: int, L: int, ids=None, owned=None) -> None:
        labels = _np(labels, np.int64).reshape(-1)
        ids_a = None if ids is None else _np(ids, np.uint32).reshape(-1)
        owned_a = None if owned is None else _np(owned, np.uint8).reshape(-1)
        assert ids_a is None or ids_a.shape == labels.shape
        assert owned_a is None or owned_a.shape == (L,)
        _check(lib().lmi_buckets_begin(self._h, labels.shape[0], int(d), int(L), _ptr(labels), _ptr(ids_a),
                                       _ptr(owned_a)))
        self.N, self.d, self.L = labels.shape[0], int(d), int(L)

    def add_rows(self, rows, row0: int) -> None:
        """rows: numpy [n,d] float32 (host) or a CUDA torch tensor (device)."""
        if isinstance(rows, np.ndarray):
            rows = _np(rows, np.float32)
            on_device = 0
        else:
            assert rows.is_cuda and rows.is_contiguous() and rows.dtype.is_floating_point and rows.element_size() == 4
            on_device = 1
        assert rows.shape[1] == self.d
        _check(lib().lmi_buckets_add_rows(self._h, _ptr(rows), int(row0), int(rows.shape[0]), on_device))

    def add_owned_rows(self, rows, index) -> None:
        """Owned-only ingest: rows[i] is object index[i] (int64 original row numbers); both numpy or both CUDA tensors."""
        if isinstance(rows, np.ndarray):
            rows, index, on_device = _np(rows, np.float32), _np(index, np.int64).reshape(-1), 0
        else:
            assert rows.is_cuda and rows.is_contiguous() and rows.element_size() == 4 and rows.dtype.is_floating_point
            assert index.is_cuda and index.is_contiguous() and index.element_size() == 8
            on_device = 1
        assert rows.shape[1] == self.d and index.shape[0] == rows.shape[0]
        _check(lib().lmi_buckets_add_owned_rows(self._h, _ptr(rows), _ptr(index), int(rows.shape[0]), on_device))

    def buckets_end(self) -> None:
        _check(lib().lmi_buckets_end(self._h))

    def set_buckets(self, data, labels, L: int, ids=None, owned=None, piece: int = 1 << 18) -> None:
        data = data if not isinstance(data, np.ndarray) else _np(data, np.float32)
        self.buckets_begin(labels, data.shape[1], L, ids, owned)
        for r0 in range(0, data.shape[0], piece):
            self.add_rows(data[r0: r0 + piece], r0)
        self.buckets_end()

    def bucket_sizes(self) -> np.ndarray:
        out = np.zeros(self.L, dtype=np.int64)
        _check(lib().lmi_bucket_sizes(self._h, _ptr(out)))
        return out

    # ---- query path, host arrays ------------------------------------------------------------
    def mlp_topk(self, queries_nav, nb: int, want_logits: bool = False):
        q = _np(queries_nav, np.float32)
        order = np.empty((q.shape[0], nb), dtype=np.int32)
        logits = np.empty((q.shape[0], self.n_classes), dtype=np.float32) if want_logits else None
        _check(lib().lmi_mlp_topk(self._h, _ptr(q), q.shape[0], int(nb), _ptr(order), _ptr(logits), 0))
        return (order, logits) if want_logits else order

    def mlp_proba(self, queries_nav):
        """(probs f32[n,L] descending, classes i32[n,L]) -- NeuralNetwork.predict_proba."""
        q = _np(queries_nav, np.float32)
        probs = np.empty((q.shape[0], self.n_classes), dtype=np.float32)
        classes = np.empty((q.shape[0], self.n_classes), dtype=np.int32)
        _check(lib().lmi_mlp_proba(self._h, _ptr(q), q.shape[0], _ptr(probs), _ptr(classes), 0))
        return probs, classes

    @staticmethod
    def kout(nb: int, k: int) -> int:
        return K_PER_BUCKET if nb == 1 else k

    def scan_topk(self, queries_search, bucket_order, k: int = 10, want_keys: bool = False):
        q = _np(queries_search, np.float32)
        bo = _np(bucket_order, np.int32).reshape(q.shape[0], -1)
        nb = bo.shape[1]
        ko = self.kout(nb, k)
        d = np.empty((q.shape[0], ko), dtype=np.float32)
        i = np.empty((q.shape[0], ko), dtype=np.uint32)
        keys = np.empty((q.shape[0], ko), dtype=np.uint32) if want_keys else None
        _check(lib().lmi_scan_topk(self._h, _ptr(q), q.shape[0], _ptr(bo), nb, int(k), _ptr(d), _ptr(i), _ptr(keys), 0))
        return (d, i, keys) if want_keys else (d, i)

    def search(self, queries_nav, queries_search, nb: int, k: int = 10, want_keys: bool = False):
        qn = _np(queries_nav, np.float32)
        qs = qn if queries_search is queries_nav else _np(queries_search, np.float32)
        nq = qn.shape[0]
        ko = self.kout(nb, k)
        d = np.empty((nq, ko), dtype=np.float32)
        i = np.empty((nq, ko), dtype=np.uint32)
        bo = np.empty((nq, nb), dtype=np.int32)
        keys = np.empty((nq, ko), dtype=np.uint32) if want_keys else None
        _check(lib().lmi_search(self._h, _ptr(qn), _ptr(qs), nq, int(nb), int(k), _ptr(d), _ptr(i), _ptr(keys),
                                _ptr(bo), 0))
        return (d, i, bo, keys) if want_keys else (d, i, bo)

    # ---- query path, device tensors (torch), asynchronous on the handle's stream --------------
    def search_device(self, qn_t, qs_t, nb: int, k: int, d_t, i_t, keys_t=None, bo_t=None) -> None:
        _check(lib().lmi_search(self._h, _ptr(qn_t), _ptr(qs_t), int(qn_t.shape[0]), int(nb), int(k), _ptr(d_t),
                                _ptr(i_t), _ptr(keys_t), _ptr(bo_t), 1))

    def mlp_topk_device(self, qn_t, nb: int, bo_t, logits_t=None) -> None:
        _check(lib().lmi_mlp_topk(self._h, _ptr(qn_t), int(qn_t.shape[0]), int(nb), _ptr(bo_t), _ptr(logits_t), 1))

    def scan_topk_device(self, qs_t, bo_t, nb: int, k: int, d_t, i_t, keys_t=None) -> None:
        _check(lib().lmi_scan_topk(self._h, _ptr(qs_t), int(qs_t.shape[0]), _ptr(bo_t), int(nb), int(k), _ptr(d_t),
                                   _ptr(i_t), _ptr(keys_t), 1))

    def merge_gathered(self, gd, gi, gk, world: int, nq: int, kout: int, out_d, out_i, world_stride: int = 0) -> None:
        on_device = 0 if isinstance(gd, np.ndarray) else 1
        _check(lib().lmi_merge_gathered(self._h, _ptr(gd), _ptr(gi), _ptr(gk), int(world), int(world_stride),
                                        int(nq), int(kout), _ptr(out_d), _ptr(out_i), on_device))

    def copy_out(self, dst_pinned_t, src_dev_t) -> None:
        """src (device tensor) -> dst (pinned host tensor of the same byte size) by a kernel on the handle's stream."""
        nbytes = src_dev_t.numel() * src_dev_t.element_size()
        assert dst_pinned_t.is_pinned() and dst_pinned_t.numel() * dst_pinned_t.element_size() == nbytes
        assert src_dev_t.is_contiguous() and dst_pinned_t.is_contiguous()
        _check(lib().lmi_copy_out(self._h, _ptr(dst_pinned_t), _ptr(src_dev_t), nbytes))

    def copy_out_many(self, pairs) -> None:
        """[(dst pinned host tensor, src device tensor), ..] (up to 4) by ONE kernel on the handle's stream."""
        n = len(pairs)
        dst = (ctypes.c_void_p * n)()
        src = (ctypes.c_void_p * n)()
        nby = (ctypes.c_int64 * n)()
        for i, (d_t, s_t) in enumerate(pairs):
            nbytes = s_t.numel() * s_t.element_size()
            assert d_t.is_pinned() and d_t.numel() * d_t.element_size() == nbytes and s_t.is_contiguous() and d_t.is_contiguous()
            dst[i], src[i], nby[i] = d_t.data_ptr(), s_t.data_ptr(), nbytes
        _check(lib().lmi_copy_out_many(self._h, n, dst, src, nby))

    def pipeline_submit(self, s_in, s_nav, s_run, ev_in, ev_nav, ev_out, qn_h, qs_h, qn_d, qs_d, nb: int, k: int, d_out, i_out, bo_d, bo_h,
                        overlap_nav: bool) -> None:
        """One batch of a host-in -> host-out pipeline as ONE C call (lmi_pipeline_submit): raw stream / event handles, torch tensors."""
        _check(lib().lmi_pipeline_submit(self._h, s_in, s_nav, s_run, ev_in, ev_nav, ev_out, _ptr(qn_h), _ptr(qs_h), _ptr(qn_d), _ptr(qs_d),
                                         int(qn_d.shape[0]), int(nb), int(k), _ptr(d_out), _ptr(i_out), _ptr(bo_d), _ptr(bo_h), 1 if overlap_nav else 0))

    # ---- RCCL inside the library (the sharded exchange without torch.distributed) ----------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _check(lib().lmi_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """ncclComm_t (opaque pointer) over `world` ranks; collective: every rank calls it with rank 0's id."""
        comm = _vp()
        _check(lib().lmi_comm_init(self._h, int(rank), int(world), ctypes.c_char_p(unique_id), ctypes.byref(comm)))
        return comm

    @staticmethod
    def comm_destroy(comm) -> None:
        _check(lib().lmi_comm_destroy(comm))

    def allgather_merge(self, comm, rank: int, world: int, d_t, i_t, k_t, out_d_t, out_i_t) -> None:
        """This rank's [nq, kout] device tensors (dists, ids, keys) -> merged (dists, ids) on every rank."""
        nq, kout = int(d_t.shape[0]), int(d_t.shape[1])
        _check(lib().lmi_allgather_merge(self._h, comm, int(rank), int(world), _ptr(d_t), _ptr(i_t), _ptr(k_t), nq, kout,
                                         _ptr(out_d_t), _ptr(out_i_t)))

    def read_bucket(self, b: int, rows_out=None, ids_out=None):
        """(rows f32[n_b,d], ids u32[n_b]) of bucket b, in bucket order; `rows_out` / `ids_out`: C-contiguous
        numpy arrays of exactly that shape to fill instead of fresh ones (e.g. slices of one host slab)."""
        n = int(self.bucket_sizes()[b])
        rows = np.empty((n, self.d), dtype=np.float32) if rows_out is None else rows_out
        ids = np.empty(n, dtype=np.uint32) if ids_out is None else ids_out
        assert rows.shape == (n, self.d) and rows.dtype == np.float32 and rows.flags.c_contiguous
        assert ids.shape == (n,) and ids.dtype == np.uint32 and ids.flags.c_contiguous
        _check(lib().lmi_bucket_read(self._h, int(b), _ptr(rows), _ptr(ids)))
        return rows, ids

    def workspace_bytes(self, nq: int, nb: int) -> int:
        """Device bytes the per-call workspaces of a search of nq queries x nb buckets need (lmi_workspace_bytes)."""
        out = ctypes.c_int64(0)
        _check(lib().lmi_workspace_bytes(self._h, int(nq), int(nb), ctypes.byref(out)))
        return out.value

    def timings(self) -> np.ndarray:
        ms = np.zeros(T_COUNT, dtype=np.float32)
        _check(lib().lmi_timings(self._h, _ptr(ms)))
        return ms

    def set_timing(self, level: int) -> None:
        """2 (default): every phase from device-side clock stamps (no bubbles); 3: every phase from hipEvents (each a ~5 us bubble);
        1: hipEvents around the whole call only; 0: nothing."""
        _check(lib().lmi_set_timing(self._h, int(level)))

    def timings_reset(self) -> None:
        _check(lib().lmi_timings_reset(self._h))

    def timings_mean(self):
        """(mean ms per slot, calls averaged) since timings_reset -- one stream sync for the whole loop."""
        ms = np.zeros(T_COUNT, dtype=np.float32)
        n = ctypes.c_int(0)
        _check(lib().lmi_timings_mean(self._h, _ptr(ms), ctypes.byref(n)))
        return ms, n.value

    def scan_stats(self):
        fl = ctypes.c_double(0)
        pairs = ctypes.c_int64(0)
        items = ctypes.c_int64(0)
        _check(lib().lmi_scan_stats(self._h, ctypes.byref(fl), ctypes.byref(pairs), ctypes.byref(items)))
        return fl.value, pairs.value, items.value


def knn_ip(xq, xb, k: int = 10, device: int = 0):
    """faiss.knn(xq, xb, k, metric=faiss.METRIC_INNER_PRODUCT) on the GPU (LearnedIndex.py:360-365)."""
    xq = _np(xq, np.float32)
    xb = _np(xb, np.float32)
    assert xq.ndim == 2 and xb.ndim == 2 and xq.shape[1] == xb.shape[1]
    D = np.empty((xq.shape[0], k), dtype=np.float32)
    I = np.empty((xq.shape[0], k), dtype=np.int64)
    _check(lib().lmi_knn_ip(int(device), _ptr(xq), xq.shape[0], _ptr(xb), xb.shape[0], xq.shape[1], int(k),
                            _ptr(D), _ptr(I)))
    return D, I
