"""GPU (`-m gpu`): the host-in -> host-out pipeline (learnedmetricindex_amd/pipeline.py) returns, batch by batch,
exactly what one synchronous lmi_search on host pointers returns (which the other tests pin to the oracle), with
several batches in flight, pageable and pinned inputs, distinct navigation / scan vectors (fixture G6)."""
import numpy as np
import pytest
import torch

from helpers import inputs_for, layers_from, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,depth,mode", [("G3", 2, "nav"), ("G6", 3, "nav"), ("G4", 1, "nav"), ("G3", 2, "plain"), ("G6", 2, "twin"),
                                             ("G3", 2, "nav+direct"), ("G4", 2, "plain+direct"), ("G3", 2, "nav+direct+lazybo"), ("G4", 2, "plain+lazybo")])
def test_pipeline_equals_synchronous_search(oracle, name, depth, mode):
    """mode: "nav" (default) -- the next batch's MLP on a navigation stream beside the current scan; "plain" -- one lmi_search
    per batch; "twin" -- batches alternate between the handle and a clone of it (lmi_clone_view: same index memory);
    "+direct" -- the search's last kernels store (dists, ids) straight into the pinned host buffers (no download kernel);
    "+lazybo" -- the bucket order is not downloaded with every batch but fetched from the slot's device buffer on demand."""
    lazybo = mode.endswith("+lazybo")
    mode = mode[:-len("+lazybo")] if lazybo else mode
    direct = mode.endswith("+direct")
    mode = mode.split("+")[0]
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.pipeline import HostPipeline

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    L = layers[-1][0].shape[0]
    idx = _capi.Index(0)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, g["data_prediction"][:, 0], L)
    nq = 96
    same = Xn.shape[1] == Xs.shape[1] and np.array_equal(Qn, Qs)
    pipe = HostPipeline(idx, nq, Qn.shape[1], Qs.shape[1], nb, k, depth=depth, same_queries=same, want_bucket_order=not lazybo,
                        overlap_inference=mode == "nav", two_handles=mode == "twin", direct_out=direct)
    assert pipe.direct_out == direct
    assert pipe.calls_per_batch == (2 if mode == "nav" else 1) and len(pipe.handles) == (2 if mode == "twin" else 1)
    rs = np.random.RandomState(0)
    batches = [np.sort(rs.choice(Qn.shape[0], nq, replace=False)) for _ in range(7)]
    tickets, expect = [], []
    for bi, sel in enumerate(batches):
        qn, qs = np.ascontiguousarray(Qn[sel]), np.ascontiguousarray(Qs[sel])
        if bi % 2:  # already pinned torch tensors: no staging copy
            tickets.append(pipe.submit(torch.from_numpy(qn).pin_memory(), torch.from_numpy(qs).pin_memory()))
        else:
            tickets.append(pipe.submit(qn, qs))
        if len(tickets) > depth:  # a ticket leaves the ring `depth` submits later: read it before that
            t = tickets[-1 - depth]
        # results of the newest ticket (waits for that batch only)
        d, i = pipe.result(tickets[-1])
        bo = pipe.bucket_order(tickets[-1])
        expect.append((d.copy(), i.copy(), bo.copy()))
    pipe.drain()
    idx.set_stream(0)
    for sel, (d, i, bo) in zip(batches, expect):
        d0, i0, bo0 = idx.search(np.ascontiguousarray(Qn[sel]), np.ascontiguousarray(Qs[sel]), nb, k)
        np.testing.assert_array_equal(i, i0)
        np.testing.assert_array_equal(d, d0)
        np.testing.assert_array_equal(bo, bo0)
    do, io, _ = oracle.search(layers, Qn[batches[0]], Xs, Qs[batches[0]], g["data_prediction"], nb, k, nthreads=4)
    np.testing.assert_array_equal(expect[0][1], io)
    np.testing.assert_array_equal(expect[0][0].astype(np.float64), do)
    idx.close()


@pytest.mark.parametrize("name,mode", [("G3", "nav"), ("G6", "plain")])
def test_pipeline_graph_replay_equals_eager(name, mode):
    """use_graph: every batch is ONE hipGraph launch (upload, MLP, routing, scan, tail, download over three streams, captured once per
    slot and pinned source).  Replays must return what the eager pipeline returns, batch after batch -- including the routing kernels'
    call tag, which lives in device memory because a replay freezes kernel arguments (lmi_front.h)."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.pipeline import HostPipeline

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    L = layers[-1][0].shape[0]
    idx = _capi.Index(0)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, g["data_prediction"][:, 0], L)
    nq = 128
    same = Xn.shape[1] == Xs.shape[1] and np.array_equal(Qn, Qs)
    rs = np.random.RandomState(1)
    sels = [np.sort(rs.choice(Qn.shape[0], nq, replace=False)) for _ in range(3)]
    pinned = [(torch.from_numpy(np.ascontiguousarray(Qn[s_])).pin_memory(), torch.from_numpy(np.ascontiguousarray(Qs[s_])).pin_memory()) for s_ in sels]
    out = {}
    for use_graph in (False, True):
        pipe = HostPipeline(idx, nq, Qn.shape[1], Qs.shape[1], nb, k, depth=2, same_queries=same, want_bucket_order=True,
                            overlap_inference=mode == "nav", use_graph=use_graph)
        assert pipe.use_graph == use_graph
        for qn, qs in pinned:
            pipe.capture(qn, qs)
        got = []
        for rep in range(9):   # every (slot, batch) pair several times: replays of the same graph with other batches in between
            qn, qs = pinned[rep % 3]
            t = pipe.submit(qn, qs)
            d, i = pipe.result(t)
            got.append((d.copy(), i.copy(), pipe.bucket_order(t).copy()))
        # a pageable batch goes through the slot's staging buffer (its own graph)
        t = pipe.submit(np.ascontiguousarray(Qn[sels[1]]), np.ascontiguousarray(Qs[sels[1]]))
        d, i = pipe.result(t)
        got.append((d.copy(), i.copy(), pipe.bucket_order(t).copy()))
        pipe.drain()
        out[use_graph] = got
        if use_graph:
            assert sum(len(s_["graphs"]) for s_ in pipe.slots) >= 6
    for (d0, i0, b0), (d1, i1, b1) in zip(out[False], out[True]):
        np.testing.assert_array_equal(i0, i1)
        np.testing.assert_array_equal(d0, d1)
        np.testing.assert_array_equal(b0, b1)
    idx.set_stream(0)
    d_ref, i_ref, _ = idx.search(np.ascontiguousarray(Qn[sels[1]]), np.ascontiguousarray(Qs[sels[1]]), nb, k)
    np.testing.assert_array_equal(out[True][-1][1], i_ref)
    np.testing.assert_array_equal(out[True][-1][0], d_ref)
    idx.close()


def test_pipelines_share_streams():
    """Every pipeline of a process runs on the same streams of its device unless asked otherwise (profiles/r05_stream_queues.txt)."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.pipeline import HostPipeline

    g = load_golden("G6")
    Xn, Qn, Xs, Qs = inputs_for("G6", g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    idx = _capi.Index(0)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, g["data_prediction"][:, 0], layers[-1][0].shape[0])
    nq = 64
    mk = lambda **kw: HostPipeline(idx, nq, Qn.shape[1], Qs.shape[1], nb, k, depth=2, **kw)   # noqa: E731
    a, b, c = mk(), mk(), mk(share_streams=False)
    for f in ("s_in", "s_run", "s_nav"):
        assert getattr(a, f).cuda_stream == getattr(b, f).cuda_stream
        assert getattr(c, f).cuda_stream != getattr(a, f).cuda_stream
    assert len({a.s_in.cuda_stream, a.s_run.cuda_stream, a.s_nav.cuda_stream}) == 3
    qn, qs = np.ascontiguousarray(Qn[:nq]), np.ascontiguousarray(Qs[:nq])
    got = []
    for pipe in (a, c, b):
        t = pipe.submit(qn, qs)
        d, i = pipe.result(t)
        got.append((d.copy(), i.copy()))
        pipe.drain()
    for d, i in got[1:]:
        np.testing.assert_array_equal(i, got[0][1])
        np.testing.assert_array_equal(d, got[0][0])
    idx.close()
