"""GPU (`-m gpu`): repeat-run determinism (SURVEY section 5 "determinism tests"; no GPU sanitizer exists on this pool).

The scan's candidate buffers are filled in atomics order and its work queues are drained in whatever order the
blocks arrive, so two runs of one batch store their intermediate lists differently; none of that may reach the
results.  One batch is answered 5 times on one handle and, at the same time, on a `lmi_clone_view` twin on a second
stream (two searches racing on the same index memory): every answer must be byte-identical -- in both scan modes, on
data with exact duplicates (ties), and equal to the oracle on a query sample."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prefilter", [True, False])
def test_same_batch_same_bytes_with_a_racing_twin(oracle, prefilter):
    from learnedmetricindex_amd import _capi

    dev = torch.device("cuda", 0)
    n, d, L, nb, nq, k = 400_000, 256, 24, 4, 3000, 10
    g = torch.Generator(device=dev).manual_seed(11)
    centres = torch.randn(L, d, generator=g, device=dev)
    lab = torch.randint(0, L, (n,), generator=g, device=dev)
    X = torch.nn.functional.normalize(centres[lab] + 0.9 * torch.randn(n, d, generator=g, device=dev), dim=1)
    X[1000:1400] = X[999]                       # 401 identical rows: exact ties inside one bucket's top-10 window
    lab[1000:1400] = lab[999]
    Q = torch.nn.functional.normalize(centres[torch.randint(0, L, (nq,), generator=g, device=dev)]
                                      + 0.9 * torch.randn(nq, d, generator=g, device=dev), dim=1).contiguous()
    Q[:40] = X[999]                             # queries sitting on the duplicates
    order = (Q @ centres.T).topk(nb, dim=1).indices.to(torch.int32).contiguous()
    order[:40, 0] = lab[999].to(torch.int32)

    a = _capi.Index(0, prefilter=prefilter)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a.set_stream(sa.cuda_stream)
    a.set_buckets(X.contiguous(), lab.cpu().numpy().astype(np.int64), L)
    b = a.clone_view()
    b.set_stream(sb.cuda_stream)
    torch.cuda.synchronize()
    outs = []
    for rep in range(5):
        da = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ia = torch.empty((nq, k), dtype=torch.int32, device=dev)
        a.scan_topk_device(Q, order, nb, k, da, ia)
        outs.append((da, ia))
        if rep in (1, 3):                        # the twin's search runs while the handle's is in flight
            db = torch.empty((nq, k), dtype=torch.float32, device=dev)
            ib = torch.empty((nq, k), dtype=torch.int32, device=dev)
            b.scan_topk_device(Q, order, nb, k, db, ib)
            outs.append((db, ib))
    torch.cuda.synchronize()
    d0, i0 = outs[0]
    for dj, ij in outs[1:]:
        assert torch.equal(ij, i0) and torch.equal(dj.view(torch.int32), d0.view(torch.int32))
    # ties: the duplicates come back lowest row first (ids are 1-based row numbers)
    ih = i0.cpu().numpy().view(np.uint32)
    assert (ih[:40, 0] == 1000).all() and (np.diff(ih[:40].astype(np.int64), axis=1) == 1).all()
    # and the answer is the oracle's, on a sample of queries
    sel = np.concatenate([np.arange(8), np.random.RandomState(3).choice(nq, 56, replace=False)])
    Xh, labh, Qh, oh = X.cpu().numpy(), lab.cpu().numpy(), Q.cpu().numpy(), order.cpu().numpy()
    do, io, _ = oracle.search(None, None, Xh, Qh[sel], labh[:, None].astype(np.int64), nb, k, nthreads=8,
                              bucket_order=oh[sel][:, :, None])
    np.testing.assert_array_equal(ih[sel], io)
    np.testing.assert_array_equal(d0.cpu().numpy()[sel].astype(np.float64), do)
    b.close()
    a.close()
