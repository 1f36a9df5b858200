"""GPU (`-m gpu`): the one-launch MLP (csrc/lmi_mlp_fused.h) against the oracle and against the per-layer kernels:
logits, bucket order and predict_proba bit-identical, for every model shape of the reference's zoo that the
fixtures hold (MLP 64->128->12, MLP-4 768->512->120, 3-layer MLP-5 on 32-d, 256 classes on 45-d), ragged batch
sizes, a wide output layer (1 024 classes: logits through global memory) and wide inputs (d = 2 048, chunked)."""
import numpy as np
import pytest

from helpers import inputs_for, layers_from, load_golden

pytestmark = pytest.mark.gpu


def both_modes(layers):
    from learnedmetricindex_amd import _capi

    out = []
    for fused in (2, 0):   # 2: the fused kernel whatever the batch size, 0: the per-layer kernels
        idx = _capi.Index(0)
        idx.set_fused_mlp(fused)
        idx.set_mlp(layers)
        out.append(idx)
    return out


@pytest.mark.parametrize("name", ["G1", "G3", "G5", "G6"])
def test_fused_equals_layers_and_oracle(oracle, name):
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb = int(g["n_buckets"])
    fused, plain = both_modes(layers)
    for nq in (Qn.shape[0], 33, 1):
        q = Qn[:nq]
        of, lf = fused.mlp_topk(q, nb, want_logits=True)
        op, lp = plain.mlp_topk(q, nb, want_logits=True)
        lo = oracle.forward_logits(layers, q, nthreads=4)
        np.testing.assert_array_equal(lf, lo)
        np.testing.assert_array_equal(lp, lo)
        np.testing.assert_array_equal(of, op)
        np.testing.assert_array_equal(of, oracle.rank_classes(lo, nb))
        np.testing.assert_array_equal(fused.mlp_topk(q, nb), of)   # without the logits output
        pf, cf = fused.mlp_proba(q)
        pp, cp = plain.mlp_proba(q)
        po, co = oracle.predict_proba(layers, q, nthreads=4)
        np.testing.assert_array_equal(cf, co.astype(np.int32))
        np.testing.assert_array_equal(pf, po)
        np.testing.assert_array_equal(pp, po)
        np.testing.assert_array_equal(cp, cf)
    np.testing.assert_array_equal(fused.mlp_topk(Qn, nb), g["ref_bucket_order"][:, :, 0])
    fused.close()
    plain.close()


@pytest.mark.parametrize("d,hidden,L", [(2048, 256, 40), (768, 512, 1024), (45, 8, 3), (100, 512, 512)])
def test_fused_shapes(oracle, d, hidden, L):
    rs = np.random.RandomState(d + L)
    layers = [((rs.randn(hidden, d) / np.sqrt(d)).astype(np.float32), rs.randn(hidden).astype(np.float32) * 0.1),
              ((rs.randn(L, hidden) / np.sqrt(hidden)).astype(np.float32), rs.randn(L).astype(np.float32) * 0.1)]
    q = rs.randn(77, d).astype(np.float32)
    fused, plain = both_modes(layers)
    nb = min(5, L)
    of, lf = fused.mlp_topk(q, nb, want_logits=True)
    lo = oracle.forward_logits(layers, q, nthreads=4)
    np.testing.assert_array_equal(lf, lo)
    np.testing.assert_array_equal(of, oracle.rank_classes(lo, nb))
    np.testing.assert_array_equal(plain.mlp_topk(q, nb), of)
    pf, cf = fused.mlp_proba(q)
    po, co = oracle.predict_proba(layers, q, nthreads=4)
    np.testing.assert_array_equal(cf, co.astype(np.int32))
    np.testing.assert_array_equal(pf, po)
    fused.close()
    plain.close()


def test_batch_with_a_thin_last_round_is_split(oracle):
    """Default mode, a batch whose last round of 32-query blocks would fill under 30 % of the CUs (here 8 192 + 600
    queries = 256 + 19 blocks): the head takes the fused kernel, the tail the per-layer kernels on a side stream.
    Same bucket order as either pure form and as the oracle."""
    from learnedmetricindex_amd import _capi

    rs = np.random.RandomState(31)
    d, hidden, L, nb = 96, 128, 40, 4
    layers = [((rs.randn(hidden, d) / np.sqrt(d)).astype(np.float32), rs.randn(hidden).astype(np.float32) * 0.1),
              ((rs.randn(L, hidden) / np.sqrt(hidden)).astype(np.float32), rs.randn(L).astype(np.float32) * 0.1)]
    q = rs.randn(8192 + 600, d).astype(np.float32)
    out = {}
    for mode in (1, 2, 0):
        idx = _capi.Index(0)
        idx.set_fused_mlp(mode)
        idx.set_mlp(layers)
        out[mode] = idx.mlp_topk(q, nb)
        if mode == 1:   # twice: the side stream and its events are created on first use
            np.testing.assert_array_equal(idx.mlp_topk(q, nb), out[mode])
        idx.close()
    np.testing.assert_array_equal(out[1], out[2])
    np.testing.assert_array_equal(out[1], out[0])
    np.testing.assert_array_equal(out[1], oracle.rank_classes(oracle.forward_logits(layers, q, nthreads=8), nb))
