"""orbfe_match_knn2_noalloc (include/orbfe.h): the kNN-2 search with its scratch in a file-scope device array, for callers
that cannot allocate.  It has no counterpart in the reference; the check is against orbfe_match_knn2 (whose results the
other tests pin to the oracle), bit for bit, on two calls in a row, the second with a smaller input -- the shared array
keeps the first call's contents, which must not leak into the second."""
import numpy as np
import pytest

from orb_slam2_amd import capi, synth

pytestmark = pytest.mark.gpu


def _run(m, fn, q, r, idx_base=0):
    import torch
    dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    out = [torch.full((len(q),), -7, dtype=torch.int32, device="cuda") for _ in range(3)]
    fn(dq.data_ptr(), len(q), dr.data_ptr(), len(r), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), idx_base=idx_base)
    return [o.cpu().numpy() for o in out]


def test_knn2_noalloc_matches_knn2_on_two_calls_in_a_row():
    pytest.importorskip("torch")
    m = capi.ORBmatcher()
    # first call: a sampled two-stage plan (>= 32768 references: pre-pass, groups); second: smaller, single-stage plan
    for nq, nr, base in [(700, 70000, 1000), (130, 900, 0), (65, 0, 5)]:
        r = synth.random_descriptors(max(nr, 1), 11 + nr)[:nr]
        q = synth.planted_queries(r, nq, 3, planted_frac=0.4) if nr else synth.random_descriptors(nq, 5)
        want = _run(m, m.knn2_device, q, r, base)
        got = _run(m, m.knn2_noalloc_device, q, r, base)
        for a, b, name in zip(got, want, ("best_dist", "best_idx", "second_dist")):
            assert np.array_equal(a, b), (nq, nr, name)
    # sizes beyond the static array are refused, loudly, before anything is launched
    import torch
    big = torch.zeros((1000000, 32), dtype=torch.uint8, device="cuda")
    o = torch.zeros(70000, dtype=torch.int32, device="cuda")
    with pytest.raises(capi.OrbfeError) as ei:
        m.knn2_noalloc_device(big.data_ptr(), 70000, big.data_ptr(), 70000, o.data_ptr(), o.data_ptr(), o.data_ptr())
    assert ei.value.code == capi.ERR_CAPACITY
    with pytest.raises(capi.OrbfeError) as ei:
        m.knn2_noalloc_device(big.data_ptr(), 60000, big.data_ptr(), 1000000, o.data_ptr(), o.data_ptr(), o.data_ptr())
    assert ei.value.code == capi.ERR_CAPACITY and "scratch" in str(ei.value)
