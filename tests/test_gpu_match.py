"""GPU parity of the Hamming matching entry points against the oracle (bit-exact: distances and indices are integers)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from orb_slam2_amd import capi, synth

pytestmark = pytest.mark.gpu
_FUZZ_FIRST = int(os.environ.get("ORBFE_FUZZ_FIRST", "0"))     # a soak continues where the last one stopped
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _knn2(m, q, r, idx_base=0):
    nq = len(q)
    dq, dr = _dev(q), _dev(r if len(r) else np.zeros((1, 32), np.uint8))
    bd, bi, sd = (torch.zeros(max(nq, 1), dtype=torch.int32, device="cuda") for _ in range(3))
    m.knn2_device(dq.data_ptr(), nq, dr.data_ptr(), len(r), bd.data_ptr(), bi.data_ptr(), sd.data_ptr(), idx_base)
    return bd.cpu().numpy()[:nq], bi.cpu().numpy()[:nq], sd.cpu().numpy()[:nq]


@pytest.mark.parametrize("nq,nr", [(1, 1), (7, 3), (300, 1000), (2000, 20000), (257, 4097)])
def test_knn2_matches_oracle(nq, nr):
    db = synth.random_descriptors(nr, 100 + nr)
    q = synth.planted_queries(db, nq, 200 + nq, planted_frac=0.3 if nq >= 10 else 0.0)
    if nr > 10:
        db[nr // 2] = db[3]          # exact duplicate rows: the first index must win, second == best
        q[0] = db[3]
    m = capi.ORBmatcher()
    got = _knn2(m, q, db)
    ref = O.knn2(q, db)
    for g, r, name in zip(got, ref, ("best_dist", "best_idx", "second_dist")):
        assert np.array_equal(g, r), name


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("ORBFE_MATCH_FUZZ_CASES", "12"))))
def test_knn2_fuzz(case):
    """Seeded random shapes on both sides of the seeded-scan threshold (32 768 references), with descriptor statistics that
    force ties: a small alphabet of rows (many equal distances, duplicates everywhere), sparse rows (few set bits, distances
    bunched near zero), or uniform rows with planted near-duplicates.  Best distance, best index (first minimum) and second
    distance equal the oracle's for every query."""
    rng = np.random.default_rng(4000 + case)
    nr = int(np.exp(rng.uniform(0.0, np.log(150000.0)))) if case % 2 else int(rng.integers(20000, 150001))
    nq = int(rng.integers(1, min(3000, max(2, 150_000_000 // max(nr, 1))) + 1))
    kind = int(rng.integers(0, 3))
    if kind == 0:                                     # small alphabet: rows repeat, distances tie
        alpha = synth.random_descriptors(int(rng.integers(2, 40)), 5000 + case)
        db = alpha[rng.integers(0, len(alpha), nr)]
        q = alpha[rng.integers(0, len(alpha), nq)].copy()
        flip = rng.integers(0, 256, nq)
        q[np.arange(nq), flip >> 3] ^= (1 << (flip & 7)).astype(np.uint8)
    elif kind == 1:                                   # sparse rows
        db = np.zeros((nr, 32), np.uint8)
        q = np.zeros((nq, 32), np.uint8)
        for a in (db, q):
            for _ in range(int(rng.integers(0, 6))):
                bit = rng.integers(0, 256, len(a))
                a[np.arange(len(a)), bit >> 3] |= (1 << (bit & 7)).astype(np.uint8)
    else:
        db = synth.random_descriptors(nr, 6000 + case)
        q = synth.planted_queries(db, nq, 7000 + case, planted_frac=0.3 if nq >= 10 else 0.0)
    idx_base = int(rng.integers(0, 1 << 20)) if case % 3 == 0 else 0
    m = capi.ORBmatcher()
    got = _knn2(m, np.ascontiguousarray(q), np.ascontiguousarray(db), idx_base=idx_base)
    ref = O.knn2(np.ascontiguousarray(q), np.ascontiguousarray(db))
    assert np.array_equal(got[0], ref[0]), (case, nq, nr, kind, "best_dist")
    assert np.array_equal(got[1], np.where(ref[1] >= 0, ref[1] + idx_base, -1)), (case, nq, nr, kind, "best_idx")
    assert np.array_equal(got[2], ref[2]), (case, nq, nr, kind, "second_dist")


def test_knn2_filtered_scan_matches_oracle():
    """Databases of 65536 references and more take the seeded scan (knn2_scan_kernel<true>: a pre-pass over the first 4096
    references bounds the second-best distance, the selection then runs only where a reference can still matter).  Exact
    against the oracle on random data with planted matches and ties, on a database sorted so that every reference improves
    on the one before (the selection runs on every trip), and where the sample holds a single distinct descriptor."""
    m = capi.ORBmatcher()
    nr, nq = 70001, 300                                      # odd size: the tail loop of the last chunk runs too
    db = synth.random_descriptors(nr, 900)
    q = synth.planted_queries(db, nq, 901, planted_frac=0.3)
    db[60000] = db[7]                                        # duplicate rows far apart: the first index wins, second == best
    q[0] = db[7]
    db[69999] = q[1]                                         # the exact match sits in the last chunk, after the sample
    for case in range(3):
        if case == 1:                                        # distances to q[2] fall steadily along the database
            order = np.argsort(-np.unpackbits(db ^ q[2][None, :], axis=1).sum(1), kind="stable")
            db = np.ascontiguousarray(db[order])
        if case == 2:                                        # the sample is 4096 copies of one row: its second-best = its best
            db = db.copy()
            db[:4096] = db[5000]
        got = _knn2(m, q, db)
        ref = O.knn2(q, db)
        for g, r, name in zip(got, ref, ("best_dist", "best_idx", "second_dist")):
            assert np.array_equal(g, r), (case, name)
    a = _knn2(m, q, db, idx_base=123456)
    assert np.array_equal(a[1], ref[1] + 123456) and np.array_equal(a[0], ref[0]) and np.array_equal(a[2], ref[2])


def test_knn2_many_queries_fewer_main_chunks_than_sample_chunks():
    """70 000 queries make the main scan use fewer chunks (29) than the pre-pass (32 chunks of 64 references): the per-chunk
    buffers are sized for the larger of the two.  No oracle at this size (2.8e9 pairs): every query block must get what
    the same queries get when they are matched on their own (a call that takes hundreds of chunks), and the first 300
    are checked against the oracle."""
    m = capi.ORBmatcher()
    nr, nq = 40000, 70000
    db = synth.random_descriptors(nr, 910)
    q = synth.planted_queries(db, nq, 911, planted_frac=0.2)
    got = _knn2(m, q, db)
    for lo in (0, 33333, nq - 1500):
        part = _knn2(m, q[lo:lo + 1500], db)
        for g, r, name in zip(got, part, ("best_dist", "best_idx", "second_dist")):
            assert np.array_equal(g[lo:lo + 1500], r), (lo, name)
    ref = O.knn2(q[:300], db)
    for g, r, name in zip(got, ref, ("best_dist", "best_idx", "second_dist")):
        assert np.array_equal(g[:300], r), name


def test_knn2_empty_reference_and_index_base():
    m = capi.ORBmatcher()
    q = synth.random_descriptors(5, 1)
    bd, bi, sd = _knn2(m, q, np.zeros((0, 32), np.uint8))
    assert (bd == 256).all() and (bi == -1).all() and (sd == 256).all()
    db = synth.random_descriptors(500, 2)
    a = _knn2(m, q, db, idx_base=1000)
    b = O.knn2(q, db)
    assert np.array_equal(a[1], b[1] + 1000) and np.array_equal(a[0], b[0])


def test_knn2_sharded_merge_equals_unsharded():
    db = synth.random_descriptors(30000, 77)
    q = synth.planted_queries(db, 500, 78, planted_frac=0.2)
    m = capi.ORBmatcher()
    ref = O.knn2(q, db)
    bounds = np.linspace(0, len(db), 9).astype(int)
    parts = [_knn2(m, q, db[bounds[s]:bounds[s + 1]], idx_base=int(bounds[s])) for s in range(8)]
    got = capi.merge_knn2(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]))
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)


def test_knn2_full_size_properties():
    """BASELINE config 5 shard size (2000 x 125000): too slow for the scalar oracle, so check size-independent properties:
    best <= second, returned distance equals the recomputed distance of the returned index, planted rows are found,
    and a numpy column-chunked scan of a query subset agrees exactly."""
    db = synth.random_descriptors(125000, 500)
    q = synth.planted_queries(db, 2000, 501, planted_frac=0.1)
    m = capi.ORBmatcher()
    bd, bi, sd = _knn2(m, q, db)
    assert (bd <= sd).all() and (bi >= 0).all()
    recomputed = np.unpackbits(q ^ db[bi], axis=1).sum(1)
    assert np.array_equal(recomputed, bd)
    sub = np.arange(0, 2000, 125)
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for i in sub:
        d = pop[q[i][None, :] ^ db].sum(1)
        order = np.lexsort((np.arange(len(db)), d))
        assert bd[i] == d[order[0]] and bi[i] == order[0] and sd[i] == d[order[1]]


def test_knn2_full_database_one_gpu():
    """The bench's single-GPU workload (2000 queries x 1 M descriptors: other chunk / group counts than the 125 000 shard):
    size-independent properties over all queries + 16 queries verified by a numpy scan of the whole database, and the
    no-alloc entry point on the same input."""
    torch = pytest.importorskip("torch")
    db = synth.random_descriptors(1_000_000, 7000)
    q = synth.planted_queries(db, 2000, 7001, planted_frac=0.1)
    m = capi.ORBmatcher()
    bd, bi, sd = _knn2(m, q, db)
    assert (bd <= sd).all() and (bi >= 0).all() and (bi < len(db)).all()
    assert np.array_equal(np.unpackbits(q ^ db[bi], axis=1).sum(1), bd)
    assert (bd[::10] <= 60).all()                                  # the planted rows (every tenth query) found a near duplicate
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for i in range(0, 2000, 125):
        d = np.zeros(len(db), np.int32)
        for c0 in range(0, len(db), 250000):
            d[c0:c0 + 250000] = pop[q[i][None, :] ^ db[c0:c0 + 250000]].sum(1)
        first = int(np.argmin(d))                                  # first minimum = lowest index among equals
        assert bd[i] == d[first] and bi[i] == first
        d[first] = 1 << 20
        assert sd[i] == d.min()
    dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    out = [torch.zeros(len(q), dtype=torch.int32, device="cuda") for _ in range(3)]
    m.knn2_noalloc_device(dq.data_ptr(), len(q), dr.data_ptr(), len(db), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    for a, b in zip(out, (bd, bi, sd)):
        assert np.array_equal(a.cpu().numpy(), b)


def test_knn2_sharded_rccl_one_rank():
    """orbfe_knn2_sharded_* (RCCL bound at run time inside liborbfe: ncclBroadcast of the queries, local scan, ncclAllGather
    of the per-shard triples, fold on the device) with a communicator of one rank -- all this box has GPUs for -- equals the
    plain call; ranks > 1 are covered on the CPU by tests/test_distributed_cpu.py (merge) and need an 8-GPU node to run."""
    torch = pytest.importorskip("torch")
    db = synth.random_descriptors(40000, 91)
    q = synth.planted_queries(db, 700, 92, planted_frac=0.3)
    m = capi.ORBmatcher()
    want = _knn2(m, q, db, idx_base=5000)
    sh = capi.ShardedKnn2(m, capi.sharded_unique_id(), 0, 1, 1024)
    dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(db).cuda()
    for nq in (700, 130):                                          # a second, smaller batch through the same buffers
        out = [torch.full((nq,), -3, dtype=torch.int32, device="cuda") for _ in range(3)]
        sh.run(dq.data_ptr(), nq, dr.data_ptr(), len(db), 5000, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        for a, b in zip(out, want):
            assert np.array_equal(a.cpu().numpy(), b[:nq])
    with pytest.raises(capi.OrbfeError):
        sh.run(dq.data_ptr(), 2000, dr.data_ptr(), len(db), 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    sh.close()


def test_csr_matches_oracle():
    rng = np.random.default_rng(5)
    r = synth.random_descriptors(3000, 9)
    q = synth.planted_queries(r, 1500, 10, planted_frac=0.5)
    counts = rng.integers(0, 50, 1500)
    counts[:5] = 0                                  # empty candidate lists -> (256, -1, 256)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    idx = rng.integers(0, 3000, off[-1]).astype(np.int32)
    m = capi.ORBmatcher()
    dq, dr, doff, didx = _dev(q), _dev(r), _dev(off), _dev(idx)
    bd, bi, sd = (torch.zeros(1500, dtype=torch.int32, device="cuda") for _ in range(3))
    m.csr_device(dq.data_ptr(), 1500, dr.data_ptr(), 3000, doff.data_ptr(), didx.data_ptr(), bd.data_ptr(), bi.data_ptr(),
                 sd.data_ptr())
    ref = O.match_csr(q, r, off, idx)
    for g, rr in zip((bd, bi, sd), ref):
        assert np.array_equal(g.cpu().numpy(), rr)


def test_stereo_rowband_matches_oracle():
    left, right = synth.stereo_pair(1241, 376, synth.BASE_SEED + 50)
    gl, gr = capi.ORBextractor(2000, 1.2, 8, 20, 7), capi.ORBextractor(2000, 1.2, 8, 20, 7)
    kl, dl = gl(left)
    kr, dr = gr(right)
    ol, orr = O.OracleExtractor(2000, 1.2, 8, 20, 7), O.OracleExtractor(2000, 1.2, 8, 20, 7)
    kol, dol = ol.extract(left)
    kor, dor = orr.extract(right)
    assert np.array_equal(kl, kol) and np.array_equal(dr, dor)
    sf = gl.GetScaleFactors()
    bf, b = 386.1448, 0.5371657                      # KITTI00-02.yaml: Camera.bf, bf / fx(718.856)
    max_d = np.float32(bf) / np.float32(b)
    ref_bd, ref_bi = O.stereo_rowband(kl, dl, kr, dr, sf, 376, 0.0, float(max_d))
    m = capi.ORBmatcher()
    dkl = _dev(kl.view(np.uint8).reshape(-1, 28))
    dkr = _dev(kr.view(np.uint8).reshape(-1, 28))
    ddl, ddr = _dev(dl), _dev(dr)
    bd, bi = (torch.zeros(len(kl), dtype=torch.int32, device="cuda") for _ in range(2))
    m.stereo_rowband_device(dkl.data_ptr(), ddl.data_ptr(), len(kl), dkr.data_ptr(), ddr.data_ptr(), len(kr), sf, 376, 0.0,
                            float(max_d), bd.data_ptr(), bi.data_ptr())
    assert np.array_equal(bd.cpu().numpy(), ref_bd)
    assert np.array_equal(bi.cpu().numpy(), ref_bi)
    assert (ref_bd < 75).sum() > 200                 # the synthetic pair really produces stereo matches


@pytest.mark.parametrize("same_handle", [False, True])
def test_stereo_match_full_matches_oracle(same_handle):
    """N3: row band + 11x11 SAD refinement + parabola + median gate (src/Frame.cc:466-640); mvuRight / mvDepth bit-exact."""
    W, H = 1241, 376
    left, right = synth.stereo_pair(W, H, synth.BASE_SEED + 77)
    ol, orr = O.OracleExtractor(2000, 1.2, 8, 20, 7), O.OracleExtractor(2000, 1.2, 8, 20, 7)
    kol, dol = ol.extract(left)
    kor, dor = orr.extract(right)
    bf, b = 386.1448, 0.5371657                      # KITTI00-02.yaml: Camera.bf, bf / fx(718.856)
    ref_ur, ref_dep = O.stereo_full(ol, orr, kol, dol, kor, dor, bf, b)
    assert (ref_dep > 0).sum() > 200                 # the pair produces real stereo matches, and the gate removes some
    m = capi.ORBmatcher()
    stride = (W + 63) // 64 * 64
    cap = 2100
    if same_handle:
        host = np.zeros((2, H, stride), np.uint8)
        host[0, :, :W], host[1, :, :W] = left, right
        frames = torch.from_numpy(host).cuda()
        ex = capi.ORBextractor(2000, 1.2, 8, 20, 7)
        kps = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(frames.data_ptr(), 2, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                                cap, sync=False)    # the matcher must order itself after the extractor's stream
        nl, nr = len(kol), len(kor)
        ur = torch.zeros(nl, dtype=torch.float32, device="cuda")
        dep = torch.zeros(nl, dtype=torch.float32, device="cuda")
        m.stereo_match_device(ex, 0, ex, 1, kps[0].data_ptr(), desc[0].data_ptr(), nl, kps[1].data_ptr(), desc[1].data_ptr(), nr,
                              bf, b, ur.data_ptr(), dep.data_ptr())
        n = cnt.cpu().numpy()
        assert n[0] == nl and n[1] == nr
    else:
        gl, gr = capi.ORBextractor(2000, 1.2, 8, 20, 7), capi.ORBextractor(2000, 1.2, 8, 20, 7)
        kl, dl = gl(left)
        kr, dr = gr(right)
        assert np.array_equal(kl, kol) and np.array_equal(dr, dor)
        dkl = _dev(kl.view(np.uint8).reshape(-1, 28))
        dkr = _dev(kr.view(np.uint8).reshape(-1, 28))
        ddl, ddr = _dev(dl), _dev(dr)
        ur = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
        dep = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
        m.stereo_match_device(gl, 0, gr, 0, dkl.data_ptr(), ddl.data_ptr(), len(kl), dkr.data_ptr(), ddr.data_ptr(), len(kr),
                              bf, b, ur.data_ptr(), dep.data_ptr())
    got_ur, got_dep = ur.cpu().numpy(), dep.cpu().numpy()
    assert np.array_equal(got_ur.view(np.uint32), ref_ur.view(np.uint32))      # bit-exact float32
    assert np.array_equal(got_dep.view(np.uint32), ref_dep.view(np.uint32))


def test_stereo_match_no_matches_and_argument_errors():
    W, H = 640, 480
    img = synth.textured_image(W, H, synth.BASE_SEED + 5)
    g = capi.ORBextractor(500, 1.2, 8, 20, 7)
    k, d = g(img)
    m = capi.ORBmatcher()
    dk, dd = _dev(k.view(np.uint8).reshape(-1, 28)), _dev(d)
    ur = torch.full((len(k),), 7.0, dtype=torch.float32, device="cuda")
    dep = torch.full((len(k),), 7.0, dtype=torch.float32, device="cuda")
    # no right keypoints: every entry is -1, the gate has nothing to do
    m.stereo_match_device(g, 0, g, 0, dk.data_ptr(), dd.data_ptr(), len(k), dk.data_ptr(), dd.data_ptr(), 0, 40.0, 0.08,
                          ur.data_ptr(), dep.data_ptr())
    assert (ur.cpu().numpy() == -1).all() and (dep.cpu().numpy() == -1).all()
    with pytest.raises(capi.OrbfeError):
        m.stereo_match_device(g, 1, g, 0, dk.data_ptr(), dd.data_ptr(), len(k), dk.data_ptr(), dd.data_ptr(), len(k), 40.0, 0.08,
                              ur.data_ptr(), dep.data_ptr())
    fresh = capi.ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(capi.OrbfeError):
        m.stereo_match_device(g, 0, fresh, 0, dk.data_ptr(), dd.data_ptr(), len(k), dk.data_ptr(), dd.data_ptr(), len(k), 40.0,
                              0.08, ur.data_ptr(), dep.data_ptr())


@pytest.mark.parametrize("k,L,irregular,levelsup", [(10, 4, False, 2), (10, 3, False, 4), (7, 5, True, 3), (2, 8, True, 0)])
def test_bow_descend_matches_oracle(k, L, irregular, levelsup):
    """N2: the DBoW2 tree descent (TemplatedVocabulary.h:1218-1259) per feature; leaf and level node bit-exact."""
    co, ci, nd, wi, w = synth.random_vocabulary(k, L, 1000 + k * L, irregular=irregular)
    n = 5000
    f = synth.random_descriptors(n, 55 + k)
    f[:500] = nd[(np.arange(500) * 7) % len(nd)]              # features that sit exactly on nodes (ties between duplicates)
    ref_leaf, ref_nid = O.bow_descend(co, ci, nd, L, f, levelsup)
    voc = capi.Vocabulary(co, ci, nd, L)
    m = capi.ORBmatcher()
    df = _dev(f)
    leaf = torch.zeros(n, dtype=torch.int32, device="cuda")
    nid = torch.zeros(n, dtype=torch.int32, device="cuda")
    m.bow_transform_device(voc, df.data_ptr(), n, levelsup, leaf.data_ptr(), nid.data_ptr())
    assert np.array_equal(leaf.cpu().numpy(), ref_leaf)
    assert np.array_equal(nid.cpu().numpy(), ref_nid)
    assert len(np.unique(ref_leaf)) > 20                       # the features really spread over the tree


def test_bow_vocab_argument_errors():
    co, ci, nd, wi, w = synth.random_vocabulary(3, 2, 5)
    bad = ci.copy()
    bad[0] = 0                                                 # a child pointing back at the root: walks would not end
    with pytest.raises(capi.OrbfeError):
        capi.Vocabulary(co, bad, nd, 2)
    voc = capi.Vocabulary(co, ci, nd, 2)
    m = capi.ORBmatcher()
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    m.bow_transform_device(voc, 0, 0, 1, out.data_ptr(), out.data_ptr())   # no features: nothing to do
    with pytest.raises(capi.OrbfeError):
        m.bow_transform_device(voc, 0, 4, 1, out.data_ptr(), out.data_ptr())


def test_distinctive_descriptors_match_oracle():
    """N4: batched MapPoint::ComputeDistinctiveDescriptors; includes empty points, single observations, > 64 observations."""
    rng = np.random.default_rng(17)
    sizes = np.concatenate([rng.integers(0, 30, 3000), [64, 65, 130, 1, 0, 2]])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = synth.random_descriptors(int(off[-1]), 21)
    # observations of one map point resemble each other: copies of a base descriptor with a few bits flipped, some identical
    for p in range(0, len(sizes), 2):
        if sizes[p] > 1:
            base = desc[off[p]].copy()
            for i in range(off[p], off[p + 1]):
                d = base.copy()
                for b in rng.integers(0, 256, int(rng.integers(0, 25))):
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                desc[i] = d
    ref = O.distinctive(desc, off)
    m = capi.ORBmatcher()
    dd, do = _dev(desc), _dev(off)
    best = torch.full((len(sizes),), -7, dtype=torch.int32, device="cuda")
    m.distinctive_device(dd.data_ptr(), do.data_ptr(), len(sizes), int(sizes.max()), best.data_ptr())
    assert np.array_equal(best.cpu().numpy(), ref)
    assert (ref == -1).sum() == (sizes == 0).sum()


def test_search_area_matches_oracle():
    """Rest of N4: device frame grid + windowed best/second search == GetFeaturesInArea + the SearchByProjection loop."""
    W, H = 1241, 376
    img = synth.textured_image(W, H, synth.BASE_SEED + 123)
    k, d = capi.ORBextractor(2000, 1.2, 8, 20, 7)(img)
    n = len(k)
    rng = np.random.default_rng(9)
    kx, ky, koct = k["x"].copy(), k["y"].copy(), k["octave"].astype(np.int32)
    ur = np.where(rng.uniform(size=n) < 0.5, kx - rng.uniform(1, 60, n), -1).astype(np.float32)
    mn_x, mn_y = 0.0, 0.0
    w_inv, h_inv = float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H))
    og = O.OracleGrid(kx, ky, koct, d, ur, mn_x, mn_y, w_inv, h_inv)
    gg = capi.FrameGrid(kx, ky, koct, d, ur, mn_x, mn_y, w_inv, h_inv)
    m = capi.ORBmatcher()
    nq = 3000
    q = np.zeros(nq, capi.AREA_QUERY_DTYPE)
    src = rng.integers(0, n, nq)
    q["x"] = kx[src] + rng.uniform(-6, 6, nq)                   # projections near real keypoints (plus some far off)
    q["y"] = ky[src] + rng.uniform(-6, 6, nq)
    q["x"][:50] = rng.uniform(-200, W + 200, 50)
    q["y"][:50] = rng.uniform(-200, H + 200, 50)
    lvl = koct[src]
    q["r"] = (rng.choice([2.5, 4.0], nq) * (1.2 ** lvl) * rng.choice([1.0, 3.0], nq)).astype(np.float32)
    q["min_level"], q["max_level"] = lvl - 1, lvl
    q["min_level"][100:200], q["max_level"][100:200] = -1, -1   # no level check
    q["min_level"][200:300], q["max_level"][200:300] = 2, -1    # lower bound only
    q["proj_xr"] = q["x"] - rng.uniform(1, 60, nq)
    q["er_max"] = np.where(rng.uniform(size=nq) < 0.5, q["r"], -1).astype(np.float32)
    qd = synth.planted_queries(d, nq, 77, planted_frac=0.0)
    qd[:] = d[src]
    flip = rng.integers(0, 256, (nq, 12))
    for i in range(nq):
        for b in flip[i]:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    blocked = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    for use_blocked in (False, True):
        res = gg.search(m, q, qd, blocked if use_blocked else None)
        for i in range(nq):
            ref = og.search(float(q["x"][i]), float(q["y"][i]), float(q["r"][i]), int(q["min_level"][i]), int(q["max_level"][i]),
                            float(q["proj_xr"][i]), float(q["er_max"][i]), qd[i], blocked if use_blocked else None)
            got = [res[f][i] for f in ("best_dist", "best_idx", "best_level", "best_dist2", "best_level2", "n_candidates", "best_idx2")]
            assert got == ref.tolist(), (i, got, ref.tolist())
    # the case is not vacuous: most queries find something even with 20 % of the keypoints blocked and the stereo
    # gate on half of them, and a good share of the windows hold several candidates
    assert (res["best_idx"] >= 0).mean() > 0.5 and (res["n_candidates"] > 3).mean() > 0.15


def _bow_case(rng, n1, n2, nn, noise_bits=40, shuffle=True):
    """Two feature sets sharing vocabulary nodes: side 1 = noisy copies of side-2 features (several may want the same one)."""
    d2 = synth.random_descriptors(n2, int(rng.integers(1, 1 << 30)))
    src = rng.integers(0, n2, n1)
    d1 = d2[src].copy()
    for i in range(n1):
        for b in rng.integers(0, 256, int(rng.integers(0, noise_bits))):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    node2 = rng.integers(0, nn, n2)
    node1 = node2[src].copy()
    wrong = rng.uniform(size=n1) < 0.1
    node1[wrong] = rng.integers(0, nn, int(wrong.sum()))

    def csr(node):
        order = [np.nonzero(node == k)[0] for k in range(nn)]
        if shuffle:
            for o in order:
                rng.shuffle(o)
        return (np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32),
                np.concatenate(order).astype(np.int32))
    return d1, d2, csr(node1), csr(node2)


@pytest.mark.parametrize("n1,n2,nn", [(1500, 2000, 100), (300, 5000, 3), (2000, 2000, 1), (50, 40, 200), (1000, 1200, 1000)])
def test_search_by_bow_matches_oracle(n1, n2, nn):
    """a-11: the SearchByBoW loops (src/ORBmatcher.cc:180-264, :550-632) on the device, one wavefront per shared node;
    includes nodes with more than 64 side-2 features, a single node holding everything, empty nodes, both threshold forms."""
    rng = np.random.default_rng(n1 + n2 + nn)
    d1, d2, (off1, idx1), (off2, idx2) = _bow_case(rng, n1, n2, nn)
    valid1 = (rng.uniform(size=n1) < 0.85).astype(np.uint8)
    blocked2 = (rng.uniform(size=n2) < 0.1).astype(np.uint8)
    m = capi.ORBmatcher()
    total = 0
    for v1, b2, md, ratio in ((valid1, None, 50, 0.7), (valid1, blocked2, 49, 0.75), (None, None, 100, 0.9), (None, blocked2, 50, 0.6)):
        got = m.search_by_bow(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        ref = O.search_by_bow(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        assert np.array_equal(got, ref), (md, ratio)
        total += int((ref >= 0).sum())
    assert total > 20


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("ORBFE_MATCH_FUZZ_CASES", "12"))))
def test_matcher_fuzz(case):
    """Seeded random shapes for the other matching entry points, in turn: SearchByBoW node lists, candidate-list argmin,
    DBoW2 descent over random trees, and the full stereo matcher on a random pair of random size.  Bit-exact against the
    oracle every time."""
    rng = np.random.default_rng(8000 + case)
    m = capi.ORBmatcher()
    which = case % 4
    if which == 0:
        n1, n2, nn = int(rng.integers(1, 2500)), int(rng.integers(1, 5000)), int(np.exp(rng.uniform(0, np.log(2000))))
        d1, d2, (off1, idx1), (off2, idx2) = _bow_case(rng, n1, n2, nn, noise_bits=int(rng.integers(1, 90)), shuffle=bool(case & 4))
        v1 = (rng.uniform(size=n1) < 0.85).astype(np.uint8) if case & 8 else None
        b2 = (rng.uniform(size=n2) < 0.1).astype(np.uint8) if case & 16 else None
        md, ratio = int(rng.integers(20, 120)), float(rng.choice([0.6, 0.7, 0.75, 0.9, 1.0]))
        got = m.search_by_bow(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        ref = O.search_by_bow(d1, v1, d2, b2, off1, idx1, off2, idx2, md, ratio)
        assert np.array_equal(got, ref), (case, n1, n2, nn, md, ratio)
    elif which == 1:
        nr, nq = int(rng.integers(1, 6000)), int(rng.integers(1, 4000))
        r = synth.random_descriptors(nr, 8100 + case)
        q = synth.planted_queries(r, nq, 8200 + case, planted_frac=0.5 if nq >= 10 else 0.0)
        counts = rng.integers(0, int(rng.integers(1, 200)), nq)
        counts[rng.uniform(size=nq) < 0.1] = 0
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        idx = rng.integers(0, nr, off[-1]).astype(np.int32)
        dq, dr, doff, didx = _dev(q), _dev(r), _dev(off), _dev(idx if len(idx) else np.zeros(1, np.int32))
        bd, bi, sd = (torch.zeros(nq, dtype=torch.int32, device="cuda") for _ in range(3))
        m.csr_device(dq.data_ptr(), nq, dr.data_ptr(), nr, doff.data_ptr(), didx.data_ptr(), bd.data_ptr(), bi.data_ptr(), sd.data_ptr())
        ref = O.match_csr(q, r, off, idx)
        for g, rr, name in zip((bd, bi, sd), ref, ("best_dist", "best_idx", "second_dist")):
            assert np.array_equal(g.cpu().numpy(), rr), (case, nq, nr, name)
    elif which == 2:
        k, L = int(rng.integers(2, 11)), int(rng.integers(1, 6))
        while k ** L > 60000:
            L -= 1
        levelsup = int(rng.integers(0, L + 2))
        co, ci, nd, _wi, _w = synth.random_vocabulary(k, L, 8300 + case, irregular=bool(case & 4))
        n = int(rng.integers(1, 4000))
        f = synth.random_descriptors(n, 8400 + case)
        f[: n // 5] = nd[rng.integers(0, len(nd), n // 5)]            # features that sit exactly on nodes
        ref_leaf, ref_nid = O.bow_descend(co, ci, nd, L, f, levelsup)
        voc = capi.Vocabulary(co, ci, nd, L)
        df = _dev(f)
        leaf = torch.zeros(n, dtype=torch.int32, device="cuda")
        nid = torch.zeros(n, dtype=torch.int32, device="cuda")
        m.bow_transform_device(voc, df.data_ptr(), n, levelsup, leaf.data_ptr(), nid.data_ptr())
        assert np.array_equal(leaf.cpu().numpy(), ref_leaf), (case, k, L, levelsup, n)
        assert np.array_equal(nid.cpu().numpy(), ref_nid), (case, k, L, levelsup, n)
    else:
        W, H = int(rng.integers(320, 1300)), int(rng.integers(240, 500))
        nfeat = int(rng.choice([300, 1000, 2000]))
        left, right = synth.stereo_pair(W, H, synth.BASE_SEED + 3000 + case)
        ol, orr = O.OracleExtractor(nfeat, 1.2, 8, 20, 7), O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        kol, dol = ol.extract(left)
        kor, dor = orr.extract(right)
        bf = float(rng.uniform(30.0, 400.0))
        b = float(rng.uniform(0.05, 0.6))
        ref_ur, ref_dep = O.stereo_full(ol, orr, kol, dol, kor, dor, bf, b)
        gl, gr = capi.ORBextractor(nfeat, 1.2, 8, 20, 7), capi.ORBextractor(nfeat, 1.2, 8, 20, 7)
        kl, dl = gl(left)
        kr, dr = gr(right)
        assert np.array_equal(kl, kol) and np.array_equal(dl, dol) and np.array_equal(kr, kor) and np.array_equal(dr, dor)
        dkl, dkr = _dev(kl.view(np.uint8).reshape(-1, 28)), _dev(kr.view(np.uint8).reshape(-1, 28))
        ddl, ddr = _dev(dl), _dev(dr)
        ur = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
        dep = torch.zeros(len(kl), dtype=torch.float32, device="cuda")
        m.stereo_match_device(gl, 0, gr, 0, dkl.data_ptr(), ddl.data_ptr(), len(kl), dkr.data_ptr(), ddr.data_ptr(), len(kr),
                              bf, b, ur.data_ptr(), dep.data_ptr())
        assert np.array_equal(ur.cpu().numpy().view(np.uint32), ref_ur.view(np.uint32)), (case, W, H, nfeat, bf, b)
        assert np.array_equal(dep.cpu().numpy().view(np.uint32), ref_dep.view(np.uint32)), (case, W, H, nfeat, bf, b)


def test_search_by_bow_argument_errors():
    m = capi.ORBmatcher()
    d = synth.random_descriptors(4, 1)
    ok_off, ok_idx = np.array([0, 4], np.int32), np.arange(4, dtype=np.int32)
    assert (m.search_by_bow(d, None, d[:0], None, ok_off, ok_idx, np.array([0, 0], np.int32), ok_idx[:0], 50, 0.7) == -1).all()
    with pytest.raises(capi.OrbfeError):       # an index past the descriptor array must be refused on the host, not fault on the GPU
        m.search_by_bow(d, None, d, None, ok_off, np.array([0, 1, 2, 4], np.int32), ok_off, ok_idx, 50, 0.7)
    with pytest.raises(capi.OrbfeError):
        m.search_by_bow(d, None, d, None, np.array([0, 5], np.int32), ok_idx, ok_off, ok_idx, 50, 0.7)
    with pytest.raises(capi.OrbfeError):
        m.search_by_bow(d, None, d, None, np.array([1, 4], np.int32), ok_idx, ok_off, ok_idx, 50, 0.7)
