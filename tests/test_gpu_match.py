"""GPU parity: HIP Hamming kernels vs the CPU oracle, bit-exact (indices and distances)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def _planted(rng, base, flip_p=0.08):
    """copies of `base` rows with Binomial(256, flip_p) flipped bits"""
    bits = np.unpackbits(base, axis=1)
    flips = rng.random(bits.shape) < flip_p
    return np.packbits(bits ^ flips, axis=1)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 500), (30, 500), (45, 500), (64, 500), (65, 512), (100, 513),
                                   (500, 500), (500, 37), (777, 1300), (4096, 64)])
def test_match_mutual_random(engine, oracle, nq, nt):
    rng = np.random.default_rng(1000 + nq * 7 + nt)
    q, t = _desc(rng, nq), _desc(rng, nt)
    got = engine.match_mutual(q, t)
    exp = oracle.match_mutual(q, t)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


def test_match_mutual_planted_and_ties(engine, oracle):
    rng = np.random.default_rng(7)
    t = _desc(rng, 500)
    q = _planted(rng, t[rng.choice(500, 64, replace=False)])
    # exact duplicates on both sides force the lowest-index tie rule
    q[10] = q[3]; t[400] = t[17]; t[401] = t[17]
    got = engine.match_mutual(q, t)
    exp = oracle.match_mutual(q, t)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    assert len(got[0]) > 40


def test_match_mutual_empty(engine):
    z = np.zeros((0, 32), np.uint8)
    d = np.zeros((5, 32), np.uint8)
    assert len(engine.match_mutual(z, d)[0]) == 0
    assert len(engine.match_mutual(d, z)[0]) == 0


def test_match_bad_input_raises(engine):
    from nclt_slam_project_amd import RelocError
    with pytest.raises(RelocError):
        engine.match_mutual(np.zeros((4, 31), np.uint8), np.zeros((4, 32), np.uint8))
    with pytest.raises(RelocError):
        engine.match_mutual(np.zeros((4, 32), np.float32), np.zeros((4, 32), np.uint8))


@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 2), (500, 64), (64, 500), (300, 3000), (1000, 70000)])
def test_knn2(engine, oracle, nq, nt):
    rng = np.random.default_rng(nq + nt)
    q, t = _desc(rng, nq), _desc(rng, nt)
    if nt > 10:
        t[5] = t[2]  # tie
    gi, gd = engine.match_knn2(q, t)
    ei, ed = oracle.match_knn2(q, t)
    np.testing.assert_array_equal(gd, ed)
    np.testing.assert_array_equal(gi, ei)


@pytest.mark.parametrize("L,nmode,Q", [(50, "fixed64", 500), (300, "ragged", 500), (40, "ragged", 37),
                                       (20, "ragged", 1200), (1000, "fixed64", 500)])
def test_db_match_counts(engine, oracle, L, nmode, Q):
    rng = np.random.default_rng(L + Q)
    if nmode == "fixed64":
        n = np.full(L, 64)
    else:
        n = np.clip(np.rint(rng.normal(60, 25, L)), 30, 500).astype(int)
        n[0] = 0  # an empty record
        n[1] = 1
    off = np.zeros(L + 1, np.int64); off[1:] = np.cumsum(n)
    cur = _desc(rng, Q)
    db = _desc(rng, int(off[-1]))
    # plant true matches in a few records
    for r in rng.choice(L, min(L, 8), replace=False):
        k = int(n[r])
        if k == 0:
            continue
        src = rng.choice(Q, min(k, Q), replace=False)
        db[off[r]:off[r] + len(src)] = _planted(rng, cur[src])
    engine.db_upload(db, np.zeros((len(db), 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
    got = engine.db_match_counts(cur)
    exp = oracle.db_match_counts(db, off, cur)
    np.testing.assert_array_equal(got, exp)
    assert got.max() >= 20


@pytest.mark.parametrize("na,nb", [(1, 8), (3, 5), (130, 2048), (257, 2056), (64, 1001), (1000, 4096)])
def test_hamming_matrix(engine, oracle, na, nb):
    rng = np.random.default_rng(na * 31 + nb)
    a, b = _desc(rng, na), _desc(rng, nb)
    got = engine.hamming_matrix(a, b)
    exp = oracle.hamming_matrix(a, b)
    np.testing.assert_array_equal(got, exp)
    # closed-form known answer on a corner of the matrix
    ref = (np.unpackbits(a[:2, None, :] ^ b[None, :3, :], axis=2).sum(axis=2)).astype(np.uint16)
    np.testing.assert_array_equal(got[:2, :3], ref)


def test_hamming_matrix_properties_full_size(engine):
    """Size-independent properties at a large shape the oracle would take minutes on:
    symmetry d(a,b) = d(b,a)^T, zero diagonal of d(a,a), and row sums equal to a bit-count identity."""
    rng = np.random.default_rng(5)
    a = _desc(rng, 6000)
    d = engine.hamming_matrix(a, a)
    assert (np.diag(d) == 0).all()
    np.testing.assert_array_equal(d, d.T)
    # sum_j d(i,j) = sum_bits [ a_ib ? (N - c_b) : c_b ], c_b = column bit counts
    bits = np.unpackbits(a, axis=1).astype(np.int64)
    c = bits.sum(axis=0)
    exp = (bits * (len(a) - c) + (1 - bits) * c).sum(axis=1)
    np.testing.assert_array_equal(d.astype(np.int64).sum(axis=1), exp)
