import numpy as np


def test_generator_is_deterministic_and_well_formed(sc):
    a = sc.make_candidate(11, 16, 4, perturb_id=2, obstacles=True)
    b = sc.make_candidate(11, 16, 4, perturb_id=2, obstacles=True)
    assert all(np.array_equal(x, y) for x, y in zip(a.h_polys, b.h_polys))
    assert all(np.array_equal(x, y) for x, y in zip(a.v_polys, b.v_polys))
    assert len(a.h_polys) == 16 and len(a.v_polys) == 31
    assert {h.shape[1] for h in a.h_polys} - set(range(8, 15)) == set()
    for i, v in enumerate(a.v_polys):                    # every vertex satisfies every half-space of its cell(s)
        hs = [a.h_polys[i // 2]] + ([a.h_polys[i // 2 + 1]] if i % 2 else [])
        for h in hs:
            n = h[:3] / np.linalg.norm(h[:3], axis=0)
            sd = np.einsum("dk,dkv->kv", n, v[:, None, :] - h[3:, :, None])
            assert sd.max() < 1e-7
        kq = np.round(v / 1e-7)                             # lexicographic order of the grid keys, v0 first
        assert np.all(np.lexsort((kq[2], kq[1], kq[0])) == np.arange(v.shape[1]))


def test_perturbed_candidates_differ_and_nominal_is_shared(sc):
    batch = sc.make_batch(0, 3, 8, 2)
    assert np.array_equal(batch[0].gates, sc.make_candidate(0, 8, 2).gates)
    assert not np.array_equal(batch[1].gates, batch[0].gates)
    assert 0.05 < np.abs(batch[1].gates - batch[0].gates).max() < 3.0


def test_splitmix_reference_values(sc):
    r = sc.SplitMix64(0)
    assert [r.next_u64() for _ in range(2)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4]   # published SplitMix64 vectors
