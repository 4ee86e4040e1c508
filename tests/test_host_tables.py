"""CPU: host-side integer tables of libstep_amd.so against the golden fixtures (bit exact) and the
oracle; the ehtim-free set-up (eht_lite, PARITY UNPINNED) against the oracle's dense evaluation."""
import os

import numpy as np
import pytest
import torch

from oracle import eht_synth
from oracle.bh_oracle import closure_tables
from step_amd import _lib, eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch, get_operator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table_from_golden(g):
    times = np.unique(g["row_time"])
    nt, nq = len(g["tri_sites"]) // 64, len(g["quad_sites"]) // 64
    return dict(times=g["row_time"], t1=list(g["row_t1"]), t2=list(g["row_t2"]),
                tri=[(tm,) + tuple(s) for tm, s in zip(np.repeat(times, nt), g["tri_sites"])],
                quad=[(tm,) + tuple(s) for tm, s in zip(np.repeat(times, nq), g["quad_sites"])])


@pytest.mark.parametrize("name", ["bh_small", "bh_allterms", "bh_allterms_norm", "bh_full_n8"])
def test_closure_tables_bit_exact_vs_reference(name, golden):
    g = golden(name)
    cp_i, cp_c, ca_i, ca_c = BlackHoleImagingTorch._closure_tables(table_from_golden(g))
    assert cp_i.dtype == torch.int64
    assert np.array_equal(cp_i.numpy(), g["cp_index"]) and np.array_equal(cp_c.numpy(), g["cp_conjugate"])
    assert np.array_equal(ca_i.numpy(), g["camp_index"]) and np.array_equal(ca_c.numpy(), g["camp_conjugate"])


def test_closure_tables_shuffled_rows_and_duplicates():
    """Dict semantics of blackhole.py:236-241: arbitrary row order, later duplicates win."""
    rng = np.random.RandomState(0)
    syn = eht_synth.synthetic_obs(6, 5, seed=1)
    perm = rng.permutation(len(syn["times"]))
    times = list(syn["times"][perm]) + [syn["times"][3]]
    t1 = [syn["t1"][i] for i in perm] + [syn["t2"][3]]        # duplicate key, stations swapped
    t2 = [syn["t2"][i] for i in perm] + [syn["t1"][3]]
    tab = dict(times=np.array(times), t1=t1, t2=t2, tri=syn["tri"], quad=syn["quad"])
    want = closure_tables(tab["times"], t1, t2, syn["tri"], syn["quad"])
    got = BlackHoleImagingTorch._closure_tables(tab)
    for a, b in zip(got, want):
        assert np.array_equal(a.numpy(), np.array(b))


def test_incidence_tables_are_the_exact_transpose(golden):
    g = golden("bh_full_n8")
    cpi, cpc, cai, cac = (np.ascontiguousarray(g[k]) for k in
                          ("cp_index", "cp_conjugate", "camp_index", "camp_conjugate"))
    n_vis = 64 * 28
    cp_ptr, ca_ptr = np.zeros(n_vis + 1, np.int32), np.zeros(n_vis + 1, np.int32)
    cp_ent, ca_ent = np.zeros(cpi.size, np.int32), np.zeros(cai.size, np.int32)
    rc = _lib.lib().bh_build_incidence(_lib.ptr(cpi), _lib.ptr(cpc), len(cpi), _lib.ptr(cai), _lib.ptr(cac),
                                       len(cai), n_vis, _lib.ptr(cp_ptr), _lib.ptr(cp_ent),
                                       _lib.ptr(ca_ptr), _lib.ptr(ca_ent))
    _lib.check(rc)
    for idx, cj, ptr, ent, legs in ((cpi, cpc, cp_ptr, cp_ent, 3), (cai, cac, ca_ptr, ca_ent, 4)):
        assert ptr[0] == 0 and ptr[-1] == idx.size
        seen = set()
        for m in range(n_vis):
            ks = [e >> 3 for e in ent[ptr[m]:ptr[m + 1]]]
            assert ks == sorted(ks)                              # fixed summation order
            for e in ent[ptr[m]:ptr[m + 1]]:
                k, leg, c = e >> 3, (e >> 1) & 3, e & 1
                assert idx[k, leg] == m and cj[k, leg] == c and leg < legs
                seen.add((k, leg))
        assert len(seen) == idx.size
    bad = cpi.copy(); bad[0, 0] = n_vis
    assert _lib.lib().bh_build_incidence(_lib.ptr(bad), _lib.ptr(cpc), len(cpi), _lib.ptr(cai), _lib.ptr(cac),
                                         len(cai), n_vis, _lib.ptr(cp_ptr), _lib.ptr(cp_ent),
                                         _lib.ptr(ca_ptr), _lib.ptr(ca_ent)) != 0


def test_mri_masks_bit_exact_vs_reference(golden):
    d = golden("mri_masks")
    for key, want in d.items():
        mode, a, c, r, f = key.rsplit("_", 4)
        op = get_operator("mri", acc_factor=int(a[1:]), mode=mode, resolution=int(r[1:]),
                          frames=int(f[1:]), num_acs_lines=int(c[1:]))
        assert op.mask.dtype == torch.float32 and op.mask.dim() == 5
        assert np.array_equal(op.mask[0, :, 0, :, 0].numpy().astype(np.uint8), want), key
    with pytest.raises(ValueError, match="Unknown mode"):
        get_operator("mri", acc_factor=4, mode="spiral")
    with pytest.raises(NameError):
        get_operator("nufft")


def test_mri_mask_random_shuffle_matches_numpy_for_many_sizes():
    """np.random.seed(42) + shuffle restated in C++ (MT19937 + masked rejection)."""
    from oracle.mri_oracle import build_mask
    for res, acc, acs, fr in [(192, 6, 24, 12), (96, 3, 10, 7), (33, 2, 4, 3), (256, 8, 32, 16), (20, 1, 0, 4)]:
        op = get_operator("mri", acc_factor=acc, mode="random_split_mask", resolution=res, frames=fr,
                          num_acs_lines=acs)
        assert torch.equal(op.mask, build_mask(acc, "random_split_mask", res, fr, acs)), (res, acc, acs, fr)


def test_separable_factors_reproduce_dense_ftmatrix():
    rng = np.random.RandomState(4)
    uv = rng.uniform(-8.5e9, 8.5e9, (3, 5, 2))
    uv[0, 0] = 0.0                                              # the pulse's omega = 0 branch
    ey, ex = eht_lite.ft_factors(uv, 24, 40)
    dense = eht_lite.dense_from_factors(ey, ex)
    for i in range(3):
        ref = eht_synth.ftmatrix_dense(eht_synth.PSIZE, 40, 24, uv[i])
        assert np.abs(dense[i] - ref).max() <= 1e-12 * np.abs(ref).max()


def test_minimal_sets_and_synthetic_table():
    tab = eht_lite.synthetic_table(8, 64, seed=37)
    ref = eht_synth.synthetic_obs(8, 64, seed=37)
    assert np.array_equal(tab["uv"], ref["uv"]) and np.array_equal(tab["sigma"], ref["sigma"])
    assert tab["tri"] == ref["tri"] and tab["quad"] == ref["quad"]
    for n in range(4, 10):
        s = ["S%d" % i for i in range(n)]
        assert len(eht_lite.minimal_triangles(s)) == (n - 1) * (n - 2) // 2
        q = eht_lite.minimal_quadrangles(s)
        assert len(q) == n * (n - 3) // 2 and all(len(set(x)) == 4 for x in q)


def test_array_file_synthesis_shapes():
    tab = eht_lite.eht_table(os.path.join(ROOT, "step_amd", "data", "array_8.txt"))
    T, F, _ = tab["uv"].shape
    assert T == 64                                              # 12.5 .. 14.3 h in 102 s steps
    n = round((1 + (1 + 8 * F) ** 0.5) / 2)
    assert n * (n - 1) // 2 == F and 4 <= n <= 8
    assert len(tab["tri"]) == T * (n - 1) * (n - 2) // 2 and len(tab["quad"]) == T * n * (n - 3) // 2
    assert np.all(tab["sigma"] > 0) and np.abs(tab["uv"]).max() < 1e10
    cp_i, cp_c, ca_i, ca_c = BlackHoleImagingTorch._closure_tables(tab)
    assert cp_i.shape == (len(tab["tri"]), 3) and int(cp_i.max()) < T * F
    # frame t's closures only touch frame t's rows (time is part of the key)
    assert torch.equal(cp_i // F, (torch.arange(len(cp_i)) // (len(cp_i) // T))[:, None].expand(-1, 3))


def test_factor_A_vis_recovers_mirror_symmetric_factors():
    """Rank-1 factor extraction from a dense complex64 A_vis (the from_tensors drop-in route):
    reproduces the matrix to fp32 rounding, yields bit-exactly conjugate-symmetric fp32 tables on a
    centred grid, and refuses matrices without the structure."""
    import torch
    from step_amd import eht_lite
    from step_amd.forward_operator.blackhole import factor_A_vis
    tab = eht_lite.synthetic_table(5, 64, seed=3)
    ey, ex = eht_lite.ft_factors(np.asarray(tab["uv"], dtype=np.float64)[:3], 32, 64, eht_lite.PSIZE)
    A = torch.from_numpy(eht_lite.dense_from_factors(ey, ex)).to(torch.complex64)
    gy, gx = factor_A_vis(A, 32, 64)
    rec = (gy[..., :, None] * gx[..., None, :]).reshape(A.shape)
    assert np.abs(rec - A.numpy()).max() <= 3e-7 * np.abs(A.numpy()).max()
    r32, i32 = gy.real.astype(np.float32), gy.imag.astype(np.float32)
    assert np.array_equal(r32[..., ::-1], r32) and np.array_equal(i32[..., ::-1], -i32)
    assert factor_A_vis(torch.randn(2, 3, 32 * 64, dtype=torch.complex64), 32, 64) is None
    assert factor_A_vis(torch.zeros(1, 2, 32 * 64, dtype=torch.complex64), 32, 64) is None
