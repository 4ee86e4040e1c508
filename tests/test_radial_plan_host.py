"""CPU: the host tables of the radial-MRI plan (step_amd.forward_operator.radial_mri.kb_plan) that the
lane-per-chain spread, the fused forward and the skipping adjoint read -- against brute force on a small
grid (R = 32, G = 64), in the kernels' own arithmetic (fp32 product of the weights, fp32 fma, list order)."""
import numpy as np
import pytest

from step_amd.forward_operator.radial_mri import golden_angle_radial, kb_plan


@pytest.fixture(scope="module")
def plan():
    R = 32
    traj = golden_angle_radial(3, 7, R)
    rng = np.random.RandomState(0)
    extra = rng.uniform(-R / 2, R / 2, (3, 40, 2))
    extra[:, :3] = [[-R / 2, -R / 2], [R / 2, R / 2], [0, 0]]              # footprints that wrap
    return kb_plan(np.concatenate([traj, extra], axis=1), R, 6)


def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def test_wide_lists_spread_bit_for_bit(plan):
    G, W, J = plan["G"], plan["W"], plan["J"]
    ntx = G // 16
    rng = np.random.RandomState(1)
    for t in range(plan["traj_frames"]):
        v = rng.standard_normal(J).astype(np.float32)
        ref = np.zeros((G, G), np.float32)
        for j in range(J):                                                 # ascending j: the order of every list
            w = plan["wy"][t, j][:, None] * plan["wx"][t, j][None, :]      # fp32 x fp32 -> fp32
            for a in range(W):
                for b in range(W):
                    y, x = (plan["iy0"][t, j] + a) % G, (plan["ix0"][t, j] + b) % G
                    ref[y, x] = fma32(w[a, b], v[j], ref[y, x])
        out = np.full((G, G), np.nan, np.float32)
        rec = plan["wide_rec"][t]
        assert sorted(rec[:, 0]) == list(range(ntx * (G // 2))) and (rec[:, 2] % 4 == 0).all()
        for tile, e0, n, outside, *rest in rec:
            blk = np.zeros((2, 16), np.float32)
            ids = plan["wide_ent"][e0:e0 + n]
            assert (np.array(rest[:8])[:min(n, 8)] == ids[:8]).all() and (np.array(rest[:8])[n:] == 0).all()
            assert (np.array(rest[:8]) >= 0).all() and (np.array(rest[:8]) < J).all() and rest[8:] == [0, 0, 0, 0]
            for e in range(e0, e0 + n):
                blk = fma32(plan["wide_w"][e], v[plan["wide_ent"][e]], blk)
            ty, tx = divmod(tile, ntx)
            out[2 * ty:2 * ty + 2, 16 * tx:16 * tx + 16] = blk
            if outside:
                assert n == 0
        assert np.array_equal(out, ref)


def test_outside_flags_and_ranges(plan):
    G = plan["G"]
    ntx, nty = G // 16, G // 2
    for t in range(plan["traj_frames"]):
        rec, rng = plan["wide_rec"][t], plan["wide_range"][t]
        cnt = np.zeros(ntx * nty, np.int64)
        cnt[rec[:, 0]] = rec[:, 2]
        outside = np.zeros(ntx * nty, bool)
        outside[rec[:, 0]] = rec[:, 3] == 1
        for tile in range(ntx * nty):
            ty, tx = divmod(tile, ntx)
            cr = ty if ty < nty // 2 else ty - nty
            inside = rng[tx, 0] <= cr <= rng[tx, 1]
            assert outside[tile] == (not inside)
            assert inside or cnt[tile] == 0                                # nothing is lost by skipping
        first_outside = np.flatnonzero(rec[:, 3] == 1)
        if len(first_outside):                                             # launch order: inside (longest first), then outside
            k = first_outside[0]
            assert (rec[k:, 3] == 1).all() and (np.diff(rec[:k, 2]) <= 0).all()


def test_strip_and_owner_bins_hold_every_sample_once(plan):
    G, J = plan["G"], plan["J"]
    for t in range(plan["traj_frames"]):
        for ent, ptr, x0, width in ((plan["strip_ent"], plan["strip_ptr"], plan["strip_x0"], 16),
                                    (plan["own_ent"], plan["own_ptr"], plan["own_x0"], 64)):
            if ent is None:
                continue
            assert sorted(ent[t]) == list(range(J))
            assert (x0[t] == plan["ix0"][t][ent[t]] % G).all()
        ptr, x0 = plan["strip_ptr"][t], plan["strip_x0"][t]
        assert ptr[0] == 0 and ptr[-1] == J
        for s in range(G // 16):
            seg = x0[ptr[s]:ptr[s + 1]]
            assert ((seg >= 16 * s) & (seg < 16 * s + 16)).all()
            assert (np.diff(plan["strip_ent"][t][ptr[s]:ptr[s + 1]]) > 0).all()
        assert np.array_equal(plan["strip_wy"][t], plan["wy"][t][plan["strip_ent"][t]])
