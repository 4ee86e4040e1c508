"""The Philox4x32-10 restatement against the published known-answer vectors (Random123's
kat_vectors: philox4x32 10 rounds), and the statistics of the normals built on it."""
import numpy as np

from oracle.philox_oracle import chain_noise, chain_noise_quad, philox4x32_10

KAT = [  # counter, key, expected
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_known_answer_vectors():
    for ctr, key, want in KAT:
        got = philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
        assert tuple(int(v) for v in got) == want


def test_normals_are_standard_and_chains_independent():
    a = chain_noise(37, 0, 1 << 18)
    b = chain_noise(38, 0, 1 << 18)
    c = chain_noise(37, 1, 1 << 18)
    for v in (a, b, c):
        assert abs(v.mean()) < 0.01 and abs(v.std() - 1) < 0.01
        assert abs((v ** 4).mean() - 3) < 0.1
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.01 and abs(np.corrcoef(a, c)[0, 1]) < 0.01
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.01
    assert np.array_equal(a, chain_noise(37, 0, 1 << 18))


def test_quad_layout_is_a_permutation_of_the_linear_draw():
    W, rows = 8, 12
    a = chain_noise(5, 3, W * rows).reshape(-1, 4)               # [call][k]
    q = chain_noise_quad(5, 3, W * rows, W).reshape(rows, W)
    for row in range(rows):
        for col in range(W):
            assert q[row, col] == a[(row // 4) * W + col, row % 4]
