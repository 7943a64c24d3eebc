"""Wire format (bincode layout of the reference's Mat) <-> dense slabs: host code, runs without a GPU.

Pinned by the reference's own serialisation test (src/mat.rs:424-438): a 1x1 Mat holding 1 + 2x + 3x^2
over i32 is 36 bytes (8 rows + 8 cols + 8 len + 3*4)."""
import struct

import numpy as np
import pytest

from ring_zk_amd import wire

N = 4  # src/mat.rs:241


def test_reference_serde_vector():
    slab = np.zeros((1, 1, N), dtype=np.int64)
    slab[0, 0, :3] = [1, 2, 3]
    data = wire.mat_encode(slab, coef_bytes=4)
    assert len(data) == 36                                   # src/mat.rs:431-434
    assert data == struct.pack("<QQQiii", 1, 1, 3, 1, 2, 3)  # bincode: LE, u64 lengths, fields in order
    back, used = wire.mat_decode(data, N, coef_bytes=4)
    assert used == 36 and np.array_equal(back, slab)         # src/mat.rs:436-437
    data8 = wire.mat_encode(slab, coef_bytes=8)
    assert data8 == struct.pack("<QQQqqq", 1, 1, 3, 1, 2, 3)


@pytest.mark.parametrize("coef_bytes", [4, 8])
def test_round_trip_trimmed_polynomials(coef_bytes):
    rng = np.random.default_rng(5)
    n_ring = 16
    lim = 2 ** 31 - 1 if coef_bytes == 4 else (3515337053 - 1) // 2
    slab = rng.integers(-lim, lim + 1, (3, 2, n_ring), dtype=np.int64)
    slab[0, 0, 5:] = 0          # degree 4 -> 5 coefficients on the wire
    slab[1, 1, :] = 0           # zero polynomial -> empty coefficient list
    slab[2, 0, -1] = -1         # full length
    data = wire.mat_encode(slab, coef_bytes)
    expect = 8 + 3 * 8 + sum(8 + coef_bytes * len(np.trim_zeros(slab[r, c], "b")) for r in range(3) for c in range(2))
    assert len(data) == expect
    back, used = wire.mat_decode(data, n_ring, coef_bytes)
    assert used == len(data) and np.array_equal(back, slab)


def test_message_is_concatenation_of_fields():
    """OpenProofCommitment { c: Commitment { c: Mat }, t: Mat } (src/prove/open.rs:181-187, src/commit.rs:134-137):
    fields back to back, decoded one after the other."""
    rng = np.random.default_rng(6)
    n_ring = 16
    c = rng.integers(-100, 100, (2, 1, n_ring), dtype=np.int64)
    t = rng.integers(-100, 100, (1, 1, n_ring), dtype=np.int64)
    msg = wire.mat_encode(c) + wire.mat_encode(t)
    c2, used = wire.mat_decode(msg, n_ring)
    t2, used2 = wire.mat_decode(msg[used:], n_ring)
    assert used + used2 == len(msg) and np.array_equal(c2, c) and np.array_equal(t2, t)


def test_malformed_inputs_are_rejected():
    slab = np.arange(2 * 2 * N, dtype=np.int64).reshape(2, 2, N) + 1
    good = wire.mat_encode(slab)
    with pytest.raises(ValueError):
        wire.mat_decode(good[:-1], N)                         # truncated
    with pytest.raises(ValueError):
        wire.mat_decode(good, N - 1)                          # polynomial longer than the ring degree
    ragged = struct.pack("<Q", 2) + struct.pack("<QQq", 1, 1, 7) + struct.pack("<QQqQq", 2, 1, 7, 1, 8)
    with pytest.raises(ValueError):
        wire.mat_decode(ragged, N)                            # rows of different width are not a matrix
    with pytest.raises(ValueError):
        wire.mat_encode(np.full((1, 1, N), 2 ** 40, dtype=np.int64), coef_bytes=4)   # does not fit i32
    empty, used = wire.mat_decode(struct.pack("<Q", 0), N)
    assert empty.shape == (0, 0, N) and used == 8


def test_decode_range_checks_coefficients_when_given_the_modulus():
    """A ZqI64 on the wire is its centred representative (src/params.rs:122-127); with q the codec rejects anything
    else — in particular v + k*2^32, whose low word is an honest value (the verifier kernels test it again)."""
    q = 3515337053
    half = (q - 1) // 2
    n_ring = 8
    slab = np.zeros((2, 1, n_ring), dtype=np.int64)
    slab[0, 0, :3] = [half, -half, 7]
    slab[1, 0, :2] = [-1, 1]
    data = wire.mat_encode(slab)
    back, used = wire.mat_decode(data, n_ring, q=q)
    assert used == len(data) and np.array_equal(back, slab)
    for bad in (half + 1, -half - 1, 7 + (1 << 32), -(1 << 32), 1 << 62):
        evil = slab.copy()
        evil[1, 0, 1] = bad
        enc = wire.mat_encode(evil)
        with pytest.raises(ValueError):
            wire.mat_decode(enc, n_ring, q=q)
        plain, _ = wire.mat_decode(enc, n_ring)             # q = 0: plain integers (the reference's i32 test ring)
        assert plain[1, 0, 1] == bad
