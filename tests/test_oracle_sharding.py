"""The sharding oracle against what pins it (CPU): the xxHash specification's known answers for XXH64, the independent
`xxhash` Python package (a build of the same library the reference vendors, 0.8.x) on random inputs of every length
around the 32-byte stripe and the 8/4/1-byte tails, and the reference's own streaming-equals-one-shot property
(ydb/core/tx/sharding/ut/ut_sharding.cpp:14-31) in the form the path uses it: fields concatenated per row.
The list split has no known-answer test in the reference (parity unpinned); its invariants are checked here."""
import numpy as np
import pytest

from oracle import sharding_oracle as so

KNOWN = [  # (input, seed, XXH64) -- published test vectors of the algorithm
    (b"", 0, 0xEF46DB3751D8E999),
    (b"a", 0, 0xD24EC4F1A98C6E5B),
    (b"abc", 0, 0x44BC2CF5AD770999),
    (b"Nobody inspects the spammish repetition", 0, 0xFBCEA83C8A378BF1),
]


@pytest.mark.parametrize("data,seed,expect", KNOWN)
def test_xxh64_known_answers(data, seed, expect):
    assert so.xxh64(data, seed) == expect


def test_xxh64_matches_the_xxhash_package():
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(1)
    for n in list(range(0, 100)) + [127, 128, 129, 255, 256, 1000, 4099]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, so.COMPUTE_SHARDING_SEED, 2**64 - 1):
            assert so.xxh64(data, seed) == xxhash.xxh64_intdigest(data, seed), (n, seed)


def test_row_hash_is_the_hash_of_the_concatenated_fields():
    xxhash = pytest.importorskip("xxhash")
    # ut_hash.cpp:44-60 (ScalarCompositeHash): "abcde" then UInt32 52
    cols = [(["abcde"], None), (np.array([52], dtype=np.uint32), None)]
    assert so.hash_rows(cols, 1, 0)[0] == xxhash.xxh64_intdigest(b"abcde" + np.uint32(52).tobytes(), 0)
    # a null field adds nothing (calcer.cpp:37)
    cols = [(["abcde"], np.array([False])), (np.array([52], dtype=np.uint32), None)]
    assert so.hash_rows(cols, 1, 0)[0] == xxhash.xxh64_intdigest(np.uint32(52).tobytes(), 0)


@pytest.mark.parametrize("n,shards", [(0, 3), (1, 2), (10, 3), (1000, 7), (5000, 64), (777, 512)])
def test_split_invariants(n, shards):
    rng = np.random.default_rng(n + shards)
    h = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    perm, sizes = so.split(h, shards)
    assert sum(sizes) == n and sorted(perm.tolist()) == list(range(n))
    at = 0
    for s, k in enumerate(sizes):
        part = perm[at:at + k]
        at += k
        assert np.all(np.diff(part.astype(np.int64)) > 0)  # ascending row order inside a shard
    # without the size correction every row sits in |hash| % shards; the correction only ever moves rows out of lists
    # longer than the expectation
    expectation = n // shards + 1
    natural = np.abs(h) % shards
    at = 0
    for s, k in enumerate(sizes):
        part = perm[at:at + k]
        at += k
        moved_in = part[natural[part] != s]
        if len(moved_in):
            assert k <= expectation
            assert all(np.sum(natural == d) > expectation for d in set(natural[moved_in].tolist()))


def test_split_moves_the_tail_of_the_longest_list():
    # 10 rows, 3 shards, expectation 4: shard 0 gets 7 rows (0..6), shard 1 gets 3, shard 2 gets 0
    h = np.array([0, 0, 0, 0, 0, 0, 0, 1, 1, 1], dtype=np.int64) + 3 * np.arange(10)
    perm, sizes = so.split(h, 3)
    # shortest list (shard 2) takes min(need 4, useless 3) = 3 records: the last three of shard 0
    assert sizes == [4, 3, 3]
    assert perm.tolist() == [0, 1, 2, 3, 7, 8, 9, 4, 5, 6]
