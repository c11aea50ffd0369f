"""Pins the CPU oracle (oracle/ref_literal.c, oracle.ref_simple_*) against every known-answer vector
the reference's own tests hold for the matcher path (tests/golden/reference_kat.json; sources:
/root/reference/src/lib/{mod,bitenc,barcode_matching}.rs tests, src/bin/commands/demux.rs tests)."""
import numpy as np
import pytest

from oracle import oracle as O


def test_encode_known_answers(kat):
    for c in kat["encode"]["cases"]:
        vals, _ = O.encode(c["base"].encode())
        assert vals == [c["mask"]], c


def test_encode_all_256_bytes_match_spec_table():
    # enc(): N/n/. -> 15; else LUT of upper-cased byte; any other byte -> 0 (mod.rs:49-61)
    for b in range(256):
        vals, _ = O.encode(bytes([b]))
        assert vals[0] == int(O.ENC[b]), b
    assert O.encode(b"X-0@[`{")[0] == [0] * 7


def test_encode_block_layout_little_endian_nibbles():
    # bitenc.rs:319-322: symbol i at bits [4*(i%8), +4) of block i/8; tail nibbles zero
    vals, blocks = O.encode(b"ACGTACGTN")
    assert vals == [1, 2, 4, 8, 1, 2, 4, 8, 15]
    assert blocks == [0x84218421, 0xF]


def test_nocall_and_valid_iupac(kat):
    for ch in kat["nocall"]["true"]:
        assert O.lib().oracle_byte_is_nocall(ord(ch))
    for ch in kat["nocall"]["false"]:
        assert not O.lib().oracle_byte_is_nocall(ord(ch))
    for ch in kat["valid_iupac"]["true"]:
        assert O.is_valid_iupac(ord(ch))
    for ch in kat["valid_iupac"]["false"]:
        assert not O.is_valid_iupac(ord(ch))


def test_hamming_known_answers(kat):
    v = kat["hamming"]["vectors"]
    for c in kat["hamming"]["cases"]:
        assert O.hamming_vals(v[c["self"]], v[c["other"]], c["max"]) == c["expect"], c


def test_count_mismatches_known_answers(kat):
    for c in kat["count_mismatches"]["cases"]:
        assert O.count_mismatches(c["observed"].encode(), c["expected"].encode()) == c["expect"], c
    for c in kat["count_mismatches"]["length_errors"]:
        with pytest.raises(O.OracleLengthError):
            O.count_mismatches(c["observed"].encode(), c["expected"].encode())


@pytest.mark.parametrize("use_cache", [True, False])
def test_assign_known_answers(kat, use_cache):
    for c in kat["assign"]["cases"]:
        m = O.RefLiteral(c["barcodes"], c["max_mismatches"], c["min_mismatch_delta"], use_cache)
        exp = None if c["expect"] is None else tuple(c["expect"])
        # twice: the second call exercises the memo-cache hit path (Some-only insertion)
        assert m.assign(c["read"].encode()) == exp, c["name"]
        assert m.assign(c["read"].encode()) == exp, c["name"]


def test_assign_known_answers_ref_simple(kat):
    for c in kat["assign"]["cases"]:
        obs = np.frombuffer(c["read"].encode(), dtype=np.uint8)[None, :]
        idx, best, nxt, _ = O.ref_simple_assign_batch(c["barcodes"], c["max_mismatches"],
                                                      c["min_mismatch_delta"], obs)
        if c["expect"] is None:
            assert idx[0] == O.NONE_IDX, c["name"]
        else:
            assert (int(idx[0]), int(best[0]), int(nxt[0])) == tuple(c["expect"]), c["name"]


def test_constructor_errors(kat):
    for c in kat["assign"]["constructor_errors"]:
        with pytest.raises(ValueError, match=c["panic"]):
            O.RefLiteral(c["barcodes"], 2, 1, True)
    for c in kat["assign"]["constructor_ok"]:
        O.RefLiteral(c["barcodes"], c["max_mismatches"], c["min_mismatch_delta"], True)
    with pytest.raises(ValueError, match="cannot be empty"):
        O.RefLiteral(["ACGT", ""], 1, 2, True)


def test_demux_level_assigns(kat):
    for c in kat["demux_assign"]["cases"]:
        m = O.RefLiteral(c["barcodes"], c["max_mismatches"], c["min_mismatch_delta"], True)
        for r in c["reads"]:
            got = m.assign(r["observed"].encode())
            assert (None if got is None else got[0]) == r["expect"], (c["name"], r)


def test_short_read_is_none_and_long_read_is_error():
    m = O.RefLiteral(["ACGT", "TTTT"], 1, 1, False)
    assert m.assign(b"ACG") is None                      # barcode_matching.rs:167-169
    with pytest.raises(O.OracleLengthError):             # falls through to count_mismatches panic
        m.assign(b"ACGTA")
    assert m.assign(b"NNNNN") is None                    # prefilter fires before the panic (:170-172)


def test_single_sample_next_is_255():
    m = O.RefLiteral(["ACGT"], 1, 2, False)
    assert m.assign(b"ACGT") == (0, 0, 255)
    assert m.assign(b"ACGA") == (0, 1, 255)
    assert m.assign(b"AGGA") is None


def test_tie_lowest_index_and_next_equals_best():
    m = O.RefLiteral(["AAAA", "AAAC", "AAAG"], 3, 0, False)
    assert m.assign(b"AAAT") == (0, 1, 1)


ALPHABET = np.frombuffer(b"ACGTACGTACGTACGTNn.acgtRYKMSWBDHVXx-*0", dtype=np.uint8)
SAMPLE_ALPHABET = list("ACGTACGTACGTNMRWSYKVHDBn.")


def _random_case(rng, S, L):
    seen = set()
    barcodes = []
    while len(barcodes) < S:
        b = "".join(rng.choice(SAMPLE_ALPHABET, size=L))
        if b not in seen:
            seen.add(b)
            barcodes.append(b)
    return barcodes


@pytest.mark.parametrize("seed", range(12))
def test_ref_literal_equals_ref_simple_on_random_inputs(seed):
    """Property: the literal restatement (u32 blocks, adaptive cap, prefilter, cache) and the distilled
    spec agree on idx/best/next for random tables and reads incl. IUPAC, no-calls, lower case and
    garbage bytes, over extreme max_mismatches/min_delta values."""
    rng = np.random.default_rng(1234 + seed)
    S = int(rng.choice([1, 2, 3, 7, 16, 33]))
    L = int(rng.choice([1, 3, 7, 8, 9, 16, 17, 24]))
    if 4 ** min(L, 6) < S * 2:
        S = 1
    barcodes = _random_case(rng, S, L)
    mm = int(rng.choice([0, 1, 2, 3, 100, 255]))
    delta = int(rng.choice([0, 1, 2, 3, 100, 255]))
    n = 1500
    obs = ALPHABET[rng.integers(0, len(ALPHABET), size=(n, L))]
    # make most reads near a sample so Some results are common
    src = rng.integers(0, S, size=n)
    bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])[src]
    near = rng.random((n, L)) < 0.85
    obs = np.where(near, bc, obs).astype(np.uint8)
    for use_cache in (True, False):
        lit = O.RefLiteral(barcodes, mm, delta, use_cache)
        i1, b1, n1, c1 = lit.assign_batch(obs)
        i2, b2, n2, c2 = O.ref_simple_assign_batch(barcodes, mm, delta, obs)
        assert np.array_equal(i1, i2)
        assert np.array_equal(b1, b2)
        assert np.array_equal(n1, n2)
        assert np.array_equal(c1, c2)
        assert int(c1.sum()) == n


def test_ragged_lengths_literal_vs_simple():
    rng = np.random.default_rng(7)
    barcodes = ["ACGTAC", "TTGCAA", "NNGCAT"]
    L = 6
    n = 400
    stride = 8
    obs = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(n, stride))]
    lens = rng.integers(0, L + 1, size=n).astype(np.uint32)   # 0..L (no over-long reads)
    lit = O.RefLiteral(barcodes, 1, 1, True)
    i1, b1, n1, c1 = lit.assign_batch(obs, lens)
    i2, b2, n2, c2 = O.ref_simple_assign_batch(barcodes, 1, 1, obs, lens)
    assert np.array_equal(i1, i2) and np.array_equal(b1, b2) and np.array_equal(n1, n2)
    assert np.all(i1[lens < L] == O.NONE_IDX)


def test_cache_is_semantically_invisible_and_some_only():
    m = O.RefLiteral(["AAAA", "CCCC"], 1, 2, True)
    assert m.assign(b"AAAA") == (0, 0, 4)
    assert m.assign(b"GGGG") is None
    assert m.assign(b"GGGG") is None
    assert m.assign(b"AAAA") == (0, 0, 4)
    hits, misses = m.cache_stats
    assert hits == 1 and misses == 3     # None results are never cached (barcode_matching.rs:177-179)
