"""Parity tests proper: the hand-written HIP path, called through the C ABI (include/fqtk_match.h),
against the CPU oracle (oracle/ref_literal.c) and the reference's golden vectors.  Bit-exact: this is
integer work, every (idx, best, next) triple and every per-sample count must be identical.

Reads like the reference's own tests (/root/reference/src/lib/barcode_matching.rs:326-447)."""
import ctypes as C
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fqtk_amd import BarcodeMatch, BarcodeMatcher, FqtkLengthError, _lib  # noqa: E402
from fqtk_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

MATCH = np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")])


def _loaded_native():
    """The test must be exercising the in-tree HIP library, not anything else."""
    maps = open("/proc/self/maps").read()
    assert "libfqtk_match.so" in maps


def _compare(barcodes, mm, delta, obs, lens=None):
    """Every device path -- use_cache=True in both memo forms (LDS-resident where it can be built, and
    the HBM/L2 table pinned with memo_kind), use_cache=False (exhaustive scan) -- against the literal oracle."""
    lit = O.RefLiteral(barcodes, mm, delta, True)
    L = len(barcodes[0])
    if lens is None:
        i, b, nx, c = lit.assign_batch(np.ascontiguousarray(obs[:, :L]))
    else:
        i, b, nx, c = lit.assign_batch(obs, lens)
    for use_cache, pin_table in ((True, False), (True, True), (False, False)):
        gm = BarcodeMatcher(barcodes, mm, delta, use_cache)
        if pin_table:
            if gm.memo_kind != BarcodeMatcher.MEMO_LDS:
                continue          # the default already was the table (or the scan)
            gm.memo_kind = BarcodeMatcher.MEMO_TABLE
            assert gm.memo_kind in (BarcodeMatcher.MEMO_TABLE, BarcodeMatcher.MEMO_NONE)
        got, counts = gm.assign_batch(obs, lens)
        tag = (use_cache, pin_table, gm.memo_kind)
        assert np.array_equal(got["idx"], i), tag
        assert np.array_equal(got["best"], b), tag
        assert np.array_equal(got["next"], nx), tag
        assert np.array_equal(counts, c), tag
    _loaded_native()
    return got, counts


# ------------------------------------------------------------------------------------------------
# golden vectors (the reference's own known answers)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_cache", [True, False])
def test_assign_known_answers(kat, use_cache):
    for c in kat["assign"]["cases"]:
        matcher = BarcodeMatcher(c["barcodes"], c["max_mismatches"], c["min_mismatch_delta"], use_cache)
        expected = None if c["expect"] is None else BarcodeMatch(*c["expect"])
        assert matcher.assign(c["read"].encode()) == expected, c["name"]


def test_count_mismatches_known_answers_via_single_sample_tables(kat):
    # count_mismatches(observed, expected) == best_mismatches of a one-sample matcher
    for c in kat["count_mismatches"]["cases"]:
        if not c["expected"]:
            continue
        m = BarcodeMatcher([c["expected"]], 255, 0)
        got = m.assign(c["observed"].encode())
        assert got == BarcodeMatch(0, c["expect"], 255), c


def test_demux_level_assigns(kat):
    for c in kat["demux_assign"]["cases"]:
        m = BarcodeMatcher(c["barcodes"], c["max_mismatches"], c["min_mismatch_delta"])
        for r in c["reads"]:
            got = m.assign(r["observed"].encode())
            assert (None if got is None else got.best_match) == r["expect"], (c["name"], r)


def test_encode_all_256_byte_values_and_all_nibble_pairs():
    """enc() over every byte value x every expected IUPAC code, through the device LUT + AND-NOT path
    (enumerate all 16x16 nibble pairs and all 256 bytes)."""
    codes = "ACGTMRWSYKVHDBN"
    obs = np.arange(256, dtype=np.uint8)[:, None]
    for e in codes:
        m = BarcodeMatcher([e], 255, 0, use_cache=False)
        got, _ = m.assign_batch(obs)
        exp = ((O.ENC[np.arange(256)] & ~O.ENC[ord(e)] & 0xF) != 0).astype(np.uint8)
        assert np.array_equal(got["best"], exp), e
        assert np.all(got["idx"] == 0)


def test_length_rules():
    m = BarcodeMatcher(["ACGT", "TTTT"], 1, 1)
    assert m.assign(b"ACG") is None                 # shorter -> None (barcode_matching.rs:167-169)
    assert m.assign(b"") is None
    # longer -> the reference panics (:95-107); the sentence is the reference's own, cf. the expected
    # strings of its should_panic tests (:252-254, :315-317)
    with pytest.raises(FqtkLengthError, match=re.escape(
            "Read barcode (ACGTA) length (5) differs from expected barcode (ACGT) length (4) for sample sample_0")):
        m.assign(b"ACGTA")
    with pytest.raises(FqtkLengthError, match=re.escape("Read barcode (ANGTCA) length (6)")):
        m.assign(b"a.gtcA")                         # decode(encode(read)): upper case, '.' -> N (mod.rs:49-80)
    with pytest.raises(FqtkLengthError, match="Invalid bit mask for base: 0"):
        m.assign(b"ACGTX")                          # decode() itself panics on a byte with no mask (mod.rs:80)
    assert m.assign(b"NNNNN") is None               # ...unless the no-call prefilter fires first
    assert m.assign(b"ACGT") == BarcodeMatch(0, 0, 3)   # the handle stays usable after an error


def test_single_sample_and_ties():
    m = BarcodeMatcher(["ACGT"], 1, 2)
    assert m.assign(b"ACGT") == BarcodeMatch(0, 0, 255)
    assert m.assign(b"AGGA") is None
    t = BarcodeMatcher(["AAAA", "AAAC", "AAAG"], 3, 0)
    assert t.assign(b"AAAT") == BarcodeMatch(0, 1, 1)   # lowest index wins, next == best


def test_all_0xff_reads_do_not_alias_empty_table_slots():
    # enc(0xFF) = 0 never mismatches: with delta 0 the reference answers Some(0, 0, 0)
    barcodes = ["ACGTACGTACGTACGT", "TTTTACGTACGTACGA"]
    obs = np.full((300, 16), 0xFF, dtype=np.uint8)
    obs[7] = np.frombuffer(b"ACGTACGTACGTACGT", dtype=np.uint8)
    for delta in (0, 2):
        got, _ = _compare(barcodes, 1, delta, obs)
    b8 = ["ACGTACGT", "TTTTACGA"]
    _compare(b8, 1, 0, np.full((300, 8), 0xFF, dtype=np.uint8))


def test_empty_batch():
    m = BarcodeMatcher(["ACGT", "TTTT"], 1, 1)
    got, counts = m.assign_batch(np.empty((0, 4), dtype=np.uint8))
    assert got.shape == (0,) and counts.sum() == 0


def test_memo_table_is_built_when_it_should_be():
    w = synth.Workload(synth.CONFIGS[3])
    m = BarcodeMatcher(w.barcodes, 1, 2)
    # every canonical string within 1 mismatch of a sample: 1 + 16*4 per sample; all are Some here
    # because the table has pairwise distance >= 3
    assert m.memo_entries == 384 * (1 + 16 * 4)
    assert BarcodeMatcher(["A" * 21, "C" * 21], 1, 2).memo_entries == 2 * (1 + 21 * 4)   # (until round 4: L > 20 was scan only)
    assert BarcodeMatcher(["A" * 33, "C" * 33], 1, 2).memo_entries == 0      # L > 32: scan only
    assert BarcodeMatcher(w.barcodes, 6, 2).memo_entries == 0                # over the build budget
    assert BarcodeMatcher(["NNNNNNN"], 0, 2).memo_entries == 5 ** 7          # catch-all barcode


def test_memo_kind_selection():
    """The LDS-resident form needs plain A/C/G/T samples, max_mismatches <= 1 and a table that fits
    160 KiB; everything else uses the HBM/L2 table (or the scan)."""
    M = BarcodeMatcher
    cfg3 = synth.CONFIGS[3]
    assert M(synth.make_barcodes(cfg3), 1, 2).memo_kind == M.MEMO_LDS           # 24 960 entries -> 128 KiB
    assert M(synth.make_barcodes(synth.CONFIGS[2]), 1, 2).memo_kind == M.MEMO_LDS
    assert M(synth.make_barcodes(synth.CONFIGS[5]), 1, 2).memo_kind == M.MEMO_TABLE    # IUPAC samples
    assert M(["ACGTACGT", "TTTTGGGG", "CCCCAAAA"], 2, 1).memo_kind == M.MEMO_TABLE     # two mismatches
    assert M(["ACGTACGN", "TTTTGGGG"], 1, 1).memo_kind == M.MEMO_TABLE                 # N in a sample
    assert M(["ACGTACGT"], 1, 1).memo_kind == M.MEMO_TABLE                             # S = 1: next = 255
    assert M(["ACGTACGT", "TTTTGGGG"], 1, 1, use_cache=False).memo_kind == M.MEMO_NONE
    assert M(["A" * 24, "C" * 24], 1, 1).memo_kind == M.MEMO_LDS                       # three key words
    assert M(["ACGT" * 8, "CCGT" * 8], 1, 1).memo_kind == M.MEMO_LDS                   # four
    assert M(["A" * 32, "C" * 32], 1, 1).memo_kind == M.MEMO_TABLE                     # next = 32 does not fit the entry's 5 bits
    assert M(["A" * 33, "C" * 33], 1, 1).memo_kind == M.MEMO_NONE                      # L > 32
    m = M(["ACGTACGT", "TTTTGGGG"], 1, 1)
    assert m.memo_kind == M.MEMO_LDS
    m.memo_kind = M.MEMO_TABLE
    assert m.memo_kind == M.MEMO_TABLE
    m.memo_kind = M.MEMO_LDS
    assert m.memo_kind == M.MEMO_LDS
    with pytest.raises(ValueError):
        m.memo_kind = 7


def test_lds_form_with_any_size_table_384_samples_x_20_bases():
    """10+10 dual index, 384 samples: 31 104 memo entries need more than 32 768 slots; the LDS form then
    uses every slot that fits (multiply-shift slot mapping).  Packed and padded strides."""
    rng = np.random.default_rng(2020)
    seen = set()
    while len(seen) < 384:
        seen.add("".join(rng.choice(list("ACGT"), size=20)))
    bcs = sorted(seen)
    m = BarcodeMatcher(bcs, 1, 2)
    assert m.memo_kind == BarcodeMatcher.MEMO_LDS and m.memo_entries > 29000
    n = 20011
    noise = np.frombuffer(b"ACGTNacgtn.R", dtype=np.uint8)
    for stride in (20, 24, 21):
        obs = noise[rng.integers(0, len(noise), size=(n, stride))]
        src = rng.integers(0, 384, size=n)
        bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[src]
        keep = rng.random((n, 20)) < 0.97
        obs[:, :20] = np.where(keep, bc, obs[:, :20])
        _compare(bcs, 1, 2, obs)


@pytest.mark.parametrize("S,L", [(384, 24), (440, 23), (300, 17), (330, 30), (300, 32)])
def test_lds_form_behind_a_perfect_hash_for_12_plus_12_dual_indexes(S, L, monkeypatch):
    """384 samples x 24 bases: 37 248 memo entries are more four-byte cuckoo slots than LDS has; round 6 gives such tables an LDS
    form of three-byte entries behind a minimal perfect hash (lds_memo_plan.hpp plan_lds_memo_mph) -- the HBM/L2 table served them
    before.  Packed and padded strides, reads with ambiguity codes / junk (second pass) and no-calls, counts; and the same reads
    through the table form.  FQTK_LDS_MPH=2 selects the form wherever it can be planned (300 x 17 fits the cuckoo form too)."""
    rng = np.random.default_rng(S + L)
    seen = set()
    while len(seen) < S:
        seen.add("".join(rng.choice(list("ACGT"), size=L)))
    bcs = sorted(seen)
    monkeypatch.setenv("FQTK_LDS_MPH", "0")
    if (S, L) == (384, 24):
        assert BarcodeMatcher(bcs, 1, 2).memo_kind == BarcodeMatcher.MEMO_TABLE, "no other LDS form has room for this table"
    monkeypatch.setenv("FQTK_LDS_MPH", "2" if S == 300 else "1")
    m = BarcodeMatcher(bcs, 1, 2)
    assert m.memo_kind == BarcodeMatcher.MEMO_LDS and m.memo_entries == S * (1 + 4 * L)
    n = 50021
    noise = np.frombuffer(b"ACGTNacgtn.R-", dtype=np.uint8)
    for stride in (L, (L + 3) // 4 * 4, L + 5):
        obs = noise[rng.integers(0, len(noise), size=(n, stride))]
        bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[rng.integers(0, S, size=n)]
        keep = rng.random((n, L)) < 0.975
        obs[:, :L] = np.where(keep, bc, obs[:, :L])
        _compare(bcs, 1, 2, obs)
        # a variable-length batch (the LENS instantiations): shorter reads are None whatever their bytes say
        lens = np.where(rng.random(n) < 0.9, L, rng.integers(0, L + 1, size=n)).astype(np.uint32)
        got, _ = _compare(bcs, 1, 2, obs, lens)
        assert np.all(got["idx"][lens < L] == 0xFFFF) and (got["idx"][lens == L] != 0xFFFF).mean() > 0.4


def test_memo_path_handles_non_canonical_reads_in_every_lane_position():
    """IUPAC / unknown bytes in the READ cannot use the memo: the wave-cooperative fallback must give
    the scan kernel's answer wherever such reads sit in the wavefront, including all 64 lanes."""
    rng = np.random.default_rng(5)
    w = synth.Workload(synth.CONFIGS[3])
    cfg = w.cfg
    obs = w.fill_host(0, 4096).copy()
    obs[rng.integers(0, 4096, 300), rng.integers(0, 16, 300)] = ord("R")     # scattered
    obs[1024:1088, 3] = ord("X")                                             # one full wavefront
    obs[2048:2049, :] = ord("Y")
    obs[4095, 0] = ord("-")                                                  # last lane of last tile
    _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs)
    _compare(w.barcodes, 2, 1, obs)


@pytest.mark.parametrize("cap", ["0", "1", "7", None, "never"])
def test_second_pass_worklist_and_its_overflow(cap, monkeypatch):
    """The memo kernels list the reads with IUPAC / junk bytes and a second kernel resolves them (LDS form: the
    memo again with ambiguity codes spelled as N; table form: the scan); what does not fit in the list is scanned
    in place by its wave.  A tiny (or absent) list forces the overflow path; results and counts must not depend
    on where a read was resolved.  (A fresh matcher launches without the list until it has met such a read:
    FQTK_SECOND_PASS=always pins the list on.)"""
    monkeypatch.setenv("FQTK_SECOND_PASS", "never" if cap == "never" else "always")
    if cap not in (None, "never"):
        monkeypatch.setenv("FQTK_WORKLIST_CAP", cap)
    rng = np.random.default_rng(11)
    for k, n in ((3, 20000), (2, 9000), (5, 9000)):
        w = synth.Workload(synth.CONFIGS[k])
        cfg = w.cfg
        L = len(w.barcodes[0])
        obs = w.fill_host(0, n).copy()
        rows = rng.choice(n, n // 5, replace=False)                       # 20 % of the reads: more than the default list holds
        obs[rows, rng.integers(0, L, rows.size)] = rng.choice(np.frombuffer(b"RYKM#x", dtype=np.uint8), rows.size)
        obs[rng.integers(0, n, 500), rng.integers(0, L, 500)] = ord(".")  # '.' reads stay with the memo
        for j in range(L):                                                # every byte value at every position,
            obs[256 * j:256 * (j + 1), j] = np.arange(256, dtype=np.uint8)   # alone and next to another odd byte
            obs[256 * j + 128:256 * (j + 1), (j + 5) % L] = rng.choice(np.frombuffer(b"Nn.RUu#", dtype=np.uint8), 128)
        _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs)
        lens = np.where(rng.random(n) < 0.9, L, rng.integers(0, L + 1, size=n)).astype(np.uint32)
        _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs, lens)


def test_second_pass_switches_itself_on_after_the_first_such_read():
    """Default policy: no list and no second launch until a kernel has met an IUPAC / junk byte in a read; every
    batch is right whichever way it ran, on the synchronous path and on the pipeline slots."""
    rng = np.random.default_rng(12)
    for k in (3, 5):
        w = synth.Workload(synth.CONFIGS[k])
        cfg = w.cfg
        L = len(w.barcodes[0])
        lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
        gm = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
        for step in range(4):
            obs = w.fill_host(step * 30000, 30000).copy()
            if step >= 1:                                  # the first batch is clean, the others carry odd bytes
                rows = rng.choice(30000, 900, replace=False)
                obs[rows, rng.integers(0, L, rows.size)] = rng.choice(np.frombuffer(b"RYKMu#", dtype=np.uint8), rows.size)
            i, b, nx, c = lit.assign_batch(np.ascontiguousarray(obs[:, :L]))
            got, counts = gm.assign_batch(obs)
            assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx), (k, step)
            assert np.array_equal(counts, c), (k, step)


# ------------------------------------------------------------------------------------------------
# seeded random parity vs the oracle
# ------------------------------------------------------------------------------------------------
ALPHABET = np.frombuffer(b"ACGTACGTACGTACGTNn.acgtRYKMSWBDHVXx-*0", dtype=np.uint8)
SAMPLE_ALPHABET = list("ACGTACGTACGTNMRWSYKVHDBn.")


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FQTK_SOAK_SEEDS", "16"))))
def test_random_tables_and_reads(seed, monkeypatch):
    if seed % 2 == 0:                                   # odd bytes in the reads: half the seeds through the second pass
        monkeypatch.setenv("FQTK_SECOND_PASS", "always")
    rng = np.random.default_rng(99 + seed)
    S = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 33, 100]))
    L = int(rng.choice([1, 3, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40, 64, 65, 100, 128]))
    if 4 ** min(L, 6) < S * 2:
        S = 1
    seen, barcodes = set(), []
    while len(barcodes) < S:
        b = "".join(rng.choice(SAMPLE_ALPHABET, size=L))
        if b not in seen:
            seen.add(b)
            barcodes.append(b)
    mm = int(rng.choice([0, 1, 2, 3, 100, 255]))
    delta = int(rng.choice([0, 1, 2, 3, 100, 255]))
    n = 5000 + int(rng.integers(0, 3000))           # not a multiple of the tile: ragged tail
    stride = L + int(rng.choice([0, 0, 1, 3, 4]))   # aligned, unaligned and padded strides
    obs = ALPHABET[rng.integers(0, len(ALPHABET), size=(n, stride))]
    src = rng.integers(0, S, size=n)
    bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])[src]
    near = rng.random((n, L)) < 0.9
    obs[:, :L] = np.where(near, bc, obs[:, :L])
    _compare(barcodes, mm, delta, obs)
    if seed % 3 == 0:                               # the same batch as a variable-length one: ~10 % shorter reads
        lens = np.where(rng.random(n) < 0.9, L, rng.integers(0, L + 1, size=n)).astype(np.uint32)
        _compare(barcodes, mm, delta, obs, lens)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FQTK_SOAK_SEEDS", "16"))))
def test_random_plain_tables_lds_form(seed, monkeypatch):
    """Random plain-A/C/G/T tables with max_mismatches <= 1 -- the shape the LDS-resident memo is built
    for -- over random S, L, delta, strides and read noise (lower case, N, '.', IUPAC and junk bytes in
    the READS, which take the in-kernel fallback)."""
    if seed % 2 == 0:
        monkeypatch.setenv("FQTK_SECOND_PASS", "always")
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.integers(1, 21))
    S = int(rng.choice([2, 3, 5, 16, 24, 96, 200, 384, 500]))
    S = max(2, min(S, 4 ** L // 2))
    seen = set()
    while len(seen) < S:
        seen.add("".join(rng.choice(list("ACGTacgt") if rng.random() < 0.1 else list("ACGT"), size=L)))
    barcodes = list(seen)
    if len({b.upper() for b in barcodes}) < S:      # case-insensitive duplicates would be duplicate samples
        barcodes = sorted({b.upper() for b in barcodes})
        if len(barcodes) < 2:
            barcodes = ["A" * L, "C" * L]
    S = len(barcodes)
    mm, delta = int(rng.choice([0, 1, 1, 1])), int(rng.choice([0, 1, 2, 2, 3, 255]))
    m = BarcodeMatcher(barcodes, mm, delta)
    if m.memo_entries and S * (1 + 4 * L) <= 20000:   # (no Some entry at all, or > LDS capacity: table form)
        assert m.memo_kind == BarcodeMatcher.MEMO_LDS, (S, L, mm, delta)
    n = 3000 + int(rng.integers(0, 6000))
    stride = L + int(rng.choice([0, 0, 0, 1, 3, 4])) if rng.random() < 0.5 else (L + 3) // 4 * 4
    noise = np.frombuffer(b"ACGTACGTACGTNNacgtn.URYK#", dtype=np.uint8)
    obs = noise[rng.integers(0, len(noise), size=(n, stride))]
    src = rng.integers(0, S, size=n)
    bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])[src]
    keep = rng.random((n, L)) < float(rng.choice([0.8, 0.95, 0.99]))
    obs[:, :L] = np.where(keep, bc, obs[:, :L])
    _compare(barcodes, mm, delta, obs)
    if seed % 3 == 1:                               # variable-length batch through the LENS instantiations
        lens = np.where(rng.random(n) < 0.9, L, rng.integers(0, L + 1, size=n)).astype(np.uint32)
        _compare(barcodes, mm, delta, obs, lens)


def test_ragged_lengths_vs_oracle():
    rng = np.random.default_rng(7)
    barcodes = ["ACGTAC", "TTGCAA", "NNGCAT"]
    n, stride, L = 4000, 8, 6
    obs = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(n, stride))]
    lens = rng.integers(0, L + 1, size=n).astype(np.uint32)
    got, _ = _compare(barcodes, 1, 1, obs, lens)
    assert np.all(got["idx"][lens < L] == 0xFFFF)


@pytest.mark.parametrize("k", [2, 3, 5])
def test_variable_length_batches_use_the_memo_and_match_the_oracle(k):
    """obs_len batches ('+B' read structures): all three device paths (LDS memo, table memo, scan) follow
    barcode_matching.rs:165-172 -- length L -> matched as usual, shorter -> None, longer with more
    no-calls than max_mismatches + max_ns -> None (the prefilter fires before the panic)."""
    cfg = synth.CONFIGS[k]
    w = synth.Workload(cfg)
    n, L = 150_000, cfg.barcode_len
    rng = np.random.default_rng(100 + k)
    for stride in (cfg.stride, cfg.stride + 4, L + 3):
        obs = np.zeros((n, stride), dtype=np.uint8)
        obs[:, :cfg.stride][:, :min(cfg.stride, stride)] = w.fill_host(0, n)[:, :min(cfg.stride, stride)]
        lens = np.full(n, L, dtype=np.uint32)
        short = rng.random(n) < 0.05
        lens[short] = rng.integers(0, L, size=int(short.sum()))
        if stride > L:                                  # over-long reads the prefilter rejects: all no-calls
            long_ = (rng.random(n) < 0.02) & ~short
            lens[long_] = stride
            obs[long_] = np.frombuffer(b"Nn.", dtype=np.uint8)[rng.integers(0, 3, size=(int(long_.sum()), stride))]
        got, _ = _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs, lens)
        assert np.all(got["idx"][lens != L] == 0xFFFF)
        assert (got["idx"][lens == L] != 0xFFFF).mean() > 0.4     # the memo did serve the full-length reads


def test_direct_indexed_table_form_for_short_barcodes():
    """Barcodes of <= 10 bases: FQTK_MEMO_TABLE is the direct-indexed variant (flat array + LDS cache of the
    exact matches + cuckoo table for reads with an N); '.' no-calls are served under N's key, IUPAC / junk
    bytes in a read still take the wave scan.  All of it bit-exact vs the oracle, in every lane position."""
    cfg = synth.CONFIGS[5]
    w = synth.Workload(cfg)
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    assert m.memo_kind == BarcodeMatcher.MEMO_TABLE and m.memo_direct_bytes == 2
    m16 = BarcodeMatcher(["ACGTACGTACGTACGT", "TTTTACGTACGTACGA"], 2, 1)
    assert m16.memo_direct_bytes == 0                         # 16 bases: no direct index
    n = 40_000
    obs = w.fill_host(0, n)
    rng = np.random.default_rng(11)
    for ch, frac in ((b".", 0.2), (b"R", 0.02), (b"-", 0.02), (b"n", 0.1), (b"u", 0.02)):
        hit = rng.random(n) < frac
        obs[hit, rng.integers(0, cfg.barcode_len, int(hit.sum()))] = ch[0]
    _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs)
    _compare(w.barcodes, 0, 0, obs)                           # demux.rs:1494-1495 runs the IUPAC tables at 0/0
    # plain plates of 8, 9, 10 bases with the table form pinned (the default would be the LDS form where it fits)
    for L, S in ((8, 700), (9, 300), (10, 1536)):
        seen = set()
        while len(seen) < S:
            seen.add("".join(rng.choice(list("ACGT"), size=L)))
        bcs = sorted(seen)
        stride = (L + 3) // 4 * 4
        o = np.zeros((30_000, stride), dtype=np.uint8)
        src = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[rng.integers(0, S, 30_000)]
        flip = rng.random(src.shape) < 0.03
        src[flip] = np.frombuffer(b"ACGTN.", dtype=np.uint8)[rng.integers(0, 6, int(flip.sum()))]
        o[:, :L] = src
        _compare(bcs, 1, 1, o)


def test_rows_shorter_than_a_barcode_are_all_none_and_never_read_past_the_buffer():
    barcodes = ["ACGTACGTACGTACGTACGTACGTACGTACGT" * 4]        # L = 128, the longest the ABI takes
    m = BarcodeMatcher(barcodes + [barcodes[0][::-1]], 1, 1)
    n, stride = 1000, 3
    obs = np.frombuffer(b"ACG" * n, dtype=np.uint8).reshape(n, stride).copy()
    lens = np.full(n, 3, dtype=np.uint32)
    got, counts = m.assign_batch(obs, lens)
    assert np.all(got["idx"] == 0xFFFF) and counts[-1] == n and counts[:-1].sum() == 0
    assert m.assign(b"ACG") is None


def test_every_slot_latches_its_own_length_error():
    """A chunk's FQTK_ELEN is reported by the wait() of ITS slot, whatever else is in flight, and the
    synchronous assign_batch() neither waits on nor disturbs the caller's slots."""
    lib = _lib.load()
    m = BarcodeMatcher(["ACGT", "TTTT"], 1, 1)
    clean = np.frombuffer(b"ACGTAA" * 4, dtype=np.uint8).reshape(4, 6).copy()
    clean_len = np.full(4, 4, dtype=np.uint32)
    dirty = np.frombuffer(b"ACGTAA" b"ACGTAA" b"TTTTAA" b"ACGTAA", dtype=np.uint8).reshape(4, 6).copy()
    dirty_len = np.array([4, 4, 6, 4], dtype=np.uint32)
    out = [np.empty(4, dtype=np.uint32) for _ in range(3)]
    enq = lambda slot, o, l, r: lib.fqtk_matcher_enqueue(m.handle, slot, o.ctypes.data, 6, l.ctypes.data, 4, r.ctypes.data)
    assert enq(0, clean, clean_len, out[0]) == 0
    assert enq(1, dirty, dirty_len, out[1]) == 0
    assert enq(2, clean, clean_len, out[2]) == 0
    got, _ = m.assign_batch(clean, clean_len)                       # synchronous call while 0-2 are in flight
    assert list(got["idx"]) == [0, 0, 0, 0]
    assert lib.fqtk_matcher_wait(m.handle, 2) == 0
    assert lib.fqtk_matcher_wait(m.handle, 0) == 0                  # slot 0's chunk was clean
    assert lib.fqtk_matcher_wait(m.handle, 1) == _lib.FQTK_ELEN     # the error stayed with slot 1
    assert "Read barcode (TTTTAA) length (6)" in _lib.last_error() and "[read index 2 of its batch]" in _lib.last_error()
    assert lib.fqtk_matcher_wait(m.handle, 1) == 0                  # reported once
    view = out[1].view(np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")]))
    assert list(view["idx"]) == [0, 0, 0xFFFF, 0]
    assert lib.fqtk_matcher_enqueue(m.handle, _lib.FQTK_MAX_SLOTS, clean.ctypes.data, 6, clean_len.ctypes.data, 4,
                                    out[0].ctypes.data) == _lib.FQTK_EINVAL   # private slots are not the caller's


def test_overlong_reads_in_a_batch_report_lowest_index_and_still_fill_results():
    barcodes = ["ACGT", "TTTT"]
    obs = np.frombuffer(b"ACGTAA" b"TTTTAA" b"NNNNNN" b"ACGTAA", dtype=np.uint8).reshape(4, 6).copy()
    lens = np.array([4, 6, 6, 5], dtype=np.uint32)
    m = BarcodeMatcher(barcodes, 1, 1)
    with pytest.raises(FqtkLengthError, match=re.escape(
            "Read barcode (TTTTAA) length (6) differs from expected barcode (ACGT) length (4) for sample sample_0 "
            "[read index 1 of its batch]")):
        m.assign_batch(obs, lens)
    from fqtk_amd import Sample
    named = BarcodeMatcher([Sample("plate7-A01", "acgt", 0), Sample("plate7-A02", "TTTT", 1)], 1, 1)
    with pytest.raises(FqtkLengthError, match=re.escape("expected barcode (ACGT) length (4) for sample plate7-A01")):
        named.assign_batch(obs, lens)
    with pytest.raises(ValueError, match="exceeds stride"):     # obs_len beyond its row is refused, not read
        m.assign_batch(obs, np.array([4, 7, 6, 4], dtype=np.uint32))
    # same batch without the offending reads is fine, and the prefiltered over-long read is None
    lens2 = np.array([4, 4, 6, 4], dtype=np.uint32)
    got, _ = m.assign_batch(obs, lens2)
    assert list(got["idx"]) == [0, 1, 0xFFFF, 0]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_baseline_configs_reduced_size_vs_oracle(k):
    """The five BASELINE.json configs at a size the oracle finishes in seconds."""
    cfg = synth.CONFIGS[k]
    w = synth.Workload(cfg)
    n = {1: 300_000, 2: 300_000, 3: 200_000, 4: 300_000, 5: 60_000}[k]
    obs = w.fill_host(0, n)
    got, counts = _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs)
    assert int(counts.sum()) == n
    if k == 5:   # also the 0/0 parameters of demux.rs:1494-1495
        _compare(w.barcodes, 0, 0, obs[:20000])


def test_many_samples_lds_and_global_histogram_paths():
    """S+1 <= 8192 keeps the per-sample histogram in LDS; above that counts go straight to global
    atomics.  Both must agree with the oracle (S = 3000 and S = 9000, L = 12, ragged tail)."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for S in (3000, 9000):
        codes = rng.choice(4 ** 12, size=S, replace=False)
        bcs = ["".join("ACGT"[(int(c) >> (2 * k)) & 3] for k in range(12)) for c in codes]
        n = 20_011
        src = rng.integers(0, S, size=n)
        obs = np.stack([np.frombuffer(bcs[i].encode(), dtype=np.uint8) for i in src]).copy()
        flip = rng.random((n, 12)) < 0.03
        obs[flip] = acgt[rng.integers(0, 4, size=int(flip.sum()))]
        obs[rng.integers(0, n, 50), rng.integers(0, 12, 50)] = ord("N")
        _compare(bcs, 1, 1, obs)


def test_maximum_sample_count_65534():
    """S = 65 534 is the largest table the 16-bit index allows (0xFFFF = None).  Samples differ in their
    first 8 bases (a base-4 counter), so the last sample's index 0xFFFD/0xFFFE must come back intact."""
    S, L = 65534, 12
    digits = np.array([(np.arange(S) >> (2 * k)) & 3 for k in range(8)]).T           # [S, 8]
    bc_bytes = np.frombuffer(b"ACGT", dtype=np.uint8)[digits]
    bc_bytes = np.concatenate([bc_bytes, np.full((S, 4), ord("A"), dtype=np.uint8)], axis=1)
    barcodes = [row.tobytes().decode() for row in bc_bytes]
    rng = np.random.default_rng(65534)
    pick = np.concatenate([[0, 1, S - 2, S - 1], rng.integers(0, S, 396)])
    obs = bc_bytes[pick].copy()
    obs[5:200, 10] = ord("C")          # one mismatch in the constant tail: best = 1 everywhere
    obs[200:260, 3] = ord("N")
    for use_cache in (True, False):
        m = BarcodeMatcher(barcodes, 1, 0, use_cache)
        got, counts = m.assign_batch(obs)
        i, b, nx, c = O.ref_simple_assign_batch(barcodes, 1, 0, obs)
        assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx)
        assert np.array_equal(counts, c)
        assert got["idx"][3] == S - 1 and got["idx"][2] == S - 2


def test_limits_are_rejected_at_create():
    """More than 65 534 samples (the 16-bit index) or barcodes longer than 128 bases (u8 mismatch counts
    must not saturate) are argument errors, not silent truncation."""
    with pytest.raises(ValueError):
        BarcodeMatcher(["ACGTACGTACGT"] * 65535, 1, 1)
    with pytest.raises(ValueError):
        BarcodeMatcher(["A" * 129, "C" * 129], 1, 1)
    BarcodeMatcher(["A" * 128, "C" * 128], 1, 1)


def test_longest_memo_key_and_longer_barcodes():
    """L = 32 is the longest barcode the memo covers (128-bit key: lo, hi, ext, ext2; 16 + 16 dual indices); L = 33..128
    are scan-only.  (The reference's cache has no length limit: barcode_matching.rs:174-181.)"""
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for L in (19, 20, 21, 24, 25, 31, 32, 33, 96, 128):
        bcs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(40)]
        m = BarcodeMatcher(bcs, 1, 1)
        assert (m.memo_entries > 0) == (L <= 32)
        n = 3000
        src = rng.integers(0, 40, size=n)
        obs = np.stack([np.frombuffer(bcs[i].encode(), dtype=np.uint8) for i in src]).copy()
        flip = rng.random((n, L)) < 0.02
        obs[flip] = acgt[rng.integers(0, 5, size=int(flip.sum()))]
        _compare(bcs, 1, 1, obs)
        _compare(bcs, 2, 1, obs[:500])


@pytest.mark.parametrize("L", range(1, 34))
def test_every_memo_key_width_and_load_path(L):
    """Every barcode length the memo covers (and one it does not), on every load path: the packed
    stride (vector loads), a dword-padded stride and an odd stride.  L <= 8 is one key word,
    9-10 the folded single word, 11-16 two words, 17-24 three, 25-32 four; reads include lower case, N, '.', U and
    IUPAC/junk bytes (the wave-cooperative fallback) and pad bytes that must be ignored."""
    rng = np.random.default_rng(500 + L)
    S = min(48, 4 ** L // 2) or 1
    seen = set()
    while len(seen) < S:
        seen.add("".join(rng.choice(list("ACGT"), size=L)))
    bcs = sorted(seen)
    m = BarcodeMatcher(bcs, 1, 1)
    assert (m.memo_entries > 0) == (L <= 32)
    n = 4099
    noise = np.frombuffer(b"ACGTNacgtn.URY#", dtype=np.uint8)
    for stride in sorted({(L + 3) // 4 * 4, (L + 3) // 4 * 4 + 4, L | 1, L + 2}):
        if stride < L:
            continue
        obs = noise[rng.integers(0, len(noise), size=(n, stride))]
        src = rng.integers(0, S, size=n)
        bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[src]
        keep = rng.random((n, L)) < 0.93
        obs[:, :L] = np.where(keep, bc, obs[:, :L])
        lower = rng.random(n) < 0.1
        obs[lower, :L] |= np.where(obs[lower, :L] >= 0x41, 0x20, 0).astype(np.uint8)   # lower case encodes the same
        _compare(bcs, 1, 1, obs)
        _compare(bcs, 2, 0, obs[:700])


def test_memo_keys_do_not_alias_near_identical_reads():
    """Reads that differ from a sample in exactly one base at every position and to every other
    canonical base (and N): each must resolve to its own memo entry, for all five key layouts."""
    for L in (8, 10, 16, 20, 24, 29, 32):
        bcs = ["ACGT" * 8, "TGCA" * 8, "GGGGGCCCCCAAAAATTTTTGGGGGCCCCCAA", "ATATATATATCGCGCGCGCGATATATCGCGCG"]
        bcs = [b[:L] for b in bcs]
        rows = []
        for b in bcs:
            for k in range(L):
                for ch in "ACGTN":
                    rows.append(b[:k] + ch + b[k + 1:])
                    for k2 in range(k + 1, L, 3):
                        rows.append(b[:k] + ch + b[k + 1:k2] + "N" + b[k2 + 1:])
        obs = np.frombuffer("".join(rows).encode(), dtype=np.uint8).reshape(-1, L)
        for mm, delta in ((1, 1), (2, 1), (2, 2), (0, 1)):
            _compare(bcs, mm, delta, obs)


def test_dual_index_12_plus_12_and_a_memo_of_millions_of_strings():
    """12 + 12 dual indices (L = 24) used to fall to the scan; with --max-mismatches 2 the memo of such a
    plate is 1.7 M strings, and cfg 3 with three mismatches enumerates 14.5 M -- streamed through the scan kernel in
    chunks at create time (the old budget was 6 M strings held in host memory at once)."""
    rng = np.random.default_rng(24)
    seen = set()
    while len(seen) < 384:
        seen.add("".join(rng.choice(list("ACGT"), size=24)))
    bcs = sorted(seen)
    acgtn = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n = 30_000
    obs = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[rng.integers(0, 384, n)].copy()
    flip = rng.random(obs.shape) < 0.03
    obs[flip] = acgtn[rng.integers(0, 5, int(flip.sum()))]
    obs[rng.random(n) < 0.1] = acgtn[rng.integers(0, 4, 24)]
    for mm, delta in ((1, 2), (2, 2)):
        m = BarcodeMatcher(bcs, mm, delta)
        assert m.memo_entries > 0 and m.memo_candidates == 384 * (1 + 24 * 4 + (mm == 2) * 276 * 16)
        _compare(bcs, mm, delta, obs)
    w = synth.Workload(synth.CONFIGS[3])
    m = BarcodeMatcher(w.barcodes, 3, 2)
    assert m.memo_candidates == 384 * (1 + 16 * 4 + 120 * 16 + 560 * 64) and m.memo_entries > 1_000_000
    o16 = w.fill_host(0, 20_000)
    flip = rng.random(o16.shape) < 0.08
    o16 = o16.copy()
    o16[flip] = acgtn[rng.integers(0, 5, int(flip.sum()))]
    got, counts = m.assign_batch(o16)
    i, b, nx, c = O.RefLiteral(w.barcodes, 3, 2, True).assign_batch(o16)
    assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx) and np.array_equal(counts, c)


def test_device_entry_point_with_misaligned_and_padded_buffers():
    """The zero-copy entry point must not assume alignment: an odd base pointer and a stride larger
    than the barcode take the generic load path and still match the oracle."""
    import torch
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    n = 100_003
    host = w.fill_host(0, n)                      # [n, 8]
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    i, b, nx, _ = lit.assign_batch(host)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    dt = np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")])
    for off, stride in ((1, 8), (0, 11), (3, 13), (4, 12), (0, 16)):
        padded = np.full((n, stride), ord("#"), dtype=np.uint8)
        padded[:, :8] = host
        raw = torch.zeros(off + n * stride + 16, dtype=torch.uint8, device=dev)
        raw[off:off + n * stride] = torch.from_numpy(padded.reshape(-1)).to(dev)
        d_out = torch.empty(n, dtype=torch.int32, device=dev)
        for use_cache in (True, False):
            m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, use_cache)
            m.assign_batch_device(raw.data_ptr() + off, stride, n, d_out.data_ptr(), stream=stream)
            m.poll_error(stream)
            got = d_out.cpu().numpy().view(dt)
            assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx), (off, stride, use_cache)


def test_cfg1_full_size_vs_oracle():
    """BASELINE config 1 at its FULL size (1 M reads x 16 samples): every result and every count."""
    cfg = synth.CONFIGS[1]
    w = synth.Workload(cfg)
    obs = w.fill_host(0, cfg.n_reads)
    _, counts = _compare(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, obs)
    assert int(counts.sum()) == cfg.n_reads


def test_device_generator_matches_host_twin_and_zero_copy_entry_point():
    import torch
    cfg = synth.CONFIGS[3]
    w = synth.Workload(cfg)
    n = 1_000_003
    dev = torch.device("cuda:0")
    d_obs = torch.empty((n, cfg.stride), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    w.fill_device(0, n, d_obs.data_ptr(), stream)
    host = w.fill_host(0, n)
    assert np.array_equal(d_obs.cpu().numpy(), host)
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    d_out = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(cfg.n_samples + 1, dtype=torch.int64, device=dev)
    m.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_counts.data_ptr(),
                          stream=stream)
    m.poll_error(stream)
    got = d_out.cpu().numpy().view(np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")]))
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    i, b, nx, c = lit.assign_batch(host)
    assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx)
    assert np.array_equal(d_counts.cpu().numpy().astype(np.uint64), c)


def test_full_size_properties_cfg3_slice():
    """Size-independent properties at a size the oracle cannot finish: counts sum to n, the histogram
    of the result array equals the device counts, a permuted input gives the permuted output, and
    running twice accumulates exactly 2x (idempotent results, additive counts)."""
    import torch
    cfg = synth.CONFIGS[3]
    w = synth.Workload(cfg)
    n = 40_000_000
    dev = torch.device("cuda:0")
    d_obs = torch.empty((n, cfg.stride), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    w.fill_device(0, n, d_obs.data_ptr(), stream)
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    d_out = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(cfg.n_samples + 1, dtype=torch.int64, device=dev)
    m.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_counts.data_ptr(), stream=stream)
    m.poll_error(stream)
    counts = d_counts.cpu().numpy()
    assert counts.sum() == n
    idx = (d_out & 0xFFFF).to(torch.int64)
    idx = torch.where(idx == 0xFFFF, torch.full_like(idx, cfg.n_samples), idx)
    hist = torch.bincount(idx, minlength=cfg.n_samples + 1).cpu().numpy()
    assert np.array_equal(hist, counts)
    # second pass: same results, counts doubled
    d_out2 = torch.empty_like(d_out)
    m.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out2.data_ptr(), d_counts.data_ptr(), stream=stream)
    m.poll_error(stream)
    assert torch.equal(d_out, d_out2)
    assert np.array_equal(d_counts.cpu().numpy(), 2 * counts)
    # permutation equivariance on a slice
    k = 5_000_000
    perm = torch.randperm(k, device=dev)
    d_p = d_obs[:k][perm].contiguous()
    d_outp = torch.empty(k, dtype=torch.int32, device=dev)
    m.assign_batch_device(d_p.data_ptr(), cfg.stride, k, d_outp.data_ptr(), 0, stream=stream)
    m.poll_error(stream)
    assert torch.equal(d_outp, d_out[:k][perm])
    # oracle spot-check of three far-apart windows of the big run
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    for start in (0, 17_000_001, n - 50_000):
        host = w.fill_host(start, 50_000)
        i, b, nx, _ = lit.assign_batch(host)
        got = d_out[start:start + 50_000].cpu().numpy().view(np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")]))
        assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx)


def test_pinned_pipeline_slots_match_sync_path():
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    lib = _lib.load()
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    n_chunk, chunks = 200_000, 5
    ref_out, ref_counts = m.assign_batch(w.fill_host(0, n_chunk * chunks))
    bufs = []
    for s in range(2):
        p_obs, p_out = C.c_void_p(), C.c_void_p()
        assert lib.fqtk_pinned_alloc(n_chunk * cfg.stride, C.byref(p_obs)) == 0
        assert lib.fqtk_pinned_alloc(n_chunk * 4, C.byref(p_out)) == 0
        bufs.append((p_obs, p_out))
    outs = []
    for c in range(chunks + 2):
        slot = c % 2
        if c >= 2:
            assert lib.fqtk_matcher_wait(m.handle, slot) == 0
            outs.append(np.ctypeslib.as_array(C.cast(bufs[slot][1], C.POINTER(C.c_uint32)), (n_chunk,)).copy())
        if c < chunks:
            host = w.fill_host(c * n_chunk, n_chunk)
            C.memmove(bufs[slot][0], host.ctypes.data, host.nbytes)
            assert lib.fqtk_matcher_enqueue(m.handle, slot, bufs[slot][0], cfg.stride, None, n_chunk,
                                            bufs[slot][1]) == 0, _lib.last_error()
    got = np.concatenate(outs).view(ref_out.dtype)
    assert np.array_equal(got, ref_out)
    counts = np.zeros(cfg.n_samples + 1, dtype=np.uint64)
    assert lib.fqtk_matcher_counts(m.handle, counts.ctypes.data) == 0
    assert np.array_equal(counts, ref_counts)
    for p_obs, p_out in bufs:
        lib.fqtk_pinned_free(p_obs)
        lib.fqtk_pinned_free(p_out)


def test_sync_batch_counts_do_not_disturb_the_pipeline_accumulator():
    """assign_batch() reports the counts of its own call; the enqueue()/counts() accumulator of the
    same handle is untouched by it (and by assign1)."""
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    lib = _lib.load()
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    a = w.fill_host(0, 50_000)
    b = w.fill_host(50_000, 30_000)
    out_a = np.empty(len(a), dtype=np.uint32)
    assert lib.fqtk_matcher_enqueue(m.handle, 3, a.ctypes.data, cfg.stride, None, len(a), out_a.ctypes.data) == 0
    assert lib.fqtk_matcher_wait(m.handle, 3) == 0
    got_b, counts_b = m.assign_batch(b)                       # synchronous call in between
    assert m.assign(bytes(b[0, :cfg.barcode_len])) is not None or True
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    _, _, _, ca = lit.assign_batch(a)
    _, _, _, cb = lit.assign_batch(b)
    assert np.array_equal(counts_b, cb)
    pipeline = np.zeros(cfg.n_samples + 1, dtype=np.uint64)
    assert lib.fqtk_matcher_counts(m.handle, pipeline.ctypes.data) == 0
    assert np.array_equal(pipeline, ca)


def test_full_parity_tool_reduced_size():
    """tools/full_parity.py is the full-size gate (results for all 751 M reads of the five configs are
    in profiles/r01_full_parity.jsonl); here it runs on a 3 M-read prefix of cfg 2 and cfg 5."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for c in ("2", "5"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "full_parity.py"), "--config", c,
                            "--reads", "3000000", "--procs", "4"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["mismatching_reads"] == 0 and d["counts_equal"] and d["reads"] == 3000000


def test_rccl_count_allreduce_entry_point():
    """fqtk_matchers_allreduce_counts: the path's one collective, natively over RCCL.  One matcher per visible
    device (a one-rank communicator is forced on a single-GPU box), chunks dealt round-robin, the all-reduced
    per-sample counts equal the oracle's for the whole stream; the accumulators are reset afterwards."""
    lib = _lib.load()
    ndev = C.c_int(0)
    assert lib.fqtk_device_count(C.byref(ndev)) == 0 and ndev.value >= 1
    G = min(ndev.value, 4)
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    ms = [BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=d) for d in range(G)]
    n_chunk, chunks = 100_000, 6
    outs = [np.empty(n_chunk, dtype=np.uint32) for _ in range(chunks)]
    hosts = [w.fill_host(c * n_chunk, n_chunk) for c in range(chunks)]
    for c in range(chunks):                       # chunk k -> device k mod G, slot k // G
        assert lib.fqtk_matcher_enqueue(ms[c % G].handle, c // G, hosts[c].ctypes.data, cfg.stride, None, n_chunk,
                                        outs[c].ctypes.data) == 0, _lib.last_error()
    for c in range(chunks):
        assert lib.fqtk_matcher_wait(ms[c % G].handle, c // G) == 0
    handles = (C.c_void_p * G)(*[m.handle for m in ms])
    counts = np.zeros(cfg.n_samples + 1, dtype=np.uint64)
    rc = lib.fqtk_matchers_allreduce_counts(handles, G, 1, counts.ctypes.data)
    assert rc == 0, _lib.last_error()
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    i, b, nx, want = lit.assign_batch(np.concatenate(hosts))
    assert np.array_equal(counts, want)
    assert np.array_equal(np.concatenate(outs).view(MATCH)["idx"], i)
    again = np.zeros_like(counts)
    assert lib.fqtk_matchers_allreduce_counts(handles, G, 1, again.ctypes.data) == 0 and again.sum() == 0   # reset
    dup = (C.c_void_p * 2)(ms[0].handle, ms[0].handle)
    assert lib.fqtk_matchers_allreduce_counts(dup, 2, 0, again.ctypes.data) == _lib.FQTK_EINVAL   # one matcher per DISTINCT device


def _few_cpus():
    import os
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = os.cpu_count() if q == "max" else max(1, int(q) // int(p))
    except Exception:
        quota = os.cpu_count() or 1
    return min(quota, len(os.sched_getaffinity(0))) < 12


@pytest.mark.skipif(bool(__import__("os").environ.get("FQTK_SKIP_FULL_PARITY")) or _few_cpus(),
                    reason="every read of all five BASELINE configs (751 M reads, ~3 min on 16 CPUs): needs >= 12 usable CPUs; FQTK_SKIP_FULL_PARITY=1 skips it")
def test_full_size_parity_of_every_baseline_config():
    """The full-size gate: tools/full_parity.py over EVERY read of cfg 1-5 with the default memo path, plus the table
    form pinned on cfg 3 -- 0 mismatching (idx, best, next) triples and identical per-sample count vectors vs the
    oracle.  (The default bench.py run checks the first 50 M triples + all 400 M counts of cfg 3 on its own.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["--config", "3"], ["--config", "2"], ["--config", "4"], ["--config", "5"], ["--config", "1"],
                 ["--config", "3", "--table"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "full_parity.py")] + args, capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["mismatching_reads"] == 0 and d["counts_equal"], d


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_packed_entry_equals_the_ascii_entry_and_the_oracle_on_every_config(k):
    """fqtk_pack_barcodes + fqtk_matcher_enqueue_packed (4 bits per base over PCIe): triples and counts equal those of
    fqtk_matcher_assign_batch and of the oracle, reads with IUPAC / junk bytes included (they travel as exceptions)."""
    cfg = synth.CONFIGS[k]
    w = synth.Workload(cfg)
    n = {1: 100_000, 2: 200_000, 3: 200_000, 4: 100_000, 5: 60_000}[k]
    obs = w.fill_host(0, n).copy()
    rng = np.random.default_rng(40 + k)
    L = cfg.barcode_len
    for i in rng.choice(n, 300, replace=False):                         # IUPAC codes, lower case, dots, junk
        obs[i, int(rng.integers(0, L))] = int(rng.choice(np.frombuffer(b"RYKMSWBDHVUnacgt.*\0", dtype=np.uint8)))
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=0)
    packed, exc_index, exc_rows = m.pack(obs)
    assert packed.shape[1] == {8: 4, 10: 8, 16: 8}[L] and 0 < exc_index.size <= 300
    got_p, counts_p = m.assign_batch_packed(packed, exc_index, exc_rows)
    got_a, counts_a = m.assign_batch(obs)
    assert np.array_equal(got_p.view(np.uint32), got_a.view(np.uint32)) and np.array_equal(counts_p, counts_a)
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    i, b, nx, c = lit.assign_batch(np.ascontiguousarray(obs[:, :L]))
    assert np.array_equal(got_p["idx"], i) and np.array_equal(got_p["best"], b) and np.array_equal(got_p["next"], nx)
    assert np.array_equal(counts_p, c)
    # no exceptions at all, and two chunks in flight on two slots
    clean = w.fill_host(n, 50_000)
    p2, e2, r2 = m.pack(clean)
    if k != 5:
        assert e2.size == 0
    got2, _ = m.assign_batch_packed(p2, e2, r2, slot=1)
    ref2, _ = m.assign_batch(clean)
    assert np.array_equal(got2.view(np.uint32), ref2.view(np.uint32))


@pytest.mark.parametrize("L,iupac", [(16, False), (24, False), (32, False), (14, True)])
def test_presence_filter_of_the_table_form_never_changes_a_result(L, iupac, monkeypatch):
    """Round 5: where a hash-table memo's exact-match entries are few, the kernel keeps a two-choice hot table and a PRESENCE FILTER over
    all memo keys in LDS, and a read whose two filter bits are not both set is None without a look at the table (memo_hash.hpp).  A filter
    must have no false negatives: the same reads through the table form with the filter, without it (FQTK_MEMO_NO_FILTER=1 at create)
    and through the oracle -- 30 % of them random (in no slot of the table: the reads the filter answers), the rest samples with
    substitutions and no-calls (barcode_matching.rs:119-186)."""
    rng = np.random.default_rng(900 + L)
    S = 384
    alphabet = list("ACGT") if not iupac else list("ACGTACGTACGTRYKMN")
    seen = set()
    while len(seen) < S:
        seen.add("".join(rng.choice(alphabet, size=L)))
    bcs = sorted(seen)
    n = 200_000
    acgtn = np.frombuffer(b"ACGTN", dtype=np.uint8)
    plain = np.frombuffer(b"ACGT", dtype=np.uint8)
    src = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[rng.integers(0, S, size=n)]
    src = np.where(np.isin(src, plain), src, plain[rng.integers(0, 4, size=src.shape)])      # (a read of a degenerate sample: some base it allows, or not)
    obs = np.where(rng.random((n, L)) < 0.985, src, acgtn[rng.integers(0, 5, size=(n, L))])
    rnd = rng.random(n) < 0.3
    obs[rnd] = plain[rng.integers(0, 4, size=(int(rnd.sum()), L))]
    obs = np.ascontiguousarray(obs)
    lit = O.RefLiteral(bcs, 1, 2, True)
    i, b, nx, c = lit.assign_batch(obs)
    results = {}
    for name, off in (("filter", ""), ("no filter", "1")):
        monkeypatch.setenv("FQTK_MEMO_NO_FILTER", off)
        gm = BarcodeMatcher(bcs, 1, 2, True)
        if gm.memo_kind == BarcodeMatcher.MEMO_LDS:
            gm.memo_kind = BarcodeMatcher.MEMO_TABLE
        assert gm.memo_kind == BarcodeMatcher.MEMO_TABLE, (name, gm.memo_kind)
        got, counts = gm.assign_batch(obs)
        assert np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx), name
        assert np.array_equal(counts, c), name
        results[name] = got
    assert np.array_equal(results["filter"].view(np.uint32), results["no filter"].view(np.uint32))
    assert int((i == 0xFFFF).sum()) > n // 5          # (the random reads are unmatched: the filter had something to answer)
    _loaded_native()
