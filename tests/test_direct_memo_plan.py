"""Host logic of the direct-indexed memo (fqtk_amd/csrc/direct_memo_plan.hpp + memo_hash.hpp), no GPU: the
planner is fed the no-call-free memo entries the CPU oracle computes; the kernel's lookup (the same index
arithmetic, 16-bit entry packing and LDS-cache probe the HIP kernel runs) is replayed for every stored key
and for random A/C/G/T reads, and must return exactly what the oracle assigns."""
import ctypes as C

import numpy as np
import pytest

from fqtk_amd import synth
from oracle import oracle as O
from tests import hostlib
from tests.test_lds_memo_plan import NONE, _keys, _neighbours


def _word(idx, best, nxt):
    return np.where(idx == O.NONE_IDX, NONE, idx.astype(np.uint32) | (best.astype(np.uint32) << 16) | (nxt.astype(np.uint32) << 24)).astype(np.uint32)


def _neighbours_iupac(barcodes, mm):
    """Every A/C/G/T string within `mm` (<= 1) mismatches of a (possibly degenerate) sample barcode."""
    masks = {"A": 1, "C": 2, "G": 4, "T": 8, "M": 3, "R": 5, "W": 9, "S": 6, "Y": 10, "K": 12, "V": 7, "H": 11, "D": 13, "B": 14, "N": 15}
    out = set()
    for b in barcodes:
        opts = [[c for c, m in zip("ACGT", (1, 2, 4, 8)) if masks[ch] & m] for ch in b]
        exact = [""]
        for o in opts:
            exact = [e + c for e in exact for c in o]
        for e in exact:
            out.add(e)
            if mm >= 1:
                for k in range(len(e)):
                    for c in "ACGT":
                        out.add(e[:k] + c + e[k + 1:])
    return np.stack([np.frombuffer(x.encode(), dtype=np.uint8) for x in sorted(out)])


def _plan_and_probe(barcodes, mm, delta, probe):
    S, L = len(barcodes), len(barcodes[0])
    cand = _neighbours_iupac(barcodes, min(mm, 1))
    lit = O.RefLiteral(barcodes, mm, delta, True)
    idx, best, nxt, _ = lit.assign_batch(cand)
    some = idx != O.NONE_IDX
    cand, vals = cand[some], _word(idx, best, nxt)[some]
    keys = np.ascontiguousarray(_keys(cand)[:, :2])
    q = np.ascontiguousarray(_keys(probe)[:, :2])
    out = np.zeros(len(q), dtype=np.uint32)
    cached = np.zeros(len(q), dtype=np.uint8)
    meta = np.zeros(8, dtype=np.uint32)
    rc = hostlib.lib().fqtk_host_direct_memo(C.c_uint32(S), C.c_uint32(L), C.c_uint64(len(keys)), keys.ctypes.data_as(C.c_void_p),
                                             np.ascontiguousarray(vals).ctypes.data_as(C.c_void_p), C.c_uint64(len(q)),
                                             q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                             cached.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p))
    assert rc == 0
    i2, b2, n2, _ = lit.assign_batch(probe)
    return meta, out, cached, _word(i2, b2, n2), cand, vals


def test_cfg5_iupac_table_every_acgt_read_resolves_through_the_direct_index():
    cfg = synth.CONFIGS[5]
    barcodes = synth.make_barcodes(cfg)
    rng = np.random.default_rng(5)
    cand = _neighbours_iupac(barcodes, 1)
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(200_000, 10))]
    probe = np.concatenate([cand, rnd])
    meta, out, cached, want, kept, vals = _plan_and_probe(barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, probe)
    assert meta[0] == 2 and meta[1] == 11 and meta[7] == 1 << 20       # 16-bit entries [idx:11 | best:1 | next:4], 4^10 of them
    assert np.array_equal(out, want)                                   # stored keys AND absent keys (None)
    # the LDS cache: <= 64 KiB, holds (nearly) every exact-match entry, and answers only exact matches
    assert meta[4] * 4 <= 64 * 1024 and meta[6] >= 0.97 * meta[5] > 0
    assert np.all(((out[cached == 1] >> 16) & 0xFF) == 0)
    exact = ((want >> 16) & 0xFF) == 0
    assert cached[exact & (want != NONE)].mean() > 0.97


@pytest.mark.parametrize("L,S,mm,delta", [(4, 12, 1, 1), (7, 40, 1, 2), (8, 96, 2, 1), (9, 300, 1, 2), (10, 1536, 1, 1), (10, 3000, 0, 0)])
def test_lengths_packings_and_parameters(L, S, mm, delta):
    rng = np.random.default_rng(L * 1000 + S)
    seen = set()
    while len(seen) < S:
        seen.add("".join(rng.choice(list("ACGT"), size=L)))
    barcodes = sorted(seen)
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(50_000, L))]
    near = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])[rng.integers(0, S, 50_000)].copy()
    near[np.arange(50_000), rng.integers(0, L, 50_000)] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50_000)]
    probe = np.concatenate([rnd, near])
    if mm <= 1:
        meta, out, cached, want, _, _ = _plan_and_probe(barcodes, mm, delta, probe)
        assert np.array_equal(out, want)
    else:   # the planner only needs SOME complete entry set: feed it every string the probe contains + neighbours
        lit = O.RefLiteral(barcodes, mm, delta, True)
        uniq = np.unique(probe, axis=0)
        idx, best, nxt, _ = lit.assign_batch(uniq)
        some = idx != O.NONE_IDX
        keys = np.ascontiguousarray(_keys(uniq[some])[:, :2])
        vals = np.ascontiguousarray(_word(idx, best, nxt)[some])
        q = np.ascontiguousarray(_keys(probe)[:, :2])
        out = np.zeros(len(q), dtype=np.uint32)
        cached = np.zeros(len(q), dtype=np.uint8)
        meta = np.zeros(8, dtype=np.uint32)
        assert hostlib.lib().fqtk_host_direct_memo(C.c_uint32(S), C.c_uint32(L), C.c_uint64(len(keys)), keys.ctypes.data_as(C.c_void_p),
                                                   vals.ctypes.data_as(C.c_void_p), C.c_uint64(len(q)), q.ctypes.data_as(C.c_void_p),
                                                   out.ctypes.data_as(C.c_void_p), cached.ctypes.data_as(C.c_void_p),
                                                   meta.ctypes.data_as(C.c_void_p)) == 0
        i2, b2, n2, _ = lit.assign_batch(probe)
        assert np.array_equal(out, _word(i2, b2, n2))
        assert meta[2] == 2                                            # best needs 2 bits
    assert meta[0] == 2 and meta[7] == (1 << 16 if L <= 8 else 1 << 18 if L == 9 else 1 << 20)


def test_tables_too_wide_for_16_bit_entries_fall_back_to_result_words_without_a_cache():
    # idx needs 15 bits (S = 20000), next 4, best 1 -> 20 bits: 4-byte entries, no LDS cache
    rng = np.random.default_rng(1)
    seen = set()
    while len(seen) < 20000:
        seen.add("".join(rng.choice(list("ACGT"), size=10)))
    barcodes = sorted(seen)
    probe = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(20_000, 10))]
    meta, out, cached, want, _, _ = _plan_and_probe(barcodes, 0, 1, probe)
    assert meta[0] == 4 and meta[4] == 0 and not cached.any()
    assert np.array_equal(out, want)


def _fold(keys):
    """memo_key_of(fold = true) for L <= 10: bases 8-9 ride in the spare bits of lo's nibbles (memo_hash.hpp: kFoldMul)."""
    lo, hi = keys[:, 0].astype(np.uint64), keys[:, 1].astype(np.uint64)
    x = (hi & 7) | (((hi >> 8) & 7) << 8)
    mul = (1 << 3) | (1 << 6) | (1 << 17)
    return (lo | ((x * mul) & 0x08088888)).astype(np.uint32)


@pytest.mark.parametrize("case", ["cfg5", "g_rich"])
def test_reads_with_a_no_call_find_their_entry_in_one_bucket_of_two_slots(case):
    """The direct form's N table (plan_nbuckets): a read's first 16-byte bucket holds its key in either slot, or the
    bucket's SPILL bit sends it to the second choice.  Keys are the folded 4-bit keys, so strings that differ only in
    WHERE their N sits -- one 2-bit index, an N reads as G there -- are told apart (g_rich: barcodes of eight G's)."""
    rng = np.random.default_rng(3)
    if case == "cfg5":
        cfg = synth.CONFIGS[5]
        barcodes, mm, delta = synth.make_barcodes(cfg), cfg.max_mismatches, cfg.min_mismatch_delta
    else:
        barcodes, mm, delta = sorted({"GGGGGGGG" + a + b for a in "ACGT" for b in "ACGT"} | {"ACGTACGTAC", "TTTTTTTTTT"}), 2, 1
    exact = _neighbours_iupac(barcodes, 0)
    cands = [exact]
    for k in range(mm):   # one more base spelled N per round
        nxt = []
        for c in cands[-1][: 40_000]:
            for p in range(c.size):
                if c[p] != ord("N"):
                    d = c.copy()
                    d[p] = ord("N")
                    nxt.append(d)
        cands.append(np.unique(np.stack(nxt), axis=0))
    cand = np.concatenate(cands[1:])
    lit = O.RefLiteral(barcodes, mm, delta, True)
    idx, best, nxt, _ = lit.assign_batch(cand)
    some = idx != O.NONE_IDX
    assert some.sum() > 100
    lo = _fold(_keys(cand))
    assert len(np.unique(lo)) == len(lo)                       # the folded key identifies the string
    vals = _word(idx, best, nxt)
    # absent keys: random strings with an N somewhere that are not stored
    L = len(barcodes[0])
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(30_000, L))].copy()
    rnd[np.arange(30_000), rng.integers(0, L, 30_000)] = ord("N")
    i2, b2, n2, _ = lit.assign_batch(rnd)
    q = np.concatenate([lo, _fold(_keys(rnd))]).astype(np.uint32)
    want = np.concatenate([np.where(some, vals, NONE), _word(i2, b2, n2)]).astype(np.uint32)
    out = np.zeros(len(q), dtype=np.uint32)
    meta = np.zeros(3, dtype=np.uint32)
    klo, kv = np.ascontiguousarray(lo[some]), np.ascontiguousarray(vals[some])
    assert hostlib.lib().fqtk_host_nbuckets(C.c_uint64(len(klo)), klo.ctypes.data_as(C.c_void_p), kv.ctypes.data_as(C.c_void_p),
                                            C.c_uint64(len(q)), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                            meta.ctypes.data_as(C.c_void_p)) == 0
    assert meta[0] == 1 and meta[1] * 2 * 0.6 >= len(klo) * 0.5
    # a random N read absent from the table may still resolve (within mm of a sample): only strings the table was built from count
    stored = set(klo.tolist())
    mask = np.array([int(x) in stored or w == NONE for x, w in zip(q.tolist(), want.tolist())])
    assert np.array_equal(out[mask], want[mask]) and mask[: len(lo)].all()
    assert meta[2] <= 0.2 * len(klo)                          # nearly every key sits in its first bucket
