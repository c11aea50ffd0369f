"""Host logic of the LDS-resident memo (fqtk_amd/csrc/lds_memo_plan.hpp), no GPU: the planner is fed the
memo entries the CPU oracle computes, and the kernel's lookup (lds_memo_lookup, the same arithmetic the
HIP kernel runs) is replayed for every stored key, for near-miss keys and for random keys."""
import ctypes as C

import numpy as np
import pytest

from fqtk_amd import synth
from oracle import oracle as O
from tests import hostlib

NONE = 0xFFFFFFFF


def _keys(strings: np.ndarray) -> np.ndarray:
    """uint8 [n, L] ASCII -> uint32 [n, 4] unfolded 4-bit keys (code = bits 1..2 of the byte, N = 7), eight bases a word."""
    n, L = strings.shape
    code = ((strings.astype(np.uint32) >> 1) & 7)
    keys = np.zeros((n, 4), dtype=np.uint32)
    for k in range(L):   # memo_hash.hpp memo_nibble_shift: byte k&3 of the word, high nibble for bases 4-7
        keys[:, k >> 3] |= code[:, k] << np.uint32(4 * (((k & 3) << 1) | ((k & 7) >> 2)))
    return keys


def _neighbours(barcodes, mm):
    """Every A/C/G/T/N string within `mm` (<= 1 here) substitutions of a sample barcode."""
    L = len(barcodes[0])
    base = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])
    out = [base]
    if mm >= 1:
        for k in range(L):
            for ch in b"ACGTN":
                v = base.copy()
                v[:, k] = ch
                out.append(v)
    return np.unique(np.concatenate(out), axis=0)


def _plan(barcodes, mm, delta, salt_offset=0, salt_trials=8):
    S, L = len(barcodes), len(barcodes[0])
    cand = _neighbours(barcodes, min(mm, 1))
    idx, best, nxt, _ = O.RefLiteral(barcodes, mm, delta, True).assign_batch(cand)
    some = idx != O.NONE_IDX
    cand, vals = cand[some], (idx[some].astype(np.uint32) | (best[some].astype(np.uint32) << 16) | (nxt[some].astype(np.uint32) << 24))
    keys = np.ascontiguousarray(_keys(cand))
    enc = np.ascontiguousarray(np.stack([O.ENC[np.frombuffer(b.upper().encode(), dtype=np.uint8)] for b in barcodes]).astype(np.uint8))
    image = np.zeros(48 * 1024, dtype=np.uint32)
    meta = np.zeros(16, dtype=np.uint32)
    lib = hostlib.lib()
    rc = lib.fqtk_host_plan_lds_memo(C.c_uint32(S), C.c_uint32(L), enc.ctypes.data_as(C.c_void_p), C.c_uint64(len(keys)),
                                     keys.ctypes.data_as(C.c_void_p), np.ascontiguousarray(vals).ctypes.data_as(C.c_void_p),
                                     image.ctypes.data_as(C.c_void_p), C.c_uint64(image.size), meta.ctypes.data_as(C.c_void_p),
                                     C.c_uint32(salt_offset), C.c_int(salt_trials))
    assert rc == 0
    return meta, image, cand, keys, vals


def _lookup(meta, image, keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.zeros(len(keys), dtype=np.uint32)
    hostlib.lib().fqtk_host_lds_memo_lookup(image.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                            C.c_uint64(len(keys)), keys.ctypes.data_as(C.c_void_p),
                                            out.ctypes.data_as(C.c_void_p))
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_baseline_config_tables_plan_and_every_lookup_is_exact(k):
    cfg = synth.CONFIGS[k]
    barcodes = synth.make_barcodes(cfg)
    meta, image, cand, keys, vals = _plan(barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    assert meta[0] == 1, "baseline configs 1-4 are plain ACGT with one mismatch: the LDS form must exist"
    lds_bytes = int(meta[9]) * 4 + 1024 + (cfg.n_samples + 1) * 4
    assert lds_bytes <= 160 * 1024
    # (1) every stored key returns exactly its (idx, best, next)
    assert np.array_equal(_lookup(meta, image, keys), vals)
    # (2) keys that are NOT in the memo return None: two-substitution neighbours and random strings
    rng = np.random.default_rng(k)
    L = cfg.barcode_len
    far = cand[rng.integers(0, len(cand), 20000)].copy()
    for _ in range(2):
        far[np.arange(len(far)), rng.integers(0, L, len(far))] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, len(far))]
    rnd = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(20000, L))]
    probe = np.concatenate([far, rnd])
    idx, best, nxt, _ = O.RefLiteral(barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True).assign_batch(probe)
    want = np.where(idx == O.NONE_IDX, NONE, idx.astype(np.uint32) | (best.astype(np.uint32) << 16) | (nxt.astype(np.uint32) << 24)).astype(np.uint32)
    assert np.array_equal(_lookup(meta, image, _keys(probe)), want)


def test_shapes_the_lds_form_does_not_cover():
    assert _plan(["ACGTACGN", "TTTTGGGG"], 1, 1)[0][0] == 0          # N in a sample
    assert _plan(["ACGTACGR", "TTTTGGGG"], 1, 1)[0][0] == 0          # IUPAC code in a sample
    assert _plan(["ACGTACGT"], 1, 1)[0][0] == 0                      # S = 1 (next = 255)
    # tiny tables still plan, at the minimum size
    meta, image, cand, keys, vals = _plan(["ACGT", "TTTT", "GGCC"], 1, 1)
    assert meta[0] == 1 and meta[1] == 256
    assert np.array_equal(_lookup(meta, image, keys), vals)


@pytest.mark.parametrize("L", [5, 8, 9, 12, 16, 17, 20, 21, 24, 25, 28, 31, 32])
def test_all_key_widths(L):
    rng = np.random.default_rng(L)
    barcodes = sorted({"".join(rng.choice(list("ACGT"), size=L)) for _ in range(60)})
    meta, image, cand, keys, vals = _plan(barcodes, 1, 1)
    assert meta[0] == 1 and meta[6] == (1 if L <= 8 else 2 if L <= 16 else 3 if L <= 24 else 4)
    assert np.array_equal(_lookup(meta, image, keys), vals)
    rnd = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(5000, L))]
    idx, best, nxt, _ = O.RefLiteral(barcodes, 1, 1, True).assign_batch(rnd)
    want = np.where(idx == O.NONE_IDX, NONE, idx.astype(np.uint32) | (best.astype(np.uint32) << 16) | (nxt.astype(np.uint32) << 24)).astype(np.uint32)
    assert np.array_equal(_lookup(meta, image, _keys(rnd)), want)


def test_table_that_needs_every_lds_slot_uses_the_any_size_mapping():
    """384 samples x 20 bases (10+10 dual index): 31 104 entries do not fit 32 768 slots at a workable
    load, and 65 536 slots do not fit LDS; the planner then takes every slot LDS has room for and the
    multiply-shift slot mapping (meta[8] = pow2 flag = 0)."""
    rng = np.random.default_rng(20)
    seen = set()
    while len(seen) < 384:
        seen.add("".join(rng.choice(list("ACGT"), size=20)))
    barcodes = sorted(seen)
    meta, image, cand, keys, vals = _plan(barcodes, 1, 2)
    assert meta[0] == 1 and meta[8] == 0 and 32768 < meta[1] < 65536
    assert int(meta[9]) * 4 + 1024 + 385 * 4 <= 160 * 1024
    assert len(vals) > 0.86 * 32768
    assert np.array_equal(_lookup(meta, image, keys), vals)
    rnd = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(20000, 20))]
    near = cand[rng.integers(0, len(cand), 20000)].copy()
    near[np.arange(20000), rng.integers(0, 20, 20000)] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, 20000)]
    probe = np.concatenate([rnd, near])
    idx, best, nxt, _ = O.RefLiteral(barcodes, 1, 2, True).assign_batch(probe)
    want = np.where(idx == O.NONE_IDX, NONE, idx.astype(np.uint32) | (best.astype(np.uint32) << 16) | (nxt.astype(np.uint32) << 24)).astype(np.uint32)
    assert np.array_equal(_lookup(meta, image, _keys(probe)), want)


def _random_barcodes(n, L, seed):
    rng = np.random.default_rng(seed)
    seen = set()
    while len(seen) < n:
        seen.add("".join(rng.choice(list("ACGT"), size=L)))
    return sorted(seen), rng


@pytest.mark.parametrize("S,L", [(384, 24), (400, 22), (60, 17), (2, 24), (340, 32), (330, 29), (40, 25)])
def test_minimal_perfect_hash_form_for_tables_the_cuckoo_slots_have_no_room_for(S, L):
    """384 samples x 24 bases (12+12 dual index): 37 248 entries are 170 KB of four-byte cuckoo slots at a workable load -- no LDS
    form until round 6.  plan_lds_memo_mph: a hash-and-displace perfect hash (one 16-bit displacement per bucket), three-byte
    entries, ONE candidate per read, verified against its sample's key.  Every stored key resolves to its entry; everything else
    -- neighbours two substitutions away, random strings -- to None."""
    barcodes, rng = _random_barcodes(S, L, 7 * S + L)
    if S == 384 and L == 24:
        assert _plan(barcodes, 1, 2)[0][0] == 0, "the cuckoo form has no room for this table (else this form is not needed)"
    meta, image, cand, keys, vals = _plan(barcodes, 1, 2, salt_trials=-1)
    assert meta[0] == 1 and meta[10] == 1 and meta[6] == (3 if L <= 24 else 4) and meta[8] == 0
    n_slots, t8_off, aux_off, skey_off, buckets = int(meta[1]), int(meta[11]), int(meta[12]), int(meta[4]), int(meta[13]) + 1
    assert len(vals) <= n_slots and buckets & (buckets - 1) == 0
    assert 2 * n_slots <= t8_off and t8_off + n_slots <= aux_off and aux_off + 2 * buckets <= skey_off and skey_off % 16 == 0
    assert int(meta[9]) * 4 == skey_off + (S + 1) * 16
    assert int(meta[9]) * 4 + 1024 + (S + 1) * 4 <= 160 * 1024
    assert np.array_equal(_lookup(meta, image, keys), vals)
    n = 20000
    rnd = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(n, L))]
    near = cand[rng.integers(0, len(cand), n)].copy()
    near[np.arange(n), rng.integers(0, L, n)] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, n)]
    probe = np.concatenate([rnd, near])
    idx, best, nxt, _ = O.RefLiteral(barcodes, 1, 2, True).assign_batch(probe)
    want = np.where(idx == O.NONE_IDX, NONE, idx.astype(np.uint32) | (best.astype(np.uint32) << 16) | (nxt.astype(np.uint32) << 24)).astype(np.uint32)
    assert np.array_equal(_lookup(meta, image, _keys(probe)), want)


def test_shapes_the_minimal_perfect_hash_form_does_not_cover():
    assert _plan(_random_barcodes(40, 16, 1)[0], 1, 2, salt_trials=-1)[0][0] == 0    # two key words: the cuckoo form's
    assert _plan(_random_barcodes(384, 32, 2)[0], 1, 2, salt_trials=-1)[0][0] == 0   # 16+16 x 384: 49 536 entries, no room even at three bytes
    assert _plan(_random_barcodes(512, 18, 3)[0], 1, 2, salt_trials=-1)[0][0] == 0   # S + 1 > 512: no room in the index field
    assert _plan(["ACGTACGTACGTACGTACGN", "TTTTGGGGTTTTGGGGTTTT"], 1, 1, salt_trials=-1)[0][0] == 0   # N in a sample


def test_codes_of_reads_with_ambiguity_codes_and_junk_bytes():
    """The LDS forms look a read with ambiguity codes up under N's code in their places (csrc/memo_hash.hpp recode_flagged_bytes:
    against plain A/C/G/T samples a code of two or more bases mismatches every sample, as N does; U is T) and only a byte of no
    IUPAC meaning -- mask 0: it matches everything (bitenc.rs:432-459) -- is left for the scan.  Every byte value at every position
    of a word, alone and among other odd bytes, against the oracle's encoding table; pad positions (masks cut down) read as absent."""
    rng = np.random.default_rng(6)
    others = np.frombuffer(b"ACGTNacgtn.RYKMSWBDHVUurykm#xX@[`{\x00\x7f\x80\xff-*0 ", dtype=np.uint8)
    rows = []
    for j in range(4):
        for v in range(256):
            for k in range(6):
                b = others[rng.integers(0, len(others), 4)] if k else np.frombuffer(b"ACGT", dtype=np.uint8).copy()
                b = b.copy()
                b[j] = v
                rows.append(b)
    rows.append(np.frombuffer(b"....", dtype=np.uint8))
    rows.append(np.frombuffer(b"NnNn", dtype=np.uint8))
    by = np.stack(rows).astype(np.uint8)
    words = np.ascontiguousarray(by).view("<u4").reshape(-1).copy()
    enc = O.ENC[by]                                                    # 4-bit masks: A 1, C 2, G 4, T 8
    want_junk = enc == 0
    single = {1: 0, 2: 1, 8: 2, 4: 3}                                  # the key's codes: A 0, C 1, T 2, G 3; anything wider: N's 7
    want_code = np.vectorize(lambda m: single.get(int(m), 7))(enc)
    lib = hostlib.lib()
    for code_mask, byte_mask, keep in ((0x07070707, 0xDFDFDFDF, 4), (0x00000707, 0x0000DFDF, 2), (0x00000007, 0x000000DF, 1)):
        codes = np.zeros(len(words), dtype=np.uint32)
        junk = np.zeros(len(words), dtype=np.uint32)
        lib.fqtk_host_recode_words(words.ctypes.data_as(C.c_void_p), C.c_uint64(len(words)), C.c_uint32(code_mask), C.c_uint32(byte_mask),
                                   codes.ctypes.data_as(C.c_void_p), junk.ctypes.data_as(C.c_void_p))
        got_code = codes.view(np.uint8).reshape(-1, 4)
        got_junk = junk.view(np.uint8).reshape(-1, 4)
        assert np.all((got_junk == 0) | (got_junk == 0x80))
        assert np.array_equal(got_junk[:, :keep] != 0, want_junk[:, :keep])
        assert not got_junk[:, keep:].any() and not got_code[:, keep:].any()          # pad positions: absent
        clean = ~want_junk[:, :keep].any(axis=1)                                       # (a read with a junk byte goes to the scan)
        assert np.array_equal(got_code[:, :keep][clean], want_code[:, :keep][clean])
