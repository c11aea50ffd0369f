"""GPU parity of the record pipeline (include/fqtk_demux.h) through the C ABI: FASTQ text in, BGZF members out, compared
record for record with the reference's semantics restated on the host -- assignments from the oracle
(oracle/ref_literal.c), headers from host/header.hpp (write_header_internal, demux.rs:171-267), records as
SampleWriters::write lays them out (:396-415), file order as demux.rs:674-688."""
import gzip
import zlib

import numpy as np
import pytest

from fqtk_amd import BarcodeMatcher
from fqtk_amd.demux import Demuxer, DemuxChunkError, parse_read_structure
from oracle import oracle as O
from tests import hostlib as H

pytestmark = pytest.mark.gpu


def _span(seg, read_len):
    off, length, _ = seg
    hi = min(off + length, read_len) if length >= 0 else read_len
    return min(off, hi), hi


def expected_files(barcodes, mm, delta, structures, output_types, templates, skip_short=False):
    """templates: list of per-input (header, bases, quals).  Returns (files[(S+1)*F] as bytes, counts, skipped)."""
    segs = [parse_read_structure(r) for r in structures]
    by = {k: [(i, s) for i, ss in enumerate(segs) for s in ss if s[2] == k] for k in "TBMC"}
    file_segs = [(k, j, i, s) for k in "TBMC" if k in output_types for j, (i, s) in enumerate(by[k])]
    F, S = len(file_segs), len(barcodes)
    files = [bytearray() for _ in range((S + 1) * F)]
    lit = O.RefLiteral(barcodes, mm, delta, True)
    counts = np.zeros(S + 1, dtype=np.uint64)
    skipped = 0
    min_len = [sum(l if l >= 0 else 1 for _, l, _ in ss) for ss in segs]
    for tpl in templates:
        if any(len(tpl[i][1]) < min_len[i] for i in range(len(segs))):
            assert skip_short
            skipped += 1
            continue
        def seg_text(i, s, what):
            lo, hi = _span(s, len(tpl[i][1]))
            return tpl[i][what][lo:hi]
        bsegs = [seg_text(i, s, 1) for i, s in by["B"]]
        msegs = [seg_text(i, s, 1) for i, s in by["M"]]
        bc = "".join(bsegs).encode()
        L = len(barcodes[0])
        if len(bc) == L:
            obs = np.frombuffer(bc, dtype=np.uint8).reshape(1, L)
            idx, _, _, _ = lit.assign_batch(obs)
            s_idx = int(idx[0])
            s_idx = S if s_idx == 0xFFFF else s_idx
        else:
            assert len(bc) < L, "longer barcodes are the matcher's length error: not part of these cases"
            s_idx = S
        counts[s_idx] += 1
        for f, (k, j, i, s) in enumerate(file_segs):
            head = H.write_header(j + 1, tpl[0][0], bsegs, msegs)
            files[s_idx * F + f] += (head + "\n" + seg_text(i, s, 1) + "\n+\n" + seg_text(i, s, 2) + "\n").encode()
    return [bytes(f) for f in files], counts, skipped


def make_templates(rng, n, barcodes, structures, header_kind=0, junk=0.0, short_every=0):
    segs = [parse_read_structure(r) for r in structures]
    out = []
    # where the sample barcode segments are, to plant real barcodes
    bpos = [(i, s) for i, ss in enumerate(segs) for s in ss if s[2] == "B"]
    for t in range(n):
        tpl = []
        bc = barcodes[int(rng.integers(0, len(barcodes)))] if rng.random() < 0.85 else "".join(rng.choice(list("ACGT"), len(barcodes[0])))
        bc = list(bc)
        if rng.random() < 0.2:
            bc[int(rng.integers(0, len(bc)))] = "ACGTN"[int(rng.integers(0, 5))]
        bc = "".join(bc)
        taken = 0
        reads = []
        for i, ss in enumerate(segs):
            fixed = sum(l for _, l, _ in ss if l >= 0)
            var = any(l < 0 for _, l, _ in ss)
            rl = fixed + (int(rng.integers(1, 40)) if var else int(rng.integers(0, 3)))
            if short_every and t % short_every == short_every - 1 and i == len(segs) - 1:
                rl = max(0, fixed - 1)
            reads.append(list(rng.choice(list("ACGT"), rl)))
        for i, s in bpos:
            lo, hi = _span(s, len(reads[i]))
            if s[1] < 0:
                hi = min(hi, lo + len(bc) - taken)
                reads[i] = reads[i][:hi]
            for k in range(lo, hi):
                if taken < len(bc):
                    reads[i][k] = bc[taken]
                    taken += 1
        for i in range(len(segs)):
            bases = "".join(reads[i])
            quals = "".join(rng.choice(list("#5?FI"), len(bases)))
            if header_kind == 0:
                head = f"inst:1:fc:{1 + t % 4}:{t}:{int(rng.integers(0, 99999))}:{t * 7} {i + 1}:N:0:{'ACGT' if t % 3 else '0'}"
            elif header_kind == 1:
                head = f"q{t}"
            elif header_kind == 2:
                head = f"r{t} some comment" if t % 2 else f"r{t} a:b:"
            else:
                head = f"a:b:c:d:e:f:g:UMI{t} 1:Y:18:" if t % 2 else f"x{t}:y 2:N:0:7"
            tpl.append((head, bases, quals))
        out.append(tpl)
    return out


def texts_of(templates, lo, hi, n_inputs, crlf=False):
    nl = "\r\n" if crlf else "\n"
    return [("".join(f"@{t[i][0]}{nl}{t[i][1]}{nl}+{nl}{t[i][2]}{nl}" for t in templates[lo:hi])).encode() for i in range(n_inputs)]


def run_case(barcodes, mm, delta, structures, output_types, templates, chunk, skip_short=False, carry=True, crlf=False, in_flight=3):
    m = BarcodeMatcher(barcodes, mm, delta, device=0)
    d = Demuxer(m, structures, output_types, skip_too_few_bases=skip_short, max_chunk_templates=max(chunk, 1), carry_blocks=carry)
    chunks = [(texts_of(templates, lo, min(lo + chunk, len(templates)), len(structures), crlf), min(chunk, len(templates) - lo))
              for lo in range(0, len(templates), chunk)]
    got = d.run(chunks, in_flight=in_flight)
    want, counts, skipped = expected_files(barcodes, mm, delta, structures, output_types, templates, skip_short)
    assert len(got) == len(want)
    for c, (g, w) in enumerate(zip(got, want)):
        assert gzip.decompress(g) == w, f"file column {c}"
    assert np.array_equal(d.counts(), counts)
    assert d.skipped == skipped
    return d, got


BARCODES8 = ["AAAAAAAA", "CCCCCCCC", "GGGGGGGG", "TTTTTTTT", "ACGTACGT", "TGCATGCA", "AACCGGTT", "GATTACAG"]


@pytest.mark.parametrize("structures, types", [
    (["8B20T"], "T"), (["8B+T"], "T"), (["4B4M+T", "4B+T"], "TBM"), (["+T", "+T", "8B"], "T"),
    (["+T", "+T", "4B", "4B"], "TB"), (["3M2S3B10T5C", "5B+T"], "TBMC"), (["8B"], "B"), (["+T", "8B"], "TB"),
])
@pytest.mark.parametrize("header_kind", [0, 1, 2, 3])
def test_small_runs_of_every_structure_and_header_shape(structures, types, header_kind):
    rng = np.random.default_rng(zlib.crc32(repr((structures, types, header_kind)).encode()))
    tpl = make_templates(rng, 700, BARCODES8, structures, header_kind)
    run_case(BARCODES8, 1, 2, structures, types, tpl, chunk=256)


def test_blocks_fill_carry_over_chunks_and_are_flushed_at_the_end():
    rng = np.random.default_rng(1)
    structures = ["8B+T", "+T"]
    tpl = make_templates(rng, 9000, BARCODES8[:3], structures)      # ~1 MB per file: many full blocks, partial ones carried
    d, got = run_case(BARCODES8[:3], 1, 2, structures, "T", tpl, chunk=1000)
    # whole blocks of 65280 bytes except the last one of every file
    for g in got:
        sizes = []
        p = 0
        while p < len(g):
            bsize = int.from_bytes(g[p + 16:p + 18], "little") + 1
            sizes.append(int.from_bytes(g[p + bsize - 4:p + bsize], "little"))
            p += bsize
        assert sizes[-1] == 0 and all(s == 65280 for s in sizes[:-2])
    run_case(BARCODES8[:3], 1, 2, structures, "T", tpl, chunk=1000, carry=False)      # every chunk ends its blocks
    run_case(BARCODES8[:3], 1, 2, structures, "T", tpl, chunk=9000)                   # one chunk


def test_chunk_sizes_around_the_tile_and_one_template_chunks():
    rng = np.random.default_rng(2)
    structures = ["8B10T"]
    tpl = make_templates(rng, 2100, BARCODES8, structures, header_kind=1)
    for chunk in (1, 1023, 1024, 1025, 2100):
        if chunk == 1:
            run_case(BARCODES8, 1, 2, structures, "T", tpl[:40], chunk=1)
        else:
            run_case(BARCODES8, 1, 2, structures, "T", tpl, chunk=chunk)


def test_crlf_input_and_too_few_bases_skipping():
    rng = np.random.default_rng(4)
    structures = ["8B12T", "6T"]
    tpl = make_templates(rng, 1500, BARCODES8, structures, short_every=7)
    run_case(BARCODES8, 1, 2, structures, "T", tpl, chunk=400, skip_short=True)
    run_case(BARCODES8, 1, 2, structures, "T", make_templates(rng, 500, BARCODES8, structures), chunk=200, crlf=True)
    with pytest.raises(DemuxChunkError) as e:
        run_case(BARCODES8, 1, 2, structures, "T", tpl, chunk=400, skip_short=False)
    assert e.value.kind == 5 and e.value.template == 6 and e.value.input_index == 1


def test_chunks_that_end_their_blocks_keep_three_in_flight():
    """carry_blocks = 0 (`--devices a,b,..`): every chunk closes at least one block AND flushes a rest per file -- two slabs
    per file and chunk.  They used to come out of the file's three persistent slabs, which the formatter of chunk k + 1
    overwrote while the compressor still read chunk k's; now such blocks live in the chunk's own slabs and
    three chunks overlap without waiting for each other."""
    rng = np.random.default_rng(11)
    structures = ["8B+T", "+T"]
    bcs = BARCODES8[:3]
    segs = 24_000
    tpl = make_templates(rng, segs, bcs, structures, header_kind=1)
    # long second reads: ~5 blocks per file and chunk, so that DEFLATE (stream B) runs far behind formatting (stream A)
    tpl = [[t[0], (t[1][0], t[1][1] * 12, t[1][2] * 12)] for t in tpl]
    for in_flight in (3, 1):
        d, got = run_case(bcs, 1, 2, structures, "T", tpl, chunk=6000, carry=False, in_flight=in_flight)
    sizes = []
    g = got[1]
    p = 0
    while p < len(g):
        bsize = int.from_bytes(g[p + 16:p + 18], "little") + 1
        sizes.append(int.from_bytes(g[p + bsize - 4:p + bsize], "little"))
        p += bsize
    assert sizes.count(65280) >= 8 and sum(1 for x in sizes if 0 < x < 65280) == 4   # one flushed rest per chunk


def test_errors_of_one_template_come_in_the_reference_order():
    """The reference zips its per-input iterators (demux.rs:285-343, 946-951): input 0's record is parsed and
    length-checked before input 1's is looked at.  So for a template broken in several inputs the LOWER input's error is
    the fatal one, whatever its stage (a parse error in input 1 used to outrank too-few-bases in input 0)."""
    structures = ["8B12T", "6T"]
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, structures, "T", max_chunk_templates=8)
    def texts(r0, r1):
        a = b"".join(b"@t%d\n%s\n+\n%s\n" % (k, b"ACGTACGT" + b"A" * 12, b"I" * 20) for k in range(8)).split(b"\n")
        b = b"".join(b"@t%d\n%s\n+\n%s\n" % (k, b"ACGTAC", b"IIIIII") for k in range(8)).split(b"\n")
        r0(a)
        r1(b)
        return [b"\n".join(a), b"\n".join(b)]
    def short(lines, t, n):
        lines[4 * t + 1] = lines[4 * t + 1][:n]
        lines[4 * t + 3] = lines[4 * t + 3][:n]
    def no_at(lines, t):
        lines[4 * t] = b"x" + lines[4 * t][1:]
    # input 0 too short, input 1 malformed, same template: the reference panics on input 0's length
    d.submit(0, texts(lambda a: short(a, 5, 10), lambda b: no_at(b, 5)), 8)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert (e.value.kind, e.value.template, e.value.input_index) == (5, 5, 0)
    # the other way round: input 0's parse error
    d.submit(0, texts(lambda a: no_at(a, 5), lambda b: short(b, 5, 3)), 8)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert (e.value.kind, e.value.template, e.value.input_index) == (1, 5, 0)
    # an earlier template always wins, whatever the input
    d.submit(0, texts(lambda a: short(a, 5, 10), lambda b: no_at(b, 4)), 8)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert (e.value.kind, e.value.template, e.value.input_index) == (1, 4, 1)


def test_malformed_text_is_reported_with_its_first_template():
    rng = np.random.default_rng(5)
    structures = ["8B12T"]
    tpl = make_templates(rng, 300, BARCODES8, structures, header_kind=1)
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, structures, "T", max_chunk_templates=300)
    good = texts_of(tpl, 0, 300, 1)[0]
    lines = good.split(b"\n")
    for line_no, kind in ((4 * 17, 1), (4 * 33 + 2, 2)):
        bad = list(lines)
        bad[line_no] = b"x" + bad[line_no][1:]
        d.submit(0, [b"\n".join(bad)], 300)
        with pytest.raises(DemuxChunkError) as e:
            d.collect(0)
        assert (e.value.kind, e.value.template) == (kind, line_no // 4)
    bad = list(lines)
    bad[4 * 50 + 3] = bad[4 * 50 + 3] + b"I"
    d.submit(0, [b"\n".join(bad)], 300)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert (e.value.kind, e.value.template) == (3, 50)
    d.submit(0, [good], 299)                      # one record more than announced
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert e.value.kind == 4
    # header errors of write_header_internal
    tpl2 = [list(t) for t in tpl]
    tpl2[77][0] = ("q77 1:N:0:A:B", tpl2[77][0][1], tpl2[77][0][2])
    d.submit(0, texts_of(tpl2, 0, 300, 1), 300)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert (e.value.kind, e.value.template, e.value.detail) == (7, 77, 3)


def test_barcode_longer_than_expected_is_the_reference_panic():
    structures = ["+B"]
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, structures, "B", max_chunk_templates=16)
    text = b"@a\nAAAAAAAA\n+\nIIIIIIII\n@b\nAAAAAAAAC\n+\nIIIIIIIII\n"
    d.submit(0, [text], 2)
    with pytest.raises(DemuxChunkError) as e:
        d.collect(0)
    assert e.value.kind == 6 and e.value.template == 1 and "length (9) differs from expected barcode (AAAAAAAA) length (8)" in e.value.message


def test_a_million_templates_cfg3_shape_counts_and_streams():
    """cfg 3's shape (dual index 8+8, 384 samples) at a size where every file closes many blocks in every chunk."""
    from fqtk_amd import synth
    cfg = synth.CONFIGS[3]
    w = synth.Workload(cfg)
    n = 200_000
    obs = w.fill_host(0, n)                                        # n x 16 barcode bytes
    rng = np.random.default_rng(6)
    r1 = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (n, 60))
    def fastq(name_fn, bases):
        parts = []
        for t in range(n):
            b = bases[t].tobytes().decode()
            parts.append(f"@{name_fn(t)}\n{b}\n+\n{'I' * len(b)}\n")
        return "".join(parts).encode()
    name = lambda t: f"inst:1:fc:1:{t}:5:7 1:N:0:0"
    texts = [fastq(name, r1), fastq(name, obs[:, :8]), fastq(name, obs[:, 8:16])]
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=0)
    d = Demuxer(m, ["+T", "8B", "8B"], "T", max_chunk_templates=n)
    # chunks of 65536 templates cut at record boundaries
    def cut(text, lo, hi):
        lines = text.split(b"\n")
        return b"\n".join(lines[4 * lo:4 * hi]) + b"\n"
    chunks = [([cut(t, lo, min(lo + 65536, n)) for t in texts], min(65536, n - lo)) for lo in range(0, n, 65536)]
    got = d.run(chunks)
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    idx, _, _, counts = lit.assign_batch(obs)
    assert np.array_equal(d.counts(), counts)
    S = cfg.n_samples
    sample_of = np.where(idx == 0xFFFF, S, idx).astype(np.int64)
    total = 0
    for s in range(S + 1):
        text = gzip.decompress(got[s]).decode()
        recs = text.split("\n")[:-1]
        assert len(recs) == 4 * int(counts[s])
        ts = np.nonzero(sample_of == s)[0]
        for k in (0, len(ts) // 2, len(ts) - 1) if len(ts) else ():
            t = int(ts[k])
            bc = obs[t, :8].tobytes().decode() + "+" + obs[t, 8:16].tobytes().decode()
            assert recs[4 * k] == f"@inst:1:fc:1:{t}:5:7 1:N:0:{bc}" and recs[4 * k + 1] == r1[t].tobytes().decode()
        total += len(recs) // 4
    assert total == n
    print("stage seconds:", d.stage_seconds())


def _bgzf_of(text, member, rng):
    """BGZF bytes of `text` in members of `member` bytes of text each (levels vary), without the EOF marker."""
    import struct
    out = b""
    for o in range(0, len(text), member):
        piece = text[o:o + member]
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15)
        payload = c.compress(piece) + c.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(piece), len(piece)))
    return out


@pytest.mark.parametrize("member, feed_members", [(700, 5), (65280, 2), (4096, 1000)])
def test_fed_bgzf_members_give_the_files_text_chunks_give(member, feed_members, monkeypatch):
    """fqtk_demuxer_feed / submit_fed / fed_tail through the C ABI: members of every input go to the device compressed, chunks are
    cut by line counts out of windows of whole members (which never line up with the chunks: inputs of 150- and 8-base reads,
    members of 700 to 65 280 bytes), and the files must equal the ones the same templates give as host text -- also when the
    fed text changes arena every few feeds, the last line has no newline, and blank lines follow the last record."""
    monkeypatch.setenv("FQTK_FED_ARENA_MIN", "60000")
    rng = np.random.default_rng(member)
    structures, types = ["8B", "+T", "6M+T"], "TBM"
    templates = make_templates(rng, 5000, BARCODES8, structures, header_kind=1)
    texts = texts_of(templates, 0, len(templates), len(structures))
    texts[1] = texts[1][:-1]            # the last line of input 1 ends with the file
    texts[2] = texts[2] + b"\n\n"       # blank lines behind input 2's last record
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, structures, types, max_chunk_templates=700)
    blobs = [_bgzf_of(t, member, rng) for t in texts]
    # feed in runs of `feed_members` members per input, the inputs in turn; cut a chunk whenever every input has its lines
    starts = []
    for b in blobs:
        pos, s = 0, []
        while pos < len(b):
            s.append(pos)
            pos += int.from_bytes(b[pos + 16:pos + 18], "little") + 1
        starts.append(s + [len(b)])
    at = [0] * len(blobs)
    fed = [0] * len(blobs)
    done = [False] * len(blobs)
    files = [bytearray() for _ in range(d.n_files)]
    taken, slot, pending, ends = 0, 0, [], None
    chunk = 700
    while True:
        for i, b in enumerate(blobs):
            if not done[i] and fed[i] < 4 * (taken + chunk):
                hi = min(at[i] + feed_members, len(starts[i]) - 1)
                done[i] = hi == len(starts[i]) - 1
                fed[i] = d.feed(i, b[starts[i][at[i]]:starts[i][hi]], last=done[i])
                at[i] = hi
        ready = all(done[i] or fed[i] >= 4 * (taken + chunk) for i in range(len(blobs)))
        if not ready:
            continue
        n = min(chunk, min(fed[i] // 4 - taken for i in range(len(blobs))))
        if n <= 0:
            break
        if len(pending) == 3:
            got, ends = d.collect_fed(pending.pop(0))
            for c, x in enumerate(got):
                files[c] += x
        d.submit_fed(slot % 3, n)
        pending.append(slot % 3)
        slot += 1
        taken += n
    while pending:
        got, ends = d.collect_fed(pending.pop(0))
        for c, x in enumerate(got):
            files[c] += x
    for c, x in enumerate(d.flush()):
        files[c] += x
    assert taken == len(templates)
    want, counts, _ = expected_files(BARCODES8, 1, 2, structures, types, templates)
    for c, w in enumerate(want):
        assert gzip.decompress(bytes(files[c]) + H_BGZF_EOF) == w, f"file column {c}"
    assert np.array_equal(d.counts(), counts)
    # what lies behind the last record: the newline the device added (+ input 2's blank lines)
    # (input 1's text ended without one: the added newline IS its last record's)
    assert d.fed_tail(0, ends[0]) == b"\n" and d.fed_tail(1, ends[1]) == b"" and d.fed_tail(2, ends[2]) == b"\n\n\n"
    # a corrupt member is refused with its number
    bad = bytearray(_bgzf_of(texts[0][:3000], 1000, rng))
    bad[40] ^= 0x55
    d2 = Demuxer(BarcodeMatcher(BARCODES8, 1, 2, device=0), structures, types, max_chunk_templates=700)
    with pytest.raises(Exception, match="corrupt BGZF block 0"):
        d2.feed(0, bytes(bad), last=True)


from fqtk_amd.demux import BGZF_EOF as H_BGZF_EOF  # noqa: E402


def test_a_serial_deflate_stream_decoded_in_chunks_on_the_device():
    """fqtk_demuxer_stream_decode / stream_commit: one DEFLATE stream (what `gzip` writes) cut at block boundaries, every chunk
    decoded by a wavefront without the 32 KiB before it (most of a chunk's symbols are references into that unknown window),
    windows handed down the chain, text / line counts / CRC-32 as zlib gives them -- in two commits (the second starts from the
    window the first one left), with a chunk that is refused in between."""
    rng = np.random.default_rng(21)
    structures = ["+T"]
    templates = make_templates(rng, 6000, BARCODES8, structures, header_kind=0)
    text = texts_of(templates, 0, len(templates), 1)[0]
    parts = [text[o:o + 50000] for o in range(0, len(text), 50000)]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp, bounds = b"", [0]
    for p in parts[:-1]:
        comp += c.compress(p) + c.flush(zlib.Z_SYNC_FLUSH)
        bounds.append(len(comp) * 8)
    comp += c.compress(parts[-1]) + c.flush()
    assert len(bounds) >= 8
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, ["8B+T"], "T", max_chunk_templates=4096)
    half = len(bounds) // 2
    # first stretch: chunks 0 .. half-1, plus a chunk that starts at a place that is no block boundary (it must not be accepted)
    chunks = [(bounds[k], bounds[k + 1]) for k in range(half)] + [(bounds[half] + 3, bounds[half + 1])]
    ends = d.stream_decode(0, comp, chunks)
    for k in range(half):
        assert ends[k][0] == 0 and ends[k][3] == bounds[k + 1] and ends[k][2] == len(parts[k]), (k, ends[k])
    assert ends[half][0] != 0 or ends[half][3] != bounds[half + 1] or ends[half][2] != len(parts[half])
    fed, crc1, n1 = d.stream_commit(0, half, member_start=True, last=False)
    assert n1 == sum(len(p) for p in parts[:half]) and crc1 == zlib.crc32(b"".join(parts[:half]))
    assert fed == b"".join(parts[:half]).count(b"\n")
    # second stretch: the rest, from where the first ended
    chunks = [(bounds[k], bounds[k + 1] if k + 1 < len(bounds) else None) for k in range(half, len(bounds))]
    ends = d.stream_decode(0, comp, chunks)
    assert all(e[0] == 0 for e in ends) and ends[-1][1] == 1
    fed, crc2, n2 = d.stream_commit(0, len(chunks), member_start=False, last=True)
    assert n1 + n2 == len(text) and fed == text.count(b"\n") + 1
    assert crc2 == zlib.crc32(b"".join(parts[half:]))
    assert d.fed_tail(0, 0, cap=len(text) + 16) == text + b"\n"


def _accept(ends, n_bits):
    """The prefix of chunks whose block boundaries chain (what `fqtk demux` accepts of a stretch): (n_accept, end bit, final)."""
    n = 0
    for k, e in enumerate(ends):
        if k and ends[k - 1][3] != e[4]:
            break
        if e[0] == 0 and e[3] <= n_bits:
            n = k + 1
            if e[1]:
                break
            continue
        if e[0] in (7, 8) and e[5] > 0:
            n = k + 1
        break
    assert n > 0, ends[:2]
    return n, ends[n - 1][3], bool(ends[n - 1][1] and ends[n - 1][0] == 0)


@pytest.mark.parametrize("level,chunk_bytes", [(6, 4096), (1, 8192), (9, 4096)])
def test_a_serial_stream_cut_on_the_device_at_block_starts_it_finds(level, chunk_bytes):
    """fqtk_demuxer_stream_scan: the device looks for the block starts itself (a lane per bit position), cuts the stretch into
    chunks there, decodes them; the chain of block boundaries says which count.  Stretch after stretch from the bit the last one
    was verified up to -- one of them decoded ELSEWHERE (zlib here, the sequential decoder in `fqtk demux`) and handed over as text,
    with the window the device gives out and the window it is given back -- the text, line counts and CRC-32 are zlib's."""
    rng = np.random.default_rng(300 + level)
    templates = make_templates(rng, 24000, BARCODES8, ["+T"], header_kind=0)
    text = texts_of(templates, 0, len(templates), 1)[0]
    comp = zlib.compress(text, level)[2:-4]          # ONE raw DEFLATE stream, blocks wherever zlib put them
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, ["8B+T"], "T", max_chunk_templates=4096)
    n_bits = len(comp) * 8
    n_slots = 32
    at, done, crc, stretches, elsewhere, chunks_seen = 0, 0, 0, 0, 0, 0
    while True:
        b0 = (at // 8) & ~3
        piece = comp[b0:b0 + n_slots * chunk_bytes + 4096]
        to_end = b0 + len(piece) == len(comp)
        if stretches == 2:   # this stretch goes to a decoder of the caller's: window out, text + window back
            window = d.stream_window(0)
            assert window[32768 - min(done, 32768):] == text[max(0, done - 32768):done]
            ends = d.stream_scan(0, piece, at - b0 * 8, chunk_bytes, n_slots, to_end)     # (only to learn where blocks end; not committed)
            n_acc, end_bit, final = _accept(ends, len(piece) * 8)
            take = sum(e[2] for e in ends[:n_acc])
            got = text[done:done + take]
            last = final
            fed, c = d.stream_commit_text(0, got, None if final else text[max(0, done + take - 32768):done + take].rjust(32768, b"\0"), last)
            assert c == zlib.crc32(got)
            elsewhere += 1
        else:
            ends = d.stream_scan(0, piece, at - b0 * 8, chunk_bytes, n_slots, to_end)
            assert ends[0][4] == at - b0 * 8 and ends[0][0] == 0, ends[0]
            chunks_seen += len(ends)
            n_acc, end_bit, final = _accept(ends, len(piece) * 8)
            last = final
            fed, c, take = d.stream_commit(0, n_acc, member_start=(at == 0), last=last)
            assert take == sum(e[2] for e in ends[:n_acc])
            assert c == zlib.crc32(text[done:done + take]), (stretches, n_acc)
        crc = zlib.crc32(text[done:done + take], crc)
        done += take
        at = b0 * 8 + end_bit
        stretches += 1
        assert fed == text[:done].count(b"\n") + (1 if last else 0)
        if last:
            break
        assert stretches < 500
    assert done == len(text) and crc == zlib.crc32(text) and stretches >= 3 and elsewhere == 1
    assert chunks_seen > stretches - elsewhere          # (block starts were found: stretches of several chunks)
    assert d.fed_tail(0, 0, cap=len(text) + 16) == text + b"\n"


def test_chunks_that_run_out_of_room_end_at_a_block_boundary_and_the_stream_goes_on_from_there():
    """a chunk that expands beyond its room for symbols must not end the run.  Text that deflates 200 : 1 with
    room for 2 symbols per compressed byte: a chunk reports FQTK_INFLATE_ERR_OUTPUT together with the last block boundary it
    reached, the caller takes it up to there and goes on -- with more room -- from that bit."""
    rng = np.random.default_rng(77)
    rec = b"@same:1:1 1:N:0:0\nACGTACGTAAAACCCCGGGGTTTT\n+\nFFFFFFFFFFFFFFFFFFFFFFFF\n"
    text = rec * 60_000                                # 4 MB that deflate to ~ 20 KB
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b""
    for o in range(0, len(text), 1 << 18):            # (a sync flush every 256 KiB: blocks the decoder can stop at)
        comp += c.compress(text[o:o + (1 << 18)]) + c.flush(zlib.Z_SYNC_FLUSH)
    comp += c.flush()
    assert len(text) // len(comp) > 100
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, ["8B+T"], "T", max_chunk_templates=4096)
    at, done, spb, seen_overflow = 0, 0, 2, 0
    for _ in range(200):
        b0 = (at // 8) & ~3
        piece = comp[b0:]
        ends = d.stream_scan(0, piece, at - b0 * 8, 4096, 4, True, sym_per_byte=spb)
        if ends[0][0] == 7 and ends[0][5] == 0:       # not even one block fits: more room, again
            spb *= 4
            seen_overflow += 1
            continue
        seen_overflow += any(e[0] == 7 for e in ends)
        n_acc, end_bit, final = _accept(ends, len(piece) * 8)
        fed, crc, take = d.stream_commit(0, n_acc, member_start=(at == 0), last=final)
        assert crc == zlib.crc32(text[done:done + take])
        done += take
        at = b0 * 8 + end_bit
        if final:
            break
    assert done == len(text) and seen_overflow >= 1


@pytest.mark.parametrize("n_dev, member", [(2, 900), (3, 65280)])
def test_fed_text_of_several_home_demuxers_runs_on_any_of_them(n_dev, member, monkeypatch):
    """SURVEY 8e for compressed inputs (`fqtk demux --devices a,b,..`): input i is fed to its HOME demuxer (i mod G), chunks are cut out
    of the homes' texts in order (fqtk_demuxer_fed_cut) and chunk k is run by demuxer k mod G (fqtk_demuxer_submit_windows), which
    copies the windows that are not its own device to device.  G demuxers on ONE GPU here -- own arenas, streams and slots each --;
    the files (every chunk ends its blocks: carry_blocks = 0) must hold what the same templates give as host text, in order, also while
    the fed text changes arena every few feeds."""
    monkeypatch.setenv("FQTK_FED_ARENA_MIN", "60000")
    rng = np.random.default_rng(100 + n_dev)
    structures, types = ["8B", "+T", "6M+T"], "TBM"
    templates = make_templates(rng, 6000, BARCODES8, structures, header_kind=1)
    texts = texts_of(templates, 0, len(templates), len(structures))
    texts[1] = texts[1][:-1]
    blobs = [_bgzf_of(t, member, rng) for t in texts]
    ms = [BarcodeMatcher(BARCODES8, 1, 2, device=0) for _ in range(n_dev)]
    ds = [Demuxer(m, structures, types, max_chunk_templates=500, carry_blocks=False) for m in ms]
    home = [i % n_dev for i in range(len(blobs))]
    starts = []
    for b in blobs:
        pos, s = 0, []
        while pos < len(b):
            s.append(pos)
            pos += int.from_bytes(b[pos + 16:pos + 18], "little") + 1
        starts.append(s + [len(b)])
    at, fed, done = [0] * len(blobs), [0] * len(blobs), [False] * len(blobs)
    files = [bytearray() for _ in range(ds[0].n_files)]
    taken, k, pending, chunk = 0, 0, [], 500
    while True:
        for i, b in enumerate(blobs):
            if not done[i] and fed[i] < 4 * (taken + 2 * chunk):
                hi = min(at[i] + 7, len(starts[i]) - 1)
                done[i] = hi == len(starts[i]) - 1
                fed[i] = ds[home[i]].feed(i, b[starts[i][at[i]]:starts[i][hi]], last=done[i])
                at[i] = hi
        if not all(done[i] or fed[i] >= 4 * (taken + chunk) for i in range(len(blobs))):
            continue
        n = min(chunk, min(fed[i] // 4 - taken for i in range(len(blobs))))
        if n <= 0:
            break
        # (cut and submitted at once: a window pins its home's text, and a feed of this same thread that had to move it would wait for
        #  ever -- the CLI tests, where feeders and submitters are threads, hold windows across feeds)
        wins = [ds[home[i]].fed_cut(i, n) for i in range(len(blobs))]
        taken += n
        if len(pending) == 3 * n_dev:
            g, slot = pending.pop(0)
            for c, x in enumerate(ds[g].collect_fed(slot)[0]):
                files[c] += x
        g, slot = k % n_dev, (k // n_dev) % 3
        ds[g].submit_windows(slot, wins, n)
        pending.append((g, slot))
        k += 1
    while pending:
        g, slot = pending.pop(0)
        for c, x in enumerate(ds[g].collect_fed(slot)[0]):
            files[c] += x
    for d in ds:
        for c, x in enumerate(d.flush()):
            files[c] += x
    assert taken == len(templates) and k >= 12
    want, counts, _ = expected_files(BARCODES8, 1, 2, structures, types, templates)
    for c, w in enumerate(want):
        assert gzip.decompress(bytes(files[c]) + H_BGZF_EOF) == w, f"file column {c}"
    assert np.array_equal(sum(d.counts() for d in ds), counts)
    # a window is a cut of ITS input, of the chunk's size
    d0 = ds[0]
    with pytest.raises(Exception, match="fewer lines|nothing has been fed"):
        d0.fed_cut(0, 400)


def test_collect_in_two_halves_gives_what_collect_gives():
    """fqtk_demuxer_collect_begin (the chunk's copy home starts) + fqtk_demuxer_collect (waits for it) == fqtk_demuxer_collect alone; a second begin is
    a no-op, a begin on a slot nothing was submitted on is refused, and a chunk that FAILED brings nothing home but reports through collect as before."""
    rng = np.random.default_rng(77)
    structures, types = ["8B+T", "+T"], "T"
    templates = make_templates(rng, 1500, BARCODES8, structures, header_kind=1)
    want, counts, _ = expected_files(BARCODES8, 1, 2, structures, types, templates)
    for halves in (False, True):
        m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
        d = Demuxer(m, structures, types, max_chunk_templates=500)
        files = [bytearray() for _ in range(d.n_files)]
        for k, lo in enumerate(range(0, 1500, 500)):
            d.submit(k % 3, texts_of(templates, lo, lo + 500, len(structures)), 500)
        for k in range(3):
            if halves:
                d.collect_begin(k)
                d.collect_begin(k)          # (already on its way)
            for c, x in enumerate(d.collect(k)):
                files[c] += x
        for c, x in enumerate(d.flush()):
            files[c] += x
        for c, w in enumerate(want):
            assert gzip.decompress(bytes(files[c]) + H_BGZF_EOF) == w, (halves, c)
        assert np.array_equal(d.counts(), counts)
        with pytest.raises(Exception, match="nothing was submitted"):
            d.collect_begin(1)
    # a chunk with a malformed record: begin succeeds and brings nothing, collect reports the record
    m = BarcodeMatcher(BARCODES8, 1, 2, device=0)
    d = Demuxer(m, ["8B+T"], "T", max_chunk_templates=10)
    bad = b"@a x\nACGTACGTAA\n+\nFFFFFFFFFF\n" + b"b x\nACGTACGTAA\n+\nFFFFFFFFFF\n"
    d.submit(0, [bad], 2)
    d.collect_begin(0)
    with pytest.raises(DemuxChunkError) as ei:
        d.collect(0)
    assert ei.value.kind == 1 and ei.value.template == 1
