"""Demuxer: Python mirror of the record pipeline's C ABI (include/fqtk_demux.h).

What the reference does per template on the host -- ReadSetIterator::next, BarcodeMatcher::assign, SampleWriters::write
with write_header_internal, BGZF compression (/root/reference/src/bin/commands/demux.rs:285-343, 968, 396-415, 171-267,
755-798) -- runs here per chunk of templates on the device: FASTQ text in, whole BGZF members per output file out.
Used by the GPU tests and bench.py; `fqtk demux` (csrc/host/demux.cpp) drives the same entry points from C++.
"""
from __future__ import annotations

import ctypes as C
import re
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .barcode_matching import BarcodeMatcher, FqtkError, _check

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
ERRORS = {1: "record does not start with '@'", 2: "third line does not start with '+'",
          3: "sequence and quality lengths differ", 4: "text is not 4 lines per template", 5: "too few bases",
          6: "barcode length", 7: "header"}


def parse_read_structure(text: str):
    """'8B92T' -> [(offset, length | -1, kind)], as the `read-structure` crate lays segments out."""
    segs, off = [], 0
    pos = 0
    for m in re.finditer(r"(\d+|\+)([TBMCS])", text.upper()):
        if m.start() != pos:
            raise ValueError(f"bad read structure {text!r}")
        pos = m.end()
        length = -1 if m.group(1) == "+" else int(m.group(1))
        segs.append((off, length, m.group(2)))
        if length > 0:
            off += length
    if pos != len(text) or not segs:
        raise ValueError(f"bad read structure {text!r}")
    return segs


class DemuxChunkError(RuntimeError):
    def __init__(self, kind: int, input_index: int, template: int, detail: int, message: str):
        super().__init__(f"chunk failed: {ERRORS.get(kind, kind)} (input {input_index}, template {template}, detail {detail}) {message}")
        self.kind, self.input_index, self.template, self.detail, self.message = kind, input_index, template, detail, message


class Demuxer:
    def __init__(self, matcher: BarcodeMatcher, read_structures: Sequence[str], output_types: str = "T",
                 skip_too_few_bases: bool = False, max_chunk_templates: int = 1 << 18, carry_blocks: bool = True,
                 compression_level: int = 5):
        self._lib = _lib.load()
        self._h = None
        self.matcher = matcher
        self.structures = [parse_read_structure(r) for r in read_structures]
        n_seg = (C.c_uint32 * len(self.structures))(*[len(s) for s in self.structures])
        flat = [x for s in self.structures for x in s]
        segs = (_lib.fqtk_demux_segment * len(flat))(*[_lib.fqtk_demux_segment(o, l, k.encode()) for o, l, k in flat])
        cfg = _lib.fqtk_demux_config()
        cfg.n_inputs = len(self.structures)
        cfg.n_segments = n_seg
        cfg.segments = segs
        for i, k in enumerate("TBMC"):
            cfg.want[i] = 1 if k in output_types else 0
        cfg.skip_too_few_bases = 1 if skip_too_few_bases else 0
        cfg.max_chunk_templates = max_chunk_templates
        cfg.carry_blocks = 1 if carry_blocks else 0
        cfg.compression_level = compression_level
        h = C.c_void_p()
        _check(self._lib.fqtk_demuxer_create(matcher.handle, C.byref(cfg), C.byref(h)))
        self._h = h
        self.files_per_sample = int(self._lib.fqtk_demuxer_files_per_sample(h))
        self.n_files = (matcher.n_samples + 1) * self.files_per_sample
        self._keep: Dict[int, list] = {}
        self.skipped = 0

    def close(self) -> None:
        if self._h:
            self._lib.fqtk_demuxer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, slot: int, texts: Sequence[bytes], n_templates: int) -> None:
        arrs = [np.frombuffer(t, dtype=np.uint8) for t in texts]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (C.c_uint64 * len(arrs))(*[a.size for a in arrs])
        self._keep[slot] = [arrs, texts]
        _check(self._lib.fqtk_demuxer_submit(self._h, slot, ptrs, lens, n_templates))

    # ---- BGZF inputs inflated on the device (fqtk_demuxer_feed / submit_fed / fed_tail) -------------------------------
    def feed(self, input_index: int, bgzf_bytes: bytes, last: bool = False) -> int:
        """Hands the whole BGZF members in `bgzf_bytes` (as they lie in a file) of one input to the device; returns the number
        of lines (newlines) fed for that input so far.  last: the input's final members (a newline is added behind the text)."""
        import struct
        members, pos = [], 0
        while pos < len(bgzf_bytes):
            bsize = struct.unpack_from("<H", bgzf_bytes, pos + 16)[0] + 1
            crc, isize = struct.unpack_from("<II", bgzf_bytes, pos + bsize - 8)
            members.append((pos + 18, bsize - 26, isize, crc))
            pos += bsize
        arr = (_lib.fqtk_inflate_member * max(len(members), 1))()
        for k, (off, plen, isize, crc) in enumerate(members):
            arr[k].payload_off, arr[k].payload_len, arr[k].isize, arr[k].crc = off, plen, isize, crc
        buf = np.frombuffer(bgzf_bytes + b"\0" * 8, dtype=np.uint8)
        fed = C.c_uint64(0)
        _check(self._lib.fqtk_demuxer_feed(self._h, input_index, buf.ctypes.data, len(bgzf_bytes), arr, len(members),
                                           1 if last else 0, C.byref(fed)))
        return int(fed.value)

    def stream_decode(self, input_index: int, data: bytes, chunks):
        """chunks: [(start_bit, stop_bit | None)] of a serial DEFLATE stream in `data`.  Returns [(status, final, n_bytes, end_bit)]."""
        arr = (_lib.fqtk_stream_chunk * len(chunks))()
        for k, (a, b) in enumerate(chunks):
            arr[k].start_bit, arr[k].stop_bit = a, (0xFFFFFFFFFFFFFFFF if b is None else b)
        ends = (_lib.fqtk_stream_end * len(chunks))()
        buf = np.frombuffer(data + b"\0" * 8, dtype=np.uint8)
        _check(self._lib.fqtk_demuxer_stream_decode(self._h, input_index, buf.ctypes.data, len(data), arr, len(chunks), ends))
        return [(int(e.status), int(e.final_block), int(e.n_bytes), int(e.end_bit)) for e in ends]

    def stream_scan(self, input_index: int, data: bytes, first_bit: int, chunk_bytes: int, n_slots: int, to_end: bool, sym_per_byte: int = 8, text_only: bool = False):
        """The chunks of a stretch of a serial DEFLATE stream cut ON THE DEVICE at block starts it finds, and decoded.
        Returns [(status, final, n_bytes, end_bit, start_bit, n_blocks, flags)]."""
        ends = (_lib.fqtk_stream_end * n_slots)()
        n = C.c_uint32(0)
        buf = np.frombuffer(data + b"\0" * 8, dtype=np.uint8)
        _check(self._lib.fqtk_demuxer_stream_scan(self._h, input_index, buf.ctypes.data, len(data), first_bit, chunk_bytes, n_slots, 1 if to_end else 0,
                                                  sym_per_byte, 1 if text_only else 0, ends, C.byref(n)))
        return [(int(e.status), int(e.final_block), int(e.n_bytes), int(e.end_bit), int(e.start_bit), int(e.n_blocks), int(e.flags)) for e in ends[:n.value]]

    def stream_window(self, input_index: int) -> bytes:
        """The 32 KiB of text in front of the next chunk of the stream."""
        buf = np.zeros(32768, dtype=np.uint8)
        _check(self._lib.fqtk_demuxer_stream_window(self._h, input_index, buf.ctypes.data))
        return buf.tobytes()

    def stream_commit_text(self, input_index: int, text: bytes, window_after, last: bool):
        """Text of the stream decoded elsewhere joins the fed text.  (lines fed so far, CRC-32 of the text)"""
        fed, crc = C.c_uint64(0), C.c_uint32(0)
        buf = np.frombuffer(text + b"\0" * 8, dtype=np.uint8)
        wa = None if window_after is None else np.frombuffer(window_after, dtype=np.uint8)
        assert wa is None or wa.size == 32768
        _check(self._lib.fqtk_demuxer_stream_commit_text(self._h, input_index, buf.ctypes.data, len(text), None if wa is None else wa.ctypes.data,
                                                         1 if last else 0, C.byref(fed), C.byref(crc)))
        return int(fed.value), int(crc.value)

    def stream_commit(self, input_index: int, n_accept: int, member_start: bool, last: bool):
        """(lines fed so far, CRC-32 of the committed text, its length)"""
        fed, crc, n = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        _check(self._lib.fqtk_demuxer_stream_commit(self._h, input_index, n_accept, 1 if member_start else 0, 1 if last else 0,
                                                    C.byref(fed), C.byref(crc), C.byref(n)))
        return int(fed.value), int(crc.value), int(n.value)

    def submit_fed(self, slot: int, n_templates: int) -> None:
        _check(self._lib.fqtk_demuxer_submit_fed(self._h, slot, n_templates))

    def fed_cut(self, input_index: int, n_templates: int):
        """The next n_templates records of this demuxer's fed text of one input: a window any demuxer of the same configuration may run."""
        w = _lib.fqtk_fed_window()
        _check(self._lib.fqtk_demuxer_fed_cut(self._h, input_index, n_templates, C.byref(w)))
        return w

    def submit_windows(self, slot: int, windows, n_templates: int) -> None:
        """One chunk out of windows[i] (a cut of input i at its home demuxer, fed_cut); windows of other demuxers are copied device to device."""
        arr = (_lib.fqtk_fed_window * len(windows))(*windows)
        _check(self._lib.fqtk_demuxer_submit_windows(self._h, slot, arr, n_templates))

    def collect_fed(self, slot: int):
        """(files, text_end per input) of a chunk of fed text."""
        res = _lib.fqtk_demux_result()
        _check(self._lib.fqtk_demuxer_collect(self._h, slot, C.byref(res)))
        ends = [int(res.text_end[i]) for i in range(len(self.structures))] if res.text_end else None
        return self._files(res), ends

    def fed_tail(self, input_index: int, pos: int, cap: int = 1 << 16) -> bytes:
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64(0)
        _check(self._lib.fqtk_demuxer_fed_tail(self._h, input_index, pos, buf, cap, C.byref(n)))
        return bytes(buf[:min(int(n.value), cap)])

    def _files(self, res) -> List[bytes]:
        if res.error:
            raise DemuxChunkError(res.error, res.error_input, res.error_template, res.error_detail,
                                  _lib.last_error() if res.error == 6 else "")
        total = int(res.file_off[res.n_files]) if res.n_files else 0
        raw = C.string_at(res.bytes, total) if total else b""
        self.skipped += int(res.n_skipped)
        return [raw[int(res.file_off[c]):int(res.file_off[c + 1])] for c in range(int(res.n_files))]

    def collect_begin(self, slot: int) -> None:
        """Optional first half of collect(): the chunk's kernels are waited for and the copy home of its members starts."""
        _check(self._lib.fqtk_demuxer_collect_begin(self._h, slot))

    def collect(self, slot: int) -> List[bytes]:
        res = _lib.fqtk_demux_result()
        _check(self._lib.fqtk_demuxer_collect(self._h, slot, C.byref(res)))
        self._keep.pop(slot, None)
        return self._files(res)

    def flush(self) -> List[bytes]:
        res = _lib.fqtk_demux_result()
        _check(self._lib.fqtk_demuxer_flush(self._h, C.byref(res)))
        return self._files(res)

    def counts(self) -> np.ndarray:
        c = np.zeros(self.matcher.n_samples + 1, dtype=np.uint64)
        _check(self._lib.fqtk_demuxer_counts(self._h, c.ctypes.data))
        return c

    def stage_seconds(self) -> Dict[str, float]:
        s = (C.c_double * _lib.FQTK_DEMUX_STAGES)()
        _check(self._lib.fqtk_demuxer_stage_seconds(self._h, s))
        return {self._lib.fqtk_demuxer_stage_name(k).decode(): float(s[k]) for k in range(_lib.FQTK_DEMUX_STAGES)}

    def run(self, chunks, in_flight: int = _lib.FQTK_DEMUX_SLOTS) -> List[bytes]:
        """chunks: iterable of (texts, n_templates).  Returns every output file's complete BGZF stream."""
        files = [bytearray() for _ in range(self.n_files)]
        pending: List[int] = []
        k = 0
        for texts, n in chunks:
            slot = k % _lib.FQTK_DEMUX_SLOTS
            if len(pending) == in_flight:
                for c, b in enumerate(self.collect(pending.pop(0))):
                    files[c] += b
            self.submit(slot, texts, n)
            pending.append(slot)
            k += 1
        while pending:
            for c, b in enumerate(self.collect(pending.pop(0))):
                files[c] += b
        for c, b in enumerate(self.flush()):
            files[c] += b
        return [bytes(f) + BGZF_EOF for f in files]
