"""Sample / SampleGroup: host-side table preconditions of the matcher.

Mirrors /root/reference/src/lib/samples.rs (Sample::new :49-57, SampleGroup::from_samples :101-133,
from_file :144-147) -- same validation rules and the same messages, raised as ValueError where the
reference panics.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

_IUPAC = set(b"ACGTUMRWSYKVHDBN")


def is_valid_iupac(byte: int) -> bool:
    """reference src/lib/mod.rs:90-92: upper-case IUPAC code or a no-call ('N', 'n', '.')."""
    return byte in _IUPAC or byte in (ord("N"), ord("n"), ord("."))


@dataclass(frozen=True)
class Sample:
    """reference samples.rs:17-26.  `ordinal` = index of the sample in its SampleGroup."""
    sample_id: str
    barcode: str
    ordinal: int = 0

    @staticmethod
    def new(ordinal: int, name: str, barcode: str) -> "Sample":
        # samples.rs:49-57
        if not name:
            raise ValueError("Sample name cannot be empty")
        if not barcode:
            raise ValueError("Sample barcode cannot be empty")
        if not all(is_valid_iupac(b) for b in barcode.encode("latin-1", errors="replace")):
            raise ValueError("All sample barcode bases must be one of A, C, G, T, U, R, Y, S, W, K, M, "
                             "D, V, H, B, N")
        return Sample(sample_id=name, barcode=barcode, ordinal=ordinal)

    @staticmethod
    def deserialize_header_line() -> str:
        # samples.rs:59-70
        return "sample_id\tbarcode"

    def __str__(self) -> str:  # samples.rs:28-39
        return f"Sample({self.ordinal:04}) - {{ name: {self.sample_id}\tbarcode: {self.barcode} }}"


class DelimFileHeaderError(ValueError):
    """fgoxide's FgError::DelimFileHeaderError {expected, found} (samples.rs tests :211-251)."""

    def __init__(self, expected: str, found: str):
        super().__init__(f"header mismatch: expected {expected!r}, found {found!r}")
        self.expected = expected
        self.found = found


class SampleGroup:
    """reference samples.rs:74-147."""

    def __init__(self, samples: List[Sample]):
        self.samples = samples

    @staticmethod
    def from_samples(samples: Sequence[Sample]) -> "SampleGroup":
        # samples.rs:101-133 -- same order of checks
        if len(samples) == 0:
            raise ValueError("Must provide one or more sample")
        ids = [s.sample_id for s in samples]
        if len(set(ids)) != len(ids):
            raise ValueError("Each sample name must be unique, duplicate identified")
        bcs = [s.barcode for s in samples]
        if len(set(bcs)) != len(bcs):
            raise ValueError("Each sample barcode must be unique, duplicate identified")
        first = len(samples[0].barcode)
        if not all(len(b) == first for b in bcs):
            raise ValueError("All barcodes must have the same length")
        return SampleGroup([Sample.new(i, s.sample_id, s.barcode) for i, s in enumerate(samples)])

    @staticmethod
    def from_file(path: str) -> "SampleGroup":
        """Headered TSV `sample_id<TAB>barcode` (samples.rs:144-147; DelimFile::read(path, b'\\t',
        false)).  Trailing blank lines are tolerated (samples.rs tests :181-201)."""
        with open(path, "r", newline="") as fh:
            lines = fh.read().split("\n")
        while lines and lines[-1].strip("\r") == "":
            lines.pop()
        if not lines:   # samples.rs test_reading_empty_file: an empty file reaches from_samples(&[])
            return SampleGroup.from_samples([])
        header = lines[0].rstrip("\r")
        expected = Sample.deserialize_header_line()
        if header != expected:
            raise DelimFileHeaderError(expected=expected, found=header)
        samples = []
        for ln, line in enumerate(lines[1:], start=2):
            line = line.rstrip("\r")
            fields = line.split("\t")
            if len(fields) != 2:
                raise ValueError(f"{path}:{ln}: expected 2 tab-separated fields, found {len(fields)}")
            samples.append(Sample(sample_id=fields[0], barcode=fields[1], ordinal=len(samples)))
        return SampleGroup.from_samples(samples)

    def __len__(self) -> int:
        return len(self.samples)

    def __str__(self) -> str:  # samples.rs:80-88
        return "SampleGroup {\n" + "".join(f"    {s}\n" for s in self.samples) + "}\n"
