"""fqtk_amd -- MI355X-native (gfx950) sample-barcode matcher for `fqtk demux`.

Host-side mirror of the reference's library API for the one hot path this repo accelerates
(reference: /root/reference/src/lib/barcode_matching.rs, src/lib/samples.rs).  All compute goes
through the C ABI in include/fqtk_match.h (fqtk_amd/lib/libfqtk_match.so, hand-written HIP); there is
no CPU fallback -- importing works without a GPU, computing does not.
"""
from .samples import Sample, SampleGroup  # noqa: F401
from .barcode_matching import BarcodeMatch, BarcodeMatcher, FqtkError, FqtkLengthError  # noqa: F401

__all__ = ["Sample", "SampleGroup", "BarcodeMatch", "BarcodeMatcher", "FqtkError", "FqtkLengthError"]
