"""Reference names and lengths from a BAM header, without pysam.

The only thing TRA genotyping takes from the BAM besides alignments is
`bamfile.get_reference_length(chr)` (cuteSV_resolveTRA.py:264, 291), which clamps the search windows.
A BAM file is a BGZF stream, i.e. a series of gzip members, so the standard `gzip` module reads it;
the header is: magic "BAM\\1", l_text:int32, text, n_ref:int32, then per reference l_name:int32,
name (NUL-terminated), l_ref:int32 (SAM specification, section 4.2).
"""
import gzip
import struct


def reference_lengths(bam_path):
    """{reference name: length} in header order."""
    with gzip.open(bam_path, "rb") as f:
        def take(n):
            buf = f.read(n)
            if len(buf) != n:
                raise ValueError("%s: truncated BAM header" % bam_path)
            return buf
        if take(4) != b"BAM\x01":
            raise ValueError("%s: not a BAM file (CRAM / SAM are not supported here)" % bam_path)
        (l_text,) = struct.unpack("<i", take(4))
        take(l_text)
        (n_ref,) = struct.unpack("<i", take(4))
        out = {}
        for _ in range(n_ref):
            (l_name,) = struct.unpack("<i", take(4))
            name = take(l_name)[:-1].decode()
            (l_ref,) = struct.unpack("<i", take(4))
            out[name] = l_ref
        return out
