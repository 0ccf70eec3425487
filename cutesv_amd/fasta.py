"""Minimal FASTA reader (plain or gzip) so the VCF emit step needs no pysam: {contig name: sequence}.
Mirrors what generate_output asks of pysam.FastaFile.fetch(chrom) (cuteSV_genotype.py:254-259)."""
import gzip


def read_fasta(path, only=None):
    opener = gzip.open if str(path).endswith(".gz") else open
    out, name, parts = {}, None, []
    with opener(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None and (only is None or name in only):
                    out[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            elif name is not None and (only is None or name in only):
                parts.append(line.strip())
    if name is not None and (only is None or name in only):
        out[name] = "".join(parts)
    return out
