"""Minimal FASTQ/FASTA ingest for the Python harness (tests, bench).

Mirrors what the reference hands to the hot path (needletail records, `src/seq_parse.rs:356-379`,
`src/kmer_comp.rs:108-128`): raw sequence bytes, raw quality bytes, and the full header text
after '@' / '>'.  Returns concatenated buffers + offsets, the layout the C-ABI takes
(`include/savont_hip.h: svt_batch_upload`).
"""
import gzip

import numpy as np


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def read_fastx(path):
    """-> (seq u8[total], qual u8[total] or None, offsets u64[n+1], ids list[str])"""
    ids, seqs, quals = [], [], []
    with _open(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    i, n = 0, len(lines)
    is_fastq = None
    while i < n:
        ln = lines[i].rstrip(b"\r")
        if not ln:
            i += 1
            continue
        if ln[:1] == b"@" and is_fastq in (None, True):
            is_fastq = True
            ids.append(ln[1:].decode("utf-8", "replace"))
            seqs.append(lines[i + 1].rstrip(b"\r"))
            quals.append(lines[i + 3].rstrip(b"\r"))
            i += 4
        elif ln[:1] == b">":
            is_fastq = False
            ids.append(ln[1:].decode("utf-8", "replace"))
            i += 1
            parts = []
            while i < n and lines[i][:1] != b">":
                parts.append(lines[i].rstrip(b"\r"))
                i += 1
            seqs.append(b"".join(parts))
        else:
            raise ValueError("unparseable FASTX line %d in %s" % (i, path))
    return pack_records(seqs, quals if is_fastq else None) + (ids,)


def pack_records(seqs, quals=None):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    seq = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    qual = None
    if quals is not None:
        qual = np.frombuffer(b"".join(quals), dtype=np.uint8).copy()
        assert qual.size == seq.size, "sequence / quality length mismatch"
    return seq, qual, offsets


def write_fastq(path, seq, qual, offsets, ids):
    """plain 4-line FASTQ of the concatenated buffers (bench.py: times the C++ ingest on the same reads it benchmarks)"""
    sb = seq.tobytes(); qb = qual.tobytes()
    with open(path, "wb") as f:
        chunk = []
        for i, rid in enumerate(ids):
            a, b = int(offsets[i]), int(offsets[i + 1])
            chunk.append(b"@" + rid.encode() + b"\n" + sb[a:b] + b"\n+\n" + qb[a:b] + b"\n")
            if len(chunk) >= 4096:
                f.write(b"".join(chunk)); chunk = []
        f.write(b"".join(chunk))
