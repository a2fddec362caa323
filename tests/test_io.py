"""C++ ingest (needletail record semantics) against the Python reader of the harness, CPU only."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _digest(ids, seq, qual, off):
    h = 1469598103934665603
    M = (1 << 64) - 1

    def mix(b):
        nonlocal h
        for x in b:
            h = ((h ^ x) * 1099511628211) & M
    for i, name in enumerate(ids):
        mix(name.encode()); mix(b"\n"); mix(seq[int(off[i]):int(off[i + 1])].tobytes()); mix(b"\n")
        if qual is not None:
            mix(qual[int(off[i]):int(off[i + 1])].tobytes())
        mix(b"\n")
    return h


def test_ingest_matches_python_reader_on_fixtures():
    from savont_amd.fastx import read_fastx
    from savont_amd.pipeline import fastx_digest
    for name in ("zymo_ref_asvs.fa.gz",):
        path = os.path.join(GOLDEN, name)
        seq, qual, off, ids = read_fastx(path)
        n, b, q, d = fastx_digest(path)
        assert (n, b, q) == (len(ids), len(seq), qual is not None) and d == _digest(ids, seq, qual, off)


def test_ingest_formats(tmp_path):
    from savont_amd.pipeline import fastx_digest
    recs = [("r1 some comment", b"ACGTACGTNN", b"IIIIIIIIII"), ("r2 rc", b"TTTT", b"!!!!"), ("r3", b"", b"")]
    fq = "".join("@%s\n%s\n+\n%s\n" % (i, s.decode(), q.decode()) for i, s, q in recs)
    p1 = tmp_path / "a.fq"; p1.write_text(fq)
    p2 = tmp_path / "a.fq.gz"
    with gzip.open(p2, "wb") as f:
        f.write(fq.replace("\n", "\r\n").encode())                     # CRLF + gzip
    ids = [r[0] for r in recs]
    seq = np.frombuffer(b"".join(r[1] for r in recs), np.uint8); qual = np.frombuffer(b"".join(r[2] for r in recs), np.uint8)
    off = np.cumsum([0] + [len(r[1]) for r in recs]).astype(np.uint64)
    want = _digest(ids, seq, qual, off)
    assert fastx_digest(str(p1)) == (3, 14, True, want)
    assert fastx_digest(str(p2)) == (3, 14, True, want)
    fa = tmp_path / "w.fa"; fa.write_text(">x desc\nACGT\nAC\n\n>y\nGG\n")     # wrapped FASTA, blank line
    n, b, q, d = fastx_digest(str(fa))
    assert (n, b, q) == (2, 8, False) and d == _digest(["x desc", "y"], np.frombuffer(b"ACGTACGG", np.uint8), None, np.array([0, 6, 8], np.uint64))
    bad = tmp_path / "bad.fq"; bad.write_text("@r\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        fastx_digest(str(bad))
    import bz2, lzma
    bz = tmp_path / "x.fq.bz2"; bz.write_bytes(bz2.compress(fq.encode()))          # bzip2 through the system libbz2
    assert fastx_digest(str(bz)) == (3, 14, True, want)
    xz = tmp_path / "x.fq.xz"; xz.write_bytes(lzma.compress(fq.encode()))          # xz: refused loudly, not read as "plain"
    with pytest.raises(ValueError):
        fastx_digest(str(xz))


def test_large_plain_fastq_parallel_reader_equals_the_line_reader(tmp_path):
    """plain FASTQ files over 4 MB are cut at record boundaries and parsed by the worker pool; the records must be those of the line reader
    (which gz input still takes): quality lines starting with '@' and '+', CRLF, blank lines between records, a last line without a newline"""
    from savont_amd.pipeline import fastx_digest
    rng = np.random.default_rng(5)
    recs = []
    for i in range(6000):
        L = int(rng.integers(1, 1600))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L))
        q = bytearray(rng.integers(33, 74, L).astype(np.uint8).tobytes())
        if i % 7 == 0:
            q[0] = ord("@")                                                       # a quality line that looks like a header
        if i % 11 == 0:
            q[0] = ord("+")
        recs.append(("read_%d some text @ here" % i, s, bytes(q)))
    txt = b""
    for k, (i, s, q) in enumerate(recs):
        nl = b"\r\n" if k % 13 == 0 else b"\n"
        txt += b"@" + i.encode() + nl + s + nl + b"+" + (i.encode() if k % 5 == 0 else b"") + nl + q + (b"" if k == len(recs) - 1 else nl) + (b"\n" if k % 97 == 0 and k != len(recs) - 1 else b"")
    assert len(txt) > (4 << 20)
    p1 = tmp_path / "big.fq"; p1.write_bytes(txt)
    p2 = tmp_path / "big.fq.gz"
    with gzip.open(p2, "wb", compresslevel=1) as f:
        f.write(txt)
    ids = [r[0] for r in recs]
    seq = np.frombuffer(b"".join(r[1] for r in recs), np.uint8); qual = np.frombuffer(b"".join(r[2] for r in recs), np.uint8)
    off = np.cumsum([0] + [len(r[1]) for r in recs]).astype(np.uint64)
    want = (len(recs), len(seq), True, _digest(ids, seq, qual, off))
    assert fastx_digest(str(p2)) == want                                          # line reader
    assert fastx_digest(str(p1)) == want                                          # pieces on the worker pool
    # a broken record in the middle: the parallel reader steps back, the line reader words the error
    bad = tmp_path / "bad_big.fq"; bad.write_bytes(txt[:len(txt) // 2] + b"@broken\nACGT\n+\nII\n" + txt[len(txt) // 2:])
    with pytest.raises(ValueError):
        fastx_digest(str(bad))
