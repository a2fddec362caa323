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


def test_pipes_and_other_non_regular_inputs_are_streamed(tmp_path):
    """a FIFO / process substitution / /dev/stdin has st_size 0 and cannot be mapped: the reference (needletail over a File, src/seq_parse.rs:356) streams it.
    The ingest opens the path ONCE (a second open would take the pipe's writer away) and reads the stream to its end: same records as the regular file,
    plain, gz (own decoder), gz the own decoder refuses (zlib over memory), truncated gz (loud), bzip2 (refused loudly), and an empty stream (no records)."""
    import threading
    import zlib
    from savont_amd.pipeline import fastx_digest
    rng = np.random.default_rng(5)
    recs = [("read_%04d x" % i, bytes(rng.choice(list(b"ACGT"), 60 + i % 7).astype(np.uint8)), bytes(rng.integers(40, 70, 60 + i % 7).astype(np.uint8))) for i in range(50)]
    fq = "".join("@%s\n%s\n+\n%s\n" % (i, s.decode(), q.decode()) for i, s, q in recs).encode()
    plain = tmp_path / "p.fq"; plain.write_bytes(fq)
    want = fastx_digest(str(plain))
    assert want[:3] == (50, sum(len(r[1]) for r in recs), True)

    def through_fifo(payload):
        ff = str(tmp_path / "pipe.fq"); os.mkfifo(ff)
        def feed():
            with open(ff, "wb") as f:
                f.write(payload)
        th = threading.Thread(target=feed); th.start()
        try:
            return fastx_digest(ff)
        finally:
            th.join(); os.unlink(ff)
    assert through_fifo(fq) == want
    assert through_fifo(gzip.compress(fq)) == want
    assert through_fifo(gzip.compress(fq[:2000]) + gzip.compress(fq[2000:])) == want          # two members
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY)
    hdr_extra = b"\x1f\x8b\x08\x10" + b"\0" * 6 + b"a comment field\0"                             # FCOMMENT set: both decoders take it
    body = zlib.compressobj(6, zlib.DEFLATED, -15); raw = body.compress(fq) + body.flush()
    member = hdr_extra + raw + (zlib.crc32(fq) & 0xFFFFFFFF).to_bytes(4, "little") + (len(fq) & 0xFFFFFFFF).to_bytes(4, "little")
    assert through_fifo(member) == want
    assert through_fifo(b"") == (0, 0, False, _digest([], np.zeros(0, np.uint8), None, np.zeros(1, np.uint64)))
    with pytest.raises(ValueError):
        through_fifo(gzip.compress(fq)[:-20])                                                   # truncated stream
    import bz2
    with pytest.raises(ValueError):
        through_fifo(bz2.compress(fq))                                                          # libbz2 takes a path; a pipe cannot be handed over by name
    with pytest.raises(ValueError):
        fastx_digest(str(tmp_path))                                                             # a directory is not an input


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


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & ((1 << 64) - 1)
    return h


def test_own_inflate_equals_zlib(tmp_path):
    """host/inflate.hpp (what svh_load_fastx uses for .gz inputs) against zlib, byte for byte: the reference's fixtures, every deflate block type (stored, fixed, dynamic),
    long matches at distance 1 and at the 32 KB window edge, incompressible bytes, an empty member, several members in one file (bgzip / `cat a.gz b.gz`), trailing zero padding"""
    import zlib
    from savont_amd.pipeline import gunzip_digest
    rng = np.random.default_rng(5)
    for name in sorted(os.listdir(GOLDEN)):
        if name.endswith(".gz") and not name.endswith(".json.gz"):
            a = gunzip_digest(os.path.join(GOLDEN, name), 0); b = gunzip_digest(os.path.join(GOLDEN, name), 1)
            assert a[:2] == b[:2] and a[0] > 0, name

    def gz_member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31):
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
        return c.compress(data) + c.flush()
    fq = b"".join(b"@read%d runid=x\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 300).tolist()), bytes(rng.integers(35, 75, 300, dtype=np.uint8).tolist())) for i in range(400))
    cases = {
        "dynamic6": gz_member(fq, 6), "dynamic9": gz_member(fq, 9), "fast1": gz_member(fq, 1), "stored": gz_member(fq, 0), "fixed": gz_member(fq, 6, zlib.Z_FIXED),
        "huffman_only": gz_member(fq, 6, zlib.Z_HUFFMAN_ONLY), "rle": gz_member(b"A" * 100000 + b"CG" * 50000 + fq[:1000], 6, zlib.Z_RLE),
        "random": gz_member(rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(), 6), "empty": gz_member(b""), "one_byte": gz_member(b"x"),
        "window_edge": gz_member((lambda blk: blk + rng.integers(0, 256, 32768 - len(blk) - 7, dtype=np.uint8).tobytes() + blk * 3)(rng.integers(0, 256, 300, dtype=np.uint8).tobytes()), 9),
        "members": gz_member(fq[:50000], 6) + gz_member(b"", 6) + gz_member(fq[50000:], 1) + gz_member(fq[:777], 0),
        "padded": gz_member(fq[:30000], 6) + b"\0" * 512,
        "skewed_code": gz_member(bytes(rng.choice(np.arange(256, dtype=np.uint8), 300000, p=(lambda w: w / w.sum())(1.0 / np.arange(1, 257) ** 2.2)).tolist()), 9),   # code lengths up to 15: the sub-tables
    }
    for name, blob in cases.items():
        path = tmp_path / (name + ".gz"); path.write_bytes(blob)
        a = gunzip_digest(path, 0); b = gunzip_digest(path, 1)
        assert a[:2] == b[:2], name
    assert gunzip_digest(tmp_path / "members.gz", 1)[0] == len(fq) + 777 and gunzip_digest(tmp_path / "members.gz", 1)[1] == _fnv(fq + fq[:777])


def test_one_gzip_member_on_several_threads_equals_zlib(tmp_path):
    """A .fq.gz sample is one deflate stream.  Round 6 inflates it on several threads (host/inflate.hpp: inflate_member_parallel -- later pieces start at guessed block
    boundaries and decode into 16-bit symbols with markers for the 32 KB they cannot know, replaced in stream order); svh_gunzip_digest(decoder = n >= 2) runs exactly that.
    Byte-equal to zlib for: FASTQ text at levels 1 / 6 / 9, a stream with sync-flush points (empty stored blocks between the dynamic ones), a large first member followed by
    a small one, text followed by incompressible bytes (stored blocks in a later piece: no block start is found there, the piece before goes on), and binary data (no text
    block start anywhere: the sequential decoder takes it).  Truncated and corrupted copies are refused, never decoded differently."""
    import zlib
    from savont_amd.pipeline import gunzip_digest
    rng = np.random.default_rng(9)
    n_rec = 16000
    L = rng.integers(900, 1600, n_rec)
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(L.sum())); quals = rng.integers(35, 75, int(L.sum())).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(L)])
    fq = b"".join(b"@read_%d runid=abc ch=%d\n%s\n+\n%s\n" % (i, i % 512, bases[off[i]:off[i + 1]].tobytes(), quals[off[i]:off[i + 1]].tobytes()) for i in range(n_rec))
    assert len(fq) > (36 << 20)

    def member(data, level=6, flush_every=0):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        if not flush_every:
            return c.compress(data) + c.flush()
        out = b""
        for a in range(0, len(data), flush_every):
            out += c.compress(data[a:a + flush_every]) + c.flush(zlib.Z_SYNC_FLUSH)
        return out + c.flush()
    cases = {"l6": member(fq), "l1": member(fq, 1), "l9": member(fq[:20 << 20], 9), "sync": member(fq, 6, 3 << 20), "two_members": member(fq) + member(fq[:100000]),
             "text_then_noise": member(fq[:24 << 20] + rng.integers(0, 256, 6 << 20, dtype=np.uint8).tobytes() + fq[:4 << 20]),
             "binary": member(rng.integers(0, 64, 12 << 20, dtype=np.uint8).tobytes())}
    for name, blob in cases.items():
        assert len(blob) > (4 << 20), name
        p = tmp_path / (name + ".gz"); p.write_bytes(blob)
        want = gunzip_digest(str(p), 0)
        for threads in (1, 2, 3, 5, 8):
            got = gunzip_digest(str(p), threads)
            assert got[:2] == want[:2], (name, threads)
        p.unlink()
    blob = cases["l6"]
    for k, bad in enumerate((blob[:len(blob) // 2], blob[:-9], blob[:len(blob) // 3] + bytes(64) + blob[len(blob) // 3 + 64:], blob[:len(blob) * 2 // 3] + b"\xff" * 3000 + blob[len(blob) * 2 // 3 + 3000:])):
        p = tmp_path / ("bad%d.gz" % k); p.write_bytes(bad)
        for threads in (1, 4):
            with pytest.raises(ValueError):
                gunzip_digest(str(p), threads)
        p.unlink()


def test_truncated_and_corrupt_gz_fail_loudly(tmp_path):
    """a .gz cut short, one with a flipped payload byte (CRC-32), one with a wrong length trailer: neither decoder hands out records; the ingest raises
    (needletail / flate2 error out: src/seq_parse.rs:356-379)"""
    from savont_amd.pipeline import fastx_digest, gunzip_digest
    rng = np.random.default_rng(6)
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 200).tolist()), b"I" * 200) for i in range(300))
    good = gzip.compress(fq, 6)
    ok = tmp_path / "ok.fq.gz"; ok.write_bytes(good)
    assert fastx_digest(str(ok))[0] == 300
    cut = tmp_path / "cut.fq.gz"; cut.write_bytes(good[:len(good) // 2])
    flip = bytearray(good); flip[len(good) // 2] ^= 0x40
    bad = tmp_path / "flip.fq.gz"; bad.write_bytes(bytes(flip))
    trl = bytearray(good); trl[-2] ^= 1
    wrong = tmp_path / "len.fq.gz"; wrong.write_bytes(bytes(trl))
    for pth in (cut, bad, wrong):
        for dec in (0, 1):
            with pytest.raises(ValueError):
                gunzip_digest(pth, dec)
        with pytest.raises(ValueError):
            fastx_digest(str(pth))


def test_large_gz_goes_through_the_parallel_parser(tmp_path):
    """a .fq.gz that inflates to more than 4 MB is parsed from memory on the worker pool exactly like the plain file: same records as the Python reader, CRLF and a
    second member included"""
    from savont_amd.fastx import read_fastx
    from savont_amd.pipeline import fastx_digest
    rng = np.random.default_rng(7)
    recs = [(b"read_%d ch=%d" % (i, i % 512), bytes(rng.choice(list(b"ACGT"), int(rng.integers(900, 1700))).tolist())) for i in range(5000)]
    body = b"".join(b"@%s\n%s\n+\n%s\n" % (h, s, bytes(rng.integers(35, 80, len(s), dtype=np.uint8).tolist())) for h, s in recs)
    assert len(body) > (4 << 20)
    plain = tmp_path / "big.fq"; plain.write_bytes(body)
    gzp = tmp_path / "big.fq.gz"; gzp.write_bytes(gzip.compress(body[:3000000], 6) + gzip.compress(body[3000000:], 1))
    want = fastx_digest(str(plain))
    assert want[0] == 5000 and fastx_digest(str(gzp)) == want
    seq, qual, off, ids = read_fastx(str(gzp))
    assert want[3] == _digest(ids, seq, qual, off)


def test_several_files_side_by_side_equal_one_after_the_other(tmp_path):
    """svh_load_fastx on several files (--pooled-samples: one per sample) inflates and parses them side by side on the pool: the records, their order and the errors are those of
    reading one file after the other -- gz and plain mixed, an empty file in the middle, a corrupt member reported for ITS file, FASTA mixed with FASTQ refused"""
    from savont_amd.fastx import read_fastx
    from savont_amd.pipeline import fastx_digest
    rng = np.random.default_rng(9)
    paths, all_ids, seqs, quals = [], [], [], []
    for f in range(7):
        recs = [("s%d_read%d extra" % (f, i), bytes(rng.choice(list(b"ACGT"), int(rng.integers(5, 900))).tolist())) for i in range(int(rng.integers(1, 400)))]
        if f == 3: recs = []
        body = b"".join(b"@%s\n%s\n+\n%s\n" % (h.encode(), s_, bytes(rng.integers(35, 75, len(s_), dtype=np.uint8).tolist())) for h, s_ in recs)
        pth = tmp_path / ("f%d.fq%s" % (f, ".gz" if f % 2 == 0 else ""))
        pth.write_bytes(gzip.compress(body, 6) if f % 2 == 0 else body)
        paths.append(str(pth))
    want_n = 0; h = None
    ids, sq, ql = [], [], []
    for pth in paths:
        s_, q_, o_, i_ = read_fastx(pth)
        for r in range(len(i_)):
            ids.append(i_[r]); sq.append(s_[int(o_[r]):int(o_[r + 1])]); ql.append(q_[int(o_[r]):int(o_[r + 1])])
    seq = np.concatenate(sq); qual = np.concatenate(ql); off = np.zeros(len(ids) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in sq])
    assert fastx_digest("\n".join(paths)) == (len(ids), len(seq), True, _digest(ids, seq, qual, off))
    bad = tmp_path / "bad.fq.gz"; blob = bytearray(gzip.compress(b"@r\nACGT\n+\nIIII\n" * 500, 6)); blob[len(blob) // 2] ^= 0x10; bad.write_bytes(bytes(blob))
    with pytest.raises(ValueError) as ei:
        fastx_digest("\n".join(paths[:2] + [str(bad)] + paths[2:]))
    assert "bad.fq.gz" in str(ei.value)
    fa = tmp_path / "x.fa"; fa.write_text(">a\nACGT\n")
    with pytest.raises(ValueError) as ei:
        fastx_digest("\n".join(paths[:2] + [str(fa)]))
    assert "FASTA after FASTQ" in str(ei.value)
