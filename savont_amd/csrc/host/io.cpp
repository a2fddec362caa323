// io.cpp -- the data formats either side of the hot path: FASTA/FASTQ ingest with the record
// semantics the reference gets from needletail, and the three output writers whose bytes the rest of savont consumes.
//   ingest   src/seq_parse.rs:356-379, src/kmer_comp.rs:108-128 (needletail::parse_fastx_file: gz or plain, FASTA may wrap,
//            FASTQ is 4 lines per record, CRLF tolerated; id = the whole header line after '@' / '>')
//   writers  write_consensus_fasta src/alignment.rs:830-860, write_feature_table src/main.rs:381-400,
//            write_clusters_tsv src/alignment.rs:799-826; final list = src/main.rs:140-200 (EM depths, zero-depth ASVs dropped,
//            stable sort by depth descending, ids renumbered for final_clusters.tsv)
// gz input (the reference's usual format) is inflated whole from a mapping of the file by host/inflate.hpp and parsed from memory like a plain file (round 5; zlib's
// line reader stays as the fallback that reads -- or words the error for -- whatever that decoder refuses).
// bzip2 input goes through the system's libbz2 (loaded at run time: the image ships the library without its header; the three
// entry points used are part of its stable ABI).  xz / zstd inputs are not supported; such a file fails loudly.
#include <sys/mman.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cerrno>
#include <cstdio>
#include <cstring>

#include "asv_pipeline.hpp"
#include "worker_pool.hpp"
#include "inflate.hpp"
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace savont {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

namespace {
struct Bz2Api {
    void* (*open)(const char*, const char*) = nullptr; int (*read)(void*, void*, int) = nullptr; void (*close)(void*) = nullptr;
    static const Bz2Api& get() {
        static const Bz2Api api = [] {
            Bz2Api a;
            void* h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
            if (h) {
                a.open = (void* (*)(const char*, const char*))dlsym(h, "BZ2_bzopen");
                a.read = (int (*)(void*, void*, int))dlsym(h, "BZ2_bzread");
                a.close = (void (*)(void*))dlsym(h, "BZ2_bzclose");
            }
            return a;
        }();
        return api;
    }
    bool ok() const { return open && read && close; }
};
struct GzLines {                                          // gz or plain through zlib, bzip2 through libbz2
    gzFile f = nullptr; void* bz = nullptr; std::vector<char> buf; size_t pos = 0, len = 0; bool eof = false, bad = false;
    bool clean_end() const { return !bad; }                // false: the gz stream ended in an error (truncated file, CRC mismatch)
    GzLines(const std::string& path, bool bzip2) : buf(1 << 20) {
        if (bzip2) bz = Bz2Api::get().open(path.c_str(), "rb");
        else { f = gzopen(path.c_str(), "rb"); if (f) gzbuffer(f, 1 << 20); }
    }
    ~GzLines() { if (f) gzclose(f); if (bz) Bz2Api::get().close(bz); }
    bool good() const { return f || bz; }
    bool fill() {
        if (eof) return false;
        const int n = bz ? Bz2Api::get().read(bz, buf.data(), (int)buf.size()) : gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) bad = true;
        if (n <= 0) { eof = true; if (f) { int en = 0; gzerror(f, &en); if (en != Z_OK && en != Z_STREAM_END) bad = true; } return false; }
        pos = 0; len = (size_t)n; return true;
    }
    // next line without its terminator ('\n' or '\r\n'); false at end of input
    bool next(std::string& line) {
        line.clear(); bool got = false;
        for (;;) {
            if (pos == len && !fill()) break;
            const char* p = buf.data() + pos; const char* e = (const char*)memchr(p, '\n', len - pos);
            got = true;
            if (e) { line.append(p, e - p); pos = (size_t)(e - buf.data()) + 1; break; }
            line.append(p, len - pos); pos = len;
        }
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return got;
    }
};
struct MemLines {                                         // the same line source over bytes in memory (an inflated gz file, a small plain file)
    const char* p; const char* end;
    bool good() const { return true; }
    bool next(std::string& line) {
        if (p >= end) { line.clear(); return false; }
        const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* stop = e ? e : end;
        line.assign(p, (size_t)(stop - p));
        p = e ? e + 1 : end;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
};
struct MappedFile {                                       // the bytes of a whole input, opened ONCE: a read-only mapping of a regular file (pages come from the page cache as
    // they are touched), or -- a FIFO, /dev/stdin, a process substitution, /proc, a filesystem without mmap -- the stream read to its end into a buffer of its own (needletail
    // streams such inputs, src/seq_parse.rs:356; a second open would consume a pipe's writer)
    const char* p = nullptr; size_t n = 0;
    bool opened = false, empty_regular = false, mapped = false;
    std::vector<char> owned;
    explicit MappedFile(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        opened = true;
        struct stat st;
        const bool have_stat = fstat(fd, &st) == 0;
        if (have_stat && S_ISDIR(st.st_mode)) { opened = false; close(fd); return; }
        if (have_stat && S_ISREG(st.st_mode) && st.st_size == 0) { empty_regular = true; close(fd); return; }
        if (have_stat && S_ISREG(st.st_mode)) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { p = (const char*)m; n = (size_t)st.st_size; mapped = true; madvise(m, n, MADV_WILLNEED); }   // read-ahead instead of MAP_POPULATE: the parsing threads do not wait for the whole file
        }
        if (!mapped) {                                                       // not a regular file, or mmap refused: read the one descriptor to its end
            owned.resize(1 << 20); size_t got = 0;
            for (;;) {
                if (got == owned.size()) owned.resize(owned.size() * 2);
                const ssize_t r = read(fd, owned.data() + got, owned.size() - got);
                if (r < 0) { if (errno == EINTR) continue; opened = false; break; }
                if (r == 0) break;
                got += (size_t)r;
            }
            owned.resize(got); p = owned.data(); n = got;
            if (opened && got == 0) empty_regular = true;                    // an empty stream holds no records either
        }
        close(fd);
    }
    ~MappedFile() { if (mapped) munmap((void*)p, n); }
    MappedFile(const MappedFile&) = delete; MappedFile& operator=(const MappedFile&) = delete;
};
// zlib over bytes in memory (every member of a multi-member stream): the fallback for gz bytes that did not come from a path zlib could reopen; false = truncated or corrupt
bool zlib_inflate_all(const unsigned char* src, size_t n, std::vector<char>& out) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    out.clear(); out.resize(std::max<size_t>(1 << 20, n * 3));
    size_t in_pos = 0, out_pos = 0; bool ok = true, member_done = false;
    while (ok) {
        if (out_pos == out.size()) out.resize(out.size() * 2);
        const size_t in_chunk = std::min<size_t>(n - in_pos, (size_t)1 << 30), out_chunk = std::min<size_t>(out.size() - out_pos, (size_t)1 << 30);
        zs.next_in = (Bytef*)(src + in_pos); zs.avail_in = (uInt)in_chunk;
        zs.next_out = (Bytef*)(out.data() + out_pos); zs.avail_out = (uInt)out_chunk;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_pos += in_chunk - zs.avail_in; out_pos += out_chunk - zs.avail_out;
        if (rc == Z_STREAM_END) {
            member_done = true;
            if (in_pos >= n) break;
            if (n - in_pos >= 2 && src[in_pos] == 0x1f && src[in_pos + 1] == 0x8b) { member_done = false; if (inflateReset(&zs) != Z_OK) ok = false; }
            else break;                                                       // trailing bytes that are not a member: ignored, as gzread does
        } else if (rc == Z_BUF_ERROR && in_pos >= n) { ok = false; }         // input ended inside a member
        else if (rc != Z_OK && rc != Z_BUF_ERROR) ok = false;
        else if (rc == Z_OK && in_pos >= n && zs.avail_out != 0) ok = false; // no progress possible: truncated
    }
    inflateEnd(&zs);
    out.resize(out_pos);
    return ok && member_done;
}
}  // namespace


// ---- plain FASTQ files, parsed by the worker pool ------------------------------------------------------------------------------
// One thread reading lines needs 0.17 s per 100k reads (310 MB) -- a fifth of a 100k-read step's CPU time and, serial, three steps of
// wall time.  A plain (uncompressed) FASTQ file is cut into pieces at record boundaries (a line starting with '@' whose third line starts
// with '+', whose second and fourth lines are equally long and whose fifth line, if any, starts with '@' again), the pieces are parsed
// twice on the pool: sizes first, then straight into the final arrays.  Same records, ids and bytes as the line reader below; anything
// it does not expect (FASTA, a malformed or truncated record, blank lines inside a record) makes it step back and leave the file to the line reader,
// which also words the error.  gz / bzip2 stay on the line reader (one inflate stream).
namespace {
struct Line { const char* p; size_t n; };                  // without terminator and without a trailing '\r'
inline const char* next_line(const char* p, const char* end, Line& l) {
    const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* stop = e ? e : end;
    l.p = p; l.n = (size_t)(stop - p);
    if (l.n && stop[-1] == '\r') l.n--;
    return e ? e + 1 : end;
}
// first record start at or after `from`; nullptr if none
const char* record_start(const char* base, const char* from, const char* end) {
    const char* p = from;
    if (p != base) { const char* e = (const char*)memchr(p - 1, '\n', (size_t)(end - (p - 1))); if (!e) return nullptr; p = e + 1; }   // to a line start
    for (int tries = 0; p < end && tries < 64; tries++) {
        Line a, b, c, d, e5;
        const char* q = next_line(p, end, a);
        if (a.n && a.p[0] == '@' && q < end) {
            const char* q2 = next_line(q, end, b); if (q2 >= end) return nullptr;
            const char* q3 = next_line(q2, end, c); if (q3 > end) return nullptr;
            const char* q4 = next_line(q3, end, d);
            bool ok = c.n && c.p[0] == '+' && b.n == d.n && q3 < end + 1;
            if (ok && q4 < end) { const char* r = q4; Line nx; do { r = next_line(r, end, nx); } while (nx.n == 0 && r < end); ok = nx.n == 0 || nx.p[0] == '@'; (void)e5; }
            if (ok) return p;
        }
        p = q;
    }
    return nullptr;
}
// pass over one piece: counts (pass 1) or fills (pass 2); false on anything unexpected
bool parse_piece(const char* p, const char* end, size_t& n_rec, size_t& n_bases, u8* seq, u8* qual, u64* offsets, std::string* ids, u64 base0) {
    size_t nr = 0, nb = 0;
    while (p < end) {
        Line h, s_, pl, q;
        p = next_line(p, end, h);
        if (h.n == 0) continue;                                              // blank line between records
        if (h.p[0] != '@' || p >= end) return false;
        p = next_line(p, end, s_); if (p >= end) return false;
        p = next_line(p, end, pl); if (p > end) return false;
        if (pl.n == 0 || pl.p[0] != '+') return false;
        if (p >= end && s_.n != 0) return false;                             // no quality line
        p = next_line(p, end, q);
        if (q.n != s_.n) return false;
        if (seq) { memcpy(seq + nb, s_.p, s_.n); memcpy(qual + nb, q.p, q.n); offsets[nr + 1] = base0 + nb + s_.n; ids[nr].assign(h.p + 1, h.n - 1); }
        nr++; nb += s_.n;
    }
    n_rec = nr; n_bases = nb;
    return true;
}
// FASTQ bytes in memory [base, end) -- a mapped plain file or an inflated gz file -- parsed on the pool; false (nothing appended) when they are not what it expects
bool parse_fastq_parallel(const char* base, const char* end, RawBytes& seq, RawBytes& qual, std::vector<u64>& offsets, std::vector<std::string>& ids, bool& any_qual, size_t& n_out) {
    if ((size_t)(end - base) < ((size_t)4 << 20) || base[0] != '@') return false;        // small inputs, FASTA, anything else: the line reader
    struct { size_t n; size_t size() const { return n; } } buf{(size_t)(end - base)};
    const size_t P = std::max<size_t>(1, std::min<size_t>(64, WorkerPool::get().threads() * 4));
    std::vector<const char*> cut(P + 1, end);
    cut[0] = base;
    for (size_t k = 1; k < P; k++) { const char* c = record_start(base, base + k * (buf.size() / P), end); if (!c) return false; cut[k] = c; }
    for (size_t k = 1; k <= P; k++) if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
    std::vector<size_t> nrec(P, 0), nbases(P, 0); std::vector<char> ok(P, 1);
    par_for(P, [&](size_t k) { ok[k] = parse_piece(cut[k], cut[k + 1], nrec[k], nbases[k], nullptr, nullptr, nullptr, nullptr, 0) ? 1 : 0; });
    for (char o : ok) if (!o) return false;
    if (!any_qual && !seq.empty()) return false;                              // FASTQ after FASTA: the line reader words the error
    if (offsets.empty()) offsets.push_back(0);
    size_t tr = 0, tb = 0; std::vector<size_t> r0(P), b0(P);
    for (size_t k = 0; k < P; k++) { r0[k] = tr; b0[k] = tb; tr += nrec[k]; tb += nbases[k]; }
    const size_t seq0 = seq.size(), rec0 = ids.size();
    seq.resize(seq0 + tb); qual.resize(seq0 + tb); offsets.resize(rec0 + 1 + tr); ids.resize(rec0 + tr);
    par_for(P, [&](size_t k) {
        size_t a, b;
        parse_piece(cut[k], cut[k + 1], a, b, seq.data() + seq0 + b0[k], qual.data() + seq0 + b0[k], offsets.data() + rec0 + r0[k], ids.data() + rec0 + r0[k], (u64)(seq0 + b0[k]));
    });
    any_qual = true; n_out = tr;
    return true;
}
}  // namespace

// the records of a line source appended to the arrays (needletail's record rules: FASTQ is four lines, FASTA may wrap, blank lines between records, CRLF tolerated)
template <class Lines>
static size_t read_records(Lines& in, const std::string& path, RawBytes& seq, RawBytes& qual, std::vector<u64>& offsets, std::vector<std::string>& ids, bool& any_qual) {
    if (offsets.empty()) offsets.push_back(0);
    std::string line, s, plus, q; size_t n = 0; bool have = in.next(line);
    while (have) {
        if (line.empty()) { have = in.next(line); continue; }
        if (line[0] == '@') {
            const std::string id = line.substr(1);
            if (!in.next(s) || !in.next(plus) || !in.next(q)) throw Error{SVT_ERR_ARG, path + ": truncated FASTQ record " + id};
            if (plus.empty() || plus[0] != '+') throw Error{SVT_ERR_ARG, path + ": malformed FASTQ record " + id};
            if (q.size() != s.size()) throw Error{SVT_ERR_ARG, path + ": sequence / quality length mismatch in " + id};
            if (!any_qual && !seq.empty()) throw Error{SVT_ERR_ARG, path + ": FASTQ after FASTA records (mixed inputs are not supported)"};
            any_qual = true;
            ids.push_back(id); seq.insert(seq.end(), s.begin(), s.end()); qual.insert(qual.end(), q.begin(), q.end()); offsets.push_back(seq.size()); n++;
            have = in.next(line);
        } else if (line[0] == '>') {
            if (any_qual) throw Error{SVT_ERR_ARG, path + ": FASTA after FASTQ records (mixed inputs are not supported)"};
            ids.push_back(line.substr(1));
            while ((have = in.next(line)) && (line.empty() || line[0] != '>')) seq.insert(seq.end(), line.begin(), line.end());
            offsets.push_back(seq.size()); n++;
        } else throw Error{SVT_ERR_ARG, path + ": not FASTA/FASTQ (line starts with '" + line.substr(0, 1) + "')"};
    }
    return n;
}

// "gz_inflate": 1 (default) = host/inflate.hpp, 0 = zlib's gzread for every gz file (comparison runs, tests); process-wide, set through svh_set_option
static int g_gz_inflate = 1;
void set_gz_inflate(int on) { g_gz_inflate = on ? 1 : 0; }
// "gz_threads": threads ONE gzip member is inflated on (host/inflate.hpp: inflate_member_parallel).  0 (default) = by the situation: up to sixteen pool threads when no other
// file is being inflated in this process (a lone sample: the cores are idle while it waits for its reads), one when several are (samples in flight: the cores are the
// bottleneck there, and the symbol decoder of the later pieces costs ~1.6 x the byte decoder's CPU); n >= 1 = exactly that.
static int g_gz_threads = 0;
void set_gz_threads(int n) { g_gz_threads = n < 0 ? 0 : n; }
static std::atomic<int> g_inflating{0};
unsigned gz_threads_now() {
    static const bool hooked = [] {
        gz::par_hooks().run = [](size_t n, void (*f)(size_t, void*), void* ctx) { par_for(n, [&](size_t i) { f(i, ctx); }); };
        gz::par_hooks().threads = (unsigned)WorkerPool::get().threads();
        return true;
    }();
    (void)hooked;
    if (g_gz_threads >= 1) return (unsigned)g_gz_threads;
    return g_inflating.load(std::memory_order_relaxed) > 1 ? 1u : std::min(16u, (unsigned)WorkerPool::get().threads());
}

// appends the records of one file; returns the number of records
size_t read_fastx_file(const std::string& path, RawBytes& seq, RawBytes& qual, std::vector<u64>& offsets, std::vector<std::string>& ids, bool& any_qual, bool keep_buffer) {
    MappedFile file(path);
    if (!file.opened) throw Error{SVT_ERR_ARG, "cannot open " + path};
    if (file.empty_regular) {                                                // an empty regular file (or an empty stream): no records
        if (offsets.empty()) offsets.push_back(0);
        return 0;
    }
    const unsigned char* m = (const unsigned char*)file.p; const size_t n = file.n;
    const bool bzip2 = n >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h';       // compressed formats zlib would pass through as "plain"
    if (bzip2 && !Bz2Api::get().ok()) throw Error{SVT_ERR_ARG, path + ": bzip2 input needs libbz2.so.1.0, which could not be loaded"};
    if (n >= 6 && m[0] == 0xFD && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z') throw Error{SVT_ERR_ARG, path + ": xz input is not supported (gz, bzip2 or plain)"};
    if (n >= 4 && m[0] == 0x28 && m[1] == 0xB5 && m[2] == 0x2F && m[3] == 0xFD) throw Error{SVT_ERR_ARG, path + ": zstd input is not supported (gz, bzip2 or plain)"};
    const bool gzip = n >= 2 && m[0] == 0x1f && m[1] == 0x8b;
    const char* base = file.p; const char* end = file.p + n;
    if (gzip) {
        // the whole file inflated into a buffer this thread keeps (its pages stay warm for the next file / the next load): then it is a plain file in memory
        static thread_local gz::BigBuf inflated;
        size_t len = 0; std::string why;
        struct Busy { Busy() { g_inflating.fetch_add(1, std::memory_order_relaxed); } ~Busy() { g_inflating.fetch_sub(1, std::memory_order_relaxed); } } busy;
        if (g_gz_inflate && gz::gunzip_all(m, n, inflated, len, why, gz_threads_now())) { base = (const char*)inflated.p; end = base + len; }
        else base = nullptr;                                                 // refused (or switched off): zlib reads it below, or words the error
        struct Release { gz::BigBuf& b; bool keep; ~Release() { if (!keep && b.p) { munmap(b.p, b.cap); b.p = nullptr; b.cap = 0; } } } release{inflated, keep_buffer};   // pool threads (several files side by side) do not sit on a file's worth of pages each
        if (base) {
            size_t n_par = 0;
            if (parse_fastq_parallel(base, end, seq, qual, offsets, ids, any_qual, n_par)) return n_par;
            MemLines in{base, end};
            return read_records(in, path, seq, qual, offsets, ids, any_qual);
        }
    }
    if (!bzip2 && base) {
        size_t n_par = 0;
        if (parse_fastq_parallel(base, end, seq, qual, offsets, ids, any_qual, n_par)) return n_par;
        MemLines in{base, end};
        return read_records(in, path, seq, qual, offsets, ids, any_qual);
    }
    if (!file.mapped) {                                                      // a stream: its bytes are here and the path cannot be opened a second time
        if (bzip2) throw Error{SVT_ERR_ARG, path + ": bzip2 input must be a regular file (a pipe cannot be handed to libbz2 by name)"};
        std::vector<char> plain;
        if (!zlib_inflate_all(m, n, plain)) throw Error{SVT_ERR_ARG, path + ": truncated or corrupt gzip stream"};
        size_t n_par = 0;
        if (parse_fastq_parallel(plain.data(), plain.data() + plain.size(), seq, qual, offsets, ids, any_qual, n_par)) return n_par;
        MemLines in{plain.data(), plain.data() + plain.size()};
        return read_records(in, path, seq, qual, offsets, ids, any_qual);
    }
    GzLines in(path, bzip2);
    if (!in.good()) throw Error{SVT_ERR_ARG, "cannot open " + path};
    const size_t got = read_records(in, path, seq, qual, offsets, ids, any_qual);
    if (!bzip2 && !in.clean_end()) throw Error{SVT_ERR_ARG, path + ": truncated or corrupt gzip stream"};
    return got;
}

// the files of one run, in list order (position = sample index; an empty name holds its position).  One file: parsed on the calling thread (its gz buffer stays warm for the
// next load).  Several (--pooled-samples: one per sample; a run's many small .fq.gz): inflated and parsed side by side on the pool, every file into arrays of its own, then
// appended in list order -- the records, their order and the errors of the one-after-the-other loop (a file's own error, the first in list order; FASTA and FASTQ files mixed:
// reported for the first file that breaks the rule).
void read_fastx_files(const std::vector<std::string>& files, RawBytes& seq, RawBytes& qual, std::vector<u64>& off, std::vector<std::string>& ids,
                      std::vector<u32>& file_idx, bool& any_qual) {
    size_t real = 0; for (auto& f : files) real += !f.empty();
    if (off.empty()) off.push_back(0);
    if (real <= 1) {
        for (size_t i = 0; i < files.size(); i++) if (!files[i].empty()) { const size_t n = read_fastx_file(files[i], seq, qual, off, ids, any_qual); file_idx.insert(file_idx.end(), n, (u32)i); }
        return;
    }
    struct One { RawBytes seq, qual; std::vector<u64> off; std::vector<std::string> ids; bool any_qual = false; std::string err; int code = 0; size_t n = 0; };
    std::vector<One> parts(files.size());
    par_for(files.size(), [&](size_t i) {
        if (files[i].empty()) return;
        One& o = parts[i];
        try { o.n = read_fastx_file(files[i], o.seq, o.qual, o.off, o.ids, o.any_qual, false); }
        catch (const Error& e) { o.err = e.msg; o.code = e.code ? e.code : SVT_ERR_ARG; }
    });
    for (size_t i = 0; i < parts.size(); i++) {
        One& o = parts[i];
        if (o.code) throw Error{o.code, o.err};
        if (o.n == 0) continue;
        if (o.any_qual && !any_qual && !seq.empty()) throw Error{SVT_ERR_ARG, files[i] + ": FASTQ after FASTA records (mixed inputs are not supported)"};
        if (!o.any_qual && any_qual) throw Error{SVT_ERR_ARG, files[i] + ": FASTA after FASTQ records (mixed inputs are not supported)"};
        any_qual = any_qual || o.any_qual;
        const u64 base = seq.size();
        seq.insert(seq.end(), o.seq.begin(), o.seq.end());
        if (o.any_qual) qual.insert(qual.end(), o.qual.begin(), o.qual.end());
        for (size_t r = 1; r < o.off.size(); r++) off.push_back(base + o.off[r]);
        for (auto& id : o.ids) ids.push_back(std::move(id));
        file_idx.insert(file_idx.end(), o.n, (u32)i);
        RawBytes().swap(o.seq); RawBytes().swap(o.qual);         // a file's arrays go as soon as they are appended
    }
}

// ---- final ASV list (src/main.rs:140-200) ------------------------------------------------------------
std::vector<FinalAsv> finalize_asvs(const std::vector<ConsensusSequence>& consensuses, const EmResult& em, const std::vector<std::vector<u64>>* per_sample) {
    std::vector<FinalAsv> out;
    for (size_t i = 0; i < consensuses.size(); i++) {
        FinalAsv a;
        a.sequence = consensuses[i].decompressed; a.debug_id = consensuses[i].id; a.cluster = consensuses[i].cluster;
        a.depth = em.kept_original || i >= em.depth.size() ? consensuses[i].depth : em.depth[i];      // src/alignment.rs:1952-1955 / :2011-2026
        if (i < em.unambig.size()) { a.unambig = em.unambig[i]; a.ambig = em.ambig[i]; a.leq10 = em.leq10[i]; }
        if (per_sample && i < per_sample->size()) a.per_sample = (*per_sample)[i];
        if (!em.kept_original && a.depth == 0) continue;                                               // retain(|c| c.depth > 0) :2029
        out.push_back(std::move(a));
    }
    std::stable_sort(out.begin(), out.end(), [](const FinalAsv& a, const FinalAsv& b) { return a.depth > b.depth; });   // src/main.rs:143
    return out;
}

static std::string f64_display(double v) {            // Rust `{}` for f64: shortest digits that round-trip, never an exponent
    char buf[64]; auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}
static std::string depth_field(const FinalAsv& a) {
    if (a.per_sample.empty()) return std::to_string(a.depth);
    std::string s; for (size_t i = 0; i < a.per_sample.size(); i++) { if (i) s += "-"; s += std::to_string(a.per_sample[i]); }
    return s;
}

// write_consensus_fasta, src/alignment.rs:830-860
void write_consensus_fasta(const std::vector<FinalAsv>& asvs, const std::string& path, const std::string& prefix) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    for (size_t i = 0; i < asvs.size(); i++) {
        const FinalAsv& a = asvs[i];
        size_t s = 0, e = a.sequence.size();
        while (s < e && a.sequence[s] == 'N') s++;
        while (e > s && a.sequence[e - 1] == 'N') e--;
        if (s >= e) { s = 0; e = a.sequence.size(); }                                   // find / rfind ... unwrap_or(0 / len)
        fprintf(f, ">%s_consensus_%zu_depth_%s debug_id:%zu chimera_score:%lld unambiguous_read_assignments:%llu ambig_read_assignments:%llu num_align_leq_10_mismatches:%llu\n",
                prefix.c_str(), i, depth_field(a).c_str(), a.debug_id, (long long)a.chimera_score, (unsigned long long)a.unambig, (unsigned long long)a.ambig, (unsigned long long)a.leq10);
        fwrite(a.sequence.data() + s, 1, e - s, f); fputc('\n', f);
    }
    fclose(f);
}
// write_feature_table, src/main.rs:381-400
void write_feature_table(const std::vector<FinalAsv>& asvs, const std::string& path, const std::vector<std::string>& sample_names) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    fputs("#OTU ID\t", f);
    for (size_t i = 0; i < sample_names.size(); i++) { if (i) fputc('\t', f); fputs(sample_names[i].c_str(), f); }
    fputc('\n', f);
    for (size_t i = 0; i < asvs.size(); i++) {
        const FinalAsv& a = asvs[i];
        if (a.per_sample.empty()) fprintf(f, "final_consensus_%zu_depth_%zu\t%zu\n", i, a.depth, a.depth);
        else {
            fprintf(f, "final_consensus_%zu_depth_%s\t", i, depth_field(a).c_str());
            for (size_t s = 0; s < a.per_sample.size(); s++) { if (s) fputc('\t', f); fprintf(f, "%llu", (unsigned long long)a.per_sample[s]); }
            fputc('\n', f);
        }
    }
    fclose(f);
}
// write_clusters_tsv, src/alignment.rs:799-826.  The label is `consensus.id`: the final list is renumbered by the caller first
// (src/main.rs:196, by_index = true), the intermediate dumps of the temp directory keep the cluster id (by_index = false)
void write_clusters_tsv(const std::vector<FinalAsv>& asvs, const ReadSet& rs, const TwinReads& tw, const std::string& path, const std::string& prefix, bool by_index) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    for (size_t i = 0; i < asvs.size(); i++) {
        const FinalAsv& a = asvs[i];
        if (a.cluster.empty()) continue;
        fprintf(f, "%s_cluster_%zu\tsize_%zu\trepresentative_%u\tmembers\n", prefix.c_str(), by_index ? i : a.debug_id, a.cluster.size(), a.cluster[0]);
        for (size_t m = 0; m < a.cluster.size(); m++) {
            const u32 t = a.cluster[m];
            const double est = tw.est_valid[t] ? tw.est_id[t] : 100.0;                  // est_id.unwrap_or(100.)
            fprintf(f, "%s %s\n", rs.ids[tw.orig[t]].c_str(), f64_display(est).c_str());
        }
    }
    fclose(f);
}

// read_to_asv_mappings.tsv (src/alignment.rs:1538-1541, :1604-1608, :1874-1886).  SNPmer path: per read the aligned lowest-mismatch ASVs in
// ascending nm, at most five, as `id \t debug_id:<consensus id> \t <SNPmer mismatches> \t <nm>` (the reference names the columns
// "mismatches, mini_matches" but destructures (asv, nm, mismatches) into them); low-polymorphism path: `id \t debug_id:<id> \t <best nm>` for
// the best ASVs of a kept read.  Reads in input order (the reference writes from a parallel loop: any order).
void write_read_to_asv_mappings(const EmResult& em, const std::vector<size_t>& consensus_ids, const ReadSet& rs, const TwinReads& tw, bool low_polymorphism, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    for (size_t r = 0; r < em.read_lines.size() && r < tw.n; r++)
        for (const EmResult::MapLine& l : em.read_lines[r]) {
            const size_t id = l.asv < consensus_ids.size() ? consensus_ids[l.asv] : l.asv;
            if (low_polymorphism) fprintf(f, "%s\tdebug_id:%zu\t%d\n", rs.ids[tw.orig[r]].c_str(), id, l.b);
            else fprintf(f, "%s\tdebug_id:%zu\t%u\t%d\n", rs.ids[tw.orig[r]].c_str(), id, l.a, l.b);
        }
    fclose(f);
}

// ---- the reference's intermediate files (`<out>/temp/`): stage-level parity probes a savont maintainer can diff against a real run ----
std::vector<FinalAsv> as_records(const std::vector<ConsensusSequence>& cons) {
    std::vector<FinalAsv> out;
    for (auto& c : cons) {
        FinalAsv a; a.sequence = c.decompressed.empty() ? c.sequence : c.decompressed;   // write_consensus_fasta decompresses a clone (hp lengths are all 1 here)
        a.depth = c.depth + c.appended_depth; a.debug_id = c.id; a.cluster = c.cluster;
        out.push_back(std::move(a));
    }
    return out;
}
static void join_members(FILE* f, const std::vector<uint32_t>& cl) { for (size_t i = 0; i < cl.size(); i++) fprintf(f, i ? ",%u" : "%u", cl[i]); }
// kmer_clusters_stage2.tsv, src/asv_cluster.rs:223-236
void write_kmer_clusters_tsv(const std::vector<std::vector<uint32_t>>& clusters, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    fputs("cluster_id\tsize\trepresentative\tmembers\n", f);
    for (size_t i = 0; i < clusters.size(); i++) { fprintf(f, "cluster_%zu\t%zu\t%u\t", i, clusters[i].size(), clusters[i][0]); join_members(f, clusters[i]); fputc('\n', f); }
    fclose(f);
}
// snpmer_clusters_before_reclust2.5.tsv, src/asv_cluster.rs:724-745 (the reference walks an FxHashMap of groups: here ascending group id)
void write_pre_recluster_tsv(const std::vector<std::vector<uint32_t>>& pre, const std::vector<uint32_t>& group, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    fputs("kmer_cluster_id\tsnpmer_cluster_id\tsize\trepresentative\tmembers\n", f);
    uint32_t cur = ~0u; size_t local = 0;
    for (size_t i = 0; i < pre.size(); i++) {
        if (group[i] != cur) { cur = group[i]; local = 0; }
        if (pre[i].empty()) { local++; continue; }
        fprintf(f, "%u\t%zu\t%zu\t%u\t", cur, local++, pre[i].size(), pre[i][0]); join_members(f, pre[i]); fputc('\n', f);
    }
    fclose(f);
}
// final_snpmer_clusters_stage3.tsv, src/asv_cluster.rs:779-792
void write_snpmer_clusters_tsv(const std::vector<std::vector<uint32_t>>& clusters, const ReadSet& rs, const TwinReads& tw, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Error{SVT_ERR_ARG, "cannot write " + path};
    for (size_t i = 0; i < clusters.size(); i++) {
        fprintf(f, "final_cluster_%zu\tsize_%zu\trepresentative_%u\tmembers\n", i, clusters[i].size(), clusters[i][0]);
        for (size_t m = 0; m < clusters[i].size(); m++) {
            const uint32_t t = clusters[i][m];
            fprintf(f, m + 1 < clusters[i].size() ? "%s %s\n" : "%s %s\n", rs.ids[tw.orig[t]].c_str(), f64_display(tw.est_valid[t] ? tw.est_id[t] : 100.0).c_str());
        }
    }
    fclose(f);
}

}  // namespace savont
