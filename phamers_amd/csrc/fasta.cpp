// fasta.cpp -- native FASTA ingest for the count path.
//
// Replaces the Biopython passes of kmer.count_file (scripts/kmer.py:124-140: fileIO.get_fasta_ids
// + SeqIO.parse) and fileIO.read_fasta (scripts/fileIO.py:28-42) with one multi-threaded pass that
// yields exactly what the count kernels consume: the concatenated sequence bytes + CSR offsets,
// plus the record titles for the host-side id rules.  Plain and gzip files (".gz", as the
// reference: scripts/kmer.py:131-134).
//
// Record semantics follow Bio.SeqIO's FASTA parser, which the reference relies on:
//   * text before the first line that starts with '>' is skipped;
//   * title = the '>' line without '>' and without trailing white space; record.id = its first
//     white-space delimited word;
//   * sequence = the following lines up to the next '>' line, each stripped of trailing white
//     space, joined, with every ' ' and '\r' removed (other characters are kept verbatim and simply
//     count as non-symbols later).
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "phk_common.h"

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// One line of the file: returns the position of its '\n' (n when there is none) and, in *first_special, the position of
// the first ' ' or '\r' before it (SIZE_MAX when there is none).  The parser looks at every line twice (measure, then
// copy) and used three memchr calls per look -- ~70-byte lines, 70 million of them in a 5 GB file: the calls were most of
// the parse.  One pass over 32 bytes at a time with AVX2 where the CPU has it (checked once), the three memchr otherwise.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static size_t scan_line_avx2(const char *b, size_t p, size_t n, size_t *first_special) {
    const __m256i nl = _mm256_set1_epi8('\n'), sp = _mm256_set1_epi8(' '), cr = _mm256_set1_epi8('\r');
    size_t fs = SIZE_MAX;
    while (p + 32 <= n) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(b + p));
        const uint32_t mn = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        uint32_t ms = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, sp), _mm256_cmpeq_epi8(v, cr)));
        if (mn) {
            const unsigned at = (unsigned)__builtin_ctz(mn);
            ms &= at ? (0xFFFFFFFFu >> (32 - at)) : 0u;
            if (ms && fs == SIZE_MAX) fs = p + (unsigned)__builtin_ctz(ms);
            *first_special = fs;
            return p + at;
        }
        if (ms && fs == SIZE_MAX) fs = p + (unsigned)__builtin_ctz(ms);
        p += 32;
    }
    for (; p < n; ++p) {
        const char c = b[p];
        if (c == '\n') break;
        if ((c == ' ' || c == '\r') && fs == SIZE_MAX) fs = p;
    }
    *first_special = fs;
    return p;
}
#endif
static size_t scan_line_plain(const char *b, size_t p, size_t n, size_t *first_special) {
    const char *nl = (const char *)memchr(b + p, '\n', n - p);
    const size_t eol = nl ? (size_t)(nl - b) : n;
    const char *s1 = (const char *)memchr(b + p, ' ', eol - p), *s2 = (const char *)memchr(b + p, '\r', eol - p);
    const size_t f1 = s1 ? (size_t)(s1 - b) : SIZE_MAX, f2 = s2 ? (size_t)(s2 - b) : SIZE_MAX;
    *first_special = f1 < f2 ? f1 : f2;
    return eol;
}
typedef size_t (*scan_line_fn)(const char *, size_t, size_t, size_t *);
static scan_line_fn pick_scan_line() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2") && !getenv("PHK_FASTA_NO_SIMD")) return scan_line_avx2;
#endif
    return scan_line_plain;
}
static const scan_line_fn scan_line = pick_scan_line();

// Byte buffer without value-initialisation (std::vector<char>::resize would zero-fill gigabytes).  Large buffers are
// anonymous mappings with transparent huge pages asked for: the sequence buffer of a 5 GB FASTA is first touched by the
// parser's threads, and 4 KB pages mean 1.2 M page faults in the phase that writes it.
struct RawBytes {
    char *p = nullptr;
    size_t n = 0, mapped = 0;
    RawBytes() = default;
    RawBytes(const RawBytes &) = delete;
    RawBytes &operator=(const RawBytes &) = delete;
    ~RawBytes() { release(); }
    void release() {
        if (!p) return;
        if (mapped) {
            // The kernel clears pages as it frees them (0.23 s for 5 GB from one thread, and munmap holds the address
            // space's lock meanwhile: every other thread's page faults and allocations wait).  MADV_DONTNEED gives the
            // pages back under the shared lock, so slices are released from several threads; the munmap that follows
            // finds nothing left to free.
            const size_t huge = (size_t)2u << 20;
            if (mapped >= ((size_t)256u << 20)) {
                unsigned nt = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
                if (const char *e = getenv("PHK_FREE_THREADS")) nt = (unsigned)std::max(0, atoi(e));
                const size_t pages = mapped / huge;
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < nt; ++t) {
                    const size_t lo = pages * t / nt * huge, hi = (t + 1 == nt) ? mapped : pages * (t + 1) / nt * huge;
                    if (hi > lo) pool.emplace_back([this, lo, hi]() { (void)madvise(p + lo, hi - lo, MADV_DONTNEED); });
                }
                for (auto &th : pool) th.join();
            }
            munmap(p, mapped);
        } else delete[] p;
        p = nullptr;
        n = mapped = 0;
    }
    void resize(size_t m) {
        release();
        if (m >= (size_t)(4u << 20)) {
            const size_t len = (m + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            void *q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q != MAP_FAILED) {
                (void)madvise(q, len, MADV_HUGEPAGE);
                p = (char *)q;
                mapped = len;
                n = m;
                return;
            }
        }
        p = new char[m];
        n = m;
    }
    char *data() { return p; }
    const char *data() const { return p; }
};

struct phk_fasta {
    RawBytes bases;                   // all sequences, concatenated
    std::vector<uint64_t> offsets;    // n + 1
    std::vector<char> titles;         // all titles, concatenated (no terminators)
    std::vector<uint64_t> title_off;  // n + 1
    std::vector<char> ids;            // PhaMers ids (phk_parse_id of each record.id), concatenated
    std::vector<uint64_t> id_off;     // n + 1
    std::vector<uint8_t> id_status;   // n: PHK_ID_OK / PHK_ID_INDEX_ERROR / PHK_ID_NONE
};

// ------------------------------------------------------------------------------------------------
// FASTA header -> PhaMers id (scripts/id_parser.py:18-100), restated as a small field scanner.
//   header holding "_ID_"      : contig   -- the '_'-field after the first field that is exactly "ID", taken from
//                                 the header stripped of outer white space and of every '>', minus every "-circular"
//   exactly four '|'           : phage    -- the fourth '|'-field, minus every '>'
//   anything else              : bacteria -- the text before the first ' ' if it looks like a GenBank accession
//                                 (not a number, '.' second to last), else the second TAB-field (minus '>') under the
//                                 same test; without a TAB the reference indexes past the end of a list.
// ------------------------------------------------------------------------------------------------
static bool py_float_parses(const char *s, size_t n) {
    // what Python's float() accepts, for the strings that occur as FASTA words: optional outer white space, then a
    // decimal float / inf / infinity / nan in any case.  strtod also takes hex floats and "nan(...)": excluded.
    while (n && is_space(s[0])) { ++s; --n; }
    while (n && is_space(s[n - 1])) --n;
    if (n == 0 || n > 400) return false;
    for (size_t i = 0; i < n; ++i)
        if (s[i] == 'x' || s[i] == 'X' || s[i] == '(' || s[i] == '_' || s[i] == '\0') return false;
    char tmp[408];
    memcpy(tmp, s, n);
    tmp[n] = 0;
    char *end = nullptr;
    (void)strtod(tmp, &end);
    return end == tmp + n;
}

// 1 accession-like, 0 not, -1 the reference's test indexes out of range (string shorter than 2, not a number)
static int looks_like_accession(const char *s, size_t n) {
    if (py_float_parses(s, n)) return 0;
    if (n < 2) return -1;
    return s[n - 2] == '.' ? 1 : 0;
}

static void append_without(std::string &out, const char *s, size_t n, const char *drop) {
    const size_t dl = strlen(drop);
    for (size_t i = 0; i < n;) {
        if (dl && i + dl <= n && memcmp(s + i, drop, dl) == 0) i += dl;
        else out.push_back(s[i++]);
    }
}

static int parse_id(const char *h, size_t n, std::string &out) {
    out.clear();
    static const char tag[4] = {'_', 'I', 'D', '_'};
    const bool contig = n >= 4 && std::search(h, h + n, tag, tag + 4) != h + n;
    if (contig) {
        // strip(), then drop every '>'
        size_t a = 0, b = n;
        while (a < b && is_space(h[a])) ++a;
        while (b > a && is_space(h[b - 1])) --b;
        std::string t;
        append_without(t, h + a, b - a, ">");
        // walk the '_'-fields: the one after the first field equal to "ID"
        size_t f0 = 0;
        bool take_next = false;
        for (size_t i = 0; i <= t.size(); ++i) {
            if (i == t.size() || t[i] == '_') {
                if (take_next) {
                    append_without(out, t.data() + f0, i - f0, "-circular");
                    return PHK_ID_OK;
                }
                take_next = (i - f0 == 2 && t[f0] == 'I' && t[f0 + 1] == 'D');
                f0 = i + 1;
            }
        }
        return PHK_ID_INDEX_ERROR;  // "ID" was the last field: nothing follows it
    }
    size_t bars = 0;
    for (size_t i = 0; i < n; ++i) bars += h[i] == '|';
    if (bars == 4) {
        size_t f0 = 0, field = 0;
        for (size_t i = 0; i <= n; ++i)
            if (i == n || h[i] == '|') {
                if (field == 3) {
                    append_without(out, h + f0, i - f0, ">");
                    return PHK_ID_OK;
                }
                ++field;
                f0 = i + 1;
            }
        return PHK_ID_INDEX_ERROR;  // unreachable with four bars
    }
    // bacteria rule
    const char *sp = (const char *)memchr(h, ' ', n);
    const size_t w = sp ? (size_t)(sp - h) : n;
    int acc = looks_like_accession(h, w);
    if (acc < 0) return PHK_ID_INDEX_ERROR;
    if (acc == 1) {
        out.assign(h, w);
        return PHK_ID_OK;
    }
    const char *t1 = (const char *)memchr(h, '\t', n);
    if (!t1) return PHK_ID_INDEX_ERROR;  // split('\t')[1] of a header without a TAB
    const char *f = t1 + 1;
    const char *t2 = (const char *)memchr(f, '\t', (size_t)(h + n - f));
    std::string second;
    append_without(second, f, (size_t)((t2 ? t2 : h + n) - f), ">");
    acc = looks_like_accession(second.data(), second.size());
    if (acc < 0) return PHK_ID_INDEX_ERROR;
    if (acc == 1) {
        out = second;
        return PHK_ID_OK;
    }
    return PHK_ID_NONE;
}

extern "C" int phk_parse_id(const char *header, uint64_t header_len, char *id_out, uint64_t cap, uint64_t *id_len,
                            int *status) {
    PHK_REQUIRE(header || header_len == 0, "phk_parse_id: NULL header");
    PHK_REQUIRE(status && id_len, "phk_parse_id: NULL output");
    std::string out;
    *status = parse_id(header ? header : "", header_len, out);
    *id_len = out.size();
    if (*status == PHK_ID_OK && id_out) {
        PHK_REQUIRE(cap >= out.size(), "phk_parse_id: id needs %llu bytes", (unsigned long long)out.size());
        memcpy(id_out, out.data(), out.size());
    }
    return PHK_OK;
}

// The file's bytes: a read-only mapping for a plain file (no copy, no zero-fill of a multi-GB buffer; the worker threads
// fault the pages in while they scan), an inflated buffer for a .gz file.
struct FileBytes {
    const char *data = nullptr;
    size_t size = 0;
    void *map = nullptr;
    size_t map_len = 0;
    RawBytes owned;           // an inflated ".gz" stream (or the part of it a rank needs)
    // a partial read (BGZF ".gz", one rank's byte range): data[0] is byte `base` of the uncompressed stream of `total` bytes
    size_t base = 0, total = 0;
    ~FileBytes() {
        if (!map) return;
        if (map_len >= ((size_t)256u << 20)) {
            // tearing down a multi-GB mapping: the page tables are emptied in slices from several threads under the SHARED
            // address-space lock (MADV_DONTNEED), so that the munmap, which holds it exclusively -- every page fault of the
            // process waits meanwhile: a detached munmap made the caller's next 1 GB array three times slower -- finds nothing
            // left to do
            const size_t page = (size_t)2u << 20, pages = map_len / page;
            const unsigned nt = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
            std::vector<std::thread> pool;
            char *base = (char *)map;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t lo = pages * t / nt * page, hi = (t + 1 == nt) ? map_len : pages * (t + 1) / nt * page;
                if (hi > lo) pool.emplace_back([base, lo, hi]() { (void)madvise(base + lo, hi - lo, MADV_DONTNEED); });
            }
            for (auto &th : pool) th.join();
        }
        munmap(map, map_len);
    }
};

static int map_plain_file(const char *path, FileBytes &fb) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return PHK_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return PHK_ERR_IO;
    }
    fb.size = (size_t)st.st_size;
    if (fb.size == 0) {
        close(fd);
        fb.data = "";
        return PHK_OK;
    }
    void *p = mmap(nullptr, fb.size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return PHK_ERR_IO;
    (void)madvise(p, fb.size, MADV_SEQUENTIAL);
    fb.map = p;
    fb.map_len = fb.size;
    fb.data = (const char *)p;
    return PHK_OK;
}

// ---- ".gz" input (scripts/kmer.py:131-134, scripts/fileIO.py:34,53,68,86 accept zipped FASTA everywhere) -------------------
// The compressed file is mapped and inflated straight into ONE buffer (rounds 1-4: gzread into a std::vector that doubled --
// every doubling a copy of all that was read so far, every resize a zero-fill).  Two shapes:
//   * BGZF (bgzip: what genomics tools write; every <= 64 KiB block a gzip member that carries its compressed size in an
//     extra field and its uncompressed size in its trailer): the block table is read from the headers alone, so the blocks
//     are inflated IN PARALLEL, each to its final place -- and one rank of a sharded run inflates only the blocks of its
//     byte range (open_gz_range);
//   * any other gzip stream (one member or several): zlib's inflate is serial -- ~0.4 GB/s, which bounds such a file whatever
//     follows -- into a buffer sized by the ISIZE trailer of the last member, grown when that was not the whole story.
struct BgzfBlock { size_t coff, csize, uoff; uint32_t usize; };

static bool bgzf_index(const uint8_t *z, size_t zn, std::vector<BgzfBlock> &blocks) {
    size_t p = 0, u = 0;
    while (p < zn) {
        if (zn - p < 28) return false;
        if (z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8 || !(z[p + 3] & 4)) return false;
        const size_t xlen = (size_t)z[p + 10] | ((size_t)z[p + 11] << 8);
        if (zn - p < 12 + xlen + 8) return false;
        size_t bsize = 0;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {   // the extra subfields: 'B' 'C' len=2 BSIZE-1
            const size_t slen = (size_t)z[q + 2] | ((size_t)z[q + 3] << 8);
            if (z[q] == 'B' && z[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = ((size_t)z[q + 4] | ((size_t)z[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > zn - p) return false;
        const uint8_t *t = z + p + bsize - 4;
        const uint32_t usize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (usize > 65536u) return false;
        blocks.push_back(BgzfBlock{p, bsize, u, usize});
        u += usize;
        p += bsize;
    }
    return !blocks.empty();
}

// one whole gzip member (header, deflate stream, CRC and size checked by zlib) -> dst; the bytes it produced
static int inflate_member(const uint8_t *src, size_t n, char *dst, size_t cap, size_t *produced) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return PHK_ERR_IO;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)n;
    zs.next_out = (Bytef *)dst;
    zs.avail_out = (uInt)cap;
    const int rc = inflate(&zs, Z_FINISH);
    *produced = cap - zs.avail_out;
    inflateEnd(&zs);
    return rc == Z_STREAM_END ? PHK_OK : PHK_ERR_IO;
}

static int inflate_bgzf_blocks(const uint8_t *z, const std::vector<BgzfBlock> &blocks, size_t b_lo, size_t b_hi, char *dst, size_t dst_u0) {
    std::atomic<int> failed(0);
    phk_parallel_for(b_hi - b_lo, [&](uint64_t i) {
        const BgzfBlock &bk = blocks[b_lo + (size_t)i];
        size_t got = 0;
        if (inflate_member(z + bk.coff, bk.csize, dst + (bk.uoff - dst_u0), bk.usize, &got) != PHK_OK || got != bk.usize) failed.store(1);
    });
    return failed.load() ? PHK_ERR_IO : PHK_OK;
}

static int inflate_gz_serial(const uint8_t *z, size_t zn, RawBytes &out, size_t *size) {
    size_t cap = 0;
    if (zn >= 4) cap = (size_t)z[zn - 4] | ((size_t)z[zn - 3] << 8) | ((size_t)z[zn - 2] << 16) | ((size_t)z[zn - 1] << 24);
    if (cap < zn) cap = 4 * zn;         // (several members, or more than 4 GB: the trailer is not the size)
    cap += 64;
    out.resize(cap);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return PHK_ERR_IO;
    size_t in_at = 0, used = 0;
    int rc = Z_OK;
    for (;;) {
        if (zs.avail_in == 0 && in_at < zn) {
            const size_t take = std::min<size_t>(zn - in_at, (size_t)1 << 30);
            zs.next_in = const_cast<Bytef *>(z + in_at);
            zs.avail_in = (uInt)take;
            in_at += take;
        }
        if (used == cap) {   // the guess was short: a larger buffer, one copy
            RawBytes bigger;
            bigger.resize(cap * 2);
            memcpy(bigger.data(), out.data(), used);
            std::swap(out.p, bigger.p); std::swap(out.n, bigger.n); std::swap(out.mapped, bigger.mapped);
            cap *= 2;
        }
        const size_t room = std::min<size_t>(cap - used, (size_t)1 << 30);
        zs.next_out = (Bytef *)(out.data() + used);
        zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        used += room - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && in_at >= zn) break;          // the last member
            if (inflateReset(&zs) != Z_OK) { rc = Z_DATA_ERROR; break; }   // the next member of a multi-member file
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) break;
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_at >= zn) break;   // truncated stream
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) return PHK_ERR_IO;
    *size = used;
    return PHK_OK;
}

struct GzMap {
    const uint8_t *z = nullptr;
    size_t zn = 0;
    ~GzMap() { if (z && zn) munmap(const_cast<uint8_t *>(z), zn); }
};
static int map_gz(const char *path, GzMap &g) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return PHK_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return PHK_ERR_IO; }
    g.zn = (size_t)st.st_size;
    if (g.zn == 0) { close(fd); return PHK_ERR_IO; }     // (an empty file is not a gzip stream: gzopen would read it as empty text)
    void *p = mmap(nullptr, g.zn, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g.zn = 0; return PHK_ERR_IO; }
    g.z = (const uint8_t *)p;
    return PHK_OK;
}

// the whole stream
static int read_gz_file(const char *path, FileBytes &fb) {
    GzMap g;
    {
        struct stat st;
        if (stat(path, &st) == 0 && S_ISREG(st.st_mode) && st.st_size == 0) {   // as gzopen: an empty file reads as no text
            fb.data = "";
            fb.size = fb.total = 0;
            return PHK_OK;
        }
    }
    PHK_TRY(map_gz(path, g));
    if (g.zn < 2 || g.z[0] != 0x1f || g.z[1] != 0x8b) {   // not gzip at all: gzopen would pass the bytes through
        fb.owned.resize(g.zn + 64);
        memcpy(fb.owned.data(), g.z, g.zn);
        fb.data = fb.owned.data();
        fb.size = fb.total = g.zn;
        return PHK_OK;
    }
    std::vector<BgzfBlock> blocks;
    if (bgzf_index(g.z, g.zn, blocks)) {
        const size_t total = blocks.back().uoff + blocks.back().usize;
        fb.owned.resize(total + 64);
        PHK_TRY(inflate_bgzf_blocks(g.z, blocks, 0, blocks.size(), fb.owned.data(), 0));
        fb.data = fb.owned.data();
        fb.size = fb.total = total;
        return PHK_OK;
    }
    size_t size = 0;
    PHK_TRY(inflate_gz_serial(g.z, g.zn, fb.owned, &size));
    fb.data = fb.owned.data();
    fb.size = fb.total = size;
    return PHK_OK;
}

// One record = [begin, end) of the file buffer, begin pointing at its '>'; written at the cursors the scan pass fixed.
struct RecSpan { size_t begin, end; };

static void write_record(const char *b, size_t begin, size_t end, char *seq_out, char *title_out) {
    size_t p = begin + 1;
    const char *nl = (const char *)memchr(b + p, '\n', end - p);
    size_t eol = nl ? (size_t)(nl - b) : end;
    size_t t_end = eol;
    while (t_end > p && is_space(b[t_end - 1])) --t_end;
    memcpy(title_out, b + p, t_end - p);
    p = eol < end ? eol + 1 : end;
    while (p < end) {
        size_t fs;
        eol = scan_line(b, p, end, &fs);
        size_t l_end = eol;
        while (l_end > p && is_space(b[l_end - 1])) --l_end;
        if (fs >= l_end) {                                   // the usual line: nothing to drop
            memcpy(seq_out, b + p, l_end - p);
            seq_out += l_end - p;
        } else {
            for (size_t i = p; i < l_end; ++i)
                if (b[i] != ' ' && b[i] != '\r') *seq_out++ = b[i];
        }
        p = eol < end ? eol + 1 : end;
    }
}

static int open_fasta_bytes(const char *path, FileBytes &fb) {
    const size_t plen = strlen(path);
    int rc;
    if (plen > 3 && strcmp(path + plen - 3, ".gz") == 0) {
        rc = read_gz_file(path, fb);
    } else {
        rc = map_plain_file(path, fb);
        fb.total = fb.size;
    }
    if (rc != PHK_OK) phk_set_error("phk_fasta_read: cannot read %s", path);
    return rc;
}

// first '>' that begins a line at or after `from` (the file's size when there is none)
static size_t next_record_start(const char *b, size_t n, size_t from) {
    size_t p = from;
    if (p >= n) return n;
    if (p > 0 && b[p - 1] != '\n') {   // inside a line: move to the next line start
        const char *nl = (const char *)memchr(b + p, '\n', n - p);
        if (!nl) return n;
        p = (size_t)(nl - b) + 1;
    }
    while (p < n) {
        if (b[p] == '>') return p;
        const char *nl = (const char *)memchr(b + p, '\n', n - p);
        if (!nl) return n;
        p = (size_t)(nl - b) + 1;
    }
    return n;
}

// Where a record's sequence sits in the file, for the device's de-lining (phk_deline_pack_kernel): base i of a REGULAR record
// is byte  seq_begin + (i / lw) * (lw + tl) + i % lw  -- every sequence line but the last holds lw bases and is followed by tl
// bytes (trailing white space + the line end) before the next one begins, and no line holds a ' ' or '\r' inside.
struct RecLayout {
    size_t seq_begin = 0;     // first byte of the line after the title line
    uint32_t lw = 0, tl = 0;  // bases per line; bytes from a line's last base to the next line's first
    bool regular = true;
};

// The bytes of one byte range of the file's (uncompressed) stream -- the records whose '>' line begins in [cut_lo, cut_hi),
// cuts given as fractions num / den of the stream or as absolute positions (den = 0) -- as *b / *n.  A plain file is mapped
// and only the range's pages are touched; a BGZF ".gz" is inflated for the blocks of the range alone (in parallel), extended
// block by block until the first record of the next range is found; any other ".gz" has to be inflated whole.
static int open_fasta_cut(const char *path, uint64_t lo_num, uint64_t hi_num, uint64_t den, FileBytes &fb, const char **b, size_t *n) {
    const size_t plen = strlen(path);
    std::vector<BgzfBlock> blocks;
    GzMap g;
    bool bgzf = false;
    if (plen > 3 && strcmp(path + plen - 3, ".gz") == 0 && map_gz(path, g) == PHK_OK) bgzf = bgzf_index(g.z, g.zn, blocks);
    if (!bgzf) {
        PHK_TRY(open_fasta_bytes(path, fb));
        const size_t tot = fb.size;
        const size_t cut_lo = den ? (size_t)((unsigned __int128)tot * lo_num / den) : (lo_num < tot ? (size_t)lo_num : tot);
        const size_t cut_hi = den ? (size_t)((unsigned __int128)tot * hi_num / den) : (hi_num < tot ? (size_t)hi_num : tot);
        const size_t lo = next_record_start(fb.data, tot, cut_lo);
        const size_t hi = cut_hi >= tot ? tot : next_record_start(fb.data, tot, cut_hi);
        *b = fb.data + lo;
        *n = hi > lo ? hi - lo : 0;
        return PHK_OK;
    }
    const size_t tot = blocks.back().uoff + blocks.back().usize;
    const size_t cut_lo = den ? (size_t)((unsigned __int128)tot * lo_num / den) : (lo_num < tot ? (size_t)lo_num : tot);
    const size_t cut_hi = den ? (size_t)((unsigned __int128)tot * hi_num / den) : (hi_num < tot ? (size_t)hi_num : tot);
    fb.total = tot;
    if (cut_lo >= tot) {
        *b = "";
        *n = 0;
        return PHK_OK;
    }
    auto block_of = [&](size_t u) {   // the block that holds stream byte u (u < tot)
        size_t a = 0, z = blocks.size();
        while (z - a > 1) {
            const size_t mid = (a + z) / 2;
            if (blocks[mid].uoff <= u) a = mid; else z = mid;
        }
        return a;
    };
    const size_t b_lo = block_of(cut_lo ? cut_lo - 1 : 0);     // (the byte before the cut says whether the cut is a line start)
    for (size_t ahead = (size_t)4 << 20;; ahead *= 8) {
        const size_t want_end = cut_hi >= tot ? tot : std::min(tot, cut_hi + ahead);
        const size_t b_hi = want_end >= tot ? blocks.size() : block_of(want_end - 1) + 1;
        const size_t u0 = blocks[b_lo].uoff, u1 = b_hi < blocks.size() ? blocks[b_hi].uoff : tot;
        fb.owned.resize(u1 - u0 + 64);
        if (inflate_bgzf_blocks(g.z, blocks, b_lo, b_hi, fb.owned.data(), u0) != PHK_OK) {
            phk_set_error("phk_fasta_read: cannot inflate %s", path);
            return PHK_ERR_IO;
        }
        const char *d = fb.owned.data();
        const size_t len = u1 - u0;
        const size_t lo = next_record_start(d, len, cut_lo - u0);
        size_t hi;
        if (cut_hi >= tot) hi = len;
        else {
            hi = cut_hi >= u1 ? len : next_record_start(d, len, cut_hi - u0);
            if (hi == len && u1 < tot) continue;               // the next range's first record lies further on: more blocks
        }
        fb.base = u0;
        fb.data = d;
        fb.size = len;
        *b = d + lo;
        *n = hi > lo ? hi - lo : 0;
        return PHK_OK;
    }
}

static int parse_fasta_bytes(const char *b, size_t n, int threads, phk_fasta **out, bool with_bases = true,
                             std::vector<size_t> *starts_out = nullptr, std::vector<RecLayout> *layout_out = nullptr);

extern "C" int phk_fasta_read(const char *path, int threads, phk_fasta **out) {
    PHK_REQUIRE(path && out, "phk_fasta_read: NULL argument");
    FileBytes fb;
    PHK_TRY(open_fasta_bytes(path, fb));
    return parse_fasta_bytes(fb.data, fb.size, threads, out);
}

// Ids, titles and sequence LENGTHS only (offsets as in phk_fasta_read; the sequence bytes are not gathered, phk_fasta_data
// returns a NULL bases pointer): what the length screen of a run that took its features from the cache needs
// (scripts/phamer.py:144-157 parses the whole file again for it).  One pass over the file instead of two and no 5 GB buffer.
extern "C" int phk_fasta_index(const char *path, int threads, phk_fasta **out) {
    PHK_REQUIRE(path && out, "phk_fasta_index: NULL argument");
    FileBytes fb;
    PHK_TRY(open_fasta_bytes(path, fb));
    return parse_fasta_bytes(fb.data, fb.size, threads, out, false);
}

// The records whose '>' line begins in bytes [byte_lo, byte_hi) of the file (of the decompressed stream for ".gz"):
// consecutive ranges that tile [0, size) give every record to exactly one range, whatever the cuts hit -- the middle of
// a sequence line, of a title, a '>' inside a title.  Nothing outside [first such '>', next range's first '>') is
// touched, so a rank that reads its range of a plain file never pages in another rank's shard.
extern "C" int phk_fasta_read_range(const char *path, uint64_t byte_lo, uint64_t byte_hi, int threads, phk_fasta **out) {
    PHK_REQUIRE(path && out, "phk_fasta_read_range: NULL argument");
    PHK_REQUIRE(byte_lo <= byte_hi, "phk_fasta_read_range: byte_lo > byte_hi");
    FileBytes fb;
    const char *b = nullptr;
    size_t n = 0;
    PHK_TRY(open_fasta_cut(path, byte_lo, byte_hi, 0, fb, &b, &n));
    return parse_fasta_bytes(b, n, threads, out);
}

// part `part` of `n_parts` equal byte ranges of the file (what rank `part` of `n_parts` reads)
extern "C" int phk_fasta_read_part(const char *path, uint32_t part, uint32_t n_parts, int threads, phk_fasta **out) {
    PHK_REQUIRE(path && out, "phk_fasta_read_part: NULL argument");
    PHK_REQUIRE(n_parts >= 1 && part < n_parts, "phk_fasta_read_part: part %u of %u", part, n_parts);
    FileBytes fb;
    const char *b = nullptr;
    size_t n = 0;
    PHK_TRY(open_fasta_cut(path, part, (uint64_t)part + 1, n_parts, fb, &b, &n));
    return parse_fasta_bytes(b, n, threads, out);
}

static int parse_fasta_bytes(const char *b, const size_t n, int threads, phk_fasta **out, const bool with_bases,
                             std::vector<size_t> *starts_out, std::vector<RecLayout> *layout_out) {
    if (threads < 1) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    // One pass finds the records and measures them: the file is cut into one slice per thread (a slice begins at the
    // first line start at or after its cut); each thread lists the '>' line starts inside its slice with the title length
    // and the number of bases of the record's lines INSIDE the slice, plus the bases of the lines before its first '>'
    // (they belong to the last record of an earlier slice).
    struct RecInfo { size_t start; uint64_t n_title, n_bases; RecLayout lay; bool closed; };   // closed: a line that must be the last was seen
    const int nt = (int)std::min<size_t>((size_t)threads, std::max<size_t>(n >> 20, 1));
    std::vector<std::vector<RecInfo>> part((size_t)nt);
    std::vector<uint64_t> lead((size_t)nt, 0);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
                if (t > 0) {   // first line start at or after lo
                    const char *nl = (const char *)memchr(b + lo - 1, '\n', n - (lo - 1));
                    lo = nl ? (size_t)(nl - b) + 1 : n;
                }
                if (t + 1 < nt) {   // the slice ends where the next one begins
                    const char *nl = hi ? (const char *)memchr(b + hi - 1, '\n', n - (hi - 1)) : nullptr;
                    hi = hi ? (nl ? (size_t)(nl - b) + 1 : n) : 0;
                }
                uint64_t *acc = &lead[t];
                RecInfo *cur = nullptr;      // the record the lines belong to (none: the slice begins inside an earlier slice's record)
                for (size_t p = lo; p < hi;) {
                    size_t fs;
                    const size_t eol = scan_line(b, p, n, &fs);
                    size_t l_end = eol;
                    while (l_end > p && is_space(b[l_end - 1])) --l_end;
                    const size_t next = eol < n ? eol + 1 : n;
                    if (b[p] == '>') {
                        part[t].push_back(RecInfo{p, (uint64_t)(l_end > p + 1 ? l_end - (p + 1) : 0), 0, RecLayout(), false});
                        cur = &part[t].back();
                        cur->lay.seq_begin = next;
                        acc = &cur->n_bases;
                    } else {
                        const size_t len = l_end - p;
                        if (fs >= l_end) {
                            *acc += len;                        // the usual line: nothing to drop
                        } else {
                            for (size_t i = p; i < l_end; ++i) *acc += (b[i] != ' ' && b[i] != '\r');
                            if (cur) cur->lay.regular = false;  // something inside the line is dropped: positions are not arithmetic
                        }
                        if (cur && cur->lay.regular && len) {   // (a line without bases moves nothing: the next base is not there yet)
                            RecLayout &L = cur->lay;
                            const size_t gap = next - l_end;    // bytes to the next line's first base
                            if (cur->closed || len > 0xFFFFFFFEull || gap > 0xFFFFFFFFull) {
                                L.regular = false;              // bases after a line that had to be the last
                            } else if (L.lw == 0) {
                                if (p != L.seq_begin) L.regular = false;   // (blank lines in front of the sequence)
                                L.lw = (uint32_t)len;
                                L.tl = (uint32_t)gap;
                            } else if (len != L.lw || gap != L.tl) {
                                if (len > L.lw) L.regular = false;
                                cur->closed = true;             // shorter, or ended differently: fine if nothing follows
                            }
                        } else if (cur && cur->lay.regular && !len && cur->lay.lw) {
                            cur->closed = true;                 // a blank line inside the sequence: nothing may follow it
                        }
                    }
                    p = next;
                }
                // the slice's last record may go on in the next slice (whose thread adds those bases as its `lead`): its
                // lines were not all seen here
                if (t + 1 < nt && !part[t].empty()) part[t].back().lay.regular = false;
            });
        for (auto &th : pool) th.join();
    }
    std::vector<size_t> starts;
    {
        size_t total = 0;
        for (auto &v : part) total += v.size();
        starts.reserve(total);
        if (layout_out) layout_out->reserve(total);
    }
    phk_fasta *f = new phk_fasta();
    f->offsets.push_back(0);
    f->title_off.push_back(0);
    for (int t = 0; t < nt; ++t) {
        if (lead[t] && !starts.empty()) f->offsets.back() += lead[t];   // (text before the first record of the file is dropped)
        for (const RecInfo &ri : part[t]) {
            starts.push_back(ri.start);
            f->offsets.push_back(ri.n_bases);
            f->title_off.push_back(ri.n_title);
            if (layout_out) layout_out->push_back(ri.lay);
        }
    }
    const size_t nrec = starts.size();
    threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(nrec, 1));
    auto span = [&](size_t r) { return RecSpan{starts[r], r + 1 < nrec ? starts[r + 1] : n}; };
    auto write_all_records = [&]() {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                const size_t lo = nrec * (size_t)t / (size_t)threads, hi = nrec * (size_t)(t + 1) / (size_t)threads;
                for (size_t r = lo; r < hi; ++r) {
                    const RecSpan s = span(r);
                    if (with_bases)
                        write_record(b, s.begin, s.end, f->bases.data() + f->offsets[r], f->titles.data() + f->title_off[r]);
                    else   // the title as the scan pass measured it: '>' excluded, trailing white space already cut
                        memcpy(f->titles.data() + f->title_off[r], b + s.begin + 1, f->title_off[r + 1] - f->title_off[r]);
                }
            });
        for (auto &th : pool) th.join();
    };
    for (size_t r = 0; r < nrec; ++r) {
        f->offsets[r + 1] += f->offsets[r];
        f->title_off[r + 1] += f->title_off[r];
    }
    if (with_bases) {
        f->bases.resize(f->offsets[nrec] + 64);  // slack: the device packer reads whole 16-byte groups
        memset(f->bases.data() + f->offsets[nrec], 0, 64);
    }
    f->titles.resize(f->title_off[nrec] + 1);
    write_all_records();
    // ids: the PhaMers id of every record.id (first white-space delimited word of the title), in parallel
    {
        std::vector<std::string> chunk((size_t)threads);
        std::vector<std::vector<uint32_t>> lens((size_t)threads);
        f->id_status.assign(nrec, PHK_ID_OK);
        f->id_off.assign(nrec + 1, 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                const size_t lo = nrec * (size_t)t / (size_t)threads, hi = nrec * (size_t)(t + 1) / (size_t)threads;
                std::string id;
                lens[t].reserve(hi - lo);
                for (size_t r = lo; r < hi; ++r) {
                    const char *tt = f->titles.data() + f->title_off[r];
                    const size_t tl = f->title_off[r + 1] - f->title_off[r];
                    size_t w = 0;
                    while (w < tl && !is_space(tt[w])) ++w;
                    f->id_status[r] = (uint8_t)parse_id(tt, w, id);
                    if (f->id_status[r] != PHK_ID_OK) id.clear();
                    chunk[t] += id;
                    lens[t].push_back((uint32_t)id.size());
                }
            });
        for (auto &th : pool) th.join();
        size_t r = 0;
        for (int t = 0; t < threads; ++t)
            for (uint32_t l : lens[t]) {
                f->id_off[r + 1] = f->id_off[r] + l;
                ++r;
            }
        f->ids.resize(f->id_off[nrec] + 1);
        size_t at = 0;
        for (int t = 0; t < threads; ++t) {
            memcpy(f->ids.data() + at, chunk[t].data(), chunk[t].size());
            at += chunk[t].size();
        }
    }
    if (starts_out) starts_out->swap(starts);
    *out = f;
    return PHK_OK;
}

extern "C" int phk_fasta_ids(const phk_fasta *f, const char **ids, const uint64_t **id_offsets, const uint8_t **id_status) {
    PHK_REQUIRE(f, "phk_fasta_ids: NULL");
    if (ids) *ids = f->ids.data();
    if (id_offsets) *id_offsets = f->id_off.data();
    if (id_status) *id_status = f->id_status.data();
    return PHK_OK;
}

extern "C" int phk_fasta_ids_fixed(const phk_fasta *f, uint64_t width, char *out) {
    PHK_REQUIRE(f && out, "phk_fasta_ids_fixed: NULL");
    const size_t nrec = f->id_off.size() - 1;
    for (size_t r = 0; r < nrec; ++r) {
        const size_t l = f->id_off[r + 1] - f->id_off[r];
        PHK_REQUIRE(l <= width, "phk_fasta_ids_fixed: id %llu is %llu bytes, width %llu", (unsigned long long)r,
                    (unsigned long long)l, (unsigned long long)width);
        memcpy(out + r * width, f->ids.data() + f->id_off[r], l);
        memset(out + r * width + l, 0, width - l);
    }
    return PHK_OK;
}

extern "C" int phk_fasta_shape(const phk_fasta *f, uint64_t *n_records, uint64_t *total_bases, uint64_t *title_bytes) {
    PHK_REQUIRE(f, "phk_fasta_shape: NULL");
    if (n_records) *n_records = f->offsets.size() - 1;
    if (total_bases) *total_bases = f->offsets.back();
    if (title_bytes) *title_bytes = f->title_off.back();
    return PHK_OK;
}

extern "C" int phk_fasta_data(const phk_fasta *f, const char **bases, const uint64_t **offsets, const char **titles,
                              const uint64_t **title_offsets) {
    PHK_REQUIRE(f, "phk_fasta_data: NULL");
    if (bases) *bases = f->bases.data();
    if (offsets) *offsets = f->offsets.data();
    if (titles) *titles = f->titles.data();
    if (title_offsets) *title_offsets = f->title_off.data();
    return PHK_OK;
}

// Sequence characters of one record from file position p (a line start, or inside a line when an earlier call stopped
// there) up to `room` of them: trailing white space of a line, ' ' and '\r' dropped as write_record does.  Returns the number
// written; p is where the next call continues.
static size_t write_sequence_part(const char *b, size_t &p, size_t end, char *out, size_t room) {
    size_t done = 0;
    while (p < end && done < room) {
        size_t fs;
        const size_t eol = scan_line(b, p, end, &fs);
        size_t l_end = eol;
        while (l_end > p && is_space(b[l_end - 1])) --l_end;
        if (fs >= l_end) {
            const size_t len = l_end - p, take = len < room - done ? len : room - done;
            memcpy(out + done, b + p, take);
            done += take;
            if (take < len) {   // the chunk ends inside this line
                p += take;
                return done;
            }
        } else {
            size_t i = p;
            for (; i < l_end && done < room; ++i)
                if (b[i] != ' ' && b[i] != '\r') out[done++] = b[i];
            if (i < l_end) {
                // (stopped inside the line: what is left may be nothing but characters that are dropped -- the next call finds out)
                p = i;
                return done;
            }
        }
        p = eol < end ? eol + 1 : end;
    }
    return done;
}

// FASTA file -> device-resident batch in one go (kmer.count_file, scripts/kmer.py:114-140, as the command line's load_data
// uses it, scripts/phamer.py:131): the scan pass measures the records, then the copy pass writes the sequences STRAIGHT INTO
// the pinned staging buffers of the upload, chunk by chunk, while the previous chunk is on the bus and the packer works on
// the one before -- no 5 GB host buffer between the parser and the device, no second copy, nothing to free afterwards.
// *index_out is the file's index (titles, ids, lengths; no sequences: as phk_fasta_index), *out the batch.
static int batch_from_fasta_bytes(phk_ctx *ctx, const char *b, size_t n, int k, const char *symbols4, int threads,
                                  phk_fasta **index_out, phk_batch **out);

extern "C" int phk_batch_from_fasta_file(phk_ctx *ctx, const char *path, int k, const char *symbols4, int threads,
                                         phk_fasta **index_out, phk_batch **out) {
    PHK_REQUIRE(ctx && path && index_out && out, "phk_batch_from_fasta_file: NULL argument");
    FileBytes fb;
    PHK_TRY(open_fasta_bytes(path, fb));
    return batch_from_fasta_bytes(ctx, fb.data, fb.size, k, symbols4, threads, index_out, out);
}

// The same for ONE RANK'S SHARE of the file: the records whose '>' line begins in byte range `part` of `n_parts` equal
// ranges (as phk_fasta_read_part) -- what rank `part` of an n_parts-GPU run of the command line loads.  A plain file is
// mapped and only the rank's own pages are touched; a ".gz" stream has to be inflated whole by every rank.
extern "C" int phk_batch_from_fasta_part(phk_ctx *ctx, const char *path, uint32_t part, uint32_t n_parts, int k,
                                         const char *symbols4, int threads, phk_fasta **index_out, phk_batch **out) {
    PHK_REQUIRE(ctx && path && index_out && out, "phk_batch_from_fasta_part: NULL argument");
    PHK_REQUIRE(n_parts >= 1 && part < n_parts, "phk_batch_from_fasta_part: part %u of %u", part, n_parts);
    FileBytes fb;
    const char *b = nullptr;
    size_t n = 0;
    PHK_TRY(open_fasta_cut(path, part, (uint64_t)part + 1, n_parts, fb, &b, &n));
    return batch_from_fasta_bytes(ctx, b, n, k, symbols4, threads, index_out, out);
}

static int batch_from_fasta_bytes(phk_ctx *ctx, const char *b, const size_t n, int k, const char *symbols4, int threads,
                                  phk_fasta **index_out, phk_batch **out) {
    // Index scan (ids, titles, lengths, each record's line layout), then the file's bytes go to the device AS THEY ARE and
    // the sequences are read out of them there (phk_batch_build_raw).  Rounds 3-4 copied every sequence line into the pinned
    // staging buffers on the host (0.25 s of 16 cores per 5 GB).  Only the records the scan finds irregular -- lines of
    // different widths, blanks inside a line, a record the scan's thread slices cut -- are de-lined here, into a side buffer.
    std::vector<size_t> starts;
    std::vector<RecLayout> lay;
    phk_fasta *f = nullptr;
    const bool timing = getenv("PHK_INGEST_TIMING") != nullptr;   // (stage times on stderr; diagnostics)
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[phk ingest] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    // the file's bytes go to the device on a thread of their own, beside the index scan (which needs nothing from the device)
    const uint8_t *d_raw = nullptr;
    int rc_up = PHK_OK;
    char up_err[512] = "";
    std::thread uploader([&]() {
        rc_up = phk_raw_to_device(ctx, b, n, &d_raw);
        if (rc_up != PHK_OK) snprintf(up_err, sizeof(up_err), "%s", phk_last_error());   // (the message is per thread)
        if (timing) fprintf(stderr, "[phk ingest] %-28s %.3f s\n", "(raw bytes on the device)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    });
    const int rc_scan = parse_fasta_bytes(b, n, threads, &f, false, &starts, &lay);
    lap("index scan + ids");
    uploader.join();
    if (rc_scan != PHK_OK || rc_up != PHK_OK) {
        delete f;
        if (rc_scan == PHK_OK) phk_set_error("%s", up_err);
        return rc_scan != PHK_OK ? rc_scan : rc_up;
    }
    lap("upload joined");
    const size_t nrec = starts.size();
    const std::vector<uint64_t> &off = f->offsets;
    auto rec_end = [&](size_t r) { return r + 1 < nrec ? starts[r + 1] : n; };
    std::vector<uint64_t> rbegin(nrec);
    std::vector<uint32_t> rlw(nrec), rtl(nrec);
    std::vector<size_t> odd;           // the irregular records that hold bases
    uint64_t side_bytes = 0;
    for (size_t r = 0; r < nrec; ++r) {
        const uint64_t nb = off[r + 1] - off[r];
        if (lay[r].regular || nb == 0) {
            rbegin[r] = lay[r].seq_begin;
            rlw[r] = lay[r].lw ? lay[r].lw : 1u;
            rtl[r] = lay[r].tl;
        } else {
            rbegin[r] = side_bytes;
            rlw[r] = 0xFFFFFFFFu;
            rtl[r] = 0;
            odd.push_back(r);
            side_bytes += nb;
        }
    }
    std::vector<char> side(side_bytes + 1);
    phk_parallel_for(odd.size(), [&](uint64_t i) {
        const size_t r = odd[(size_t)i];
        const size_t p0 = starts[r] + 1, e = rec_end(r);
        const char *nl = (const char *)memchr(b + p0, '\n', e - p0);
        size_t p = nl ? (size_t)(nl - b) + 1 : e;
        write_sequence_part(b, p, e, side.data() + rbegin[r], (size_t)(off[r + 1] - off[r]));
    });
    lap("layout + irregular records");
    phk_batch *batch = nullptr;
    const int rc = phk_batch_build_raw(ctx, d_raw, side.data(), side_bytes, rbegin.data(), rlw.data(), rtl.data(), off.data(), nrec, k,
                                       symbols4, &batch);
    lap("de-line + count");
    if (timing) fprintf(stderr, "[phk ingest] %zu records, %zu irregular, %llu side bytes\n", nrec, odd.size(), (unsigned long long)side_bytes);
    if (rc != PHK_OK) {
        delete f;
        return rc;
    }
    *index_out = f;
    *out = batch;
    return PHK_OK;
}

extern "C" int phk_fasta_free(phk_fasta *f) {
    delete f;
    return PHK_OK;
}

extern "C" int phk_batch_from_ascii(phk_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n, int k,
                                    const char *symbols4, phk_batch **out);

extern "C" int phk_batch_from_fasta(phk_ctx *ctx, const phk_fasta *f, int k, const char *symbols4, phk_batch **out) {
    PHK_REQUIRE(ctx && f, "phk_batch_from_fasta: NULL");
    PHK_REQUIRE(f->bases.data() || f->offsets.back() == 0, "phk_batch_from_fasta: the file was only indexed (phk_fasta_index)");
    return phk_batch_from_ascii(ctx, f->bases.data(), f->offsets.data(), f->offsets.size() - 1, k, symbols4, out);
}

extern "C" int phk_count_fasta(phk_ctx *ctx, const phk_fasta *f, int k, const char *symbols4, int64_t *counts) {
    PHK_REQUIRE(ctx && f, "phk_count_fasta: NULL");
    PHK_REQUIRE(f->bases.data() || f->offsets.back() == 0, "phk_count_fasta: the file was only indexed (phk_fasta_index)");
    return phk_count_ascii(ctx, f->bases.data(), f->offsets.data(), f->offsets.size() - 1, k, symbols4, counts);
}
