// csvio.cpp -- native writers for the two on-disk formats either side of the path, byte
// compatible with what the reference's NumPy calls produce:
//   features CSV   fileIO.save_counts (scripts/fileIO.py:169-181): np.savetxt of [id, counts...] rows, ',' delimiter
//   scores CSV     fileIO.save_phamer_scores (scripts/fileIO.py:241-253): np.savetxt of [id, str(score)] rows, ', '
// The '#'-header block is prepared by the Python side (it is np.savetxt's header/comments rule applied to
// basic.generate_summary's text) and passed in as `prefix`; rows are formatted here on all cores -- at 1 M contigs the
// features cache is ~0.7 GB of text, minutes through np.savetxt.
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <charconv>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "phk_common.h"

static int pwrite_all(int fd, const char *p, size_t n, uint64_t off) {
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)off);
        if (w < 0) return PHK_ERR_IO;
        p += w;
        n -= (size_t)w;
        off += (uint64_t)w;
    }
    return PHK_OK;
}

// rows [lo, hi) -> text, by `fmt_row(r, out)`.  Worker t formats every nt-th chunk of 4096 rows; a chunk's file offset is
// known once all earlier chunks have been formatted, so the workers hand the running offset from chunk to chunk in
// order (a turn counter) and then write their chunk with pwrite -- formatting AND the copies into the page cache run on
// all cores (the sequential write() of the first version was two thirds of the 0.6 s a 770 MB features file took).
template <typename F>
static int write_rows(const char *path, const char *prefix, uint64_t n, F fmt_row) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        phk_set_error("cannot open %s for writing", path);
        return PHK_ERR_IO;
    }
    const size_t plen = prefix ? strlen(prefix) : 0;
    int rc = plen ? pwrite_all(fd, prefix, plen, 0) : PHK_OK;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    const uint64_t rows_per_chunk = 4096;
    const uint64_t nchunk = (n + rows_per_chunk - 1) / rows_per_chunk;
    if (nchunk < nt) nt = nchunk ? (unsigned)nchunk : 1;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t turn = 0;          // the chunk whose offset is next to be fixed
    uint64_t next_off = plen;   // ... and that offset
    std::atomic<int> failed{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt && rc == PHK_OK; ++t) {
        pool.emplace_back([&, t]() {
            std::string text;
            for (uint64_t c = t; c < nchunk; c += nt) {
                text.clear();
                const uint64_t lo = c * rows_per_chunk, hi = lo + rows_per_chunk < n ? lo + rows_per_chunk : n;
                for (uint64_t r = lo; r < hi; ++r) fmt_row(r, text);
                uint64_t off;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return turn == c; });
                    off = next_off;
                    next_off += text.size();
                    ++turn;
                }
                cv.notify_all();
                if (!failed.load() && pwrite_all(fd, text.data(), text.size(), off) != PHK_OK) failed.store(1);
            }
        });
    }
    for (auto &th : pool) th.join();
    if (failed.load()) rc = PHK_ERR_IO;
    if (close(fd) != 0 && rc == PHK_OK) rc = PHK_ERR_IO;
    if (rc != PHK_OK) phk_set_error("write to %s failed", path);
    return rc;
}

// ids of the rows: concatenated bytes + offsets, or a NumPy 'U<width>' array as it lies in memory (UCS-4 code points,
// NUL padded) -- written as str(id).encode('latin-1', 'replace') would write them, without a Python loop over 10^6 ids
struct IdSource {
    const char *blob;
    const uint64_t *off;
    const uint32_t *ucs4;
    uint64_t width;
    void append(uint64_t r, std::string &out) const {
        if (ucs4) {
            const uint32_t *p = ucs4 + r * width;
            uint64_t l = width;
            while (l && p[l - 1] == 0) --l;
            for (uint64_t i = 0; i < l; ++i) out.push_back(p[i] < 256 ? (char)p[i] : '?');
        } else {
            out.append(blob + off[r], off[r + 1] - off[r]);
        }
    }
};

static int write_counts(const char *path, const char *prefix, IdSource ids, const void *counts, int elem_bytes, uint64_t n,
                        uint64_t D) {
    return write_rows(path, prefix, n, [=](uint64_t r, std::string &out) {
        ids.append(r, out);
        char buf[24];
        for (uint64_t j = 0; j < D; ++j) {
            out.push_back(',');
            char *e;
            if (elem_bytes == 4)
                e = std::to_chars(buf, buf + sizeof(buf), static_cast<const uint32_t *>(counts)[r * D + j]).ptr;
            else
                e = std::to_chars(buf, buf + sizeof(buf), static_cast<const int64_t *>(counts)[r * D + j]).ptr;
            out.append(buf, (size_t)(e - buf));
        }
        out.push_back('\n');
    });
}

extern "C" int phk_write_counts_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                                    const void *counts, int elem_bytes, uint64_t n, uint64_t D) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_offsets && counts)), "phk_write_counts_csv: NULL argument");
    PHK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "phk_write_counts_csv: counts must be uint32 or int64");
    return write_counts(path, prefix, IdSource{ids, id_offsets, nullptr, 0}, counts, elem_bytes, n, D);
}

extern "C" int phk_write_counts_csv_ucs4(const char *path, const char *prefix, const uint32_t *ids, uint64_t id_width,
                                         const void *counts, int elem_bytes, uint64_t n, uint64_t D) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_width && counts)), "phk_write_counts_csv_ucs4: NULL argument");
    PHK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "phk_write_counts_csv_ucs4: counts must be uint32 or int64");
    return write_counts(path, prefix, IdSource{nullptr, nullptr, ids, id_width}, counts, elem_bytes, n, D);
}

// str(numpy.float64) / Python repr: the shortest digit string that round-trips, laid out as Python does --
// scientific when the decimal exponent is < -4 or >= 16, otherwise positional with at least one fractional digit
static void append_py_float(std::string &out, double v) {
    if (v != v) { out += "nan"; return; }
    if (v == __builtin_inf()) { out += "inf"; return; }
    if (v == -__builtin_inf()) { out += "-inf"; return; }
    char buf[40];
    char *e = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific).ptr;   // [-]d[.ddd]e[+-]XX
    *e = 0;
    const char *p = buf;
    if (*p == '-') { out.push_back('-'); ++p; }
    const char *ep = strchr(p, 'e');
    std::string digits;
    for (const char *q = p; q < ep; ++q)
        if (*q != '.') digits.push_back(*q);
    const int ex = atoi(ep + 1);
    if (ex < -4 || ex >= 16) {
        out.push_back(digits[0]);
        if (digits.size() > 1) {
            out.push_back('.');
            out.append(digits, 1, std::string::npos);
        }
        char eb[8];
        snprintf(eb, sizeof(eb), "e%c%02d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
        out += eb;
    } else if (ex >= 0) {
        const size_t ip = (size_t)ex + 1;
        if (digits.size() <= ip) {
            out += digits;
            out.append(ip - digits.size(), '0');
            out += ".0";
        } else {
            out.append(digits, 0, ip);
            out.push_back('.');
            out.append(digits, ip, std::string::npos);
        }
    } else {
        out += "0.";
        out.append((size_t)(-ex - 1), '0');
        out += digits;
    }
}

static int write_scores(const char *path, const char *prefix, IdSource ids, const double *scores, uint64_t n) {
    return write_rows(path, prefix, n, [=](uint64_t r, std::string &out) {
        ids.append(r, out);
        out += ", ";
        append_py_float(out, scores[r]);
        out.push_back('\n');
    });
}

extern "C" int phk_write_scores_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                                    const double *scores, uint64_t n) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_offsets && scores)), "phk_write_scores_csv: NULL argument");
    return write_scores(path, prefix, IdSource{ids, id_offsets, nullptr, 0}, scores, n);
}

extern "C" int phk_write_scores_csv_ucs4(const char *path, const char *prefix, const uint32_t *ids, uint64_t id_width,
                                         const double *scores, uint64_t n) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_width && scores)), "phk_write_scores_csv_ucs4: NULL argument");
    return write_scores(path, prefix, IdSource{nullptr, nullptr, ids, id_width}, scores, n);
}

extern "C" int phk_format_float(double v, char *out, int cap) {
    PHK_REQUIRE(out && cap > 0, "phk_format_float: NULL");
    std::string s;
    append_py_float(s, v);
    PHK_REQUIRE((int)s.size() < cap, "phk_format_float: buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return PHK_OK;
}

// ---- features CSV reader (fileIO.read_feature_file, scripts/fileIO.py:134-166) ----------------------------------
// np.loadtxt(dtype=str, delimiter=',') of a 1M-contig features cache takes minutes; the files save_counts writes
// have one shape, which is parsed here on all cores.  Every deviation from that shape is refused (not guessed at).
struct phk_features {
    const char *data = nullptr;
    size_t size = 0;
    void *map = nullptr;
    std::vector<uint64_t> row_lo, row_hi;   // data rows: [lo, hi) without the line end
    uint64_t D = 0, id_width = 0;
    ~phk_features() {
        if (map) munmap(map, size);
    }
};

extern "C" int phk_features_open(const char *path, phk_features **out, uint64_t *n, uint64_t *D, uint64_t *id_width) {
    PHK_REQUIRE(path && out, "phk_features_open: NULL argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        phk_set_error("cannot open %s", path);
        return PHK_ERR_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        phk_set_error("cannot stat %s", path);
        return PHK_ERR_IO;
    }
    phk_features *f = new phk_features;
    f->size = (size_t)st.st_size;
    if (f->size) {
        void *p = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            close(fd);
            delete f;
            phk_set_error("cannot map %s", path);
            return PHK_ERR_IO;
        }
        f->map = p;
        f->data = (const char *)p;
    }
    close(fd);
    // line index: per-thread scans of equal byte ranges, concatenated
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (f->size < (1u << 20)) nt = 1;
    std::vector<std::vector<uint64_t>> ends(nt);
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                const size_t lo = f->size * t / nt, hi = f->size * (t + 1) / nt;
                const char *p = f->data + lo, *e = f->data + hi;
                while (p < e) {
                    const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
                    if (!q) break;
                    ends[t].push_back((uint64_t)(q - f->data));
                    p = q + 1;
                }
            });
        for (auto &th : pool) th.join();
    }
    uint64_t lo = 0;
    bool bad = false;
    auto add_line = [&](uint64_t a, uint64_t b) {   // [a, b): without '\n'
        if (b > a && f->data[b - 1] == '\r') --b;
        if (a == b || f->data[a] == '#') return;            // blank or comment line
        f->row_lo.push_back(a);
        f->row_hi.push_back(b);
    };
    for (unsigned t = 0; t < nt; ++t)
        for (uint64_t e : ends[t]) {
            add_line(lo, e);
            lo = e + 1;
        }
    if (lo < f->size) add_line(lo, f->size);
    const uint64_t rows = f->row_lo.size();
    if (rows && !bad) {
        uint64_t commas = 0;
        for (uint64_t i = f->row_lo[0]; i < f->row_hi[0]; ++i) commas += f->data[i] == ',';
        f->D = commas;
        std::vector<uint64_t> wmax(nt, 0);
        std::vector<int> tbad(nt, 0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                for (uint64_t r = rows * t / nt; r < rows * (t + 1) / nt; ++r) {
                    const char *p = f->data + f->row_lo[r], *e = f->data + f->row_hi[r];
                    const char *c = (const char *)memchr(p, ',', (size_t)(e - p));
                    const uint64_t w = (uint64_t)((c ? c : e) - p);
                    for (uint64_t i = 0; i < w; ++i)
                        if ((unsigned char)p[i] >= 0x80) tbad[t] = 1;   // a 'U' array would count characters, not bytes
                    // np.loadtxt cuts a line at '#', and what it does with blanks inside fields is its business
                    if (memchr(p, '#', (size_t)(e - p)) || memchr(p, ' ', (size_t)(e - p)) || memchr(p, '\t', (size_t)(e - p)))
                        tbad[t] = 1;
                    if (w > wmax[t]) wmax[t] = w;
                }
            });
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < nt; ++t) {
            if (wmax[t] > f->id_width) f->id_width = wmax[t];
            bad = bad || tbad[t];
        }
    }
    if (bad || (rows && f->D == 0)) {
        delete f;
        phk_set_error("%s is not of the shape save_counts writes", path);
        return PHK_ERR_UNSUPPORTED;
    }
    if (f->id_width == 0) f->id_width = 1;
    if (n) *n = rows;
    if (D) *D = f->D;
    if (id_width) *id_width = f->id_width;
    *out = f;
    return PHK_OK;
}

extern "C" int phk_features_read(const phk_features *f, int64_t *counts, char *ids, uint64_t id_width) {
    PHK_REQUIRE(f && (f->row_lo.empty() || (counts && ids)), "phk_features_read: NULL argument");
    PHK_REQUIRE(id_width >= f->id_width, "phk_features_read: id_width %llu below the longest id (%llu)",
                (unsigned long long)id_width, (unsigned long long)f->id_width);
    const uint64_t rows = f->row_lo.size(), D = f->D;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (rows < 64) nt = 1;
    std::vector<int> tbad(nt, 0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() {
            for (uint64_t r = rows * t / nt; r < rows * (t + 1) / nt && !tbad[t]; ++r) {
                const char *p = f->data + f->row_lo[r], *e = f->data + f->row_hi[r];
                const char *c = (const char *)memchr(p, ',', (size_t)(e - p));
                if (!c) { tbad[t] = 1; break; }
                const uint64_t w = (uint64_t)(c - p);
                memcpy(ids + r * id_width, p, w);
                memset(ids + r * id_width + w, 0, id_width - w);
                p = c + 1;
                int64_t *row = counts + r * D;
                for (uint64_t j = 0; j < D; ++j) {
                    // a plain decimal integer, optionally signed: what int() of the reference accepts without surprises
                    bool neg = false;
                    if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
                    if (p >= e || *p < '0' || *p > '9') { tbad[t] = 1; break; }
                    uint64_t v = 0;
                    int digits = 0;
                    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (uint64_t)(*p++ - '0'); ++digits; }
                    if (digits > 18) { tbad[t] = 1; break; }
                    row[j] = neg ? -(int64_t)v : (int64_t)v;
                    if (j + 1 < D) {
                        if (p >= e || *p != ',') { tbad[t] = 1; break; }
                        ++p;
                    } else if (p != e) { tbad[t] = 1; break; }
                }
            }
        });
    for (auto &th : pool) th.join();
    for (unsigned t = 0; t < nt; ++t)
        if (tbad[t]) {
            phk_set_error("features file: a row is not 'id' + %llu plain integers", (unsigned long long)D);
            return PHK_ERR_UNSUPPORTED;
        }
    return PHK_OK;
}

extern "C" int phk_features_close(phk_features *f) {
    delete f;
    return PHK_OK;
}
