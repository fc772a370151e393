// csvio.cpp -- native writers for the two on-disk formats either side of the path, byte
// compatible with what the reference's NumPy calls produce:
//   features CSV   fileIO.save_counts (scripts/fileIO.py:169-181): np.savetxt of [id, counts...] rows, ',' delimiter
//   scores CSV     fileIO.save_phamer_scores (scripts/fileIO.py:241-253): np.savetxt of [id, str(score)] rows, ', '
// The '#'-header block is prepared by the Python side (it is np.savetxt's header/comments rule applied to
// basic.generate_summary's text) and passed in as `prefix`; rows are formatted here on all cores -- at 1 M contigs the
// features cache is ~0.7 GB of text, minutes through np.savetxt.
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <charconv>
#include <string>
#include <thread>
#include <vector>

#include "phk_common.h"

static int write_all(int fd, const char *p, size_t n) {
    while (n) {
        const ssize_t w = write(fd, p, n);
        if (w < 0) return PHK_ERR_IO;
        p += w;
        n -= (size_t)w;
    }
    return PHK_OK;
}

// rows [lo, hi) -> text, by `fmt_row(r, out)`; chunks are written in order
template <typename F>
static int write_rows(const char *path, const char *prefix, uint64_t n, F fmt_row) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        phk_set_error("cannot open %s for writing", path);
        return PHK_ERR_IO;
    }
    int rc = prefix ? write_all(fd, prefix, strlen(prefix)) : PHK_OK;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    const uint64_t rows_per_chunk = 4096;
    for (uint64_t base = 0; base < n && rc == PHK_OK; base += rows_per_chunk * nt) {
        std::vector<std::string> text(nt);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) {
            const uint64_t lo = base + t * rows_per_chunk;
            if (lo >= n) break;
            const uint64_t hi = lo + rows_per_chunk < n ? lo + rows_per_chunk : n;
            pool.emplace_back([&, t, lo, hi]() {
                std::string &out = text[t];
                for (uint64_t r = lo; r < hi; ++r) fmt_row(r, out);
            });
        }
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < nt && rc == PHK_OK; ++t)
            if (!text[t].empty()) rc = write_all(fd, text[t].data(), text[t].size());
    }
    if (close(fd) != 0 && rc == PHK_OK) rc = PHK_ERR_IO;
    if (rc != PHK_OK) phk_set_error("write to %s failed", path);
    return rc;
}

extern "C" int phk_write_counts_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                                    const void *counts, int elem_bytes, uint64_t n, uint64_t D) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_offsets && counts)), "phk_write_counts_csv: NULL argument");
    PHK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "phk_write_counts_csv: counts must be uint32 or int64");
    return write_rows(path, prefix, n, [=](uint64_t r, std::string &out) {
        out.append(ids + id_offsets[r], id_offsets[r + 1] - id_offsets[r]);
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

extern "C" int phk_write_scores_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                                    const double *scores, uint64_t n) {
    PHK_REQUIRE(path && (n == 0 || (ids && id_offsets && scores)), "phk_write_scores_csv: NULL argument");
    return write_rows(path, prefix, n, [=](uint64_t r, std::string &out) {
        out.append(ids + id_offsets[r], id_offsets[r + 1] - id_offsets[r]);
        out += ", ";
        append_py_float(out, scores[r]);
        out.push_back('\n');
    });
}

extern "C" int phk_format_float(double v, char *out, int cap) {
    PHK_REQUIRE(out && cap > 0, "phk_format_float: NULL");
    std::string s;
    append_py_float(s, v);
    PHK_REQUIRE((int)s.size() < cap, "phk_format_float: buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return PHK_OK;
}
