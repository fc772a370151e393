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
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "phk_common.h"

struct phk_fasta {
    std::vector<char> bases;          // all sequences, concatenated
    std::vector<uint64_t> offsets;    // n + 1
    std::vector<char> titles;         // all titles, concatenated (no terminators)
    std::vector<uint64_t> title_off;  // n + 1
};

static int read_whole_file(const char *path, std::vector<char> &buf) {
    const size_t plen = strlen(path);
    const bool gz = plen > 3 && strcmp(path + plen - 3, ".gz") == 0;
    if (gz) {
        gzFile f = gzopen(path, "rb");
        if (!f) return PHK_ERR_IO;
        gzbuffer(f, 1 << 20);
        size_t used = 0;
        buf.resize(1 << 24);
        for (;;) {
            if (buf.size() - used < (1 << 22)) buf.resize(buf.size() * 2);
            const int got = gzread(f, buf.data() + used, (unsigned)std::min<size_t>(buf.size() - used, 1u << 30));
            if (got < 0) {
                gzclose(f);
                return PHK_ERR_IO;
            }
            if (got == 0) break;
            used += (size_t)got;
        }
        gzclose(f);
        buf.resize(used);
        return PHK_OK;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return PHK_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return PHK_ERR_IO;
    }
    buf.resize((size_t)st.st_size);
    size_t used = 0;
    while (used < buf.size()) {
        const ssize_t got = read(fd, buf.data() + used, std::min<size_t>(buf.size() - used, 1u << 30));
        if (got < 0) {
            close(fd);
            return PHK_ERR_IO;
        }
        if (got == 0) break;
        used += (size_t)got;
    }
    close(fd);
    buf.resize(used);
    return PHK_OK;
}

static inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// One record = [begin, end) of the file buffer, begin pointing at its '>'.  Pass 0 measures, pass 1
// writes at the given cursors.
struct RecSpan { size_t begin, end; };

static void measure_record(const char *b, size_t begin, size_t end, uint64_t &n_bases, uint64_t &n_title) {
    size_t p = begin + 1;
    const char *nl = (const char *)memchr(b + p, '\n', end - p);
    size_t eol = nl ? (size_t)(nl - b) : end;
    size_t t_end = eol;
    while (t_end > p && is_space(b[t_end - 1])) --t_end;
    n_title = t_end - p;
    n_bases = 0;
    p = eol < end ? eol + 1 : end;
    while (p < end) {
        nl = (const char *)memchr(b + p, '\n', end - p);
        eol = nl ? (size_t)(nl - b) : end;
        size_t l_end = eol;
        while (l_end > p && is_space(b[l_end - 1])) --l_end;
        for (size_t i = p; i < l_end; ++i) n_bases += (b[i] != ' ' && b[i] != '\r');
        p = eol < end ? eol + 1 : end;
    }
}

static void write_record(const char *b, size_t begin, size_t end, char *seq_out, char *title_out) {
    size_t p = begin + 1;
    const char *nl = (const char *)memchr(b + p, '\n', end - p);
    size_t eol = nl ? (size_t)(nl - b) : end;
    size_t t_end = eol;
    while (t_end > p && is_space(b[t_end - 1])) --t_end;
    memcpy(title_out, b + p, t_end - p);
    p = eol < end ? eol + 1 : end;
    while (p < end) {
        nl = (const char *)memchr(b + p, '\n', end - p);
        eol = nl ? (size_t)(nl - b) : end;
        size_t l_end = eol;
        while (l_end > p && is_space(b[l_end - 1])) --l_end;
        for (size_t i = p; i < l_end; ++i)
            if (b[i] != ' ' && b[i] != '\r') *seq_out++ = b[i];
        p = eol < end ? eol + 1 : end;
    }
}

extern "C" int phk_fasta_read(const char *path, int threads, phk_fasta **out) {
    PHK_REQUIRE(path && out, "phk_fasta_read: NULL argument");
    std::vector<char> buf;
    const int rc = read_whole_file(path, buf);
    if (rc != PHK_OK) {
        phk_set_error("phk_fasta_read: cannot read %s", path);
        return rc;
    }
    const char *b = buf.data();
    const size_t n = buf.size();
    // record starts: '>' at the beginning of a line
    std::vector<size_t> starts;
    for (size_t p = 0; p < n;) {
        if (b[p] == '>') starts.push_back(p);
        const char *nl = (const char *)memchr(b + p, '\n', n - p);
        if (!nl) break;
        p = (size_t)(nl - b) + 1;
    }
    const size_t nrec = starts.size();
    phk_fasta *f = new phk_fasta();
    f->offsets.assign(nrec + 1, 0);
    f->title_off.assign(nrec + 1, 0);
    if (threads < 1) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(nrec, 1));
    auto span = [&](size_t r) { return RecSpan{starts[r], r + 1 < nrec ? starts[r + 1] : n}; };
    auto run = [&](int pass) {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                const size_t lo = nrec * (size_t)t / (size_t)threads, hi = nrec * (size_t)(t + 1) / (size_t)threads;
                for (size_t r = lo; r < hi; ++r) {
                    const RecSpan s = span(r);
                    if (pass == 0) {
                        uint64_t nb, nt;
                        measure_record(b, s.begin, s.end, nb, nt);
                        f->offsets[r + 1] = nb;
                        f->title_off[r + 1] = nt;
                    } else {
                        write_record(b, s.begin, s.end, f->bases.data() + f->offsets[r], f->titles.data() + f->title_off[r]);
                    }
                }
            });
        for (auto &th : pool) th.join();
    };
    run(0);
    for (size_t r = 0; r < nrec; ++r) {
        f->offsets[r + 1] += f->offsets[r];
        f->title_off[r + 1] += f->title_off[r];
    }
    f->bases.resize(f->offsets[nrec] + 64);  // slack: the device packer reads whole 16-byte groups
    f->titles.resize(f->title_off[nrec] + 1);
    run(1);
    *out = f;
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

extern "C" int phk_fasta_free(phk_fasta *f) {
    delete f;
    return PHK_OK;
}

extern "C" int phk_count_fasta(phk_ctx *ctx, const phk_fasta *f, int k, const char *symbols4, int64_t *counts) {
    PHK_REQUIRE(ctx && f, "phk_count_fasta: NULL");
    return phk_count_ascii(ctx, f->bases.data(), f->offsets.data(), f->offsets.size() - 1, k, symbols4, counts);
}
