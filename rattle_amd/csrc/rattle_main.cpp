// `rattle` command line: drop-in for `rattle cluster` (/root/reference/main.cpp:133-324),
// `rattle correct` (:325-412) and `rattle polish` (:612-762) over librattle_hip.so's C ABI.
// Same flags and defaults, same `clusters.out` (hps stream) and FASTQ outputs.  Host C++ only:
// every compute step goes through include/rattle_hip.h.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <chrono>

#include "../../include/rattle_hip.h"

// RATTLE_TIMING=1: wall time of the CLI's own phases on stderr (the library prints its phases the same way)
struct cli_timer {
    const char *name;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit cli_timer(const char *n) : name(n) {}
    ~cli_timer() {
        static const bool on = getenv("RATTLE_TIMING") != nullptr;
        if (on) fprintf(stderr, "[rattle cli] %-26s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// An output file of known size: the text is laid out by record offsets, formatted by all threads into an UNinitialised
// buffer (the page faults spread over the threads; a zero-filled std::vector would touch the 2 GB once more, alone), and
// written with one pwrite stream per thread.
struct sized_output {
    std::string path;
    char *p = nullptr;
    size_t size = 0;
    sized_output(const std::string &path_, size_t n) : path(path_), size(n) {
        p = (char *)malloc(n ? n : 1);
        if (!p) { fprintf(stderr, "Error: out of memory for %s\n", path.c_str()); exit(EXIT_FAILURE); }
    }
    ~sized_output() { free(p); }
    bool finish() {
        const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return false;
        std::atomic<bool> ok(true);
        const size_t piece = 32u << 20;
        const size_t n_pieces = (size + piece - 1) / piece;
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (size_t i = next++; i < n_pieces && ok; i = next++) {
                size_t at = i * piece;
                const size_t end = std::min(size, at + piece);
                while (at < end) {
                    const ssize_t w = pwrite(fd, p + at, end - at, (off_t)at);
                    if (w <= 0) { ok = false; break; }
                    at += (size_t)w;
                }
            }
        };
        const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), n_pieces}));
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
        return (close(fd) == 0) && ok;
    }
};

namespace {

struct read_t { std::string header, seq, ann, quality; };      // fasta.hpp:7-12
typedef std::vector<read_t> read_set_t;
struct cseq_t { int seq_id; bool rev; int gene_id; };            // cluster.hpp:10-13
struct cluster_t { cseq_t main_seq; std::vector<cseq_t> seqs; };
typedef std::vector<cluster_t> cluster_set_t;

[[noreturn]] void die(const std::string &m) { std::cerr << m << std::endl; exit(EXIT_FAILURE); }
void chk(int rc) { if (rc != 0) die(std::string("\nError: ") + rattle_hip_last_error()); }

// ---- argument parsing: --name v, --name=v, -n v, -nv, flags (argagg behaviour used by main.cpp) ----
struct opt_def { const char *key; std::vector<std::string> names; bool has_arg; };
struct args_t {
    std::map<std::string, std::string> v;
    bool has(const std::string &k) const { return v.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d) const { auto i = v.find(k); return i == v.end() ? d : i->second; }
    int i(const std::string &k, int d) const { auto it = v.find(k); return it == v.end() ? d : std::stoi(it->second); }
    double d(const std::string &k, double dv) const { auto it = v.find(k); return it == v.end() ? dv : std::stod(it->second); }
};

args_t parse(int argc, char **argv, const std::vector<opt_def> &defs) {
    args_t a;
    for (int i = 2; i < argc; ++i) {
        std::string tok = argv[i], val;
        bool has_val = false;
        if (tok.rfind("--", 0) == 0) {
            size_t eq = tok.find('=');
            if (eq != std::string::npos) { val = tok.substr(eq + 1); tok = tok.substr(0, eq); has_val = true; }
        } else if (tok.size() > 2 && tok[0] == '-') {
            val = tok.substr(2); tok = tok.substr(0, 2); has_val = true;
        }
        const opt_def *d = nullptr;
        for (auto &o : defs) for (auto &n : o.names) if (n == tok) d = &o;
        if (!d) die("unknown option \"" + tok + "\"");
        if (!d->has_arg) { a.v[d->key] = "1"; continue; }
        if (!has_val) { if (i + 1 >= argc) die("option \"" + tok + "\" needs an argument"); val = argv[++i]; }
        a.v[d->key] = val;
    }
    return a;
}

std::vector<std::string> split_string(const std::string &s, char d) {            // correct.cpp:20-30
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string t;
    while (getline(ss, t, d)) out.push_back(t);
    return out;
}

// ---- input: fasta.cpp:7-31 (gz), :33-205 (fasta), :207-370 (fastq) ------------------------------
std::string unzip_file(const std::string &filename, int index) {
    gzFile in = gzopen(filename.c_str(), "rb");
    std::cerr << "Start decompressing file" << std::endl;
    std::string out = filename.substr(0, index);
    if (!in) die("Error: Failed to decompress the file");
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) die("Error: Failed to decompress the file");
    std::vector<unsigned char> buf(1 << 20);
    int n;
    while ((n = gzread(in, buf.data(), (unsigned)buf.size())) > 0) fwrite(buf.data(), 1, n, f);
    fclose(f);
    gzclose(in);
    std::cerr << "Decompressing file Complete" << std::endl;
    return out;
}

struct line_reader {
    std::ifstream in;
    bool dos = false, first = true;
    explicit line_reader(const std::string &p) : in(p) {}
    bool next(std::string &l) {
        if (!std::getline(in, l)) return false;
        if (first) { dos = !l.empty() && l[l.size() - 1] == '\r'; first = false; }
        if (dos && !l.empty()) l.erase(l.size() - 1);
        return true;
    }
};

// Records of a FASTQ (4 lines each) or FASTA (header + joined, upper-cased sequence lines; quality '~').
void for_each_record(const std::string &file, bool fastq,
                     const std::function<void(const std::string &, const std::string &, const std::string &, const std::string &)> &f) {
    if (fastq) {
        // whole file in memory, lines by memchr (std::getline was the slowest part of a run once the kernels were fast);
        // same line semantics: '\n' separated, a final line without one counts, DOS endings detected on line 1
        FILE *fp = fopen(file.c_str(), "rb");
        if (!fp) return;
        std::string buf;
        {
            fseek(fp, 0, SEEK_END);
            const long sz = ftell(fp);
            fseek(fp, 0, SEEK_SET);
            buf.resize(sz > 0 ? (size_t)sz : 0);
            size_t got = 0;
            while (got < buf.size()) { const size_t n = fread(&buf[got], 1, buf.size() - got, fp); if (n == 0) break; got += n; }
            buf.resize(got);
            fclose(fp);
        }
        std::string fld[4];
        bool dos = false, first = true;
        int id = 0;
        size_t pos = 0;
        while (pos < buf.size()) {
            const char *nl = (const char *)memchr(buf.data() + pos, '\n', buf.size() - pos);
            size_t end = nl ? (size_t)(nl - buf.data()) : buf.size();
            size_t len = end - pos;
            if (first) { dos = len > 0 && buf[end - 1] == '\r'; first = false; }
            if (dos && len > 0) --len;
            fld[id].assign(buf.data() + pos, len);
            pos = end + 1;
            if (++id == 4) { f(fld[0], fld[1], fld[2], fld[3]); id = 0; }
        }
        return;
    }
    line_reader R(file);
    std::string l;
    {
        std::string h, s;
        bool have = false;
        while (R.next(l)) {
            if (l.empty()) continue;
            if (l[0] == '>') {
                if (have) f(h, s, "", std::string(s.size(), '~'));
                h = l; s.clear(); have = true;
            } else {
                for (char &c : l) c = (char)toupper((unsigned char)c);          // fasta.cpp:60,131
                s += l;
            }
        }
        if (have) f(h, s, "", std::string(s.size(), '~'));
    }
}

void resolve_input(std::string &filename, bool &fastq) {                           // main.cpp:35-57
    if (access(filename.c_str(), F_OK)) die("\nError: Input file not found! \n");
    int index = (int)filename.find_last_of(".");
    std::string ext = filename.substr(index + 1);
    if (ext == "gz") {
        filename = unzip_file(filename, index);
        index = (int)filename.find_last_of(".");
        ext = filename.substr(index + 1);
    }
    if (ext == "fq" || ext == "fastq") fastq = true;
    else if (ext == "fasta" || ext == "fa") fastq = false;
    else die("\nError: Input file format incorrect! Please use fasta/fastq file. \n");
}

// main.cpp:66-112 + fasta.cpp:207-270: everything, file order, with qualities.
read_set_t read_inputs(const std::vector<std::string> &files, const std::vector<std::string> &labels) {
    if (!labels.empty() && labels.size() != files.size()) die("\nError: Number of input files and number of label files do not match\n");
    read_set_t reads;
    int sample = 0;
    for (std::string fn : files) {
        bool fastq;
        resolve_input(fn, fastq);
        std::string lab = labels.empty() ? "" : "," + labels[sample];
        for_each_record(fn, fastq, [&](const std::string &h, const std::string &s, const std::string &a, const std::string &q) {
            reads.push_back(read_t{h + lab, s, a, q});
        });
        ++sample;
    }
    return reads;
}

void write_fastq_file(const read_set_t &reads, const std::string &file) {        // fasta.cpp:436-445
    std::ofstream f(file);
    for (auto &r : reads) f << r.header << "\n" << r.seq << "\n" << r.ann << "\n" << r.quality << "\n";
}

// ---- clusters.out: hps stream of cluster_set_t (cluster.hpp:15-18,30-33; grammar SURVEY 5) ---------
void put_uvarint(std::string &o, uint64_t x) { while (x >= 0x80) { o.push_back((char)((x & 0x7F) | 0x80)); x >>= 7; } o.push_back((char)x); }
void put_svarint(std::string &o, int32_t x) { put_uvarint(o, (uint32_t)((x << 1) ^ (x >> 31))); }

void write_clusters(const cluster_set_t &cs, const std::string &path) {
    std::string o;
    put_uvarint(o, cs.size());
    auto cseq = [&o](const cseq_t &c) { put_svarint(o, c.seq_id); o.push_back(c.rev ? 1 : 0); put_svarint(o, c.gene_id); };
    for (auto &c : cs) { cseq(c.main_seq); put_uvarint(o, c.seqs.size()); for (auto &s : c.seqs) cseq(s); }
    std::ofstream f(path, std::ofstream::binary);
    f.write(o.data(), (std::streamsize)o.size());
}

bool decode_clusters(const std::string &b, int fields, cluster_set_t &out) {
    size_t p = 0;
    bool ok = true;
    auto uv = [&]() -> uint64_t {
        uint64_t x = 0; int s = 0;
        while (true) {
            if (p >= b.size() || s > 63) { ok = false; return 0; }
            uint8_t c = (uint8_t)b[p++];
            x |= (uint64_t)(c & 0x7F) << s;
            if (!(c & 0x80)) return x;
            s += 7;
        }
    };
    auto sv = [&]() -> int32_t { uint32_t z = (uint32_t)uv(); return (int32_t)((z >> 1) ^ (~(z & 1) + 1)); };
    auto cseq = [&]() -> cseq_t {
        cseq_t c{0, false, -1};
        c.seq_id = sv();
        if (p >= b.size()) { ok = false; return c; }
        uint8_t r = (uint8_t)b[p++];
        if (r > 1) ok = false;
        c.rev = r != 0;
        if (fields == 3) c.gene_id = sv();
        return c;
    };
    out.clear();
    uint64_t n = uv();
    for (uint64_t i = 0; ok && i < n; ++i) {
        cluster_t c;
        c.main_seq = cseq();
        uint64_t m = uv();
        for (uint64_t j = 0; ok && j < m; ++j) c.seqs.push_back(cseq());
        out.push_back(c);
    }
    return ok && p == b.size();
}

cluster_set_t read_clusters(const std::string &path) {            // current 3-field layout, else the old 2-field one
    std::ifstream in(path, std::ifstream::binary);
    if (!in) die("\nError: clusters file not found! \n");
    std::string b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    cluster_set_t cs;
    if (decode_clusters(b, 3, cs) || decode_clusters(b, 2, cs)) return cs;
    die("\nError: clusters file is not an hps cluster stream\n");
}

// ---- ingest for `cluster` / `correct` (fasta.cpp:7-31,207-370, main.cpp:16-112) -----------------------
// The reference reads line by line into four std::string per record and inflates .gz inputs into a sibling file first.  Here
// a record is four spans into the file's bytes: a plain file is mapped, a .gz one inflated in memory (the sibling file is
// only written with --write-unzipped), records are found with memchr, and the concatenated base / quality buffers the library
// takes are gathered from the spans in parallel.  Same semantics: '\n' separated lines, a last line without one counts, DOS
// endings detected on the first line, FASTA sequence lines joined and upper-cased, quality '~' for FASTA.
struct span { const char *p; uint32_t n; };
struct read_table {
    std::vector<std::unique_ptr<std::string>> pools;       // owned bytes (inflated input, joined FASTA sequences, label suffixes)
    std::vector<std::pair<void *, size_t>> maps;           // mapped files
    std::vector<span> header, seq, ann, qual;
    std::vector<uint32_t> label;                           // 0: none, else 1 + index into labels
    std::vector<std::string> labels;                       // "," + label
    size_t n() const { return seq.size(); }
    std::string head(size_t i) const { return std::string(header[i].p, header[i].n) + (label[i] ? labels[label[i] - 1] : std::string()); }
    ~read_table() { for (auto &m : maps) munmap(m.first, m.second); }
    read_table() = default;
    read_table(const read_table &) = delete;
    read_table &operator=(const read_table &) = delete;
};

template <typename F>
void parallel_chunks(size_t n, F f) {                     // f(begin, end) on host threads
    const size_t T = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), (n + 4095) / 4096));
    if (T <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t) th.emplace_back([=] { f(n * t / T, n * (t + 1) / T); });
    for (auto &x : th) x.join();
}

// bytes of an input file: mapped, or inflated in memory for .gz (main.cpp:35-57 decides by extension)
void file_bytes(read_table &T, std::string &filename, bool &fastq, bool write_unzipped, const char *&data, size_t &size) {
    if (access(filename.c_str(), F_OK)) die("\nError: Input file not found! \n");
    int index = (int)filename.find_last_of(".");
    std::string ext = filename.substr(index + 1);
    bool gz = false;
    std::string inner = filename;
    if (ext == "gz") {
        gz = true;
        inner = filename.substr(0, index);
        index = (int)inner.find_last_of(".");
        ext = inner.substr(index + 1);
    }
    if (ext == "fq" || ext == "fastq") fastq = true;
    else if (ext == "fasta" || ext == "fa") fastq = false;
    else die("\nError: Input file format incorrect! Please use fasta/fastq file. \n");
    if (gz) {
        std::cerr << "Start decompressing file" << std::endl;
        gzFile in = gzopen(filename.c_str(), "rb");
        if (!in) die("Error: Failed to decompress the file");
        gzbuffer(in, 1 << 20);
        std::unique_ptr<std::string> buf(new std::string());
        size_t got = 0;
        buf->resize(64u << 20);
        int k;
        while ((k = gzread(in, &(*buf)[got], (unsigned)std::min<size_t>(buf->size() - got, 1u << 30))) > 0) {
            got += (size_t)k;
            if (buf->size() - got < (16u << 20)) buf->resize(buf->size() * 2);
        }
        gzclose(in);
        buf->resize(got);
        if (write_unzipped) {                           // the reference's side effect (fasta.cpp:7-31): the inflated sibling file
            FILE *f = fopen(inner.c_str(), "wb");
            if (!f) die("Error: Failed to decompress the file");
            fwrite(buf->data(), 1, buf->size(), f);
            fclose(f);
        }
        std::cerr << "Decompressing file Complete" << std::endl;
        data = buf->data(); size = buf->size();
        T.pools.push_back(std::move(buf));
        return;
    }
    const int fd = open(filename.c_str(), O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) die("\nError: Input file not found! \n");
    size = (size_t)st.st_size;
    data = "";
    if (size) {
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m == MAP_FAILED) die("Error: cannot map " + filename);
        (void)madvise(m, size, MADV_SEQUENTIAL);
        T.maps.emplace_back(m, size);
        data = (const char *)m;
    }
    close(fd);
}

// all records of the input files, in file order (main.cpp:66-112: `correct`, `cluster_summary`, ... read everything)
void read_table_inputs(read_table &T, const std::vector<std::string> &files, const std::vector<std::string> &labels, bool write_unzipped) {
    if (!labels.empty() && labels.size() != files.size()) die("\nError: Number of input files and number of label files do not match\n");
    for (auto &l : labels) T.labels.push_back("," + l);
    uint32_t sample = 0;
    for (std::string fn : files) {
        bool fastq = true;
        const char *d; size_t sz;
        file_bytes(T, fn, fastq, write_unzipped, d, sz);
        const uint32_t lab = labels.empty() ? 0u : sample + 1;
        if (fastq) {
            span fld[4];
            bool dos = false, first = true;
            int id = 0;
            size_t pos = 0;
            while (pos < sz) {
                const char *nl = (const char *)memchr(d + pos, '\n', sz - pos);
                const size_t end = nl ? (size_t)(nl - d) : sz;
                size_t len = end - pos;
                if (first) { dos = len > 0 && d[end - 1] == '\r'; first = false; }
                if (dos && len > 0) --len;
                fld[id] = span{d + pos, (uint32_t)len};
                pos = end + 1;
                if (++id == 4) { T.header.push_back(fld[0]); T.seq.push_back(fld[1]); T.ann.push_back(fld[2]); T.qual.push_back(fld[3]); T.label.push_back(lab); id = 0; }
            }
        } else {
            // FASTA: header + joined, upper-cased sequence lines (fasta.cpp:60,131); quality '~' x length
            std::unique_ptr<std::string> pool(new std::string());
            pool->reserve(sz + 16);
            struct rec { size_t h0; uint32_t hn; size_t s0, s1; };
            std::vector<rec> recs;
            bool dos = false, first = true, have = false;
            rec cur{0, 0, 0, 0};
            size_t pos = 0;
            while (pos < sz) {
                const char *nl = (const char *)memchr(d + pos, '\n', sz - pos);
                const size_t end = nl ? (size_t)(nl - d) : sz;
                size_t len = end - pos;
                if (first) { dos = len > 0 && d[end - 1] == '\r'; first = false; }
                if (dos && len > 0) --len;
                if (len) {
                    if (d[pos] == '>') {
                        if (have) { cur.s1 = pool->size(); recs.push_back(cur); }
                        cur = rec{pos, (uint32_t)len, pool->size(), 0}; have = true;
                    } else {
                        for (size_t t = 0; t < len; ++t) pool->push_back((char)toupper((unsigned char)d[pos + t]));
                    }
                }
                pos = end + 1;
            }
            if (have) { cur.s1 = pool->size(); recs.push_back(cur); }
            size_t longest = 0;
            for (auto &r : recs) longest = std::max(longest, r.s1 - r.s0);
            std::unique_ptr<std::string> tilde(new std::string(longest, '~'));
            for (auto &r : recs) {
                T.header.push_back(span{d + r.h0, r.hn}); T.seq.push_back(span{pool->data() + r.s0, (uint32_t)(r.s1 - r.s0)});
                T.ann.push_back(span{"", 0}); T.qual.push_back(span{tilde->data(), (uint32_t)(r.s1 - r.s0)}); T.label.push_back(lab);
            }
            T.pools.push_back(std::move(pool)); T.pools.push_back(std::move(tilde));
        }
        ++sample;
    }
}

// concatenated bases (and qualities, padded with '!' / cut to the sequence length) of the records `ids` in that order
// bytes that are filled right away by all threads: no zero fill by one thread first (a GB-sized std::vector costs 0.1 s that way)
struct byte_buf {
    std::unique_ptr<uint8_t[]> p;
    void resize(size_t n) { p.reset(new uint8_t[n]); }
    uint8_t *data() const { return p.get(); }
};

void gather_reads(const read_table &T, const std::vector<uint32_t> &ids, byte_buf &cat, byte_buf *qcat, std::vector<uint64_t> &off) {
    const size_t n = ids.size();
    off.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + T.seq[ids[i]].n;
    cat.resize(off[n] + 1);
    if (qcat) qcat->resize(off[n] + 1);
    parallel_chunks(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const span sq = T.seq[ids[i]];
            memcpy(cat.data() + off[i], sq.p, sq.n);
            if (qcat) {
                const span q = T.qual[ids[i]];
                const uint32_t m = std::min(q.n, sq.n);
                memcpy(qcat->data() + off[i], q.p, m);
                if (m < sq.n) memset(qcat->data() + off[i] + m, '!', sq.n - m);
            }
        }
    });
}

// ---- one job over several GPUs of the node: --devices 0,1,... --------------------------------------------------
// One host thread per device, each with its own context; every thread makes the same library calls and the library shards
// the work (candidates of a seed batch, --iso gene clusters, `correct` packs; include/rattle_hip.h).  The exchange is RCCL
// over xGMI; --host-exchange swaps in an in-process all-gather on host buffers (no RCCL needed; also lets several ranks
// share one device, which RCCL refuses).
struct host_exchange {
    int n = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};
struct rank_user { host_exchange *X; int rank; };
int host_allgatherv(void *user, const void *send, uint64_t, void *recv, const uint64_t *recv_bytes) {
    rank_user *u = (rank_user *)user;
    host_exchange &X = *u->X;
    X.send[u->rank] = send;
    X.barrier();                                   // every rank has published its piece
    uint64_t at = 0;
    for (int r = 0; r < X.n; ++r) { if (recv_bytes[r]) memcpy((char *)recv + at, X.send[r], recv_bytes[r]); at += recv_bytes[r]; }
    X.barrier();                                   // every rank has copied: the pieces may be released
    return 0;
}

struct device_team {
    std::vector<int> devs;
    std::vector<rattle_ctx *> ctx;
    host_exchange hx;
    std::vector<rank_user> users;
    int n() const { return (int)devs.size(); }
    // fn(rank, ctx) on every rank at once (rank 0 on the calling thread)
    void run(const std::function<void(int, rattle_ctx *)> &fn) {
        std::vector<std::thread> th;
        for (int r = 1; r < n(); ++r) th.emplace_back([&, r] { fn(r, ctx[r]); });
        fn(0, ctx[0]);
        for (auto &t : th) t.join();
    }
    void open(const args_t &a) {
        if (a.has("devices")) for (auto &t : split_string(a.str("devices", ""), ',')) devs.push_back(std::stoi(t));
        else devs.push_back(a.i("device", 0));
        if (devs.empty()) die("\nError: --devices needs a list of device indices\n");
        ctx.assign(devs.size(), nullptr);
        const bool host = a.has("host-exchange");
        uint8_t id[RATTLE_COMM_ID_BYTES] = {0};
        if (n() > 1 && !host) chk(rattle_hip_comm_unique_id(id));
        hx.n = n(); hx.send.assign(devs.size(), nullptr);
        users.resize(devs.size());
        for (int r = 0; r < n(); ++r) users[r] = rank_user{&hx, r};
        run([&](int r, rattle_ctx *) {
            chk(rattle_hip_ctx_create(devs[r], &ctx[r]));
            if (n() > 1) chk(host ? rattle_hip_set_exchange(ctx[r], r, n(), host_allgatherv, &users[r]) : rattle_hip_comm_init(ctx[r], r, n(), id));
        });
        if (n() > 1) std::cerr << "One job over " << n() << " devices (" << (host ? "host exchange" : "RCCL") << ")" << std::endl;
    }
    void close() { for (auto *c : ctx) rattle_hip_ctx_destroy(c); ctx.clear(); }
};

// ---- device calls --------------------------------------------------------------------------------
void load(rattle_ctx *ctx, const read_set_t &reads, int k, bool both) {
    std::string cat;
    std::vector<uint64_t> off(1, 0);
    for (auto &r : reads) { cat += r.seq; off.push_back(cat.size()); }
    chk(rattle_hip_load_reads(ctx, (const uint8_t *)cat.data(), off.data(), (uint32_t)reads.size(), k, both ? 1 : 0));
}

cluster_set_t to_set(rattle_cluster_set *cs) {
    cluster_set_t out(cs->n_clusters);
    for (uint32_t c = 0; c < cs->n_clusters; ++c) {
        out[c].main_seq = cseq_t{cs->main_id[c], cs->main_rev[c] != 0, -1};
        for (uint32_t i = cs->offsets[c]; i < cs->offsets[c + 1]; ++i) out[c].seqs.push_back(cseq_t{cs->member_id[i], cs->member_rev[i] != 0, -1});
    }
    rattle_hip_cluster_set_free(cs);
    return out;
}

// context creation (HIP start-up, ~0.5 s) and, for `correct`, the POA arena allocation (seconds for > 100 GB) run on a helper
// thread while the main thread reads the input
struct team_opener {
    device_team &team;
    std::thread th;
    team_opener(device_team &t, const args_t &a, uint64_t arena_hint) : team(t) {
        th = std::thread([&t, &a, arena_hint] {
            t.open(a);
            if (arena_hint) t.run([&](int, rattle_ctx *c) { (void)rattle_hip_reserve_arena(c, arena_hint); });
        });
    }
    void wait() { if (th.joinable()) th.join(); }
    ~team_opener() { wait(); }
};

uint64_t input_bytes(const std::vector<std::string> &files) {
    uint64_t tot = 0;
    for (auto &f : files) { struct stat st; if (stat(f.c_str(), &st) == 0) tot += (uint64_t)st.st_size * (f.size() > 3 && f.substr(f.size() - 3) == ".gz" ? 4 : 1); }
    return tot;
}

// (Round 5 tried leaving through _exit() once every output is closed, instead of giving ~10 GB of host buffers and a ~100 GB device
// arena back one by one: no gain -- the time outside the timers is process START, 0.5 s of library loading and HIP start-up -- and a
// loss for the next process: the kernel reclaims the device memory of a process that just vanished in the background, and the
// following `rattle correct` waited 1.1 s instead of 0.04 s for its arena, profiles/round5_cli_e2e_1M.txt.)
int mode_cluster(int argc, char **argv) {
    std::vector<opt_def> defs = {
        {"help", {"-h", "--help"}, false}, {"input", {"-i", "--input"}, true}, {"label", {"-l", "--label"}, true},
        {"output", {"-o", "--output"}, true}, {"threads", {"-t", "--threads"}, true}, {"kmer_size", {"-k", "--kmer-size"}, true},
        {"t_s", {"-s", "--score-threshold"}, true}, {"t_v", {"-v", "--max-variance"}, true}, {"iso", {"--iso"}, false},
        {"iso_kmer_size", {"--iso-kmer-size"}, true}, {"iso_t_s", {"--iso-score-threshold"}, true},
        {"iso_t_v", {"--iso-max-variance"}, true}, {"bv_threshold", {"-B", "--bv-start-threshold"}, true},
        {"bv_min_threshold", {"-b", "--bv-end-threshold"}, true}, {"bv_falloff", {"-f", "--bv-falloff"}, true},
        {"min_reads_cluster", {"-r", "--min-reads-cluster"}, true}, {"repr_percentile", {"-p", "--repr-percentile"}, true},
        {"rna", {"--rna"}, false}, {"verbose", {"--verbose"}, false}, {"raw", {"--raw"}, false},
        {"lower_len", {"--lower-length"}, true}, {"upper_len", {"--upper-length"}, true}, {"device", {"--device"}, true},
        {"devices", {"--devices"}, true}, {"host-exchange", {"--host-exchange"}, false}, {"write-unzipped", {"--write-unzipped"}, false}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) {
        std::cerr << "rattle cluster -i reads.fq [-o dir] [--rna] [--iso] ... (flags of RATTLE's cluster mode)\n"
                     "  --devices 0,1,..   one job over several GPUs (RCCL; --host-exchange: in-process exchange on host buffers)\n"
                     "  --write-unzipped   also write the inflated copy of a .gz input next to it, as the reference does\n";
        return EXIT_SUCCESS;
    }
    if (!a.has("input")) die("ERROR: No input file provided");
    int k = a.i("kmer_size", 10), iso_k = a.i("iso_kmer_size", 11);
    if (k > 16 || iso_k > 16) die("\nError: maximum kmer size = 16 \n");
    std::string outdir = a.str("output", ".");
    if (a.has("output") && access(outdir.c_str(), F_OK)) die("\nOutput folder doesn't exit. Please create it first. \n");
    bool is_rna = a.has("rna");
    std::cerr << "RNA mode: " << std::boolalpha << is_rna << std::endl;
    std::cerr << "Reading fasta file... " << std::endl;
    device_team team;
    team_opener opener(team, a, 0);
    // main.cpp:16-64 + fasta.cpp:272-370: ann = running record index over ALL records, length filter unless --raw, reads that
    // contain 'N' skipped; then sort_read_set (fasta.cpp:458-464): stable, longest first (a counting sort over the lengths)
    read_table T;
    { cli_timer t("read input"); read_table_inputs(T, split_string(a.str("input", ""), ','), split_string(a.str("label", ""), ','), a.has("write-unzipped")); }
    std::unique_ptr<cli_timer> t_sort(new cli_timer("filter + sort + gather"));
    std::vector<uint32_t> order;                  // processing position -> record index (= the reference's `ann`)
    {
        const bool raw = a.has("raw");
        const int lo = a.i("lower_len", 150), hi = a.i("upper_len", 100000);
        const size_t n = T.n();
        std::vector<uint8_t> keep(n, 0);
        std::atomic<size_t> n_skipped(0);
        parallel_chunks(n, [&](size_t b, size_t e) {
            size_t sk = 0;
            for (size_t i = b; i < e; ++i) {
                const span sq = T.seq[i];
                if (!(raw || ((int)sq.n >= lo && (int)sq.n <= hi))) continue;
                if (memchr(sq.p, 'N', sq.n)) { ++sk; continue; }
                keep[i] = 1;
            }
            n_skipped += sk;
        });
        if (n_skipped) std::cerr << "\n" << n_skipped << "  reads contains N are skipped!" << std::endl;
        uint32_t max_len = 0;
        for (size_t i = 0; i < n; ++i) if (keep[i]) max_len = std::max(max_len, T.seq[i].n);
        std::vector<uint32_t> start((size_t)max_len + 2, 0);
        for (size_t i = 0; i < n; ++i) if (keep[i]) ++start[max_len - T.seq[i].n + 1];
        for (size_t b = 0; b <= max_len; ++b) start[b + 1] += start[b];
        order.resize(start[(size_t)max_len + 1]);
        for (size_t i = 0; i < n; ++i) if (keep[i]) order[start[max_len - T.seq[i].n]++] = (uint32_t)i;
    }
    std::cout << "Reads: " << order.size() << std::endl;
    byte_buf cat;
    std::vector<uint64_t> off;
    gather_reads(T, order, cat, nullptr, off);
    std::cerr << "Done" << std::endl;
    const uint32_t n_reads = (uint32_t)order.size();

    t_sort.reset();
    { cli_timer t("wait for device"); opener.wait(); }
    std::unique_ptr<cli_timer> t_lib(new cli_timer("library: load + cluster"));
    rattle_cluster_params P;
    P.t_s = a.d("t_s", 0.2); P.t_v = a.d("t_v", 1000000); P.bv_threshold = a.d("bv_threshold", 0.4);
    P.min_bv_threshold = a.d("bv_min_threshold", 0.2); P.bv_falloff = a.d("bv_falloff", 0.05);
    P.min_reads_cluster = a.i("min_reads_cluster", 0); P.use_hc = 0; P.repr_percentile = a.d("repr_percentile", 0.15);
    P.is_rna = is_rna ? 1 : 0;
    rattle_cluster_set *raw = nullptr;
    team.run([&](int r, rattle_ctx *ctx) {                                      // every rank ends up with the same clusters
        chk(rattle_hip_load_reads(ctx, cat.data(), off.data(), n_reads, k, is_rna ? 0 : 1));
        rattle_cluster_set *mine = nullptr;
        chk(rattle_hip_cluster_reads(ctx, &P, &mine));
        if (r == 0) raw = mine; else rattle_hip_cluster_set_free(mine);
    });
    t_lib.reset();
    cli_timer t_out("to clusters.out");
    cluster_set_t gene = to_set(raw);
    std::cerr << "Gene clustering done" << std::endl;
    std::cerr << gene.size() << " gene clusters found" << std::endl;
    const std::string out_path = outdir + "/clusters.out";
    if (!a.has("iso")) {                                                         // main.cpp:264-277
        for (auto &c : gene) {
            c.main_seq.seq_id = (int)order[c.main_seq.seq_id];
            for (auto &s : c.seqs) s.seq_id = (int)order[s.seq_id];
        }
        write_clusters(gene, out_path);
        team.close();
        return EXIT_SUCCESS;
    }
    // main.cpp:281-323: second level per gene cluster with the iso parameters
    P.t_s = a.d("iso_t_s", 0.3); P.t_v = a.d("iso_t_v", 25);
    cluster_set_t iso;
    // all gene clusters at once: the subsets are independent (rattle_hip_cluster_subsets runs them concurrently, and
    // over the ranks when there are several)
    std::vector<uint32_t> ids;
    std::vector<uint64_t> sub_off(1, 0);
    for (auto &c : gene) {
        std::stable_sort(c.seqs.begin(), c.seqs.end(), [](const cseq_t &x, const cseq_t &y) { return x.seq_id > y.seq_id; });
        std::stable_sort(c.seqs.begin(), c.seqs.end(), [&](const cseq_t &x, const cseq_t &y) { return off[x.seq_id + 1] - off[x.seq_id] > off[y.seq_id + 1] - off[y.seq_id]; });
        for (auto &s : c.seqs) ids.push_back((uint32_t)s.seq_id);
        sub_off.push_back(ids.size());
    }
    std::vector<rattle_cluster_set *> subs(gene.size() ? gene.size() : 1, nullptr);
    team.run([&](int r, rattle_ctx *ctx) {
        chk(rattle_hip_load_reads(ctx, cat.data(), off.data(), n_reads, iso_k, is_rna ? 0 : 1));
        std::vector<rattle_cluster_set *> mine(gene.size() ? gene.size() : 1, nullptr);
        chk(rattle_hip_cluster_subsets(ctx, &P, ids.data(), sub_off.data(), (uint32_t)gene.size(), mine.data(), 0));
        if (r == 0) subs = mine; else for (auto *x : mine) rattle_hip_cluster_set_free(x);
    });
    int gi = 0;
    for (auto &c : gene) {
        for (auto &ic : to_set(subs[gi])) {
            cluster_t o;
            o.main_seq = cseq_t{(int)order[c.seqs[ic.main_seq.seq_id].seq_id], ic.main_seq.rev, gi};
            for (auto &s : ic.seqs) o.seqs.push_back(cseq_t{(int)order[c.seqs[s.seq_id].seq_id], s.rev, gi});
            iso.push_back(o);
        }
        ++gi;
    }
    std::cerr << "Isoform clustering done" << std::endl;
    std::cerr << iso.size() << " isoform clusters found" << std::endl;
    write_clusters(iso, out_path);
    team.close();
    return EXIT_SUCCESS;
}

int mode_correct(int argc, char **argv) {
    std::vector<opt_def> defs = {
        {"help", {"-h", "--help"}, false}, {"input", {"-i", "--input"}, true}, {"label", {"-l", "--label"}, true},
        {"clusters", {"-c", "--clusters"}, true}, {"output", {"-o", "--output"}, true}, {"gap-occ", {"-g", "--gap-occ"}, true},
        {"min-occ", {"-m", "--min-occ"}, true}, {"split", {"-s", "--split"}, true}, {"min-reads", {"-r", "--min-reads"}, true},
        {"threads", {"-t", "--threads"}, true}, {"verbose", {"--verbose"}, false}, {"device", {"--device"}, true},
        {"vote-order", {"--vote-order"}, true}, {"max-pack-cells", {"--max-pack-cells"}, true}, {"devices", {"--devices"}, true},
        {"host-exchange", {"--host-exchange"}, false}, {"write-unzipped", {"--write-unzipped"}, false}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) {
        std::cerr << "rattle correct -i reads.fq -c clusters.out [-o dir] ... (flags of RATTLE's correct mode)\n"
                     "  --max-pack-cells N   leave packs whose largest alignment needs more than N DP cells uncorrected (default: only\n"
                     "                       packs that do not fit the device are skipped); skipped packs are listed in skipped_packs.tsv\n"
                     "  --devices 0,1,..     one job over several GPUs (RCCL; --host-exchange: in-process exchange on host buffers)\n";
        return EXIT_SUCCESS;
    }
    if (!a.has("input")) die("ERROR: No input file provided");
    if (!a.has("clusters")) die("ERROR: No clusters file provided");
    std::cerr << "Reading fasta file... ";
    std::vector<std::string> labels = split_string(a.str("label", ""), ',');
    const std::vector<std::string> files = split_string(a.str("input", ""), ',');
    device_team team;
    // arena hint: ~440 kB of FASTQ per pack of 200 one-kb reads, at most a device full of resident packs in each of two
    // column classes (1792 places x 18.4 MB + 2048 x 34.5 MB = 104 GB at 1e6 one-kb reads, round 4; a hint that falls short
    // costs a second allocation -- 2.7 s in the first end-to-end run of round 4, when the hint was 89 GB -- not the result).  The
    // driver clears what it hands out at ~20 ms per GB, so a generous hint is seconds of start-up.
    const uint64_t packs_hint = std::min<uint64_t>(input_bytes(files) / 440000 + 1, 3840);
    team_opener opener(team, a, packs_hint * (30ull << 20));
    read_table T;
    { cli_timer t("read input"); read_table_inputs(T, files, labels, a.has("write-unzipped")); }
    std::cerr << "Done" << std::endl;
    std::unique_ptr<cli_timer> t_cl(new cli_timer("read clusters + gather"));
    cluster_set_t clusters = read_clusters(a.str("clusters", ""));
    if (clusters.empty()) die("\nError: empty clusters file\n");
    const bool gene_mode = clusters[0].main_seq.gene_id == -1;                   // correct.cpp:322
    const uint32_t n_reads = (uint32_t)T.n();

    byte_buf cat, qcat;
    std::vector<uint64_t> off;
    {
        std::vector<uint32_t> all(n_reads);
        for (uint32_t i = 0; i < n_reads; ++i) all[i] = i;
        gather_reads(T, all, cat, &qcat, off);
    }
    std::vector<uint32_t> coff(1, 0);
    std::vector<int32_t> mid;
    std::vector<uint8_t> mrev;
    for (auto &c : clusters) {
        for (auto &s : c.seqs) { mid.push_back(s.seq_id); mrev.push_back(s.rev ? 1 : 0); }
        coff.push_back((uint32_t)mid.size());
    }
    rattle_correct_params P;
    memset(&P, 0, sizeof(P));
    P.min_occ = a.d("min-occ", 0.3); P.gap_occ = a.d("gap-occ", 0.3); P.err_ratio = 30.0;
    P.split = a.i("split", 200); P.min_reads = a.i("min-reads", 5); P.n_threads = 0;
    std::string vo = a.str("vote-order", "");
    if (vo.size() == 6) memcpy(P.vote_order, vo.data(), 6);
    if (a.has("max-pack-cells")) P.max_pack_cells = std::stoull(a.str("max-pack-cells", "0"));
    t_cl.reset();
    { cli_timer t("wait for device + arena"); opener.wait(); }
    auto tag = [&](int cid) {                                                    // correct.cpp:348-353
        int gid = clusters[cid].main_seq.gene_id;
        if (gid == -1) return ",gene_cluster_" + std::to_string(cid);
        return ",gene_cluster_" + std::to_string(gid) + ",transcript_cluster_" + std::to_string(cid);
    };
    // corrected.fq / uncorrected.fq (fasta.cpp:436-445 record layout) straight from the library's buffers: the text of each
    // file is laid out by record offsets and filled in parallel
    auto write_set = [&](const rattle_read_set &S, bool corrected, const std::string &path) -> bool {
        std::vector<std::string> tags(clusters.size());
        std::vector<uint64_t> at((size_t)S.n + 1, 0);
        for (uint32_t i = 0; i < S.n; ++i) {
            const uint32_t rid = (uint32_t)S.read_id[i];
            std::string &tg = tags[S.cluster_id[i]];
            if (tg.empty()) tg = tag(S.cluster_id[i]);
            const uint64_t len = S.off[i + 1] - S.off[i];
            at[i + 1] = at[i] + T.header[rid].n + (T.label[rid] ? T.labels[T.label[rid] - 1].size() : 0) + tg.size() + 1 + len + 1 +
                        (corrected ? 1 : T.ann[rid].n) + 1 + len + 1;
        }
        sized_output text(path, at[S.n]);
        parallel_chunks(S.n, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const uint32_t rid = (uint32_t)S.read_id[i];
                char *o = text.p + at[i];
                auto put = [&o](const char *p, size_t n) { memcpy(o, p, n); o += n; };
                put(T.header[rid].p, T.header[rid].n);
                if (T.label[rid]) put(T.labels[T.label[rid] - 1].data(), T.labels[T.label[rid] - 1].size());
                put(tags[S.cluster_id[i]].data(), tags[S.cluster_id[i]].size()); *o++ = '\n';
                put(S.seq + S.off[i], S.off[i + 1] - S.off[i]); *o++ = '\n';
                if (corrected) *o++ = '+'; else put(T.ann[rid].p, T.ann[rid].n);
                *o++ = '\n';
                put(S.qual + S.off[i], S.off[i + 1] - S.off[i]); *o++ = '\n';
            }
        });
        return text.finish();
    };
    const std::string outdir = a.str("output", ".");
    // corrected.fq is formatted and written from the library's corrected_ready callback, while POA #2 / #3 still run on the
    // device (one device: the sharded job reassembles its reads at the end).  The callback runs on a library thread with kernels
    // in flight, so it never exits the process: it writes corrected.fq.tmp and records the outcome; the main thread renames the
    // file once correct_reads has returned 0 (the reference writes its three files only after correct_reads, main.cpp:405-411: a
    // failed run must not leave a complete-looking corrected.fq) or removes it and reports the failure.
    struct early_out { std::function<bool(const rattle_read_set &)> write; std::atomic<bool> done{false}, failed{false}; } early;
    const std::string corrected_path = outdir + "/corrected.fq", corrected_tmp = corrected_path + ".tmp";
    early.write = [&](const rattle_read_set &S) { cli_timer t("corrected.fq (behind the consensus stages)"); return write_set(S, true, corrected_tmp); };
    rattle_correction *R = nullptr;
    std::unique_ptr<cli_timer> t_lib(new cli_timer("library: correct_reads"));
    if (team.n() == 1) {
        P.corrected_ready = [](void *u, const rattle_read_set_s *S, const uint32_t *) {
            early_out *E = (early_out *)u;
            if (E->write(*S)) E->done = true; else E->failed = true;
        };
        P.corrected_ready_user = &early;
    }
    team.run([&](int r, rattle_ctx *ctx) {                                      // packs sharded over the ranks, result reassembled on rank 0
        rattle_correction *mine = nullptr, *merged = nullptr;
        const int rc = rattle_hip_correct_reads(ctx, cat.data(), qcat.data(), off.data(), n_reads, (uint32_t)clusters.size(), coff.data(), mid.data(),
                                                mrev.data(), &P, &mine);
        if (rc != 0 && team.n() == 1) unlink(corrected_tmp.c_str());           // no half of a result stays behind
        chk(rc);
        if (team.n() == 1) { R = mine; return; }
        chk(rattle_hip_correction_gather(ctx, mine, 0, &merged));
        rattle_hip_correction_free(mine);
        if (r == 0) R = merged;
    });
    if (early.failed) { unlink(corrected_tmp.c_str()); die("Error: cannot write " + corrected_path); }
    if (early.done && rename(corrected_tmp.c_str(), corrected_path.c_str()) != 0) die("Error: cannot write " + corrected_path);
    t_lib.reset();
    std::unique_ptr<cli_timer> t_out(new cli_timer("format + write outputs"));
    read_set_t consensi;
    // consensus headers, correct.cpp:453-469,495-549: labels counted over the reads of the cluster's packs
    std::vector<std::vector<int>> label_counts(clusters.size(), std::vector<int>(labels.size(), 0));
    if (!labels.empty()) {
        // only packs that produced a pack consensus count (as `reads=` does): packs the library skipped (budget rule: listed by
        // their index among ALL packs of the cluster; POA #1 / #2 beyond the device: by their index among the QUEUED packs) do not
        std::set<std::pair<int, uint32_t>> skip_all, skip_queued;
        for (uint32_t i = 0; i < R->skipped.n; ++i) {
            if (R->skipped.stage[i] == 0) skip_all.insert({R->skipped.cluster_id[i], R->skipped.pack[i]});
            else if (R->skipped.stage[i] <= 2) skip_queued.insert({R->skipped.cluster_id[i], R->skipped.pack[i]});
        }
        for (size_t c = 0; c < clusters.size(); ++c) {
            int n = (int)clusters[c].seqs.size();
            if (n == 0) continue;
            int n_files = (n - 1) / P.split + 1;
            uint32_t queued = 0;
            for (int nf = 0; nf < n_files; ++nf) {
                int sz = (n - 1 - nf) / n_files + 1;
                if (sz <= P.min_reads) continue;
                if (skip_all.count({(int)c, (uint32_t)nf})) continue;
                if (skip_queued.count({(int)c, queued++})) continue;
                for (int j = nf; j < n; j += n_files) {
                    const std::string h = T.head(clusters[c].seqs[j].seq_id);
                    size_t p = h.find_first_of(",");
                    std::string rest = p == std::string::npos ? "" : h.substr(p + 1);
                    std::string lab = rest.substr(0, rest.find_first_of(","));
                    for (size_t l = 0; l < labels.size(); ++l) if (labels[l] == lab) label_counts[c][l]++;
                }
            }
        }
    }
    for (uint32_t i = 0; i < R->consensi.n; ++i) {
        int cid = R->consensi.cluster_id[i];
        std::string lr;
        for (size_t l = 0; l < labels.size(); ++l) lr += labels[l] + ":" + std::to_string(label_counts[cid][l]) + ",";
        read_t r;
        if (gene_mode) r.header = "@gene_cluster_" + std::to_string(cid) + " reads=" + std::to_string(R->consensi.n_reads[i]) + " labels=" + lr;
        else r.header = "@transcript_cluster_" + std::to_string(cid) + " gene_cluster_" + std::to_string(clusters[cid].main_seq.gene_id) +
                        " reads=" + std::to_string(R->consensi.n_reads[i]) + " labels=" + lr;
        r.seq.assign(R->consensi.seq + R->consensi.off[i], R->consensi.seq + R->consensi.off[i + 1]);
        r.quality.assign(R->consensi.qual + R->consensi.off[i], R->consensi.qual + R->consensi.off[i + 1]);
        r.ann = "+";
        consensi.push_back(r);
    }
    std::cerr << std::endl << "Generating consensi..." << std::endl;
    if (!early.done && !write_set(R->corrected, true, corrected_path)) die("Error: cannot write " + corrected_path);
    if (!write_set(R->uncorrected, false, outdir + "/uncorrected.fq")) die("Error: cannot write " + outdir + "/uncorrected.fq");
    write_fastq_file(consensi, outdir + "/consensi.fq");
    if (R->skipped.n) {
        // packs whose POA did not fit the device (or the --max-pack-cells budget): their reads are in uncorrected.fq
        std::ofstream f(outdir + "/skipped_packs.tsv");
        f << "cluster\tpack\tstage\treads\n";
        for (uint32_t i = 0; i < R->skipped.n; ++i)
            f << R->skipped.cluster_id[i] << "\t" << R->skipped.pack[i] << "\t" << R->skipped.stage[i] << "\t"
              << (R->skipped.read_off[i + 1] - R->skipped.read_off[i]) << "\n";
        std::cerr << R->skipped.n << " pack(s) with " << R->counters[4] << " reads were not corrected (DP beyond the device or the budget): skipped_packs.tsv" << std::endl;
    }
    t_out.reset();
    std::cerr << "Done" << std::endl;
    rattle_hip_correction_free(R);
    team.close();
    return EXIT_SUCCESS;
}

std::string reverse_complement(const std::string &seq) {          // utils.cpp:15-24
    std::string r(seq.size(), 'A');
    for (size_t i = 0; i < seq.size(); ++i) {
        char c = seq[seq.size() - 1 - i];
        r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'T' ? 'A' : c == 'G' ? 'C' : c == 'U' ? 'A' : c;
    }
    return r;
}

// `rattle cluster_summary`, /root/reference/main.cpp:413-483 (host only, no device)
int mode_cluster_summary(int argc, char **argv) {
    std::vector<opt_def> defs = {{"help", {"-h", "--help"}, false}, {"input", {"-i", "--input"}, true}, {"label", {"-l", "--label"}, true},
                                 {"clusters", {"-c", "--clusters"}, true}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) { std::cerr << "rattle cluster_summary -i reads.fq -c clusters.out\n"; return EXIT_SUCCESS; }
    if (!a.has("input")) die("ERROR: No input file provided");
    if (!a.has("clusters")) die("ERROR: No clusters file provided");
    std::cerr << "Reading fasta file... ";
    read_set_t reads = read_inputs(split_string(a.str("input", ""), ','), split_string(a.str("label", ""), ','));
    std::cerr << "Done" << std::endl;
    cluster_set_t clusters = read_clusters(a.str("clusters", ""));
    int cid = 0;
    for (auto &c : clusters) {
        for (auto &s : c.seqs) {
            if ((size_t)s.seq_id >= reads.size()) die("\nError: cluster member id out of range\n");
            if (c.main_seq.gene_id == -1) std::cout << reads[s.seq_id].header << ",gene_cluster_" << cid << "\n";
            else std::cout << reads[s.seq_id].header << ",gene_cluster_" << s.gene_id << ",transcript_cluster_" << cid << "\n";
        }
        ++cid;
    }
    return EXIT_SUCCESS;
}

// `rattle extract_clusters`, /root/reference/main.cpp:484-611 (host only, no device)
int mode_extract_clusters(int argc, char **argv) {
    std::vector<opt_def> defs = {{"help", {"-h", "--help"}, false}, {"input", {"-i", "--input"}, true}, {"label", {"-l", "--label"}, true},
                                 {"clusters", {"-c", "--clusters"}, true}, {"output", {"-o", "--output-folder"}, true},
                                 {"minreads", {"-m", "--min-reads"}, true}, {"fastq", {"--fastq"}, false}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) { std::cerr << "rattle extract_clusters -i reads.fq -c clusters.out [-o dir] [-m N] [--fastq]\n"; return EXIT_SUCCESS; }
    if (!a.has("input")) die("ERROR: No input file provided");
    if (!a.has("clusters")) die("ERROR: No clusters file provided");
    if (a.has("output") && access(a.str("output", ".").c_str(), F_OK)) die("\nOutput folder doesn't exit. Please create it first. \n");
    std::cerr << "Reading fasta file... ";
    read_set_t reads = read_inputs(split_string(a.str("input", ""), ','), split_string(a.str("label", ""), ','));
    std::cerr << "Done" << std::endl;
    cluster_set_t clusters = read_clusters(a.str("clusters", ""));
    const int min_reads = a.i("minreads", 0);
    const bool fastq = a.has("fastq");
    int cid = 0;
    for (auto &c : clusters) {
        if ((int)c.seqs.size() > min_reads) {
            std::string fn = (a.has("output") ? a.str("output", ".") + "/" : std::string()) + "cluster_" + std::to_string(cid) + (fastq ? ".fq" : ".fa");
            std::ofstream f(fn);
            for (auto &s : c.seqs) {
                if ((size_t)s.seq_id >= reads.size()) die("\nError: cluster member id out of range\n");
                const read_t &r = reads[s.seq_id];
                if (c.main_seq.gene_id == -1) f << r.header << "\n";
                else f << r.header << "," << s.gene_id << "\n";
                f << (s.rev ? reverse_complement(r.seq) : r.seq) << "\n";          // quality is NOT reversed (main.cpp:585-588)
                if (fastq) f << r.ann << "\n" << r.quality << "\n";
            }
        }
        ++cid;
    }
    return EXIT_SUCCESS;
}

// `rattle polish`, /root/reference/main.cpp:612-762: cluster the consensi (k=6, 0.5, 25, bv 0.4/0.4 so
// no merge pass), correct with min_reads=0, rewrite the consensus headers.
int mode_polish(int argc, char **argv) {
    std::vector<opt_def> defs = {
        {"help", {"-h", "--help"}, false}, {"input", {"-i", "--input"}, true}, {"output", {"-o", "--output-folder"}, true},
        {"label", {"-l", "--label"}, true}, {"threads", {"-t", "--threads"}, true}, {"rna", {"--rna"}, false},
        {"verbose", {"--verbose"}, false}, {"summary", {"--summary"}, false}, {"device", {"--device"}, true}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) { std::cerr << "rattle polish -i consensi.fq [-o dir] [--rna] [--summary]\n"; return EXIT_SUCCESS; }
    if (!a.has("input")) die("ERROR: No input file provided");
    std::cerr << "Reading fasta file... ";
    std::string in = a.str("input", "");
    if (access(in.c_str(), F_OK)) die("\nError: Input file not found! \n");
    read_set_t reads;
    for_each_record(in, true, [&](const std::string &h, const std::string &s, const std::string &an, const std::string &q) {
        reads.push_back(read_t{h, s, an, q});
    });
    std::stable_sort(reads.begin(), reads.end(), [](const read_t &x, const read_t &y) { return x.seq.size() > y.seq.size(); });
    std::cerr << "Done" << std::endl;
    const bool is_rna = a.has("rna"), summary = a.has("summary");
    std::vector<std::string> labels = split_string(a.str("label", ""), ',');
    std::cerr << "Clustering consensus sequences..." << std::endl;
    rattle_ctx *ctx = nullptr;
    chk(rattle_hip_ctx_create(a.i("device", 0), &ctx));
    rattle_cluster_params P;
    P.t_s = 0.5; P.t_v = 25; P.bv_threshold = 0.4; P.min_bv_threshold = 0.4; P.bv_falloff = 0.05;
    P.min_reads_cluster = 0; P.use_hc = 0; P.repr_percentile = 0.15; P.is_rna = is_rna ? 1 : 0;
    load(ctx, reads, 6, !is_rna);
    rattle_cluster_set *raw = nullptr;
    chk(rattle_hip_cluster_reads(ctx, &P, &raw));
    cluster_set_t clusters = to_set(raw);

    std::string cat, qcat;
    std::vector<uint64_t> off(1, 0);
    for (auto &r : reads) {
        cat += r.seq;
        std::string q = r.quality;
        q.resize(r.seq.size(), '!');
        qcat += q;
        off.push_back(cat.size());
    }
    std::vector<uint32_t> coff(1, 0);
    std::vector<int32_t> mid;
    std::vector<uint8_t> mrev;
    for (auto &c : clusters) {
        for (auto &s : c.seqs) { mid.push_back(s.seq_id); mrev.push_back(s.rev ? 1 : 0); }
        coff.push_back((uint32_t)mid.size());
    }
    rattle_correct_params CP;
    memset(&CP, 0, sizeof(CP));
    CP.min_occ = 0.3; CP.gap_occ = 0.3; CP.err_ratio = 30.0; CP.split = 200; CP.min_reads = 0; CP.n_threads = 0;
    rattle_correction *R = nullptr;
    chk(rattle_hip_correct_reads(ctx, (const uint8_t *)cat.data(), (const uint8_t *)qcat.data(), off.data(), (uint32_t)reads.size(),
                                 (uint32_t)clusters.size(), coff.data(), mid.data(), mrev.data(), &CP, &R));
    read_set_t out;
    std::vector<std::string> summary_results;
    std::map<int, int> gene_map;
    int gid = -1;
    for (uint32_t i = 0; i < R->consensi.n; ++i) {
        const int cid = R->consensi.cluster_id[i];           // == i: every cluster has a pack with min_reads = 0
        int total_reads = 0;
        std::vector<int> label_counts(labels.size(), 0);
        for (auto &m : clusters[cid].seqs) {
            const std::string &h = reads[m.seq_id].header;
            auto info = split_string(h, '=');
            total_reads += std::stoi(info.at(1));
            for (size_t l = 0; l < labels.size(); ++l) {
                size_t idx = h.find(labels[l]);
                if (idx != std::string::npos) {
                    std::string sub = h.substr(idx + 1);
                    label_counts[l] += std::stoi(sub.substr(sub.find_first_of(":") + 1));
                }
            }
            auto info_c = split_string(h, '_');
            if (h.find("transcript_cluster") != std::string::npos) {
                int id = std::stoi(info_c.at(4));
                if (gene_map.find(id) == gene_map.end()) { if (gid == -1) gid = id; gene_map.insert({id, gid}); }
                else gid = gene_map.find(id)->second;
                if (summary) summary_results.push_back("transcript_cluster_" + std::to_string(std::stoi(info_c.at(2))) + ", gene_cluster_" +
                                                       std::to_string(id) + ", new_cluster_" + std::to_string(cid));
            } else if (summary) {
                summary_results.push_back("gene_cluster_" + std::to_string(std::stoi(info_c.at(2))) + ", new_cluster_" + std::to_string(cid));
            }
        }
        const int rcount = R->consensi.n_reads[i];           // the "reads=" field correct_reads writes (correct.cpp:542)
        read_t r;
        if (gid != -1)
            r.header = "@transcript_cluster_" + std::to_string(cid) + " gene_cluster_" + std::to_string(gid) + " generated_from_transcript_clusters=" +
                       std::to_string(rcount) + " total_reads=" + std::to_string(total_reads) + " labels=";
        else
            r.header = "@cluster_" + std::to_string(cid) + " generated_from_consensi_clusters=" + std::to_string(rcount) + " total_reads=" +
                       std::to_string(total_reads) + " labels=";
        for (size_t l = 0; l < labels.size(); ++l) r.header += labels[l] + ":" + std::to_string(label_counts[l]) + ",";
        r.seq.assign(R->consensi.seq + R->consensi.off[i], R->consensi.seq + R->consensi.off[i + 1]);
        r.quality.assign(R->consensi.qual + R->consensi.off[i], R->consensi.qual + R->consensi.off[i + 1]);
        r.ann = "+";
        out.push_back(r);
        gid = -1;
    }
    const std::string outdir = a.str("output", ".");
    if (summary) {
        std::ofstream f(outdir + "/polish_summary.tsv");
        for (auto &l : summary_results) f << l << "\n";
    }
    write_fastq_file(out, outdir + "/transcriptome.fq");
    rattle_hip_correction_free(R);
    rattle_hip_ctx_destroy(ctx);
    std::cerr << "Done" << std::endl;
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
    setenv("GPU_MAX_HW_QUEUES", "12", 0);      // before the HIP runtime starts: the POA column classes of a pass run concurrently, one hardware queue each (poa.hip)
    if (argc < 2) {
        std::cout << "Run with mode: ./rattle <cluster|cluster_summary|extract_clusters|correct|polish>" << std::endl;
        return EXIT_FAILURE;
    }
    try {
        if (!strcmp(argv[1], "cluster")) return mode_cluster(argc, argv);
        if (!strcmp(argv[1], "correct")) return mode_correct(argc, argv);
        if (!strcmp(argv[1], "polish")) return mode_polish(argc, argv);
        if (!strcmp(argv[1], "cluster_summary")) return mode_cluster_summary(argc, argv);
        if (!strcmp(argv[1], "extract_clusters")) return mode_extract_clusters(argc, argv);
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    std::cout << "Run with mode: ./rattle <cluster|cluster_summary|extract_clusters|correct|polish>" << std::endl;
    return EXIT_FAILURE;
}
