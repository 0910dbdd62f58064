// `rattle` command line: drop-in for `rattle cluster` (/root/reference/main.cpp:133-324),
// `rattle correct` (:325-412) and `rattle polish` (:612-762) over librattle_hip.so's C ABI.
// Same flags and defaults, same `clusters.out` (hps stream) and FASTQ outputs.  Host C++ only:
// every compute step goes through include/rattle_hip.h.
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rattle_hip.h"

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

// main.cpp:16-64 + fasta.cpp:272-370: quality dropped, ann = running record index over ALL
// records, length filter unless raw, reads containing 'N' skipped.
read_set_t read_inputs_cluster(const std::vector<std::string> &files, const std::vector<std::string> &labels, bool raw, int lo, int hi) {
    if (!labels.empty() && labels.size() != files.size()) die("\nError: Number of input files and number of label files do not match\n");
    read_set_t reads;
    int index = 0, sample = 0;
    for (std::string fn : files) {
        bool fastq;
        resolve_input(fn, fastq);
        std::string lab = labels.empty() ? "" : "," + labels[sample];
        int n_skipped = 0;
        for_each_record(fn, fastq, [&](const std::string &h, const std::string &s, const std::string &, const std::string &) {
            int my = index++;
            bool len_ok = raw || ((int)s.length() >= lo && (int)s.length() <= hi);
            if (!len_ok) return;
            if (s.find('N') != std::string::npos) { ++n_skipped; return; }
            reads.push_back(read_t{h + lab, s, std::to_string(my), ""});
        });
        if (n_skipped) std::cerr << "\n" << n_skipped << "  reads contains N are skipped!" << std::endl;
        ++sample;
    }
    return reads;
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
        {"devices", {"--devices"}, true}, {"host-exchange", {"--host-exchange"}, false}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) { std::cerr << "rattle cluster -i reads.fq [-o dir] [--rna] [--iso] ... (flags of RATTLE's cluster mode)\n"; return EXIT_SUCCESS; }
    if (!a.has("input")) die("ERROR: No input file provided");
    int k = a.i("kmer_size", 10), iso_k = a.i("iso_kmer_size", 11);
    if (k > 16 || iso_k > 16) die("\nError: maximum kmer size = 16 \n");
    std::string outdir = a.str("output", ".");
    if (a.has("output") && access(outdir.c_str(), F_OK)) die("\nOutput folder doesn't exit. Please create it first. \n");
    bool is_rna = a.has("rna");
    std::cerr << "RNA mode: " << std::boolalpha << is_rna << std::endl;
    std::cerr << "Reading fasta file... " << std::endl;
    read_set_t reads = read_inputs_cluster(split_string(a.str("input", ""), ','), split_string(a.str("label", ""), ','), a.has("raw"),
                                           a.i("lower_len", 150), a.i("upper_len", 100000));
    std::cout << "Reads: " << reads.size() << std::endl;
    std::stable_sort(reads.begin(), reads.end(), [](const read_t &x, const read_t &y) { return x.seq.size() > y.seq.size(); });
    std::cerr << "Done" << std::endl;

    device_team team;
    team.open(a);
    rattle_cluster_params P;
    P.t_s = a.d("t_s", 0.2); P.t_v = a.d("t_v", 1000000); P.bv_threshold = a.d("bv_threshold", 0.4);
    P.min_bv_threshold = a.d("bv_min_threshold", 0.2); P.bv_falloff = a.d("bv_falloff", 0.05);
    P.min_reads_cluster = a.i("min_reads_cluster", 0); P.use_hc = 0; P.repr_percentile = a.d("repr_percentile", 0.15);
    P.is_rna = is_rna ? 1 : 0;
    rattle_cluster_set *raw = nullptr;
    team.run([&](int r, rattle_ctx *ctx) {                                      // every rank ends up with the same clusters
        load(ctx, reads, k, !is_rna);
        rattle_cluster_set *mine = nullptr;
        chk(rattle_hip_cluster_reads(ctx, &P, &mine));
        if (r == 0) raw = mine; else rattle_hip_cluster_set_free(mine);
    });
    cluster_set_t gene = to_set(raw);
    std::cerr << "Gene clustering done" << std::endl;
    std::cerr << gene.size() << " gene clusters found" << std::endl;
    const std::string out_path = outdir + "/clusters.out";
    if (!a.has("iso")) {                                                         // main.cpp:264-277
        for (auto &c : gene) {
            c.main_seq.seq_id = std::stoi(reads[c.main_seq.seq_id].ann);
            for (auto &s : c.seqs) s.seq_id = std::stoi(reads[s.seq_id].ann);
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
        std::stable_sort(c.seqs.begin(), c.seqs.end(), [&reads](const cseq_t &x, const cseq_t &y) {
            return reads[x.seq_id].seq.size() > reads[y.seq_id].seq.size();
        });
        for (auto &s : c.seqs) ids.push_back((uint32_t)s.seq_id);
        sub_off.push_back(ids.size());
    }
    std::vector<rattle_cluster_set *> subs(gene.size() ? gene.size() : 1, nullptr);
    team.run([&](int r, rattle_ctx *ctx) {
        load(ctx, reads, iso_k, !is_rna);
        std::vector<rattle_cluster_set *> mine(gene.size() ? gene.size() : 1, nullptr);
        chk(rattle_hip_cluster_subsets(ctx, &P, ids.data(), sub_off.data(), (uint32_t)gene.size(), mine.data(), 0));
        if (r == 0) subs = mine; else for (auto *x : mine) rattle_hip_cluster_set_free(x);
    });
    int gi = 0;
    for (auto &c : gene) {
        for (auto &ic : to_set(subs[gi])) {
            cluster_t o;
            o.main_seq = cseq_t{std::stoi(reads[c.seqs[ic.main_seq.seq_id].seq_id].ann), ic.main_seq.rev, gi};
            for (auto &s : ic.seqs) o.seqs.push_back(cseq_t{std::stoi(reads[c.seqs[s.seq_id].seq_id].ann), s.rev, gi});
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
        {"host-exchange", {"--host-exchange"}, false}};
    args_t a = parse(argc, argv, defs);
    if (a.has("help")) {
        std::cerr << "rattle correct -i reads.fq -c clusters.out [-o dir] ... (flags of RATTLE's correct mode)\n"
                     "  --max-pack-cells N   leave packs whose largest alignment needs more than N DP cells uncorrected (default: only\n"
                     "                       packs that do not fit the device are skipped); skipped packs are listed in skipped_packs.tsv\n";
        return EXIT_SUCCESS;
    }
    if (!a.has("input")) die("ERROR: No input file provided");
    if (!a.has("clusters")) die("ERROR: No clusters file provided");
    std::cerr << "Reading fasta file... ";
    std::vector<std::string> labels = split_string(a.str("label", ""), ',');
    read_set_t reads = read_inputs(split_string(a.str("input", ""), ','), labels);
    std::cerr << "Done" << std::endl;
    cluster_set_t clusters = read_clusters(a.str("clusters", ""));
    if (clusters.empty()) die("\nError: empty clusters file\n");
    const bool gene_mode = clusters[0].main_seq.gene_id == -1;                   // correct.cpp:322

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
    rattle_correct_params P;
    memset(&P, 0, sizeof(P));
    P.min_occ = a.d("min-occ", 0.3); P.gap_occ = a.d("gap-occ", 0.3); P.err_ratio = 30.0;
    P.split = a.i("split", 200); P.min_reads = a.i("min-reads", 5); P.n_threads = 0;
    std::string vo = a.str("vote-order", "");
    if (vo.size() == 6) memcpy(P.vote_order, vo.data(), 6);
    if (a.has("max-pack-cells")) P.max_pack_cells = std::stoull(a.str("max-pack-cells", "0"));
    device_team team;
    team.open(a);
    rattle_correction *R = nullptr;
    team.run([&](int r, rattle_ctx *ctx) {                                      // packs sharded over the ranks, result reassembled on rank 0
        rattle_correction *mine = nullptr, *merged = nullptr;
        chk(rattle_hip_correct_reads(ctx, (const uint8_t *)cat.data(), (const uint8_t *)qcat.data(), off.data(), (uint32_t)reads.size(),
                                     (uint32_t)clusters.size(), coff.data(), mid.data(), mrev.data(), &P, &mine));
        if (team.n() == 1) { R = mine; return; }
        chk(rattle_hip_correction_gather(ctx, mine, 0, &merged));
        rattle_hip_correction_free(mine);
        if (r == 0) R = merged;
    });
    auto tag = [&](int cid) {                                                    // correct.cpp:348-353
        int gid = clusters[cid].main_seq.gene_id;
        if (gid == -1) return ",gene_cluster_" + std::to_string(cid);
        return ",gene_cluster_" + std::to_string(gid) + ",transcript_cluster_" + std::to_string(cid);
    };
    // corrected.fq / uncorrected.fq (fasta.cpp:436-445 record layout) straight from the library's buffers
    auto write_set = [&](const rattle_read_set &S, bool corrected, const std::string &path) {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) die("Error: cannot write " + path);
        std::string buf;
        buf.reserve(64u << 20);
        for (uint32_t i = 0; i < S.n; ++i) {
            buf += reads[S.read_id[i]].header; buf += tag(S.cluster_id[i]); buf += '\n';
            buf.append(S.seq + S.off[i], S.seq + S.off[i + 1]); buf += '\n';
            buf += corrected ? std::string("+") : reads[S.read_id[i]].ann; buf += '\n';
            buf.append(S.qual + S.off[i], S.qual + S.off[i + 1]); buf += '\n';
            if (buf.size() > (60u << 20)) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
        }
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
    };
    read_set_t consensi;
    // consensus headers, correct.cpp:453-469,495-549: labels counted over the reads of the cluster's packs
    std::vector<std::vector<int>> label_counts(clusters.size(), std::vector<int>(labels.size(), 0));
    if (!labels.empty()) {
        std::vector<char> in_pack(reads.size(), 0);
        (void)in_pack;
        for (size_t c = 0; c < clusters.size(); ++c) {
            int n = (int)clusters[c].seqs.size();
            int n_files = (n - 1) / P.split + 1;
            for (int nf = 0; nf < n_files; ++nf) {
                int sz = (n - 1 - nf) / n_files + 1;
                if (sz <= P.min_reads) continue;
                for (int j = nf; j < n; j += n_files) {
                    const std::string &h = reads[clusters[c].seqs[j].seq_id].header;
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
    std::string outdir = a.str("output", ".");
    write_set(R->corrected, true, outdir + "/corrected.fq");
    write_set(R->uncorrected, false, outdir + "/uncorrected.fq");
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
    rattle_hip_correction_free(R);
    team.close();
    std::cerr << "Done" << std::endl;
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
