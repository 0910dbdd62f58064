// ORACLE (test infrastructure, never shipped in the product path).
//
// File boundary of the reference restated for the oracle CLI: FASTQ readers
// (fasta.cpp:207-270 plain, :272-370 cluster variant), FASTQ writer
// (fasta.cpp:436-445) and the hps `clusters.out` stream (grammar recovered from
// the shipped fixtures; hps itself is an absent submodule).
#pragma once
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "orc_cluster.hpp"

namespace orc {

inline void chomp_cr(std::string &s, bool dos) { if (dos && !s.empty()) s.erase(s.size() - 1); }

// fasta.cpp:207-270: every 4-line record, file order, header += sample_id.
inline read_set_t read_fastq_plain(const std::string &file, const std::string &sample_id) {
    read_set_t out;
    std::ifstream in(file);
    std::string line, header, seq, ann;
    if (!std::getline(in, line)) return out;
    bool dos = !line.empty() && line[line.size() - 1] == '\r';
    chomp_cr(line, dos);
    header = line + sample_id;
    int id = 1;
    while (std::getline(in, line)) {
        chomp_cr(line, dos);
        if (id == 0) { header = line + sample_id; id = 1; }
        else if (id == 1) { seq = line; id = 2; }
        else if (id == 2) { ann = line; id = 3; }
        else { out.push_back(read_t{header, seq, ann, line}); id = 0; }
    }
    return out;
}

// fasta.cpp:272-370: ann = running record index (filtered records consume an index),
// quality dropped, length filter unless raw, reads containing 'N' skipped.
// `index` is updated to the next free index (the reference smuggles it out through
// result.back().quality, fasta.cpp:363 / main.cpp:47).
inline read_set_t read_fastq_cluster(const std::string &file, const std::string &sample_id, int &index, bool raw,
                                     int lower_len, int upper_len) {
    read_set_t out;
    std::ifstream in(file);
    std::string line, header, seq, ann;
    if (!std::getline(in, line)) return out;
    bool dos = !line.empty() && line[line.size() - 1] == '\r';
    chomp_cr(line, dos);
    header = line + sample_id;
    int id = 1;
    while (std::getline(in, line)) {
        chomp_cr(line, dos);
        if (id == 0) { header = line + sample_id; id = 1; }
        else if (id == 1) { seq = line; id = 2; }
        else if (id == 2) { ann = std::to_string(index++); id = 3; }
        else {
            id = 0;
            bool len_ok = raw || ((int)seq.length() >= lower_len && (int)seq.length() <= upper_len);
            if (len_ok && seq.find('N') == std::string::npos) out.push_back(read_t{header, seq, ann, ""});
        }
    }
    return out;
}


// fasta.cpp:33-96: FASTA, plain variant (`correct` / `polish` input): every record, sequence upper-cased (:131 / :57),
// ann "+", quality '~' per base; a DOS file loses the last byte of every line.  The first header is taken as is,
// whatever its first character.
inline read_set_t read_fasta_plain(const std::string &file, const std::string &sample_id) {
    read_set_t out;
    std::ifstream in(file);
    std::string line, header, seq;
    if (!std::getline(in, line)) return out;
    bool dos = !line.empty() && line[line.size() - 1] == '\r';
    chomp_cr(line, dos);
    header = line + sample_id;
    auto flush = [&]() {
        for (auto &c : seq) c = (char)::toupper((unsigned char)c);
        out.push_back(read_t{header, seq, "+", std::string(seq.size(), '~')});
    };
    while (std::getline(in, line)) {
        if (line.size() == 0) continue;
        if (line[0] == '>') {
            if (!header.empty()) flush();
            seq.clear();
            chomp_cr(line, dos);
            header = line + sample_id;
        } else {
            chomp_cr(line, dos);
            seq += line;
        }
    }
    flush();
    return out;
}

// fasta.cpp:98-205: FASTA, cluster variant: ann = running record index, upper-cased, length filter unless raw, reads with
// 'N' skipped.  The Unix branch advances the index at EVERY header line (:146, outside the `!header.empty()` test),
// the DOS branch only when the pending header is not empty (:176); the last record takes the index as it stands and
// the next free index is one more (:203).
inline read_set_t read_fasta_cluster(const std::string &file, const std::string &sample_id, int &index, bool raw, int lower_len, int upper_len) {
    read_set_t out;
    std::ifstream in(file);
    std::string line, header, seq;
    if (!std::getline(in, line)) return out;
    bool dos = !line.empty() && line[line.size() - 1] == '\r';
    chomp_cr(line, dos);
    header = line + sample_id;
    auto keep = [&]() {
        for (auto &c : seq) c = (char)::toupper((unsigned char)c);
        const bool len_ok = raw || ((int)seq.length() >= lower_len && (int)seq.length() <= upper_len);
        if (len_ok && seq.find('N') == std::string::npos) out.push_back(read_t{header, seq, std::to_string(index), ""});
    };
    while (std::getline(in, line)) {
        if (line.size() == 0) continue;
        if (line[0] == '>') {
            if (!header.empty()) { keep(); if (dos) ++index; }
            if (!dos) ++index;
            seq.clear();
            chomp_cr(line, dos);
            header = line + sample_id;
        } else {
            chomp_cr(line, dos);
            seq += line;
        }
    }
    if (!header.empty()) keep();
    ++index;
    return out;
}

inline std::vector<std::string> split_commas(const std::string &s) {          // utils.cpp splitString(str, ',')
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) out.push_back(tok);
    return out;
}

inline std::string file_extension(const std::string &f) { return f.substr(f.find_last_of(".") + 1); }

// main.cpp:16-64: every input file in turn, header += "," + label, one running record index over all files
inline read_set_t read_multiple_inputs_cluster(const std::vector<std::string> &files, const std::vector<std::string> &labels, bool raw, int lower_len, int upper_len) {
    if (!labels.empty() && labels.size() != files.size()) throw std::runtime_error("Number of input files and number of label files do not match");
    read_set_t reads;
    int index = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        const std::string lab = labels.empty() ? "" : "," + labels[i];
        const std::string ext = file_extension(files[i]);
        read_set_t part;
        if (ext == "fq" || ext == "fastq") part = read_fastq_cluster(files[i], lab, index, raw, lower_len, upper_len);
        else if (ext == "fasta" || ext == "fa") part = read_fasta_cluster(files[i], lab, index, raw, lower_len, upper_len);
        else throw std::runtime_error("Input file format incorrect! Please use fasta/fastq file.");
        reads.insert(reads.end(), part.begin(), part.end());
    }
    return reads;
}

// main.cpp:66-109
inline read_set_t read_multiple_inputs(const std::vector<std::string> &files, const std::vector<std::string> &labels) {
    if (!labels.empty() && labels.size() != files.size()) throw std::runtime_error("Number of input files and number of label files do not match");
    read_set_t reads;
    for (size_t i = 0; i < files.size(); ++i) {
        const std::string lab = labels.empty() ? "" : "," + labels[i];
        const std::string ext = file_extension(files[i]);
        read_set_t part;
        if (ext == "fq" || ext == "fastq") part = read_fastq_plain(files[i], lab);
        else if (ext == "fasta" || ext == "fa") part = read_fasta_plain(files[i], lab);
        else throw std::runtime_error("Input file format incorrect! Please use fasta/fastq file.");
        reads.insert(reads.end(), part.begin(), part.end());
    }
    return reads;
}

inline void write_fastq(const read_set_t &reads, const std::string &file) {   // fasta.cpp:436-445
    std::ofstream f(file);
    for (auto &r : reads) f << r.header << "\n" << r.seq << "\n" << r.ann << "\n" << r.quality << "\n";
}

inline void put_uvarint(std::string &o, uint64_t x) {
    while (x >= 0x80) { o.push_back((char)((x & 0x7F) | 0x80)); x >>= 7; }
    o.push_back((char)x);
}
inline void put_svarint(std::string &o, int32_t x) { put_uvarint(o, (uint32_t)((x << 1) ^ (x >> 31))); }

inline std::string hps_encode(const cluster_set_t &cs) {        // cluster.hpp:15-18,30-33
    std::string o;
    put_uvarint(o, cs.size());
    auto cseq = [&o](const cseq_t &c) { put_svarint(o, c.seq_id); o.push_back(c.rev ? 1 : 0); put_svarint(o, c.gene_id); };
    for (auto &c : cs) {
        cseq(c.main_seq);
        put_uvarint(o, c.seqs.size());
        for (auto &s : c.seqs) cseq(s);
    }
    return o;
}

inline bool hps_decode(const std::string &b, int fields, cluster_set_t &out) {
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
        cseq_t c;
        c.seq_id = sv();
        if (p >= b.size()) { ok = false; return c; }
        uint8_t r = (uint8_t)b[p++];
        if (r > 1) ok = false;
        c.rev = r;
        c.gene_id = fields == 3 ? sv() : -1;
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

inline cluster_set_t hps_read_file(const std::string &path) {
    std::ifstream in(path, std::ifstream::binary);
    std::string b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    cluster_set_t cs;
    if (hps_decode(b, 3, cs)) return cs;
    if (hps_decode(b, 2, cs)) return cs;
    throw std::runtime_error("not a clusters.out stream: " + path);
}

}  // namespace orc
