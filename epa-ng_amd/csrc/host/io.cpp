// FASTA reading and jplace writing (formats of src/io/jplace_util.cpp:20-98 and
// src/io/jplace_writer.hpp:79-148; fixed-point doubles with `precision` digits).
#include <cctype>
#include <fstream>
#include <ostream>

#include <cctype>
#include <cstdio>
#include <cstring>

#include "epa_host.hpp"

namespace epa {

// ---- FASTA: block reads + memchr line splitting; sites are upper-cased by the reader
// (src/seq/MSA_Stream.cpp:41), whitespace inside sequence lines is dropped.
Fasta_Stream::Fasta_Stream(const std::string& path) : f_(std::fopen(path.c_str(), "rb")) {
  if (!f_) throw std::runtime_error{"file_check failed: " + path};
  buf_.resize(1 << 22);
  for (int c = 0; c < 256; ++c) up_[c] = std::isspace(c) ? 0 : (char)std::toupper(c);
}

Fasta_Stream::~Fasta_Stream() { if (f_) std::fclose(f_); }

bool Fasta_Stream::next_line(const char*& b, const char*& e) {
  for (;;) {
    const char* nl = (const char*)std::memchr(buf_.data() + pos_, '\n', len_ - pos_);
    if (nl) {
      b = buf_.data() + pos_;
      e = nl;
      pos_ = (size_t)(nl - buf_.data()) + 1;
      return true;
    }
    if (eof_) {
      if (pos_ == len_) return false;
      b = buf_.data() + pos_;
      e = buf_.data() + len_;
      pos_ = len_;
      return true;
    }
    // refill: keep the partial line at the front, grow if a single line exceeds the buffer
    std::memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
    len_ -= pos_;
    pos_ = 0;
    if (len_ == buf_.size()) buf_.resize(buf_.size() * 2);
    const size_t got = std::fread(buf_.data() + len_, 1, buf_.size() - len_, f_);
    len_ += got;
    if (got == 0) eof_ = true;
  }
}

size_t Fasta_Stream::read_next(MSA& out, size_t max_seqs) {
  size_t n = 0;
  const char *b, *e;
  while (n < max_seqs || !pending_header_) {
    if (!next_line(b, e)) {
      if (pending_header_) { out.emplace_back(std::move(header_), std::move(seq_)); ++n; }
      pending_header_ = false;
      header_.clear(); seq_.clear();
      break;
    }
    while (e > b && (e[-1] == '\r' || e[-1] == ' ')) --e;
    if (e == b) continue;
    if (*b == '>') {
      if (pending_header_) { out.emplace_back(std::move(header_), std::move(seq_)); ++n; seq_.clear(); }
      const char* h = b + 1;
      const char* he = h;
      while (he < e && *he != ' ' && *he != '\t') ++he;
      header_.assign(h, he);
      pending_header_ = true;
      if (n >= max_seqs) break;   // the header just read belongs to the next call
    } else if (pending_header_) {
      const size_t old = seq_.size();
      seq_.resize(old + (size_t)(e - b));
      char* d = &seq_[old];
      size_t k = 0;
      for (const char* p = b; p < e; ++p) { const char c = up_[(unsigned char)*p]; d[k] = c; k += c != 0; }
      seq_.resize(old + k);
    }
  }
  return n;
}

MSA read_fasta(const std::string& path) {
  Fasta_Stream in(path);
  MSA out;
  while (in.read_next(out, (size_t)1 << 20)) {}
  return out;
}

void write_jplace(std::ostream& os, const std::vector<Sample>& chunks, const std::string& newick,
                  const std::string& invocation, unsigned int precision) {
  os.precision(precision);
  os.setf(std::ios::fixed, std::ios::floatfield);
  os << "{\n  \"tree\": \"" << newick << "\",\n  \"placements\": \n  [\n";
  bool first_chunk = true;
  for (const auto& sample : chunks) {
    if (sample.empty()) continue;
    if (!first_chunk) os << ",\n";  // chunks separated by ",\n" (jplace_writer.hpp:141)
    first_chunk = false;
    size_t i = 0;
    for (const auto& pq : sample) {
      os << "    {\"p\": [\n";
      size_t j = 0;
      for (const auto& p : pq) {
        os << "      [" << p.branch_id() << ", " << p.likelihood() << ", " << p.lwr() << ", "
           << p.distal_length() << ", " << p.pendant_length() << "]";
        if (++j < pq.size()) os << ",";
        os << "\n";
      }
      os << "      ],\n    \"n\": [\"" << pq.header() << "\"]\n    }";
      if (++i < sample.size()) os << ",";
      os << "\n";
    }
  }
  os << "  ],\n  \"metadata\": {\"invocation\": \"" << invocation << "\"},\n  \"version\": 3,\n"
     << "  \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\""
     << ", \"pendant_length\"]\n}\n";
}

}  // namespace epa
