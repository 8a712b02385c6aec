// FASTA reading and jplace writing (formats of src/io/jplace_util.cpp:20-98 and
// src/io/jplace_writer.hpp:79-148; fixed-point doubles with `precision` digits).
#include <algorithm>
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

// One chunk of the "placements" array as text (sample_to_jplace_string, src/io/jplace_util.cpp:
// 60-98): formatted per pquery in parallel with snprintf into per-thread strings, then joined.
// Numbers are fixed-point with `precision` digits like the reference's stream settings.
std::string jplace_chunk_text(const Sample& sample, unsigned int precision) {
  configure_host_threads();
  const long n = (long)sample.size();
  std::vector<std::string> part(n);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    const auto& pq = sample[i];
    std::string& o = part[i];
    o.reserve(64 + pq.size() * (40 + 4 * (precision + 8)));
    o += "    {\"p\": [\n";
    char buf[512];
    size_t j = 0;
    for (const auto& p : pq) {
      const int len = std::snprintf(buf, sizeof(buf), "      [%zu, %.*f, %.*f, %.*f, %.*f]", p.branch_id(),
                                    (int)precision, p.likelihood(), (int)precision, p.lwr(), (int)precision,
                                    p.distal_length(), (int)precision, p.pendant_length());
      o.append(buf, (size_t)std::max(0, std::min(len, (int)sizeof(buf) - 1)));
      if (++j < pq.size()) o += ",";
      o += "\n";
    }
    o += "      ],\n    \"n\": [\"";
    o += pq.header();
    o += "\"]\n    }";
    if (i + 1 < n) o += ",";
    o += "\n";
  }
  size_t total = 0;
  for (const auto& x : part) total += x.size();
  std::string out;
  out.reserve(total);
  for (const auto& x : part) out += x;
  return out;
}

void write_jplace_text(std::ostream& os, const std::vector<std::string>& chunk_texts, const std::string& newick,
                       const std::string& invocation) {
  os << "{\n  \"tree\": \"" << newick << "\",\n  \"placements\": \n  [\n";
  bool first_chunk = true;
  for (const auto& t : chunk_texts) {
    if (t.empty()) continue;
    if (!first_chunk) os << ",\n";  // chunks separated by ",\n" (jplace_writer.hpp:141)
    first_chunk = false;
    os.write(t.data(), (std::streamsize)t.size());
  }
  os << "  ],\n  \"metadata\": {\"invocation\": \"" << invocation << "\"},\n  \"version\": 3,\n"
     << "  \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\""
     << ", \"pendant_length\"]\n}\n";
}

void write_jplace(std::ostream& os, const std::vector<Sample>& chunks, const std::string& newick,
                  const std::string& invocation, unsigned int precision) {
  std::vector<std::string> texts;
  texts.reserve(chunks.size());
  for (const auto& sample : chunks) texts.push_back(jplace_chunk_text(sample, precision));
  write_jplace_text(os, texts, newick, invocation);
}

}  // namespace epa
