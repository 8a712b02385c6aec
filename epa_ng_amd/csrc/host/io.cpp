// FASTA reading and jplace writing (formats of src/io/jplace_util.cpp:20-98 and
// src/io/jplace_writer.hpp:79-148; fixed-point doubles with `precision` digits).
#include <algorithm>
#include <cctype>
#include <charconv>
#include <system_error>
#include <fstream>
#include <ostream>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "epa_host.hpp"

namespace epa {

// ---- FASTA: block reads + memchr line splitting; sites are upper-cased by the reader
// (src/seq/MSA_Stream.cpp:41), whitespace inside sequence lines is dropped.
// ---- binary fasta (".bfast", the reference's --bfast conversion, src/io/Binary_Fasta.hpp):
//   "BFAST\0\0" (7 bytes) | u64 n | [u64 len + that many '0'/'1' chars: the all-gap column mask;
//   files written before the mask existed have none] | n x (u64 id, u64 offset) |
//   per sequence at its offset: u64 len, header bytes, u64 sites, ceil(sites / 2) bytes of 4-bit
//   codes (FourBit: index in NT_MAP, earlier site in the high nibble).  Integers little-endian.
namespace {
constexpr char kBfastMagic[7] = {'B', 'F', 'A', 'S', 'T', 0, 0};
constexpr char kNtMap[] = "-TGKCYSBAWRDMHVN";  // src/util/maps.hpp:9-26

uint64_t get_u64(std::FILE* f) {
  unsigned char b[8];
  if (std::fread(b, 1, 8, f) != 8) throw std::runtime_error{"bfast: unexpected end of file"};
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | b[i];
  return v;
}
}  // namespace

bool Fasta_Stream::open_bfast() {
  char magic[7];
  if (std::fread(magic, 1, 7, f_) != 7 || std::memcmp(magic, kBfastMagic, 7) != 0) {
    std::rewind(f_);
    return false;
  }
  const uint64_t n = get_u64(f_);
  // with or without the mask string?  The first table entry names the data section's offset.
  const long table_plain = 7 + 8;
  std::vector<uint64_t> offs(n);
  auto read_table = [&](long at) {
    if (std::fseek(f_, at, SEEK_SET) != 0) return false;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t id = get_u64(f_), off = get_u64(f_);
      if (id >= n) return false;
      offs[id] = off;
    }
    return true;
  };
  bool ok = false;
  if (n > 0) {
    std::fseek(f_, table_plain, SEEK_SET);
    const uint64_t mask_len = get_u64(f_);
    const uint64_t data_masked = 7 + 8 + 8 + mask_len + n * 16;
    if (mask_len < (1ull << 40) && read_table((long)(table_plain + 8 + mask_len)) && offs[0] == data_masked) ok = true;
    if (!ok && read_table(table_plain) && offs[0] == 7 + 8 + n * 16) ok = true;
  } else {
    ok = true;
  }
  if (!ok) throw std::runtime_error{"File is not an epa::Binary_Fasta file"};
  bfast_ = true;
  bfast_offsets_ = std::move(offs);
  bfast_next_ = 0;
  struct stat sb;   // mapped as well: read_next_wire() takes the records' nibbles straight from the page cache
  if (::fstat(::fileno(f_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
    void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, ::fileno(f_), 0);
    if (m != MAP_FAILED) {
      map_ = static_cast<const char*>(m);
      map_len_ = (size_t)sb.st_size;
    }
  }
  return true;
}

size_t Fasta_Stream::read_next_wire(MSA& out, Encoded_Chunk& enc, size_t sites, size_t max_seqs, bool premasking) {
  configure_host_threads();
  if (!bfast_ || !map_ || max_seqs == 0 || sites == 0) return 0;
  const size_t m = std::min(max_seqs, bfast_offsets_.size() - bfast_next_);
  if (m == 0) return 0;
  const size_t first = out.size();
  out.resize(first + m);
  enc.win_begin.assign(m, 0);
  enc.win_span.assign(m, 0);
  std::vector<const unsigned char*> rows(m, nullptr);
  const size_t nbytes = (sites + 1) / 2;
  auto u64_at = [&](size_t off) { uint64_t v; std::memcpy(&v, map_ + off, 8); return v; };   // little-endian hosts (x86-64)
  std::vector<uint8_t> stat(m, 0);   // 1 truncated record, 2 another width than the reference, 3 all gap
  uint32_t mx = 1;
#pragma omp parallel for schedule(static) reduction(max : mx)
  for (long i = 0; i < (long)m; ++i) {
    const size_t off = bfast_offsets_[bfast_next_ + (size_t)i];
    if (off > map_len_ || map_len_ - off < 16) { stat[i] = 1; continue; }   // (no sums of untrusted 64-bit values)
    const uint64_t hlen = u64_at(off);
    if (hlen > (1u << 20) || map_len_ - off < 16 + hlen) { stat[i] = 1; continue; }
    out[first + i] = Sequence(std::string(map_ + off + 8, (size_t)hlen), std::string());
    if (u64_at(off + 8 + hlen) != sites) { stat[i] = 2; continue; }
    if (map_len_ - off - 16 - hlen < nbytes) { stat[i] = 1; continue; }
    const unsigned char* p = reinterpret_cast<const unsigned char*>(map_ + off + 16 + hlen);
    rows[i] = p;
    size_t lo = 0, hi = sites;
    if (premasking) {   // nibble 0 = '-' (NT_MAP[0]); the padding nibble of an odd row is 0 as well
      size_t k = 0;
      while (k + 8 <= nbytes) { uint64_t w; std::memcpy(&w, p + k, 8); if (w) break; k += 8; }
      while (k < nbytes && p[k] == 0) ++k;
      lo = std::min(sites, 2 * k + ((k < nbytes && (p[k] >> 4) == 0) ? 1 : 0));
      size_t e = nbytes;
      while (e >= k + 8 && e >= 8) { uint64_t w; std::memcpy(&w, p + e - 8, 8); if (w) break; e -= 8; }
      while (e > k && p[e - 1] == 0) --e;
      hi = e == 0 ? 0 : std::min(sites, 2 * e - ((p[e - 1] & 15) == 0 ? 1 : 0));
      if (hi <= lo) { stat[i] = 3; continue; }
    }
    enc.win_begin[i] = (uint32_t)lo;
    enc.win_span[i] = (uint32_t)(hi - lo);
    mx = std::max(mx, (uint32_t)(hi - lo));
  }
  for (size_t i = 0; i < m; ++i) {   // the first offender in file order, with the reference's messages
    if (stat[i] == 1) throw std::runtime_error{"bfast: truncated sequence"};
    if (stat[i] == 2) throw std::runtime_error{"Query sequence length not same as reference alignment!"};   // Tiny_Tree.cpp:145-147
    if (stat[i] == 3)   // Tiny_Tree.cpp:153-156
      throw std::runtime_error{"Sequence with header '" + out[first + i].header() + "' does not appear to have any non-gap sites!"};
  }
  enc.stride = (mx + 15) / 16 * 16;
  enc.bits = 4;
  const size_t ps = (enc.stride + 1) / 2;
  enc.codes.assign(m * ps, 0);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)m; ++i) {
    const unsigned char* p = rows[i];
    uint8_t* o = enc.codes.data() + (size_t)i * ps;
    const size_t lo = enc.win_begin[i], n = enc.win_span[i], nb = (n + 1) / 2;
    if (!(lo & 1)) {
      std::memcpy(o, p + lo / 2, nb);
    } else {   // the window starts in a low nibble: shift the nibble stream by one
      const size_t k0 = lo / 2;
      for (size_t b = 0; b < nb; ++b) {
        const unsigned nxt = (k0 + b + 1 < nbytes) ? p[k0 + b + 1] : 0u;
        o[b] = (uint8_t)(((p[k0 + b] & 15u) << 4) | (nxt >> 4));
      }
    }
    if (n & 1) o[nb - 1] &= 0xf0;   // past the window: '-'
  }
  bfast_next_ += m;
  return m;
}

size_t Fasta_Stream::read_next_bfast(MSA& out, size_t max_seqs) {
  size_t got = 0;
  std::vector<unsigned char> packed;
  while (got < max_seqs && bfast_next_ < bfast_offsets_.size()) {
    if (std::fseek(f_, (long)bfast_offsets_[bfast_next_], SEEK_SET) != 0)
      throw std::runtime_error{"bfast: bad sequence offset"};
    const uint64_t hlen = get_u64(f_);
    std::string header(hlen, '\0');
    if (hlen && std::fread(&header[0], 1, hlen, f_) != hlen) throw std::runtime_error{"bfast: truncated header"};
    const uint64_t sites = get_u64(f_);
    packed.resize((sites + 1) / 2);
    if (!packed.empty() && std::fread(packed.data(), 1, packed.size(), f_) != packed.size())
      throw std::runtime_error{"bfast: truncated sequence"};
    std::string seq(sites, '-');
    for (uint64_t i = 0; i < sites; ++i) {
      const unsigned char b = packed[i / 2];
      seq[i] = kNtMap[(i & 1) ? (b & 15) : (b >> 4)];
    }
    out.emplace_back(std::move(header), std::move(seq));
    ++bfast_next_;
    ++got;
  }
  return got;
}

Fasta_Stream::Fasta_Stream(const std::string& path) : f_(std::fopen(path.c_str(), "rb")) {
  if (!f_) throw std::runtime_error{"file_check failed: " + path};
  for (int c = 0; c < 256; ++c) up_[c] = std::isspace(c) ? 0 : (char)std::toupper(c);
  if (open_bfast()) return;  // like the reference: try bfast first, fall back to fasta (msa_reader.hpp:15-24)
  // a regular file is mapped: the parser reads the page cache directly (a 1.5 GB query file spent
  // 0.3 s of its 0.55 s in fread's copy); pipes and the like go through the read buffer
  struct stat sb;
  if (::fstat(::fileno(f_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
    void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, ::fileno(f_), 0);
    if (m != MAP_FAILED) {
      ::madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
      map_ = static_cast<const char*>(m);
      map_len_ = len_ = (size_t)sb.st_size;
      eof_ = true;   // nothing more to read; `scan_` tells how far the record index has got
      return;
    }
  }
  buf_.resize(1 << 24);
}

Fasta_Stream::~Fasta_Stream() {
  if (map_) ::munmap(const_cast<char*>(map_), map_len_);
  if (f_) std::fclose(f_);
}

// appends more of the file behind the unparsed region [pos_, len_); false at end of file
bool Fasta_Stream::refill() {
  if (eof_) return false;
  if (pos_ > 0) {
    std::memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
    len_ -= pos_;
    scan_ -= pos_;
    for (auto& s : starts_) s -= pos_;
    pos_ = 0;
  }
  if (buf_.size() - len_ < (buf_.size() >> 2)) buf_.resize(buf_.size() * 2);
  const size_t got = std::fread(buf_.data() + len_, 1, buf_.size() - len_, f_);
  len_ += got;
  if (got == 0) eof_ = true;
  return got != 0;
}

// Extends the index of record starts as far as a chunk of max_seqs needs and returns how many
// complete records are available (a record is complete once the next one has started, or the file
// has ended); `done` = nothing is left behind them.
size_t Fasta_Stream::index_records(size_t max_seqs, bool& done, bool predict) {
  for (;;) {
    const char* base = map_ ? map_ : buf_.data();
    while (scan_ < len_ && starts_.size() <= max_seqs) {
      if (predict && predict_ok_ && map_ && seq_part_ && !starts_.empty() && starts_.back() + 1 == scan_) {
        // the record that starts at starts_.back(): end of its header line + the learned length of the rest
        const size_t b = starts_.back();
        const char* nl = (const char*)std::memchr(base + b, '\n', std::min<size_t>(len_ - b, 4096));
        if (nl) {
          const size_t cand = (size_t)(nl - base) + seq_part_;
          if (cand < len_ && base[cand] == '>' && base[cand - 1] == '\n') { starts_.push_back(cand); scan_ = cand + 1; continue; }
          if (cand == len_) { scan_ = len_; break; }   // the last record ends with the file
        }
      }
      const char* p = (const char*)std::memchr(base + scan_, '>', len_ - scan_);
      if (!p) { scan_ = len_; break; }
      const size_t o = (size_t)(p - base);
      if (o == 0 ? first_block_ : base[o - 1] == '\n') {
        if (map_ && !starts_.empty()) {   // learn: header end of the previous record -> this record
          const size_t b = starts_.back();
          const char* nl = (const char*)std::memchr(base + b, '\n', o - b);
          if (nl) seq_part_ = o - (size_t)(nl - base);
        }
        starts_.push_back(o);
      }
      scan_ = o + 1;
    }
    done = eof_ && scan_ >= len_;
    const size_t complete = done ? starts_.size() : (starts_.empty() ? 0 : starts_.size() - 1);
    if (complete >= max_seqs || done) {
      first_block_ = false;
      return std::min(max_seqs, complete);
    }
    if (starts_.empty() && len_ > 0) pos_ = len_ - 1;  // junk before the first record: keep 1 byte of context
    first_block_ = first_block_ && len_ == 0;
    refill();
  }
}

void Fasta_Stream::consume(size_t m) {
  pos_ = m < starts_.size() ? starts_[m] : len_;
  starts_.erase(starts_.begin(), starts_.begin() + m);
}

// Records are located first ('>' at the start of a line), then parsed in parallel: the header
// line is the label, the sequence lines are upper-cased with white space dropped.
size_t Fasta_Stream::read_next(MSA& out, size_t max_seqs) {
  configure_host_threads();
  if (max_seqs == 0) return 0;
  if (bfast_) return read_next_bfast(out, max_seqs);
  bool done = false;
  const size_t m = index_records(max_seqs, done);
  if (m == 0) { if (done) { pos_ = len_; starts_.clear(); } return 0; }
  const size_t first = out.size();
  out.resize(first + m);
  const char* base = map_ ? map_ : buf_.data();
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)m; ++i) {
    const char* b = base + starts_[i] + 1;
    const char* e = base + (i + 1 < (long)starts_.size() ? starts_[i + 1] : len_);
    const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
    const char* hend = nl ? nl : e;
    // the label is the whole header line, as the reference's reader (genesis FastaReader via
    // src/seq/MSA_Stream.cpp:39-41) keeps it; only the line end and trailing blanks are dropped
    const char* h = hend;
    while (h > b && (h[-1] == '\r' || h[-1] == ' ' || h[-1] == '\t')) --h;
    std::string header(b, h), seq;
    if (nl) {
      seq.resize((size_t)(e - nl));
      size_t k = 0;
      for (const char* p = nl; p < e; ++p) { const char c = up_[(unsigned char)*p]; seq[k] = c; k += c != 0; }
      seq.resize(k);
    }
    out[first + i] = Sequence(std::move(header), std::move(seq));
  }
  consume(m);
  return m;
}

// Zero-copy variant for mapped files whose next records are all one header line + ONE sequence line
// of exactly `sites` characters (what alignment tools write for short reads): `out` gets the
// headers (empty sequences), rows[i] points at record i's sequence inside the mapping (not
// NUL-terminated, upper / lower case as in the file; valid as long as this reader lives).  Returns 0
// and consumes nothing when the file is not mapped or a record has another shape: the caller then
// takes the chunk with read_next().
size_t Fasta_Stream::read_next_views(MSA& out, std::vector<const char*>& rows, size_t sites, size_t max_seqs) {
  configure_host_threads();
  if (!map_ || bfast_ || max_seqs == 0 || sites == 0) return 0;
  bool done = false;
  const size_t m = index_records(max_seqs, done, true);
  if (m == 0) return 0;
  const size_t first = out.size();
  out.resize(first + m);
  rows.assign(m, nullptr);
  int ok = 1;
#pragma omp parallel for schedule(static) reduction(&& : ok)
  for (long i = 0; i < (long)m; ++i) {
    const char* b = map_ + starts_[i] + 1;
    const char* e = map_ + (i + 1 < (long)starts_.size() ? starts_[i + 1] : len_);
    const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
    if (!nl) { ok = 0; continue; }
    const char* sb = nl + 1;
    const char* se = e;
    while (se > sb && (se[-1] == '\n' || se[-1] == '\r')) --se;
    // exactly `sites` bytes between the header line and the next record: a row with a line break or a blank INSIDE
    // those bytes is short of real characters -- the encoder, which looks at every one of them, refuses it
    // ("char is invalid!"), so the bytes are not walked twice
    const bool good = (size_t)(se - sb) == sites;
    if (!good) { ok = 0; continue; }
    const char* h = nl;
    while (h > b && (h[-1] == '\r' || h[-1] == ' ' || h[-1] == '\t')) --h;
    out[first + i] = Sequence(std::string(b, h), std::string());
    rows[i] = sb;
  }
  if (!ok) {
    out.resize(first);
    rows.clear();
    if (predict_ok_ && starts_.size() > 1) {
      // the looked-up record starts cannot be trusted for this file: search from the first record on, from now on
      predict_ok_ = false;
      scan_ = starts_[0] + 1;
      starts_.resize(1);
    }
    return 0;
  }
  consume(m);
  return m;
}

// ---- column premasking (src/seq/MSA_Info.hpp)
namespace {
inline bool is_mask_gap(char c) {
  switch (c) {
    case 'N': case 'n': case 'O': case 'o': case 'X': case 'x': case '.': case '-': case '?': return true;
    default: return false;
  }
}
}  // namespace

MSA_Info::MSA_Info(const MSA& msa) {
  for (const auto& s : msa) add(s.sequence());
}

void MSA_Info::add(const std::string& seq) {
  if (sequences_ == 0) {
    sites_ = seq.size();
    gap_mask_.assign(sites_, 1);
  } else if (seq.size() != sites_) {
    throw std::runtime_error{"MSA does not contain equal size sequences! (" + std::to_string(seq.size()) +
                             " vs. " + std::to_string(sites_) + " sites)"};
  }
  ++sequences_;
  for (size_t i = 0; i < sites_; ++i) gap_mask_[i] &= (uint8_t)is_mask_gap(seq[i]);
}

size_t MSA_Info::gap_count() const {
  size_t n = 0;
  for (uint8_t g : gap_mask_) n += g;
  return n;
}

MSA_Info MSA_Info::from_file(const std::string& path) {
  MSA_Info info;
  Fasta_Stream in(path);
  MSA part;
  for (;;) {
    part.clear();
    if (in.read_next(part, 4096) == 0) break;
    for (const auto& s : part) info.add(s.sequence());
    // the mask only ever loses bits: once it is empty the rest of the file cannot change it
    if (info.gap_count() == 0) break;
  }
  return info;
}

void MSA_Info::or_mask(MSA_Info& lhs, MSA_Info& rhs) {
  if (lhs.sites() != rhs.sites())
    throw std::runtime_error{"MSA_Infos are unequal site width: " + std::to_string(lhs.sites()) + " vs. " +
                             std::to_string(rhs.sites())};
  for (size_t i = 0; i < lhs.gap_mask_.size(); ++i) lhs.gap_mask_[i] = rhs.gap_mask_[i] = lhs.gap_mask_[i] | rhs.gap_mask_[i];
}

std::string subset_sequence(const std::string& seq, const MSA_Info::mask_type& mask) {
  if (seq.size() != mask.size()) throw std::runtime_error{"In subset_sequence: mask and seq incompatible"};
  std::string out;
  out.reserve(seq.size());
  for (size_t i = 0; i < seq.size(); ++i) if (!mask[i]) out.push_back(seq[i]);
  return out;
}

MSA subset_msa(const MSA& msa, const MSA_Info::mask_type& mask) {
  MSA out;
  out.reserve(msa.size());
  for (const auto& s : msa) out.emplace_back(s.header(), subset_sequence(s.sequence(), mask));
  return out;
}

MSA read_fasta(const std::string& path) {
  Fasta_Stream in(path);
  MSA out;
  while (in.read_next(out, (size_t)1 << 20)) {}
  return out;
}

// v in fixed notation with `precision` digits, correctly rounded like printf's %.*f (and the reference's
// std::fixed << std::setprecision).  std::to_chars does that exactly, at ~80 ns per number; the jplace text of a
// million reads holds six million of them.  Fast path: |v| x 10^p in 80-bit arithmetic (64-bit mantissa) is within
// x 2^-63 of the true product, so unless its fraction lies that close to one half the rounded integer is KNOWN to be
// the correctly rounded one and is printed as an integer with a decimal point; anything near a tie, huge, non-finite
// or with more than 18 digits goes to std::to_chars.  tests/test_host_cpu.py compares both against printf.
size_t format_fixed(char* buf, size_t cap, double v, unsigned int precision) {
  static const long double kPow10[19] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L, 1e11L, 1e12L,
                                         1e13L, 1e14L, 1e15L, 1e16L, 1e17L, 1e18L};
  static const unsigned long long kPow10u[19] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                                                 100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
                                                 10000000000000ull, 100000000000000ull, 1000000000000000ull,
                                                 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull};
  if (precision >= 1 && precision <= 18 && cap >= 48 && sizeof(long double) > sizeof(double)) {
    const long double x = fabsl((long double)v) * kPow10[precision];   // NaN fails the comparison below
    if (x < 9.0e18L) {
      unsigned long long n = (unsigned long long)x;          // floor
      const long double frac = x - (long double)n;           // exact
      const long double err = x * 2.2e-19L + 1e-30L;         // > x 2^-63: the product's rounding
      if (fabsl(frac - 0.5L) > err) {
        n += frac > 0.5L;
        const unsigned long long ip = n / kPow10u[precision], fp = n % kPow10u[precision];
        char* p = buf;
        if (std::signbit(v)) *p++ = '-';
        p = std::to_chars(p, buf + cap, ip).ptr;
        *p++ = '.';
        char* e = p + precision;
        unsigned long long r = fp;
        for (char* q = e; q > p;) { *--q = (char)('0' + r % 10); r /= 10; }
        return (size_t)(e - buf);
      }
    }
  }
  const auto r = std::to_chars(buf, buf + cap, v, std::chars_format::fixed, (int)precision);
  if (r.ec == std::errc()) return (size_t)(r.ptr - buf);
  const int len = std::snprintf(buf, cap, "%.*f", (int)precision, v);
  return (size_t)std::max(0, std::min(len, (int)cap - 1));
}

// One chunk of the "placements" array as text (sample_to_jplace_string, src/io/jplace_util.cpp:
// 60-98): formatted per pquery in parallel with snprintf into per-thread strings, then joined.
// Numbers are fixed-point with `precision` digits like the reference's stream settings.
std::string jplace_chunk_text(const Sample& sample, unsigned int precision, const Rtree_Mapper* mapper) {
  const bool remap = mapper && (bool)*mapper;  // placement_to_jplace_string, jplace_util.cpp:20-26
  const int nt_max = std::max(1, configure_host_threads());
  const long n = (long)sample.size();
  // one contiguous buffer per thread over a contiguous range of pqueries (no string per pquery: 50 000 small heap
  // blocks per chunk were a third of this stage), every placement line assembled in a stack buffer with pointer bumps
  std::vector<std::string> part((size_t)nt_max);
#pragma omp parallel num_threads(nt_max)
  {
    const int t = omp_get_thread_num(), nthr = omp_get_num_threads();
    const long i0 = n * t / nthr, i1 = n * (t + 1) / nthr;
    std::string& o = part[(size_t)t];
    size_t est = 0;
    for (long i = i0; i < i1; ++i) est += 48 + sample[i].header().size() + sample[i].size() * (24 + 4 * (size_t)(precision + 12));
    o.reserve(est);
    char line[1024];
    for (long i = i0; i < i1; ++i) {
      const auto& pq = sample[i];
      o += "    {\"p\": [\n";
      size_t j = 0;
      for (const auto& p : pq) {
        size_t edge = p.branch_id();
        double distal = p.distal_length();
        if (remap) {
          const auto m = mapper->in_rtree((unsigned int)edge, distal);
          edge = m.first;
          distal = m.second;
        }
        char* w = line;
        std::memcpy(w, "      [", 7); w += 7;
        w = std::to_chars(w, line + 64, edge).ptr;
        const double vals[4] = {p.likelihood(), p.lwr(), distal, p.pendant_length()};
        for (int k = 0; k < 4; ++k) {
          *w++ = ','; *w++ = ' ';
          w += format_fixed(w, (size_t)(line + sizeof(line) - w) - 8, vals[k], precision);
        }
        *w++ = ']';
        if (++j < pq.size()) *w++ = ',';
        *w++ = '\n';
        o.append(line, (size_t)(w - line));
      }
      o += "      ],\n    \"n\": [\"";
      o += pq.header();
      o += "\"]\n    }";
      if (i + 1 < n) o += ",";
      o += "\n";
    }
  }
  size_t total = 0;
  std::vector<size_t> at(part.size());
  for (size_t k = 0; k < part.size(); ++k) { at[k] = total; total += part[k].size(); }
  std::string out(total, '\0');
#pragma omp parallel for schedule(static) num_threads(nt_max)
  for (long k = 0; k < (long)part.size(); ++k) std::memcpy(&out[at[(size_t)k]], part[(size_t)k].data(), part[(size_t)k].size());
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
                  const std::string& invocation, unsigned int precision, const Rtree_Mapper* mapper) {
  std::vector<std::string> texts;
  texts.reserve(chunks.size());
  for (const auto& sample : chunks) texts.push_back(jplace_chunk_text(sample, precision, mapper));
  write_jplace_text(os, texts, newick, invocation);
}

}  // namespace epa
