// FASTA reading and jplace writing (formats of src/io/jplace_util.cpp:20-98 and
// src/io/jplace_writer.hpp:79-148; fixed-point doubles with `precision` digits).
#include <cctype>
#include <fstream>
#include <ostream>

#include "epa_host.hpp"

namespace epa {

MSA read_fasta(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error{"file_check failed: " + path};
  MSA out;
  std::string line, header, seq;
  bool have = false;
  auto flush = [&]() {
    if (have) out.emplace_back(header, seq);
    seq.clear();
  };
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '>') {
      flush();
      have = true;
      const size_t e = line.find_first_of(" \t");
      header = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
    } else {
      // sites are upper-cased by the reader (src/seq/MSA_Stream.cpp:41)
      for (char c : line)
        if (!std::isspace((unsigned char)c)) seq.push_back((char)std::toupper((unsigned char)c));
    }
  }
  flush();
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
