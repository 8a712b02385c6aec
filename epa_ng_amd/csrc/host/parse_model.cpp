// Model descriptor from a model FILE (the reference's --model accepts a file name as well as a
// descriptor, src/main.cpp:433-436): RAxML 8 info files (-f e), RAxML-NG .bestModel files and
// IQ-TREE reports are scraped into the "GTR{..}+FU{..}+IU{..}+G4{..}" form that Model parses.
// Behaviour follows src/util/parse_model.hpp:12-268; the expected strings of the reference's
// own tests (test/src/parse_model.cpp:7-63) are asserted in tests/test_host_cpu.py.
#include <fstream>
#include <sstream>

#include "epa_host.hpp"

namespace epa {
namespace {

// Sequential "key ... end of line" scraper: every lookup continues where the previous one ended
// (the files list e.g. "rate A <-> C: " once per partition; order matters).
class Scraper {
public:
  explicit Scraper(std::string text) : text_(std::move(text)) {}
  bool ahead(const std::string& key) const { return text_.find(key, pos_) != std::string::npos; }
  std::string after(const std::string& key) {
    const size_t at = text_.find(key, pos_);
    if (at == std::string::npos)
      throw std::invalid_argument{"Couldn't parse model file! (can't find '" + key + "'!)"};
    const size_t begin = at + key.size();
    const size_t end = text_.find('\n', begin);
    if (end == std::string::npos) throw std::runtime_error{"couldnt find terminating newline?!"};
    pos_ = end;
    return text_.substr(begin, end - begin);
  }
  std::string between(const std::string& key, const std::string& stop) {
    const std::string tail = after(key);
    const size_t at = tail.find(stop);
    if (at == std::string::npos)
      throw std::invalid_argument{"Couldn't parse model file! (can't find '" + stop + "'!)"};
    if (at == 0) throw std::runtime_error{"Nothing inbetween '" + stop + "' and '" + key + "'?"};
    return tail.substr(0, at);
  }
private:
  std::string text_;
  size_t pos_ = 0;
};

const char* alphabet(bool dna) { return dna ? "ACGT" : "ARNDCQEGHILKMFPSTWYV"; }

// "{r01/r02/.../r(n-2)(n-1)}" in upper-triangle order, each value behind the key made by `key`
template <class KeyFn>
std::string upper_triangle(Scraper& in, bool dna, KeyFn key) {
  const std::string chars = alphabet(dna);
  std::string out = "{";
  bool first = true;
  for (size_t i = 0; i + 1 < chars.size(); ++i)
    for (size_t k = i + 1; k < chars.size(); ++k) {
      if (!first) out += "/";
      first = false;
      out += in.after(key(chars[i], chars[k]));
    }
  return out + "}";
}

template <class KeyFn>
std::string frequencies(Scraper& in, bool dna, KeyFn key) {
  const std::string chars = alphabet(dna);
  std::string out = "+FU{";
  for (size_t i = 0; i < chars.size(); ++i) {
    if (i) out += "/";
    out += in.after(key(chars[i]));
  }
  return out + "}";
}

std::string from_raxml8(const std::string& text) {
  Scraper in(text);
  const bool dna = in.after("DataType: ") == "DNA";
  std::string matrix = in.after("Substitution Matrix: ");
  if (!dna && matrix == "GTR") matrix = "PROTGTR";
  // alpha and invar precede the rates in the file, but follow them in the descriptor
  const std::string alpha = in.ahead("alpha: ") ? "+G4{" + in.after("alpha: ") + "}" : "";
  const std::string pinv = in.ahead("invar: ") ? "+IU{" + in.after("invar: ") + "}" : "";
  std::string desc = matrix;
  desc += upper_triangle(in, dna, [](char a, char b) { return std::string("rate ") + a + " <-> " + b + ": "; });
  desc += frequencies(in, dna, [](char a) { return std::string("freq pi(") + a + "): "; });
  return desc + pinv + alpha;
}

std::string from_raxml_ng(const std::string& text) {
  // "<descriptor>, <partition name> = <range>": the descriptor is everything before the first comma
  const std::string line = text.substr(0, text.find('\n'));
  const size_t comma = line.find(',');
  if (comma == std::string::npos) throw std::runtime_error{"Model string in provided file seems wrong."};
  return line.substr(0, comma);
}

std::string from_iqtree(const std::string& text) {
  Scraper in(text);
  const std::string full = in.after("Model of substitution: ");
  const std::string matrix = full.substr(0, full.find('+'));
  const bool dna = matrix == "GTR";
  std::string desc = matrix;
  desc += upper_triangle(in, dna, [](char a, char b) { return std::string() + a + "-" + b + ": "; });
  desc += frequencies(in, dna, [](char a) { return std::string("pi(") + a + ") = "; });
  const bool gamma = in.ahead("Gamma with ");
  const std::string cats = gamma ? in.between("Gamma with ", " categories") : "";
  if (in.ahead("Proportion of invariable sites: "))
    desc += "+IU{" + in.after("Proportion of invariable sites: ") + "}";
  if (gamma) desc += "+G" + cats + "{" + in.after("Gamma shape alpha: ") + "}";
  return desc;
}

}  // namespace

std::string parse_model(const std::string& file) {
  std::ifstream f(file);
  if (!f) throw std::runtime_error{"file_check failed: " + file};
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  // src/util/parse_model.hpp:219-232: IQ-TREE reports start with "IQ-TREE ", RAxML 8 info files
  // name their version somewhere, anything else is taken for a RAxML-NG .bestModel line
  if (text.compare(0, 8, "IQ-TREE ") == 0) return from_iqtree(text);
  if (text.find("This is RAxML version 8.") != std::string::npos) return from_raxml8(text);
  return from_raxml_ng(text);
}

}  // namespace epa
