// TEST INFRASTRUCTURE: a driver over the REFERENCE's own classes for the placement hot path, compiled by
// `make -C oracle ref` from the reference's sources where they lie (EPA_REF, default /root/reference) against
// an installed libpll-2 / pll-modules / genesis.  It is what pins oracle/epa_oracle.c against the real thing:
// tests/test_ref_pin.py diffs the two on the reference's bundled data whenever oracle/_ref/epa_ref_driver
// exists.  In THIS image it cannot be built (libs/pll-modules and libs/genesis are empty submodules, no pll.h
// anywhere): the recipe is committed for whoever has the libraries; nothing here stands in for them.
//
//   epa_ref_driver <tree.newick> <ref_msa.fasta> <query.fasta> <model string> [--raxml-blo]
//
// For every branch b (utree_query_branches order = jplace edge_num, src/core/pll/pll_util.cpp:182-205) and
// every query q it prints
//   P b q lnL                     Tiny_Tree(opt_branches = false).place(): the Lookup_Store sum
//                                 (src/tree/Tiny_Tree.cpp:131-218, src/core/Lookup_Store.hpp:110-141)
//   T b q lnL pendant distal      Tiny_Tree(opt_branches = true).place(): optimize_branch_triplet
//                                 (src/core/pll/optimize.cpp:253-286)
// with %.17g, premasking off (the oracle sees the same W), site repeats off, default numerical scaling.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "core/Lookup_Store.hpp"
#include "core/pll/pll_util.hpp"
#include "core/pll/pllhead.hpp"
#include "core/raxml/Model.hpp"
#include "io/file_io.hpp"
#include "seq/MSA.hpp"
#include "seq/MSA_Info.hpp"
#include "tree/Tiny_Tree.hpp"
#include "tree/Tree.hpp"
#include "util/Options.hpp"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s tree ref_msa query_msa model [--raxml-blo]\n", argv[0]);
    return 2;
  }
  Options options;
  options.premasking = false;
  options.repeats = false;
  options.opt_branches = false;
  options.opt_model = false;
  for (int i = 5; i < argc; ++i)
    if (!std::strcmp(argv[i], "--raxml-blo")) options.sliding_blo = false;
  try {
    raxml::Model model{std::string(argv[4])};
    auto ref_msa = build_MSA_from_file(argv[2], MSA_Info(argv[2]), options.premasking);
    auto queries = build_MSA_from_file(argv[3], MSA_Info(argv[3]), options.premasking);
    Tree ref_tree(argv[1], ref_msa, model, options);
    const auto num_branches = ref_tree.nums().branches;
    std::vector<pll_unode_t*> branches(num_branches);
    if (utree_query_branches(ref_tree.tree(), &branches[0]) != num_branches) {
      std::fprintf(stderr, "utree_query_branches: unexpected branch count\n");
      return 1;
    }
    std::printf("L %.17g\n", ref_tree.ref_tree_logl());
    for (int thorough = 0; thorough < 2; ++thorough) {
      auto lookups = std::make_shared<Lookup_Store>(num_branches, ref_tree.partition()->states);
      for (unsigned int b = 0; b < num_branches; ++b) {
        Tiny_Tree tt(branches[b], b, ref_tree, thorough != 0, options, lookups);
        size_t q = 0;
        for (auto const& s : queries) {
          const auto p = tt.place(s);
          if (thorough)
            std::printf("T %u %zu %.17g %.17g %.17g\n", b, q, p.likelihood(), p.pendant_length(), p.distal_length());
          else
            std::printf("P %u %zu %.17g\n", b, q, p.likelihood());
          ++q;
        }
      }
    }
  } catch (std::exception const& e) {
    std::fprintf(stderr, "epa_ref_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
