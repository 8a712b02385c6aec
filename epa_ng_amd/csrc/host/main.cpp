// epa-ng-amd: command-line front end keeping EPA-ng's flags for the placement path
// (src/main.cpp:96-270).  Flags outside the hot path (binary dump, bfast conversion, --split)
// are rejected with a message, not silently ignored.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "epa_host.hpp"

using namespace epa;

static void usage() {
  std::cout <<
      "epa-ng-amd - Evolutionary Placement Algorithm, MI355X placement evaluator\n"
      "  -t,--tree FILE        reference tree (newick, unrooted or rooted)\n"
      "  -s,--ref-msa FILE     reference MSA (fasta)\n"
      "  -q,--query FILE       query MSA (fasta, aligned to the reference)\n"
      "  -m,--model STR|FILE   model descriptor, e.g. GTR{..}+FU{..}+G4{a} (default GTR+G), or a RAxML 8\n"
      "                        info / RAxML-NG .bestModel / IQ-TREE report file\n"
      "  -w,--outdir DIR       output directory (default ./)\n"
      "  -g,--dyn-heur X       accumulated-LWR preplacement threshold (default 0.99999)\n"
      "  -G,--fix-heur X       fixed fraction of branches\n"
      "  --baseball-heur       baseball heuristic\n"
      "  --no-heur             thorough placement on every branch\n"
      "  --filter-acc-lwr X | --filter-min-lwr X | --filter-min N | --filter-max N\n"
      "  --precision N         output digits (default 10)\n"
      "  --chunk-size N        queries per chunk (default 50000; EPA-ng's CPU default is 5000)\n"
      "  --device-min-chunk N  a device chunk holds at least N queries whatever --chunk-size says\n"
      "                        (default 40000; 0: chunks of exactly --chunk-size)\n"
      "  --no-pre-mask         evaluate all sites of every query\n"
      "  --raxml-blo           radius-1 local branch-length optimisation instead of the sliding rule\n"
      "  --rate-scalers auto|on|off  per-rate-category numerical scaling (auto: on above 2000 tips)\n"
      "  --preserve-rooting on|off  rooted reference tree: report on the rooted tree (default on)\n"
      "  -T,--threads N        upper limit on the host threads (parsing, encoding, LWR / filter, jplace text)\n"
      "  --device N            GPU ordinal (default 0)\n"
      "  --devices a,b,..      place on several GPUs of the node (chunks are dealt to them in turn)\n"
      "  --rank R --world N --comm-file F   one process per GPU: rank R places its contiguous slice of the\n"
      "                        queries, results are gathered to rank 0 over RCCL; rank 0 leaves the RCCL id in F\n"
      "                        (defaults: RANK / WORLD_SIZE / LOCAL_RANK / EPA_COMM_FILE of the environment)\n"
      "  --stats-json FILE     the time breakdown of the run as JSON\n"
      "  --comm-nonce STR      marks this run's id record in F (default: EPA_COMM_NONCE / TORCHELASTIC_RUN_ID); give\n"
      "                        one whenever the launcher can: without it a record is only checked for its age\n"
      "  --comm-probe-seconds S  bound of the handshake every rank runs before placing (default 120)\n"
      "  --comm-timeout S      bound of every later wait for a peer (default EPA_COMM_TIMEOUT_S, else 600)\n"
      "  --comm-rows-per-read N  rows per rank and gather = chunk x N (default 8; more rows take the carry path)\n"
      "                        RCCL is bound at run time: the library named by EPA_RCCL_LIB, else librccl.so.1 on the\n"
      "                        loader's search path (LD_LIBRARY_PATH), else /opt/rocm/lib/librccl.so.1\n";
}

int main(int argc, char** argv) {
  const auto start = std::chrono::steady_clock::now();
  std::string invocation;
  for (int i = 0; i < argc; ++i) { invocation += argv[i]; invocation += " "; }
  std::string tree_file, ref_file, query_file, outdir = "./", model_desc = "GTR+G", stats_json;
  Options opt;
  int device = 0;
  bool device_given = false;
  std::vector<int> devices;
  // one process per GPU (place_ranks.cpp): --rank / --world / --comm-file, or what torchrun / mpirun export
  auto env_int = [](const char* a, const char* b, int dflt) {
    const char* v = std::getenv(a);
    if (!v && b) v = std::getenv(b);
    return v ? std::atoi(v) : dflt;
  };
  int rank = env_int("EPA_RANK", "RANK", 0), world = env_int("EPA_WORLD", "WORLD_SIZE", 1);
  const int local_rank = env_int("EPA_LOCAL_RANK", "LOCAL_RANK", rank);
  std::string comm_file = std::getenv("EPA_COMM_FILE") ? std::getenv("EPA_COMM_FILE") : "";
  if (const char* e = std::getenv("EPA_COMM_NONCE")) opt.comm_nonce = e;
  if (opt.comm_nonce.empty()) if (const char* e = std::getenv("TORCHELASTIC_RUN_ID")) opt.comm_nonce = e;
  auto need = [&](int& i) -> std::string {
    if (i + 1 >= argc) { std::cerr << "missing value for " << argv[i] << "\n"; std::exit(1); }
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-t" || a == "--tree") tree_file = need(i);
    else if (a == "-s" || a == "--ref-msa" || a == "--msa") ref_file = need(i);
    else if (a == "-q" || a == "--query") query_file = need(i);
    else if (a == "-m" || a == "--model") model_desc = need(i);
    else if (a == "-w" || a == "--outdir" || a == "--out-dir") outdir = need(i);
    else if (a == "-g" || a == "--dyn-heur") { opt.prescoring_threshold = std::stod(need(i)); opt.prescoring_by_percentage = false; }
    else if (a == "-G" || a == "--fix-heur") { opt.prescoring_threshold = std::stod(need(i)); opt.prescoring_by_percentage = true; }
    else if (a == "--baseball-heur") opt.baseball = true;
    else if (a == "--no-heur") opt.prescoring = false;
    else if (a == "--filter-acc-lwr") { opt.support_threshold = std::stod(need(i)); opt.acc_threshold = true; }
    else if (a == "--filter-min-lwr") { opt.support_threshold = std::stod(need(i)); opt.acc_threshold = false; }
    else if (a == "--filter-min") opt.filter_min = (unsigned)std::stoul(need(i));
    else if (a == "--filter-max") opt.filter_max = (unsigned)std::stoul(need(i));
    else if (a == "--precision") opt.precision = (unsigned)std::stoul(need(i));
    else if (a == "--chunk-size") { opt.chunk_size = (unsigned)std::stoul(need(i)); opt.chunk_size_given = true; }
    else if (a == "--device-min-chunk") { opt.device_min_chunk = (unsigned)std::stoul(need(i)); opt.device_min_chunk_given = true; }
    else if (a == "--no-pre-mask") opt.premasking = false;
    else if (a == "--raxml-blo") opt.sliding_blo = false;   // src/main.cpp:239-242
    else if (a == "--rate-scalers") {                       // src/main.cpp:248-250,399-407
      const std::string v = need(i);
      if (v == "auto") opt.scaling = Options::NumericalScaling::kAuto;
      else if (v == "on") opt.scaling = Options::NumericalScaling::kOn;
      else if (v == "off") opt.scaling = Options::NumericalScaling::kOff;
      else { std::cerr << "--rate-scalers: " << v << " not in {auto,on,off}\n"; return 1; }
    }
    else if (a == "--preserve-rooting") {  // src/main.cpp:196-199, 410-418
      const std::string v = need(i);
      if (v == "off") opt.preserve_rooting = false;
      else if (v == "on") opt.preserve_rooting = true;
      else { std::cerr << "--preserve-rooting: " << v << " not in {on,off}\n"; return 1; }
    }
    else if (a == "-T" || a == "--threads") {
      opt.num_threads = (unsigned)std::stoul(need(i));
      set_host_thread_limit((int)opt.num_threads);   // caps the OpenMP host stages; the placement itself runs on the GPU
    }
    else if (a == "--device") { device = std::stoi(need(i)); device_given = true; }
    else if (a == "--devices") {
      std::stringstream ls(need(i));
      std::string tok;
      while (std::getline(ls, tok, ',')) if (!tok.empty()) devices.push_back(std::stoi(tok));
    }
    else if (a == "--rank") rank = std::stoi(need(i));
    else if (a == "--world") world = std::stoi(need(i));
    else if (a == "--comm-file") comm_file = need(i);
    else if (a == "--comm-nonce") opt.comm_nonce = need(i);
    else if (a == "--comm-probe-seconds") opt.comm_probe_seconds = std::stod(need(i));
    else if (a == "--comm-timeout") opt.comm_timeout_seconds = std::stod(need(i));
    else if (a == "--comm-rows-per-read") opt.comm_rows_per_read = std::stoi(need(i));
    else if (a == "--comm-self-send") opt.comm_self_send = true;       // test hook
    else if (a == "--host-heuristic") opt.host_heuristic = true;       // diagnostics
    else if (a == "--no-pipeline") opt.no_pipeline = true;
    else if (a == "--stats-json") stats_json = need(i);
    else if (a == "--redo" || a == "--verbose") {}
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else { std::cerr << "option " << a << " is outside the placement hot path of this build\n"; return 1; }
  }
  if (tree_file.empty() || ref_file.empty() || query_file.empty()) { usage(); return 1; }
  try {
    std::ifstream tf(tree_file);
    if (!tf) throw std::runtime_error{"file_check failed: " + tree_file};
    std::stringstream ss;
    ss << tf.rdbuf();
    MSA ref = read_fasta(ref_file);
    // "Peeking into MSA files and generating masks" (src/main.cpp:468-494): columns that are all-gap in
    // the reference or in the whole query file leave both alignments (unless --no-pre-mask)
    MSA_Info ref_info(ref), qry_info = MSA_Info::from_file(query_file);
    if (ref_info.sites() != qry_info.sites()) {
      std::cerr << "The reference and query alignment files do not seem to have the same alignment width! ("
                << ref_info.sites() << " vs. " << qry_info.sites() << "). Are the query sequences not aligned?\n";
      return 1;
    }
    MSA_Info::or_mask(ref_info, qry_info);
    if (opt.premasking && ref_info.gap_count() > 0) {
      std::cout << "Premasking: " << ref_info.gap_count() << " of " << ref_info.sites()
                << " columns are gaps in the whole reference or the whole query alignment and are removed." << std::endl;
      ref = subset_msa(ref, ref_info.gap_mask());
    }
    {  // --model may name a RAxML 8 info / RAxML-NG .bestModel / IQ-TREE report file (src/main.cpp:433-436)
      std::ifstream mf(model_desc);
      if (mf.good()) model_desc = parse_model(model_desc);
    }
    const Model model(model_desc);
    std::cout << "Using model parameters: " << model.to_string() << std::endl;
    const auto t_tree = std::chrono::steady_clock::now();
    const double secs_parse = std::chrono::duration<double>(t_tree - start).count();   // reference MSA, query-file peek, masks, model
    const Tree tree(ss.str(), ref, model, opt);
    if (tree.rooted_input())
      std::cout << "Rooted reference tree: placements are reported on the "
                << (tree.mapper() ? "rooted tree (--preserve-rooting on)." : "unrooted tree (--preserve-rooting off).")
                << std::endl;
    const double secs_tree = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tree).count();
    if (devices.empty()) devices.push_back(device);
    const bool rank_mode = world > 1 || !comm_file.empty();   // a 1-rank communicator is legal (tests, 1-GPU nodes)
    const Run_Stats st = rank_mode
        ? simple_mpi_ranks(tree, query_file, qry_info, outdir, opt, invocation, device_given ? device : local_rank, rank, world, comm_file)
        : simple_mpi(tree, query_file, qry_info, outdir, opt, invocation, devices);
    if (rank_mode && rank != 0) {
      std::cout << "rank " << rank << " of " << world << ": " << st.queries << " sequences placed" << std::endl;
      return 0;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::cout << "Reference tree log-likelihood: " << st.ref_tree_logl << "\n";
    std::cout << st.queries << " Sequences done! (" << st.pairs << " thorough pairs)\n"
              << "Time breakdown [s]: tree + MSA " << secs_tree << ", device setup " << st.seconds_setup
              << ", read queries " << st.seconds_read << ", wait for encoded chunk " << st.seconds_stage_wait
              << ", device " << st.seconds_place + st.seconds_thorough << ", lwr+filter " << st.seconds_post
              << ", write jplace " << st.seconds_write << "\n"
              << "Elapsed Time: " << secs << "s" << std::endl;
    if (!stats_json.empty()) {   // the same numbers, machine-readable (bench.py's cli_e2e leg)
      std::ofstream sj(stats_json);
      sj << "{\"queries\": " << st.queries << ", \"pairs\": " << st.pairs << ", \"host_threads\": " << st.host_threads
         << ", \"elapsed_s\": " << secs << ", \"tree_msa_s\": " << secs_tree << ", \"parse_inputs_s\": " << secs_parse
         << ", \"device_setup_s\": " << st.seconds_setup
         << ", \"loop_s\": " << st.seconds_loop << ", \"read_index_s\": " << st.seconds_read << ", \"encode_s\": " << st.seconds_encode
         << ", \"stage_wait_s\": " << st.seconds_stage_wait << ", \"device_calls_s\": " << st.seconds_place + st.seconds_thorough
         << ", \"build_sample_s\": " << st.seconds_sample << ", \"lwr_filter_s\": " << st.seconds_post
         << ", \"jplace_text_s\": " << st.seconds_text << ", \"write_s\": " << st.seconds_write << "}\n";
    }
  } catch (const std::exception& e) {
    std::cerr << e.what() << "\nAborting with a failure." << std::endl;
    return 1;
  }
  return 0;
}
