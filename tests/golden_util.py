"""Helpers shared by the tests: golden fixtures (tests/golden/*.json) -> inputs."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
CASES = ["dna8_gtr_g_default", "dna8_gtr_fu_g4", "aa8_protgtr_g4"]


def read_fasta(path):
    out = []
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if line[0] == ">":
            out.append([line[1:].split()[0], ""])
        else:
            out[-1][1] += line.upper()
    return [(a, b) for a, b in out]


def load_case(name):
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    g["newick"] = open(os.path.join(GOLDEN, "data", g["tree_file"])).read().strip()
    g["msa"] = read_fasta(os.path.join(GOLDEN, "data", g["aln_file"]))
    return g
