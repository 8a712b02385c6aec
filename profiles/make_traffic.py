#!/usr/bin/env python3
"""rN_pmc_summary.txt (+ the AA summary) -> rN_traffic.json: FETCH_SIZE / WRITE_SIZE per launch of the
kernels bench.py prices, stamped with a hash of the kernel sources the counters were taken from --
bench.py reports `roofline.traffic` only while that hash matches the sources it runs (a kernel change
without a profile refresh gives null, not stale bytes).
    python profiles/make_traffic.py r3 <reads_per_step> <pairs_per_launch> [aa_reads aa_pairs]"""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "epa_ng_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for l in open(path):
        m = re.match(r"== (\S+)", l)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", l)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


if __name__ == "__main__":
    tag = sys.argv[1]
    reads, pairs = int(sys.argv[2]), float(sys.argv[3])
    dna = parse(os.path.join(ROOT, "gpurun_out", tag + "_pmc_summary.txt"))
    out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate "
                       "runs; KB), bench.py default workload.  bytes = 2 x FETCH_SIZE + WRITE_SIZE: the factor is "
                       "calibrated on known byte counts (profiles/r2_traffic_calibration.txt: on gfx950 FETCH_SIZE "
                       "reports 0.500 of coalesced streaming reads at 8 and 16 bytes per lane, WRITE_SIZE is exact).",
           "fetch_correction": 2.0, "kernel_sources_sha16": src_hash()}
    for k, name in (("k_thorough", "k_thorough_dna"), ("k_preplace_pairs", "k_preplace_pairs"), ("k_select", "k_select")):
        if k in dna and "FETCH_SIZE" in dna[k]:
            out[name] = {"reads_per_step": reads, "fetch_kb": dna[k]["FETCH_SIZE"], "write_kb": dna[k].get("WRITE_SIZE", 0.0)}
            if name == "k_thorough_dna":
                out[name]["pairs_per_launch"] = pairs
    if len(sys.argv) > 5:
        aa = parse(os.path.join(ROOT, "gpurun_out", tag + "_aa_pmc_summary.txt"))
        if "k_preplace_sites" in aa and "FETCH_SIZE" in aa["k_preplace_sites"]:
            out["k_preplace_sites"] = {"reads_per_step": int(sys.argv[4]), "fetch_kb": aa["k_preplace_sites"]["FETCH_SIZE"],
                                       "write_kb": aa["k_preplace_sites"].get("WRITE_SIZE", 0.0)}
        if "k_thorough" in aa and "FETCH_SIZE" in aa["k_thorough"]:
            out["k_thorough_aa_mfma"] = {"reads_per_step": int(sys.argv[4]), "pairs_per_launch": float(sys.argv[5]),
                                         "fetch_kb": aa["k_thorough"]["FETCH_SIZE"], "write_kb": aa["k_thorough"].get("WRITE_SIZE", 0.0)}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", tag + "_traffic.json"), "w"), indent=1)
    print(json.dumps(out)[:400])
