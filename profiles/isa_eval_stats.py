#!/usr/bin/env python3
"""Static instruction mix of ONE Newton evaluation of k_thorough_dna's cfg2 instantiation
(<NCH 3, ZERO0, no +I, 1 wave, sliding, half-chunk tail, 1 category group>), from the built object:
    python profiles/isa_eval_stats.py > profiles/r6_newton_eval_isa.txt
The evaluation is the stretch from the table-driven exp (v_rndne_f64) of a Newton loop to the four v_readlane_b32
behind the paired DPP reduction; the first such stretch of the kernel (the pendant solve's loop body) is printed."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
obj = os.path.join(ROOT, "epa_ng_amd", "build", "thorough_dna.o")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    asm = subprocess.check_output([LLVM + "llvm-objdump", "-d", co], text=True)
name = "k_thorough_dnaILi3ELb1ELb0ELi1ELb0ELb1ELi1EE"
lines = asm.split("\n")
start = next(i for i, l in enumerate(lines) if name in l and l.rstrip().endswith(">:"))
ins = []
for l in lines[start + 1:]:
    if re.match(r"^[0-9a-f]+ <", l):
        break
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//", l)
    if m:
        ins.append((m.group(1), m.group(2)))
print("kernel %s: %d instructions" % (name, len(ins)))
# first evaluation body inside a loop: from the v_rndne_f64 that follows a backward s_cbranch to the 4th v_readlane_b32 behind it
i0 = next(i for i, (op, _) in enumerate(ins) if op.startswith("v_rndne_f64") and any(o.startswith("v_div_fmas") for o, _ in ins[max(0, i - 60):i]))
while not ins[i0 - 1][0].startswith("s_c") and not ins[i0 - 1][0].startswith("s_branch"):
    i0 -= 1
i1, seen = i0, 0
while seen < 4:
    if ins[i1][0].startswith("v_readlane_b32"):
        seen += 1
    i1 += 1
body = ins[i0:i1]
cls = collections.Counter()
for op, _ in body:
    if op.startswith(("v_fma_f64", "v_fmac_f64")): k = "fp64 fma"
    elif op.startswith(("v_mul_f64", "v_add_f64")): k = "fp64 mul / add"
    elif op.startswith("v_rcp_f64"): k = "v_rcp_f64"
    elif op.startswith(("ds_read", "ds_write")): k = op.split("_e")[0]
    elif op.startswith("v_mov_b32_dpp") or "permlane" in op or op.startswith("v_readlane"): k = "cross-lane (dpp / permlane / readlane)"
    elif op.startswith("s_"): k = "scalar (waitcnt, nop, branch, select)"
    elif op.startswith("v_"): k = "other vector (cndmask, cvt, ldexp, rndne, cmp, mov)"
    else: k = op
    cls[k] += 1
print("one Newton evaluation (exp of the proposal -> f, f' in scalar registers): %d instructions" % len(body))
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print("  %4d  %s" % (v, k))
valu = sum(v for k, v in cls.items() if not k.startswith(("scalar", "ds_")))
print("  vector instructions: %d, of them fp64 fma %d; LDS reads %d (the 36-entry table: 18 ds_read_b128 for the two full chunks"
      " + 9 for the half chunk, whose lanes need other entries; 1 ds_read_b64 = exp_tab's 2^(j/64))"
      % (valu, cls["fp64 fma"], sum(v for k, v in cls.items() if k.startswith("ds_read"))))
print("-- listing")
for op, a in body:
    print("  %-28s %s" % (op, a[:70]))
