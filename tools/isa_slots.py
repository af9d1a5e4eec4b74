#!/usr/bin/env python3
"""VALU issue cost of a kernel's innermost loops, read off the gfx950 ISA hipcc emits - the numerator of the roofline
fraction of the compute-bound kernels (echo synthesis, back-projection), whose bound is the vector issue rate, not HBM.

    python3 tools/isa_slots.py nis-sar-amtigmti-video_amd/csrc/echo.hip echo_synth_kernel [--json out.json]

Compiles the file to device assembly with the library's flags, takes the named kernel, and for every innermost loop
(a block the compiler marks "Inner Loop Header" up to its back edge, conditional blocks inside included) counts the
vector instructions by issue class.  Issue cost per wave64 instruction on one SIMD (MI355X_MICROARCH.md, per-instruction
cycle constants; tools/valubench.hip): fp32 / integer VALU 2 cycles, fp64 arithmetic (v_*_f64, conversions from / to
f64) 4, fp32 transcendentals (v_sin/cos/exp/log/rcp/rsq/sqrt_f32) 8, fp64 transcendentals (v_rcp/rsq/sqrt_f64) 16.
cycles_per_iteration = sum of those = SIMD cycles one wave's iteration occupies when the pipe never idles; with the loop's
work units per iteration (targets, pulses ...) and the measured rate this gives

    valu_fraction = rate [units/s] / 64 lanes * cycles_per_unit / (n_SIMD * clock)

(n_SIMD = 1024, clock 2.4 GHz: tools/clockprobe.hip reads 2.38-2.39 GHz under load).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
TRANS32 = re.compile(r"^v_(sin|cos|exp|log|rcp|rsq|sqrt)(_iflag)?_f32")
TRANS64 = re.compile(r"^v_(rcp|rsq|sqrt)_f64")
F64 = re.compile(r"^v_\w*f64")
COST = {"fp32_int": 2, "fp64": 4, "trans32": 8, "trans64": 16}


def classify(op):
    if TRANS64.match(op):
        return "trans64"
    if TRANS32.match(op):
        return "trans32"
    if F64.match(op):
        return "fp64"
    return "fp32_int"


def kernel_asm(path, kernel):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, path, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def loops(asm):
    """(header label, [instructions]) for every innermost loop: the header block and every block the compiler tags
    "in Loop: Header=<that label>" (conditional blocks inside the loop included)."""
    out = []
    for i, l in enumerate(asm):
        if "Inner Loop Header" not in l:
            continue
        j = i
        while not asm[j].startswith(".LBB"):
            j -= 1
        lab = asm[j].split(":")[0]                     # .LBB0_11
        tag = "Header=" + lab[2:]                      # Header=BB0_11
        body = []
        for k, m in enumerate(asm):
            starts = m.startswith(lab + ":") or ((m.startswith(".LBB") or m.lstrip().startswith("; %bb.")) and tag in m)
            if not starts:
                continue
            q = k + 1
            while q < len(asm) and not asm[q].startswith(".LBB") and not asm[q].lstrip().startswith("; %bb."):
                body.append(asm[q])
                q += 1
        out.append((lab, body))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--json")
    a = ap.parse_args()
    asm = kernel_asm(a.source, a.kernel)
    res = {"source": a.source, "kernel": a.kernel, "cost_cycles": COST, "loops": []}
    for lab, body in loops(asm):
        counts = {k: 0 for k in COST}
        lds = vmem = salu = 0
        for l in body:
            t = l.strip()
            if not t or t.startswith(";"):
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                counts[classify(op)] += 1
            elif op.startswith("ds_"):
                lds += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                vmem += 1
            elif op.startswith("s_"):
                salu += 1
        cyc = sum(COST[k] * v for k, v in counts.items())
        res["loops"].append({"header": lab, "valu": counts, "valu_instructions": sum(counts.values()), "lds": lds, "vmem": vmem,
                             "scalar": salu, "cycles_per_iteration": cyc})
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    sys.exit(main())
