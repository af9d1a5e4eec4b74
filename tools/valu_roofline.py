#!/usr/bin/env python3
"""Roofline fraction of the compute-bound kernels from a rocprofv3 kernel-stats summary and the ISA issue-cost table:

    python3 tools/valu_roofline.py profiles/r03_f_videosar_kernel_stats.csv [--echo-targets 35 --pulses 2500 --samples 22004 --nx 512]

frac = (work units / 64 lanes) * (SIMD issue cycles per unit-wave, profiles/r05_isa_slots.json) / (n_SIMD * clock * kernel time):
the share of the chip's vector issue cycles the kernel's own instruction stream needs (the rest is stalls).  PMC check
of the instruction counts: tools/pmc_compute.sh (SQ_INSTS_VALU per dispatch)."""
import argparse
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats_csv")
    ap.add_argument("--echo-targets", type=int, default=35)
    ap.add_argument("--pulses", type=int, default=2500)
    ap.add_argument("--samples", type=int, default=22004)
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--isa", default=os.path.join(ROOT, "profiles", "r05_isa_slots.json"))
    a = ap.parse_args()
    isa = json.load(open(a.isa))
    avg = {}
    with open(a.stats_csv) as fh:
        for row in csv.DictReader(fh):
            avg[row["Name"]] = float(row["AverageNs"]) * 1e-9
    units = {"echo_synth_kernel": float(a.echo_targets) * a.pulses * a.samples, "tdbp_kernel": float(a.pulses) * a.nx * a.nx,
             "tdbp_tile_kernel": float(a.pulses) * a.nx * a.nx}
    peak = isa["n_simd"] * isa["clock_hz"]
    out = {}
    for k, info in isa["kernels"].items():
        name = next((n for n in avg if ("::" + k + "(") in n), None)
        if name is None:
            continue
        t = avg[name]
        need = units[k] / 64.0 * info["cycles_per_unit"]
        out[k] = {"bound": "valu_issue", "unit": "SIMD issue cycles / s", "work_units": units[k], "work_unit": info["unit"],
                  "kernel_s": t, "rate_units_per_s": units[k] / t, "cycles_per_unit_wave": info["cycles_per_unit"],
                  "valu_mix_per_iteration": info["loop"]["valu"], "achieved": need / t, "peak": peak, "frac": need / t / peak}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
