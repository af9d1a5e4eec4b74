#!/usr/bin/env python3
"""Which kernels wait for their global loads one at a time?  For every kernel of a .hip file (device assembly with the library's
flags): vector-memory loads, stores, and the s_waitcnt instructions that wait on vmcnt, by the count they wait down to.  A kernel
whose loads are each followed by `s_waitcnt vmcnt(0)` before the next load is issued pays one memory round trip per load - the
form the copy-in of az_tile_kernel had while its weight multiply sat inside the bounds test (0.306 -> 0.216 ms once it did not).

    python3 tools/isa_load_waits.py nis-sar-amtigmti-video_amd/csrc/az_pfa.hip [more.hip ...] [--min-loads 8]

Prints per kernel: loads, stores, waits, and `serial` = loads that are directly followed (before any other load) by a wait to
vmcnt(0) - the suspicious ones - for kernels with at least --min-loads loads, worst ratio first."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load)\w*")
STORE = re.compile(r"^\s*(global_store|buffer_store|flat_store)\w*")
WAIT = re.compile(r"^\s*s_waitcnt\s+(.*)")
VMC = re.compile(r"vmcnt\((\d+)\)")


def census(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.dirname(path), path, "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    rows = []
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        loads = stores = waits0 = waits = serial = 0
        pending = False            # a load has been issued and no other load since
        for line in body.splitlines():
            if LOAD.match(line):
                loads += 1
                pending = True
            elif STORE.match(line):
                stores += 1
            else:
                w = WAIT.match(line)
                if w:
                    v = VMC.search(w.group(1))
                    if v:
                        waits += 1
                        if int(v.group(1)) == 0:
                            waits0 += 1
                            if pending:
                                serial += 1
                            pending = False
        rows.append((name, loads, stores, waits, waits0, serial))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--min-loads", type=int, default=8)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = []
    for f in a.files:
        rows += [(os.path.basename(f),) + r for r in census(f)]
    rows = [r for r in rows if r[2] >= a.min_loads]
    rows.sort(key=lambda r: -(r[6] / max(r[2], 1)))
    print(f"{'file':18s} {'loads':>5s} {'stores':>6s} {'waits':>5s} {'to 0':>5s} {'serial':>6s}  kernel")
    for r in rows[: a.top]:
        demangled = subprocess.run(["c++filt", r[1]], capture_output=True, text=True).stdout.strip()
        print(f"{r[0]:18s} {r[2]:5d} {r[3]:6d} {r[4]:5d} {r[5]:5d} {r[6]:6d}  {demangled[:130]}")


if __name__ == "__main__":
    sys.exit(main())
