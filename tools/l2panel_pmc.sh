# FETCH_SIZE / WRITE_SIZE per dispatch of tools/l2panel.bin (TCC -> fabric bytes: what the XCD's L2 did not serve)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/l2p_$C -- $R/tools/l2panel.bin 16384 16 > $R/gpurun_out/l2p_$C.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
rows = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/l2p_{C}/*/*_counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if "fused_kernel" in r["Kernel_Name"] or "static_kernel" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][:50]})[C] = float(r["Counter_Value"]) / 2**20
for d in sorted(rows):
    r = rows[d]
    print(d, r["k"], f"fetch(x2) {2 * r.get('FETCH_SIZE', 0):.2f} GiB  write {r.get('WRITE_SIZE', 0):.2f} GiB")
PY
