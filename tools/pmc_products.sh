# HBM traffic of the launches of one two-channel 8192^2 frame in its fused form (FETCH_SIZE / WRITE_SIZE in separate passes,
# FETCH_SIZE doubled as on gfx950): usage (GPU box, repo root): bash tools/pmc_products.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/prod_$C -- python3 $R/tools/bench_twochannel.py 8192 2 fused > $R/gpurun_out/prod_$C.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/prod_{C}/*/*_counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][C].append(float(r["Counter_Value"]) * 1024)
for k, v in acc.items():
    if "az_tile" in k or "range_pass" in k or "ati" in k:
        fe = 2 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        print(f"{k[:70]:70s} read {fe / 2**30:6.3f} GiB  write {wr / 2**30:6.3f} GiB  per launch")
PY
