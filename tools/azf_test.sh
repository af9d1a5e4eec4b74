set -e
cd /root/repo
export PYTHONPATH=/root/repo
for f in 0 1; do echo "SARX_AZ_FUSED=$f"; SARX_AZ_FUSED=$f timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-batch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done
