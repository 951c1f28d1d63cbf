# A/B on one GPU box, library variants libgsamd$v.so (built with EXTRA=-D...): the repeat-rich batch under the heavy
# instantiation of k_search (tools/rep_share_sweep.py), the m <= 6 batch and the headline bench with it forced.
cd /root/repo
for v in "$@"; do
  export GS_LIB_PATH=/root/repo/guidescan-cli_amd/libgsamd$v.so
  echo "== variant '$v'"
  timeout -k 10 400 python tools/rep_share_sweep.py hg38rep 20000 3 3 512:2048 2>&1 | grep -h "^{"
  timeout -k 10 400 python tools/rep_share_sweep.py hg38 20000 6 3 512:2048 2>&1 | grep -h "^{"
  GS_HEAVY=1 timeout -k 10 400 python bench.py --cpu-sample 0 --steps 5 --warmup 2 --extra-rows off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'heavy_forced': 1, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'roofline': {k: d['roofline'].get(k) for k in ('avg_launch_ms','frac')}}))"
done
