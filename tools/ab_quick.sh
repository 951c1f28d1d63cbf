# quick check on one GPU box: scoring tests, then the scoring kernels timed (rocprofv3 kernel stats) on the repeat-rich batch and the m <= 6 batch
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_score_kmers.py -x -q -m gpu 2>&1 | tail -3
for w in "hg38rep 20000 3" "hg38 20000 6"; do
  rm -rf /tmp/sc_prof
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/sc_prof -- python3 tools/score_bench.py $w 4 2>/dev/null | grep -h "^{"
  f=$(find /tmp/sc_prof -name '*kernel_stats.csv' | head -1)
  grep -h "k_score" $f | cut -c1-160
done
