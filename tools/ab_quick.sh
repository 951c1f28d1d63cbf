# quick check on one GPU box: the sharing / ordering parity tests, then the repeat-rich batch under the plain and the heavy
# instantiation (GS_LIB_PATH may name a variant build; with a -DGS_SH_PROFILE build GS_DEBUG=1 prints the heavy launch's phases)
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "repeat_guide or heavy_item or tile_sizes or gives_up" 2>&1 | tail -3
timeout -k 10 400 python tools/rep_share_sweep.py hg38rep 20000 3 3 0:2048 512:2048 256:1024 2>&1 | grep -h "^{\|heavy launch" | tail -8
