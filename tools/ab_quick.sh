# quick check on one GPU box
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond_the_tiles or half_a_million or heavy_item or repeat_guide or tile_sizes or gives_up" 2>&1 | tail -3
GS_DEBUG=1 timeout -k 10 400 python tools/rep_steps.py 5 2>&1 | grep -h "^step\|arena ran out\|released\|dropped" | tail -12
GS_DEBUG=1 timeout -k 10 900 python tools/rep_share_sweep.py hg38alu 20000 3 3 512:2048 > gpurun_out/r05_alu3.log 2>&1; grep -h "^{\|arena ran out\|released\|dropped\|Error" gpurun_out/r05_alu3.log | tail -8; grep -h "tile ordering: [0-9]* items\|guide(s) with an item" gpurun_out/r05_alu3.log | tail -3
