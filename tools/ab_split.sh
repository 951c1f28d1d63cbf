#!/bin/bash
# the two-launch form of the sharing (GS_SPLIT_SHARE: 1 behind, 2 beside the search launch) against the one-launch forms, on the
# headline batch (what the publishing code costs a batch without heavy passes) and on the repeat-rich batch
set -e
mkdir -p gpurun_out
timeout -k 10 400 python tools/rep_share_sweep.py hg38 1000000 3 3 default GS_SPLIT_SHARE=2 default GS_SPLIT_SHARE=2 > gpurun_out/r05_split_hg38.txt 2>gpurun_out/r05_split_hg38.err
cat gpurun_out/r05_split_hg38.txt
timeout -k 10 500 python tools/rep_share_sweep.py hg38rep 20000 3 4 GS_HEAVY=1 GS_SPLIT_SHARE=2 > gpurun_out/r05_split_rep.txt 2>gpurun_out/r05_split_rep.err
cat gpurun_out/r05_split_rep.txt
