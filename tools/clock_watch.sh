#!/bin/bash
# the shader / memory / fabric clocks and the power the board reports while k_search runs back to back on the headline batch
# (one resident index, the same 1 M guides 300 times: ~8 s of launches; rocm-smi twice a second beside it)
mkdir -p gpurun_out
( for i in $(seq 1 240); do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | grep -v '====' | sed 's/  */ /g; s/GPU\[0\]//' | tr '\n' ';')"
    sleep 0.4
  done ) > gpurun_out/r05_clock_samples.txt &
SP=$!
python3 - <<'PY'
import sys, time, json
sys.path.insert(0, ".")
from importlib import import_module
import numpy as np, torch
bench = import_module("bench"); api = import_module("guidescan-cli_amd.api"); synth = import_module("guidescan-cli_amd.synth")
lens_name, _, probs = bench.WORKLOADS["hg38"]
text, names, lengths = bench.make_workload_genome(synth, "hg38", getattr(synth, lens_name), probs)
g = api.GenomeIndex.build(text, device=0)
seqs, pams, _, _ = synth.sample_guides(text, 1_000_000, seed=1000)
d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
print("loop starts", time.time(), flush=True)
ms = []
for i in range(300):
    _, _, st = g.enumerate_device(d_s.data_ptr(), 1_000_000, 20, d_p.data_ptr(), 3, mismatches=3)
    ms.append(st["ms_search"])
torch.cuda.synchronize()
print("loop ends", time.time(), flush=True)
print(json.dumps({"k_search_ms_first10": [round(x, 2) for x in ms[2:12]], "last10": [round(x, 2) for x in ms[-10:]], "mean": round(float(np.mean(ms[2:])), 2)}), flush=True)
g.close()
PY
kill $SP 2>/dev/null
