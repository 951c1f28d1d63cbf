"""Two ranks with REAL kernels on the one GPU of the box: two fresh child processes (started before either touches the GPU,
rendezvous over gloo on 127.0.0.1), each with a chr1-sized index of its own in HBM, run parallel.enumerate_dealt with the HIP
enumerate function - chunks of one guide set drawn from the shared counter, per-rank chunk files - and the merged files
equal what one rank returns for the whole set, byte for byte.  What the strong-scaling form of a node does per GPU
(src/guidescan.cxx:226-231 deals guides to threads; here to processes, each with the whole index), with the ranks
sharing one device instead of owning one each: N > 1 GPUs stays unmeasured (no node was ever available), this covers the
code path - dealing, two processes driving the kernels side by side, the merge."""
import json
import os
import subprocess
import sys
import tempfile
from importlib import import_module
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import json, os, sys, time
from importlib import import_module
import numpy as np
root, rank, world, port, out_dir, n_guides, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
sys.path.insert(0, root)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
import torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)       # before anything touches the GPU
api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")
parallel = import_module("guidescan-cli_amd.parallel")
text, names, lengths = synth.make_genome([synth.CHR1_LENGTH], seed=1)
t0 = time.perf_counter()
gidx = api.GenomeIndex.build(text, device=0)
t_build = time.perf_counter() - t0
seqs, pams, _, _ = synth.sample_guides(text, n_guides, seed=31)
gidx.enumerate(seqs[:chunk], pams[:chunk], mismatches=3)           # the handle's one-off work (tables, workspace), untimed
dist.barrier()
t0 = time.perf_counter()
path, mine, busy = parallel.enumerate_dealt(gidx.enumerate, seqs, pams, out_dir, chunk, "two_ranks", dist=dist, mismatches=3)
t_job = time.perf_counter() - t0
dist.barrier()
if rank == 0:                                                      # the whole set on one rank: what the merge must equal
    off, hits, _ = gidx.enumerate(seqs, pams, mismatches=3)
    np.save(os.path.join(out_dir, "whole_offsets.npy"), off)
    np.save(os.path.join(out_dir, "whole_hits.npy"), hits)
json.dump({"rank": rank, "chunks": len(mine), "busy_s": busy, "job_s": t_job, "index_build_s": t_build, "file": str(path)},
          open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
gidx.close()
dist.destroy_process_group()
'''


def test_two_ranks_with_real_kernels_on_one_gpu_deal_one_guide_set():
    api = import_module("guidescan-cli_amd.api")
    parallel = import_module("guidescan-cli_amd.parallel")
    n_guides, chunk, world = 100_000, 5_000, 2
    port = str(33500 + os.getpid() % 2000)
    with tempfile.TemporaryDirectory(prefix="gs_two_ranks_") as d:
        worker = Path(d) / "worker.py"
        worker.write_text(WORKER)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, str(worker), str(ROOT), str(r), str(world), port, d, str(n_guides), str(chunk)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append(o)
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-2000:]
        reps = [json.load(open(Path(d) / f"rank{r}.json")) for r in range(world)]
        n_chunks = len(parallel.chunk_bounds(n_guides, chunk))
        assert sum(r["chunks"] for r in reps) == n_chunks
        assert all(r["chunks"] >= 1 for r in reps), reps   # both ranks drove kernels
        off, hits = parallel.merge_chunk_files([r["file"] for r in reps], n_guides, api.HIT_DTYPE)
        whole_off, whole_hits = np.load(Path(d) / "whole_offsets.npy"), np.load(Path(d) / "whole_hits.npy")
        assert np.array_equal(off, whole_off)
        assert hits.tobytes() == whole_hits.tobytes()
        assert int(off[-1]) >= n_guides                      # every guide finds at least its own site
        print("two ranks on one GPU:", json.dumps({"per_rank": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()
                                                                  if k != "file"} for r in reps],
                                                   "imbalance": round(parallel.imbalance([r["busy_s"] for r in reps]), 3),
                                                   "hits": int(off[-1])}))
