"""N>1 path on CPU: world_size-2 gloo.  The sharding/gather layer is exercised with the
oracle standing in as the per-rank enumerate function AND as the unsharded checker
(the HIP library cannot run here).  CPU only."""
import os
import sys
from importlib import import_module

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol

parallel = import_module("guidescan-cli_amd.parallel")
synth = import_module("guidescan-cli_amd.synth")
api = import_module("guidescan-cli_amd.api")


def oracle_enumerate_fn(text):
    oidx = ol.OracleIndex(text)
    opts = ol.make_opts(mismatches=2)

    def fn(seqs, pams, **kw):
        offsets = [0]
        recs = []
        for i in range(seqs.shape[0]):
            hits, ctr, raw = oidx.enumerate(seqs[i].tobytes().decode(), pams[i].tobytes().decode(), opts)
            ol.lib().gso_free(raw[0])
            for pos, mm, idx, seq, row in hits:
                recs.append((pos, (mm << 61) | (idx << 60)))
            offsets.append(len(recs))
        return (np.asarray(offsets, dtype=np.uint64), np.array(recs, dtype=api.HIT_DTYPE).reshape(-1),
                dict(n=seqs.shape[0]))
    return fn


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text, _, _ = synth.make_genome([150_000], seed=6)
    seqs, pams, _, _ = synth.sample_guides(text, 23, seed=2)   # odd count: uneven chunks
    fn = oracle_enumerate_fn(text)
    out_dir = os.path.dirname(out)
    # chunks of 4 guides drawn from the shared counter: every rank writes the hit lists of what it drew to its own file
    path, mine, busy = parallel.enumerate_dealt(fn, seqs, pams, out_dir, 4, "t1", dist=dist)
    el = parallel.timed_steps(lambda i: None, 2, 1, lambda: None, dist=dist)
    assert el >= 0
    took = [None] * world
    dist.all_gather_object(took, mine)
    assert sorted(c for t in took for c in t) == list(range(6))      # every chunk exactly once
    dist.barrier()
    if rank == 0:
        full_off, full_hits, _ = fn(seqs, pams)
        off, hits = parallel.merge_chunk_files([os.path.join(out_dir, f"hits.rank{r}.gschunks") for r in range(world)], 23,
                                               api.HIT_DTYPE)
        assert np.array_equal(off, full_off)
        assert np.array_equal(hits, full_hits)
        open(out, "w").write("ok %d" % len(full_hits))
    dist.destroy_process_group()


def test_shard_bounds():
    assert parallel.shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert parallel.shard_bounds(2, 4) == [0, 1, 2, 2, 2]
    assert parallel.shard_bounds(0, 2) == [0, 0, 0]
    assert parallel.chunk_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.chunk_bounds(0, 4) == []
    assert abs(parallel.imbalance([1.0, 1.0, 2.0]) - (2.0 - 4.0 / 3.0) / 2.0) < 1e-12


def test_world2_gloo_dealt_chunks_equal_the_whole_batch(tmp_path):
    out = tmp_path / "r0.txt"
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    assert out.read_text().startswith("ok")


def _skew_worker(rank, world, port, out_dir):
    """a job whose first stretch of guides is repeat-dense (ten times the cost per guide): contiguous shards leave one rank
    with nearly all of the work; chunks drawn from the shared counter spread it"""
    import json
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, chunk = 6400, 100
    cost = np.where(np.arange(n) < n // 8, 10.0, 1.0) * 4e-5          # seconds per guide

    def work(c, lo, hi):
        time.sleep(float(cost[lo:hi].sum()))
    dist.barrier()
    mine, busy = parallel.deal_chunks(work, n, chunk, "skew", dist=dist)
    dealt = parallel.gather_floats(busy, dist)
    # the same key once more (the same kmers file enumerated twice, a retry): a job of its own, every chunk dealt again
    again, _ = parallel.deal_chunks(lambda c, lo, hi: None, n, chunk, "skew", dist=dist)
    n_again = parallel.gather_floats(float(len(again)), dist)
    assert sum(n_again) == len(parallel.chunk_bounds(n, chunk)), n_again
    b = parallel.shard_bounds(n, world)
    static = parallel.gather_floats(float(cost[b[rank]:b[rank + 1]].sum()), dist)
    if rank == 0:
        json.dump({"dealt": dealt, "static": static, "chunks": len(mine)}, open(os.path.join(out_dir, "skew.json"), "w"))
    dist.destroy_process_group()


def test_dealt_chunks_balance_a_skewed_job(tmp_path):
    """world 2 and 4 on gloo: the imbalance (what the slowest rank adds to the job) stays within 10 % with chunks drawn from
    the counter, where contiguous shards of the same job are 35-60 % out of balance"""
    import json
    for world in (2, 4):
        port = 31500 + (os.getpid() + world) % 2000
        mp.spawn(_skew_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        r = json.load(open(tmp_path / "skew.json"))
        assert parallel.imbalance(r["dealt"]) <= 0.10, r
        assert parallel.imbalance(r["static"]) >= 0.30, r


def _run_bench_stub(cmd, env_extra):
    import json
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GS_BENCH_STUB="1", **env_extra)
    env.pop("RANK", None)
    env.pop("GS_BENCH_TEXT", None)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout        # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no external launcher) starts two ranks, shares one genome file
    through /dev/shm and prints one line; rehearsed on gloo with the stub step (no GPU here)"""
    j = _run_bench_stub([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--workload", "saccer3", "--batch", "64"], {})
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["shared_text"] is True
    assert j["steps_run_per_rank"] == [4, 4]
    assert len(set(j["text_checksums"])) == 1


def test_bench_under_the_drivers_torchrun_command():
    """the driver's own N>1 command line (torch.distributed.run ... bench.py --gpus N): local rank 0
    generates the genome, the other rank maps it"""
    port = 23000 + os.getpid() % 2000
    j = _run_bench_stub([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                         "--steps", "2", "--warmup", "1", "--workload", "saccer3", "--batch", "64"], {})
    assert j["n_gpus"] == 2 and j["shared_text"] is False
    assert len(set(j["text_checksums"])) == 1


def test_bench_strong_scaling_splits_one_guide_set():
    """`--scaling strong`: --batch is the whole job's guides per step, dealt to the ranks in contiguous shards of ONE
    seeded set (configs 4 and 5: a fixed candidate set over 8 GPUs).  Two ranks on gloo: uneven shards, the union is
    the set a single rank would have taken, the line says strong."""
    j2 = _run_bench_stub([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "saccer3", "--batch", "65", "--scaling", "strong"], {})
    assert j2["scaling"] == "strong" and j2["n_gpus"] == 2 and j2["guides_per_step"] == 65
    assert j2["guides_per_rank"] == [33, 32]
    port = 25000 + os.getpid() % 2000
    j1 = _run_bench_stub([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1",
                          "--steps", "2", "--warmup", "1", "--workload", "saccer3", "--batch", "65", "--scaling", "strong"], {})
    assert j1["guides_per_rank"] == [65]
    assert sum(j2["guide_checksums"]) == j1["guide_checksums"][0]     # the same guides, split
    # chunks of 10 guides from the shared counter instead of two shards: every guide on exactly one rank
    jd = _run_bench_stub([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--workload", "saccer3", "--batch", "65", "--scaling", "strong", "--deal", "10"], {})
    assert sum(jd["guides_per_rank"]) == 65 and all(g % 10 in (0, 5) for g in jd["guides_per_rank"])
    assert sum(jd["guide_checksums"]) == j1["guide_checksums"][0]
    # weak scaling keeps --batch per rank
    jw = _run_bench_stub([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--workload", "saccer3", "--batch", "65"], {})
    assert jw["scaling"] == "weak" and jw["guides_per_rank"] == [65, 65] and jw["guides_per_step"] == 130


def _run_tool_stub(tool, world, extra):
    import json
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GS_TOOLS_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    port = 21000 + (os.getpid() * 7 + world * 131 + len(tool)) % 4000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), f"tools/{tool}", "--workload", "saccer3"] + extra
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout        # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["stub"] is True
    return j


def test_genomewide_and_config5_tools_under_world2_gloo():
    """tools/genomewide_enumerate.py (config 4) and tools/config5_stream.py (config 5) as the driver would start them on a
    node: torch.distributed.run, rank r takes the batches b with b % world == r, totals summed and times MAX-reduced, one
    line from rank 0.  Rehearsed on gloo with tools/_stub.py standing in for the index (its numbers depend on the guides
    alone): two ranks report the totals one rank reports - no batch dropped, none taken twice."""
    g1 = _run_tool_stub("genomewide_enumerate.py", 1, ["--batch", "50000", "--max-guides", "0", "--score"])
    g2 = _run_tool_stub("genomewide_enumerate.py", 2, ["--batch", "50000", "--max-guides", "0", "--score"])
    assert g1["n_gpus"] == 1 and g2["n_gpus"] == 2
    assert g1["guides_enumerated"] == g1["candidates_scanned"] > 300_000
    for k in ("candidates_scanned", "guides_enumerated", "hits"):
        assert g1[k] == g2[k], k
    assert abs(g1["mean_specificity"] - g2["mean_specificity"]) < 1e-6   # float32 partial sums per batch
    c1 = _run_tool_stub("config5_stream.py", 1, ["--guides", "10007", "--batch", "1000"])
    c2 = _run_tool_stub("config5_stream.py", 2, ["--guides", "10007", "--batch", "1000"])
    assert c1["guides"] == c2["guides"] == 10007 and c1["hits"] == c2["hits"] > 10007
    assert abs(c1["mean_specificity"] - c2["mean_specificity"]) < 1e-9
