"""Several host threads on ONE handle.  The reference calls its seam from N std::threads on one const index
(src/guidescan.cxx:240-247); a gs_index owns workspace, lazily built tables and result buffers, so every entry
point holds the handle's lock for its whole call (include/guidescan_amd.h, gs_index).  Threads that hammer one
handle with different batches - host-pointer calls, and device-pointer calls bracketed by gs_index_lock - must
each get exactly the bytes a single thread gets.  ctypes releases the GIL inside the library, so the calls really
overlap."""
import threading
from importlib import import_module

import numpy as np
import pytest

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")

pytestmark = pytest.mark.gpu


def test_two_threads_on_one_handle_get_the_single_thread_bytes():
    import torch
    text, names, lengths = synth.make_repeat_genome([1_500_000, 700_000], seed=9)
    gidx = api.GenomeIndex.build(text, device=0)
    gs = api.make_genome_structure(names, lengths)
    try:
        jobs = []
        for j, (n, m, alt) in enumerate([(300, 3, ()), (120, 4, ("NAG",)), (500, 2, ()), (200, 3, ("NAG", "NGA"))]):
            seqs, pams, _, _ = synth.sample_guides(text, n, seed=50 + j)
            jobs.append((seqs, pams, m, alt))
        # single thread first: what every call must return
        want = []
        for seqs, pams, m, alt in jobs:
            off, hits, _ = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt)
            _, spec = gidx.score(gs, seqs, 3, off, hits, want_cfd=False)
            want.append((off.tobytes(), hits.tobytes(), spec.tobytes()))
        errors = []
        rounds = 6

        def host_pointer_worker(k):
            try:
                for r in range(rounds):
                    j = (k + r) % len(jobs)
                    seqs, pams, m, alt = jobs[j]
                    off, hits, _ = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt)
                    _, spec = gidx.score(gs, seqs, 3, off, hits, want_cfd=False)
                    if (off.tobytes(), hits.tobytes(), spec.tobytes()) != want[j]:
                        errors.append(("host", k, r, j))
            except Exception as e:   # noqa: BLE001 - reported below
                errors.append(("host", k, repr(e)))

        def device_pointer_worker(k):
            try:
                torch.cuda.set_device(0)
                for r in range(rounds):
                    j = (k + 2 * r + 1) % len(jobs)
                    seqs, pams, m, alt = jobs[j]
                    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
                    with gidx.locked():   # the results stay in the handle's buffers until we have copied them
                        d_off, d_hits, st = gidx.enumerate_device(d_s.data_ptr(), seqs.shape[0], 20, d_p.data_ptr(), 3,
                                                                  mismatches=m, alt_pams=alt)
                        off = torch.empty(seqs.shape[0] + 1, dtype=torch.int64, device="cuda")
                        hits = torch.empty((st["n_hits"], 2), dtype=torch.int64, device="cuda")
                        import ctypes as C
                        hip = C.CDLL("libamdhip64.so")
                        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                        assert hip.hipMemcpy(off.data_ptr(), d_off, 8 * (seqs.shape[0] + 1), 3) == 0
                        if st["n_hits"]:
                            assert hip.hipMemcpy(hits.data_ptr(), d_hits, 16 * st["n_hits"], 3) == 0
                    if off.cpu().numpy().astype(np.uint64).tobytes() != want[j][0] or hits.cpu().numpy().tobytes() != want[j][1]:
                        errors.append(("device", k, r, j))
            except Exception as e:   # noqa: BLE001
                errors.append(("device", k, repr(e)))

        threads = [threading.Thread(target=host_pointer_worker, args=(0,)), threading.Thread(target=host_pointer_worker, args=(1,)),
                   threading.Thread(target=device_pointer_worker, args=(2,))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in threads), "a thread did not finish"
        assert not errors, errors
    finally:
        gidx.close()
