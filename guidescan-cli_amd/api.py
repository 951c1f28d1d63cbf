"""ctypes binding of libgsamd.so (include/guidescan_amd.h) plus the thin host layer
that mirrors the reference's per-guide pipeline interface
(include/genomics/process.hpp:35-158) over the batch C-ABI.

The product path is the HIP library only: there is no CPU fallback.  Importing
this module without the built library raises; calling it without a GPU returns
GS_ERR_DEVICE from the library, surfaced as GsError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
# GS_LIB_PATH: another build of the same library (kernel tuning experiments); default in-tree
LIB_PATH = Path(os.environ.get("GS_LIB_PATH", str(PKG / "libgsamd.so")))

GS_FLAG_PAM_AT_START = 1
GS_FLAG_FAITHFUL_WALK = 2
GS_FLAG_COUNT_REQUESTS = 4
GS_FLAG_RAW_COUNTS = 8
GS_FLAG_NO_NEW_TABLES = 16


class GsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"gs_status {status}: {msg}")
        self.status = status


class GsHit(C.Structure):
    _fields_ = [("pos", C.c_int64), ("key", C.c_uint64)]


HIT_DTYPE = np.dtype([("pos", "<i8"), ("key", "<u8")])
HIT_EX_DTYPE = np.dtype([("pos", "<i8"), ("seq", "S32"), ("mismatches", "<u4"),
                         ("dna_bulges", "u1"), ("rna_bulges", "u1"), ("index", "u1"), ("seq_len", "u1")])


class GsResultView(C.Structure):
    _fields_ = [("n_guides", C.c_uint64), ("n_hits", C.c_uint64),
                ("guide_offsets", C.POINTER(C.c_uint64)), ("hits", C.POINTER(GsHit)),
                ("n_ext", C.c_uint64), ("n_matches", C.c_uint64),
                ("ms_search", C.c_float), ("ms_total", C.c_float),
                ("n_unsupported", C.c_uint64), ("guide_flags", C.POINTER(C.c_uint8)),
                ("raw_hits", C.POINTER(C.c_uint32))]


class GsSaReport(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("not_permutation", C.c_uint64), ("sampled", C.c_uint64),
                ("out_of_order", C.c_uint64), ("undecided", C.c_uint64), ("bwt_mismatch", C.c_uint64)]


class GsGenomeStructure(C.Structure):
    _fields_ = [("chr_names", C.POINTER(C.c_char_p)), ("chr_lengths", C.POINTER(C.c_uint64)),
                ("n_chr", C.c_uint32)]


class GsKmer(C.Structure):
    _fields_ = [("id", C.c_char_p), ("sequence", C.c_char_p), ("pam", C.c_char_p),
                ("sense_positive", C.c_int)]


GS_TEXT_SAM = 0x100
GS_TEXT_COMPLETE = 0x200


def build_library():
    subprocess.run(["make", "-s", "-C", str(PKG / "csrc")], check=True, timeout=3600)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is mandatory; there is no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.gs_index_build.restype = i32
    L.gs_index_build.argtypes = [vp, u64, i32, C.POINTER(vp)]
    L.gs_index_build_with_sa.restype = i32
    L.gs_index_build_with_sa.argtypes = [vp, u64, vp, vp, i32, C.POINTER(vp)]
    L.gs_index_open_sdsl.restype = i32
    L.gs_index_open_sdsl.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.gs_sdsl_extract_text.restype = i32
    L.gs_sdsl_extract_text.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(u64)]
    L.gs_index_save_sa.restype = i32
    L.gs_index_save_sa.argtypes = [vp, vp, u64, C.c_char_p]
    L.gs_index_open_sa.restype = i32
    L.gs_index_open_sa.argtypes = [vp, u64, C.c_char_p, i32, C.POINTER(vp)]
    L.gs_index_close.argtypes = [vp]
    L.gs_index_genome_length.restype = u64
    L.gs_index_genome_length.argtypes = [vp]
    L.gs_index_device_bytes.restype = u64
    L.gs_index_device_bytes.argtypes = [vp]
    L.gs_enumerate.restype = i32
    L.gs_enumerate.argtypes = [vp, vp, u64, u32, vp, u32, C.c_char_p, u32, u32, u32, C.POINTER(vp)]
    L.gs_enumerate_device.restype = i32
    L.gs_enumerate_device.argtypes = [vp, vp, u64, u32, vp, u32, C.c_char_p, u32, u32, u32, vp,
                                      C.POINTER(vp), C.POINTER(vp), C.POINTER(GsResultView)]
    L.gs_result_get.restype = i32
    L.gs_result_get.argtypes = [vp, C.POINTER(GsResultView)]
    L.gs_result_free.argtypes = [vp]
    L.gs_decode_sequence.restype = i32
    L.gs_decode_sequence.argtypes = [C.c_char_p, u32, u32, u32, u64, C.c_char_p]
    L.gs_rank_bwt4.restype = i32
    L.gs_rank_bwt4.argtypes = [vp, i32, vp, u64, vp]
    L.gs_resolve.restype = i32
    L.gs_resolve.argtypes = [vp, i32, vp, u64, vp]
    L.gs_index_meta.restype = i32
    L.gs_index_meta.argtypes = [vp, i32, vp, C.POINTER(u64)]
    L.gs_index_copy_sa.restype = i32
    L.gs_index_copy_sa.argtypes = [vp, i32, vp]
    L.gs_index_last_counters.restype = i32
    L.gs_index_last_counters.argtypes = [vp, vp]
    L.gs_index_last_sharing.restype = i32
    L.gs_index_last_sharing.argtypes = [vp, vp]
    L.gs_index_prepare.restype = i32
    L.gs_index_prepare.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.gs_index_set_option.restype = i32
    L.gs_index_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.gs_index_get_option.restype = i32
    L.gs_index_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, u64]
    L.gs_index_lock.restype = i32
    L.gs_index_lock.argtypes = [vp]
    L.gs_index_unlock.restype = i32
    L.gs_index_unlock.argtypes = [vp]
    L.gs_index_verify_sa.restype = i32
    L.gs_index_verify_sa.argtypes = [vp, i32, vp, u64, u64, u64, C.POINTER(GsSaReport)]
    L.gs_calculate_cfd.restype = C.c_float
    L.gs_calculate_cfd.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.gs_format_guide.restype = i32
    L.gs_format_guide.argtypes = [C.POINTER(GsGenomeStructure), C.POINTER(GsKmer), vp, u64, u32, u32,
                                  C.c_int64, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gs_format_guide_scored.restype = i32
    L.gs_format_guide_scored.argtypes = [C.POINTER(GsGenomeStructure), C.POINTER(GsKmer), vp, u64, u32, u32,
                                         C.c_int64, C.c_float, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gs_format_guides_scored.restype = i32
    L.gs_format_guides_scored.argtypes = [C.POINTER(GsGenomeStructure), vp, u64, vp, vp, vp, vp, u32, u32, C.c_int64,
                                          C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gs_format_header.restype = i32
    L.gs_format_header.argtypes = [C.POINTER(GsGenomeStructure), u32, C.POINTER(vp),
                                   C.POINTER(C.c_size_t)]
    L.gs_free.argtypes = [vp]
    L.gs_enumerate_bulges.restype = i32
    L.gs_enumerate_bulges.argtypes = [vp, vp, u64, u32, vp, u32, C.c_char_p, u32, u32, u32, u32, u32,
                                      C.POINTER(vp)]
    L.gs_result_ex_get.restype = i32
    L.gs_result_ex_get.argtypes = [vp, C.POINTER(u64), C.POINTER(vp), C.POINTER(vp)]
    L.gs_result_ex_free.argtypes = [vp]
    L.gs_result_ex_raw_hits.restype = i32
    L.gs_result_ex_raw_hits.argtypes = [vp, C.POINTER(vp)]
    L.gs_decode_sequence_ex.restype = i32
    L.gs_decode_sequence_ex.argtypes = [vp, C.c_char_p]
    L.gs_enumerate_general.restype = i32
    L.gs_enumerate_general.argtypes = [vp, vp, u64, u32, vp, u32, C.c_char_p, u32, u32, u32, u32, u32,
                                       C.POINTER(vp)]
    L.gs_enumerate_general_pams.restype = i32
    L.gs_enumerate_general_pams.argtypes = [vp, vp, u64, u32, vp, u32, C.c_char_p, vp, u32, u32, u32, u32, u32,
                                            C.POINTER(vp)]
    L.gs_index_last_guide_flags.restype = i32
    L.gs_index_last_guide_flags.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.gs_format_guide_ex.restype = i32
    L.gs_format_guide_ex.argtypes = [C.POINTER(GsGenomeStructure), C.POINTER(GsKmer), vp, u64, u32, u32,
                                     C.c_int64, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gs_score_device.restype = i32
    L.gs_score_device.argtypes = [vp, vp, u64, u32, u32, u32, C.c_int64, C.POINTER(GsGenomeStructure), vp, vp,
                                  vp, vp, vp]
    L.gs_score.restype = i32
    L.gs_score.argtypes = [vp, vp, u64, u32, u32, u32, C.c_int64, C.POINTER(GsGenomeStructure), vp, vp, vp, vp]
    L.gs_kmers_generate.restype = i32
    L.gs_kmers_generate.argtypes = [i32, vp, u64, i32, C.c_char_p, u32, u32, vp, C.POINTER(vp)]
    L.gs_kmers_get.restype = i32
    L.gs_kmers_get.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                               C.POINTER(vp)]
    L.gs_kmers_free.argtypes = [vp]
    L.gs_status_string.restype = C.c_char_p
    L.gs_status_string.argtypes = [i32]
    L.gs_version.restype = C.c_char_p
    _lib = L
    return L


EXPORTS = ["gs_index_build", "gs_index_build_with_sa", "gs_index_open_sdsl", "gs_index_close",
           "gs_index_genome_length", "gs_index_device_bytes", "gs_enumerate", "gs_enumerate_device",
           "gs_result_get", "gs_result_free", "gs_decode_sequence", "gs_rank_bwt4", "gs_resolve",
           "gs_index_meta", "gs_index_copy_sa", "gs_calculate_cfd", "gs_status_string", "gs_version",
           "gs_format_guide", "gs_format_header", "gs_free", "gs_sdsl_extract_text",
           "gs_enumerate_bulges", "gs_result_ex_get", "gs_result_ex_free", "gs_decode_sequence_ex",
           "gs_format_guide_ex", "gs_score_device", "gs_score", "gs_kmers_generate", "gs_kmers_get",
           "gs_kmers_free", "gs_format_guide_scored", "gs_index_verify_sa", "gs_index_last_counters", "gs_enumerate_general",
           "gs_index_last_guide_flags", "gs_enumerate_general_pams", "gs_index_save_sa", "gs_index_open_sa", "gs_format_guides_scored", "gs_result_ex_raw_hits",
           "gs_debug_seed_recipes", "gs_debug_choose_thresholds", "gs_debug_tile_plan", "gs_debug_guide_descriptor", "gs_index_lock", "gs_index_unlock",
           "gs_index_last_sharing", "gs_index_set_option", "gs_index_get_option", "gs_index_prepare"]


def _check(rc):
    if rc != 0:
        raise GsError(rc, lib().gs_status_string(rc).decode())


def seed_recipes(k, L, P, m, n_x, astar=None, deep=False):
    """the seed plan of k_search for a batch shape (gs_debug_seed_recipes; host only):
    (one-sided, this strand's share, the other strand's share) as uint64 arrays"""
    L_ = lib()
    L_.gs_debug_seed_recipes.restype = C.c_int
    L_.gs_debug_seed_recipes.argtypes = [C.c_uint32] * 5 + [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    a = None
    if astar is not None:
        a = (C.c_uint32 * 8)(*(list(astar) + [15] * 8)[:8])
    counts = (C.c_uint64 * 3)()
    _check(L_.gs_debug_seed_recipes(k, L, P, m, n_x, a, 1 if deep else 0, None, 0, counts))
    n = sum(counts)
    out = np.zeros(n, dtype=np.uint64)
    _check(L_.gs_debug_seed_recipes(k, L, P, m, n_x, a, 1 if deep else 0, out.ctypes.data, n, counts))
    c0, c1 = int(counts[0]), int(counts[1])
    return out[:c0], out[c0:c0 + c1], out[c0 + c1:]


def guide_descriptor(q: int, pams, L: int, P: int, k: int, x_len: int, codes=(0, 0xFFFFFFFF), n_pt: int = 1, valid: bool = True):
    """the sixteen words an item of the seeding launches starts from (gs_debug_guide_descriptor; host only)"""
    L_ = lib()
    L_.gs_debug_guide_descriptor.restype = C.c_int
    L_.gs_debug_guide_descriptor.argtypes = [C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    pam = (C.c_uint32 * 4)(*(list(pams) + [0] * (4 - len(pams))))
    code = (C.c_uint32 * 2)(*codes)
    out = (C.c_uint32 * 16)()
    _check(L_.gs_debug_guide_descriptor(q, pam, len(pams), 1 if valid else 0, L, P, k, x_len, n_pt, code, out))
    names = ("q_lo", "q_hi", "pam0", "pam1", "pam2", "pam3", "meta", "pidx0", "pidxg", "qrem_b", "bsel_z", "bsel_w", "qhot",
             "key_a", "key_b", "guide")
    return dict(zip(names, (int(x) for x in out)))


def choose_thresholds(m, n_x, n_o, n_r, pam_expansions=4.0, verify_a=1.5, verify_b=1.9):
    """the cost model's thresholds a*(o) (gs_debug_choose_thresholds)"""
    L_ = lib()
    L_.gs_debug_choose_thresholds.restype = None
    L_.gs_debug_choose_thresholds.argtypes = [C.c_uint32] * 4 + [C.c_double] * 3 + [C.c_void_p]
    out = (C.c_uint32 * 8)()
    L_.gs_debug_choose_thresholds(m, n_x, n_o, n_r, pam_expansions, verify_a, verify_b, out)
    return list(out)


def tile_plan(records):
    """the tile ordering's plan for an item of `records` match records (gs_debug_tile_plan; host only):
    dict(buckets, slot, per, wave_tile, max_buckets)"""
    L_ = lib()
    L_.gs_debug_tile_plan.restype = None
    L_.gs_debug_tile_plan.argtypes = [C.c_uint32, C.c_void_p]
    out = (C.c_uint32 * 5)()
    L_.gs_debug_tile_plan(records, out)
    return dict(buckets=out[0], slot=out[1], per=out[2], wave_tile=out[3], max_buckets=out[4])


def make_genome_structure(names, lengths):
    arr_n = (C.c_char_p * len(names))(*[n.encode() for n in names])
    arr_l = (C.c_uint64 * len(lengths))(*lengths)
    g = GsGenomeStructure(arr_n, arr_l, len(names))
    g._keep = (arr_n, arr_l)
    return g


def format_header(gs, sam=False, complete=True) -> str:
    out, n = C.c_void_p(), C.c_size_t()
    flags = (GS_TEXT_SAM if sam else 0) | (GS_TEXT_COMPLETE if complete else 0)
    _check(lib().gs_format_header(C.byref(gs), flags, C.byref(out), C.byref(n)))
    s = C.string_at(out, n.value).decode()
    lib().gs_free(out)
    return s


def format_guide(gs, gid, sequence, pam, sense_positive, hits, mismatches, sam=False, complete=True,
                 start=False, max_off_targets=-1, specificity=None) -> str:
    """hits: numpy HIT_DTYPE array of this guide (canonical order, as gs_enumerate returns);
    specificity: the guide's float from GenomeIndex.score (then the host does no CFD arithmetic)"""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    k = GsKmer(gid.encode(), sequence.encode(), pam.encode(), int(sense_positive))
    out, n = C.c_void_p(), C.c_size_t()
    flags = ((GS_TEXT_SAM if sam else 0) | (GS_TEXT_COMPLETE if complete else 0) |
             (GS_FLAG_PAM_AT_START if start else 0))
    if specificity is None:
        _check(lib().gs_format_guide(C.byref(gs), C.byref(k), hits.ctypes.data, hits.shape[0], mismatches,
                                     flags, max_off_targets, C.byref(out), C.byref(n)))
    else:
        _check(lib().gs_format_guide_scored(C.byref(gs), C.byref(k), hits.ctypes.data, hits.shape[0],
                                            mismatches, flags, max_off_targets, float(specificity),
                                            C.byref(out), C.byref(n)))
    s = C.string_at(out, n.value).decode()
    lib().gs_free(out)
    return s


def format_guides(gs, ids, seqs, pams, senses_positive, offsets, hits, specificity, mismatches, sam=False,
                  complete=True, start=False, max_off_targets=-1, skip=None) -> bytes:
    """the lines of a whole batch in one call (gs_format_guides_scored)"""
    n = len(ids)
    arr = (GsKmer * n)(*[GsKmer(ids[i].encode(), seqs[i].encode(), pams[i].encode(), int(senses_positive[i]))
                         for i in range(n)])
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    spec = np.ascontiguousarray(specificity, dtype=np.float32)
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    out, ln = C.c_void_p(), C.c_size_t()
    flags = ((GS_TEXT_SAM if sam else 0) | (GS_TEXT_COMPLETE if complete else 0) |
             (GS_FLAG_PAM_AT_START if start else 0))
    _check(lib().gs_format_guides_scored(C.byref(gs), arr, n, offsets.ctypes.data,
                                         hits.ctypes.data if hits.shape[0] else None, spec.ctypes.data,
                                         sk.ctypes.data if sk is not None else None, mismatches, flags,
                                         max_off_targets, C.byref(out), C.byref(ln)))
    s = C.string_at(out, ln.value)
    lib().gs_free(out)
    return s


class DeviceKmers:
    """Candidate guides of one chromosome, resident in HBM (gs_kmers_generate).  `seqs_ptr` /
    `pams_ptr` are raw device addresses in the layout GenomeIndex.enumerate_device takes."""

    def __init__(self, handle, k, P):
        self._h, self.k, self.P = handle, k, P
        n, a, b, c, d = C.c_uint64(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().gs_kmers_get(handle, 1, C.byref(n), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        self.n = int(n.value)
        self.seqs_ptr, self.pams_ptr, self.pos_ptr, self.sense_ptr = a.value, b.value, c.value, d.value

    def to_host(self):
        """-> (seqs uint8[n,k], pams uint8[n,P], positions uint32[n] 1-based, senses uint8[n])"""
        n, a, b, c, d = C.c_uint64(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().gs_kmers_get(self._h, 0, C.byref(n), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        m = int(n.value)
        if m == 0:
            return (np.empty((0, self.k), np.uint8), np.empty((0, self.P), np.uint8), np.empty(0, np.uint32),
                    np.empty(0, np.uint8))
        seqs = np.frombuffer(C.string_at(a, m * self.k), dtype=np.uint8).reshape(m, self.k).copy()
        pams = np.frombuffer(C.string_at(b, m * self.P), dtype=np.uint8).reshape(m, self.P).copy()
        pos = np.frombuffer(C.string_at(c, 4 * m), dtype=np.uint32).copy()
        sense = np.frombuffer(C.string_at(d, m), dtype=np.uint8).copy()
        return seqs, pams, pos, sense

    def close(self):
        if self._h:
            lib().gs_kmers_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_kmers(chrm, pam="NGG", k=20, start=False, device=0, chrm_device_ptr=None, chrm_len=None):
    """scripts/generate_kmers.py:70-118 for ONE chromosome on the GPU -> DeviceKmers.
    chrm: bytes / uint8 array (host), or pass chrm_device_ptr + chrm_len for text already in HBM."""
    h = C.c_void_p()
    flags = GS_FLAG_PAM_AT_START if start else 0
    if chrm_device_ptr is not None:
        _check(lib().gs_kmers_generate(device, chrm_device_ptr, int(chrm_len), 1, pam.encode(), k, flags, None,
                                       C.byref(h)))
    else:
        arr = np.frombuffer(bytes(chrm), dtype=np.uint8) if not isinstance(chrm, np.ndarray) else \
            np.ascontiguousarray(chrm, dtype=np.uint8)
        _check(lib().gs_kmers_generate(device, arr.ctypes.data if arr.size else None, arr.shape[0], 0,
                                       pam.encode(), k, flags, None, C.byref(h)))
    return DeviceKmers(h, k, len(pam))


def sdsl_extract_text(index_file) -> np.ndarray:
    """genome text held in a reference `.forward` / `.reverse` index file"""
    out, n = C.c_void_p(), C.c_uint64()
    _check(lib().gs_sdsl_extract_text(str(index_file).encode(), C.byref(out), C.byref(n)))
    t = np.frombuffer(C.string_at(out, n.value), dtype=np.uint8).copy()
    lib().gs_free(out)
    return t


def decode_sequence_ex(hit) -> str:
    """match.sequence of one HIT_EX_DTYPE record"""
    return bytes(hit["seq"])[:int(hit["seq_len"])].decode()


def format_guide_ex(gs, gid, sequence, pam, sense_positive, hits, mismatches, sam=False, complete=True,
                    start=False, max_off_targets=-1) -> str:
    """hits: numpy HIT_EX_DTYPE array of this guide (bulge path)"""
    hits = np.ascontiguousarray(hits, dtype=HIT_EX_DTYPE)
    k = GsKmer(gid.encode(), sequence.encode(), pam.encode(), int(sense_positive))
    out, n = C.c_void_p(), C.c_size_t()
    flags = ((GS_TEXT_SAM if sam else 0) | (GS_TEXT_COMPLETE if complete else 0) |
             (GS_FLAG_PAM_AT_START if start else 0))
    _check(lib().gs_format_guide_ex(C.byref(gs), C.byref(k), hits.ctypes.data, hits.shape[0], mismatches,
                                    flags, max_off_targets, C.byref(out), C.byref(n)))
    s = C.string_at(out, n.value).decode()
    lib().gs_free(out)
    return s


def decode_sequence(guide: str, P: int, key: int, flags: int = 0) -> str:
    buf = C.create_string_buffer(len(guide) + P + 1)
    _check(lib().gs_decode_sequence(guide.encode(), len(guide), P, flags, key, buf))
    return buf.value.decode()


class GenomeIndex:
    """Both strand indexes of one genome in one GPU's HBM.  Mirrors the pair of
    genome_index objects of src/guidescan.cxx:210-211."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device

    @classmethod
    def build(cls, text: np.ndarray, device: int = 0, sa_fwd=None, sa_rev=None):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        h = C.c_void_p()
        if sa_fwd is not None:
            sa_fwd = np.ascontiguousarray(sa_fwd, dtype=np.uint32)
            sa_rev = np.ascontiguousarray(sa_rev, dtype=np.uint32)
            _check(lib().gs_index_build_with_sa(text.ctypes.data, text.shape[0], sa_fwd.ctypes.data,
                                                sa_rev.ctypes.data, device, C.byref(h)))
        else:
            _check(lib().gs_index_build(text.ctypes.data, text.shape[0], device, C.byref(h)))
        return cls(h, device)

    def save_sa(self, text, path):
        """store both suffix arrays next to the text (gs_index_save_sa)"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        _check(lib().gs_index_save_sa(self._h, text.ctypes.data, text.shape[0], str(path).encode()))

    @classmethod
    def open_sa(cls, text, path, device: int = 0):
        """text + stored suffix arrays -> index without the suffix sort (gs_index_open_sa)"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        h = C.c_void_p()
        _check(lib().gs_index_open_sa(text.ctypes.data, text.shape[0], str(path).encode(), device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def open_sdsl(cls, prefix, device: int = 0):
        """import the reference's <prefix>.forward index file (sdsl::load_from_file replacement)"""
        h = C.c_void_p()
        _check(lib().gs_index_open_sdsl(str(prefix).encode(), device, C.byref(h)))
        return cls(h, device)

    def close(self):
        if self._h:
            lib().gs_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def genome_length(self):
        return int(lib().gs_index_genome_length(self._h))

    @property
    def device_bytes(self):
        return int(lib().gs_index_device_bytes(self._h))

    def meta(self, strand=0):
        c = (C.c_uint64 * 5)()
        n = C.c_uint64()
        _check(lib().gs_index_meta(self._h, strand, c, C.byref(n)))
        return list(c), int(n.value)

    def rank_bwt4(self, rows, strand=0):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty((rows.shape[0], 4), dtype=np.uint64)
        _check(lib().gs_rank_bwt4(self._h, strand, rows.ctypes.data, rows.shape[0], out.ctypes.data))
        return out

    def resolve(self, rows, strand=0):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty(rows.shape[0], dtype=np.uint64)
        _check(lib().gs_resolve(self._h, strand, rows.ctypes.data, rows.shape[0], out.ctypes.data))
        return out

    def suffix_array(self, strand=0):
        _, n = self.meta(strand)
        out = np.empty(n, dtype=np.uint32)
        _check(lib().gs_index_copy_sa(self._h, strand, out.ctypes.data))
        return out

    def verify_sa(self, text, strand=0, samples=1 << 20, seed=1):
        """self-check from the text alone -> dict of gs_sa_report (all counters but rows/sampled must be 0).
        samples="all": every adjacent pair of rows by the linear-time rule (GS_VERIFY_ALL_ROWS) - a complete proof"""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        rep = GsSaReport()
        if samples == "all":
            samples = (1 << 64) - 1
        _check(lib().gs_index_verify_sa(self._h, strand, text.ctypes.data, text.shape[0], samples, seed,
                                        C.byref(rep)))
        return {k: int(getattr(rep, k)) for k, _ in GsSaReport._fields_}

    def enumerate(self, seqs: np.ndarray, pams: np.ndarray, mismatches=3, alt_pams=(), start=False,
                  faithful=False, raw_counts=False, no_new_tables=False):
        """seqs uint8[n,L], pams uint8[n,P] -> (offsets uint64[n+1], hits HIT_DTYPE[], stats dict).
        Hits of guide i are hits[offsets[i]:offsets[i+1]] in the reference's canonical order."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        n, L = seqs.shape
        pams = np.ascontiguousarray(pams, dtype=np.uint8)
        P = pams.shape[1] if pams.ndim == 2 else 0
        pams = pams.reshape(n, P)
        alt = b"".join(p.encode() for p in alt_pams)
        for p in alt_pams:
            if len(p) != P:
                raise ValueError("alt PAM length differs from the guides' PAM length")
        r = C.c_void_p()
        flags = ((GS_FLAG_PAM_AT_START if start else 0) | (GS_FLAG_FAITHFUL_WALK if faithful else 0) |
                 (GS_FLAG_RAW_COUNTS if raw_counts else 0) | (GS_FLAG_NO_NEW_TABLES if no_new_tables else 0))
        _check(lib().gs_enumerate(self._h, seqs.ctypes.data, n, L, pams.ctypes.data if P else None, P,
                                  alt if alt_pams else None, len(alt_pams), mismatches, flags,
                                  C.byref(r)))
        try:
            v = GsResultView()
            _check(lib().gs_result_get(r, C.byref(v)))
            offsets = np.ctypeslib.as_array(v.guide_offsets, shape=(n + 1,)).copy()
            if v.n_hits:
                raw = C.string_at(C.cast(v.hits, C.c_void_p), int(v.n_hits) * 16)
                hits = np.frombuffer(raw, dtype=HIT_DTYPE).copy()
            else:
                hits = np.empty(0, dtype=HIT_DTYPE)
            stats = dict(n_ext=int(v.n_ext), n_matches=int(v.n_matches), n_hits=int(v.n_hits),
                         ms_search=float(v.ms_search), ms_total=float(v.ms_total),
                         raw_hits=(np.ctypeslib.as_array(v.raw_hits, shape=(n,)).copy() if raw_counts and n and v.raw_hits
                                   else None),
                         needs_general=(np.nonzero(np.ctypeslib.as_array(v.guide_flags, shape=(n,)) & 1)[0].tolist()
                                        if v.n_unsupported and n else []))
        finally:
            lib().gs_result_free(r)
        return offsets, hits, stats

    def score(self, gs, seqs, P, offsets, hits, sam=False, start=False, max_off_targets=-1, want_cfd=True):
        """CFD per hit and specificity per guide on the device (printer.hpp:98-113, 115-170, 251-297)
        -> (cfd float32[n_hits] or None, specificity float32[n])"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        n, L = seqs.shape
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        cfd = np.empty(hits.shape[0], dtype=np.float32) if want_cfd else None
        spec = np.empty(n, dtype=np.float32)
        flags = (GS_TEXT_SAM if sam else 0) | (GS_FLAG_PAM_AT_START if start else 0)
        _check(lib().gs_score(self._h, seqs.ctypes.data, n, L, P, flags, max_off_targets, C.byref(gs),
                              offsets.ctypes.data, hits.ctypes.data if hits.shape[0] else None,
                              cfd.ctypes.data if want_cfd and hits.shape[0] else None, spec.ctypes.data))
        return cfd, spec

    def score_device(self, gs, d_guides_ptr, n, L, P, d_offsets_ptr, d_hits_ptr, d_cfd_ptr, d_spec_ptr,
                     sam=False, start=False, max_off_targets=-1, stream=None):
        """device-resident variant: all pointers are raw device addresses"""
        flags = (GS_TEXT_SAM if sam else 0) | (GS_FLAG_PAM_AT_START if start else 0)
        _check(lib().gs_score_device(self._h, d_guides_ptr, n, L, P, flags, max_off_targets, C.byref(gs),
                                     d_offsets_ptr, d_hits_ptr, stream, d_cfd_ptr, d_spec_ptr))

    def enumerate_general(self, seqs, pams, mismatches=3, rna_bulges=0, dna_bulges=0, alt_pams=(),
                          start=False):
        """the general path (any symbol, any number of PAMs, bulges) -> (offsets uint64[n+1], hits HIT_EX_DTYPE[])"""
        return self.enumerate_bulges(seqs, pams, mismatches, rna_bulges, dna_bulges, alt_pams, start)

    def enumerate_bulges(self, seqs, pams, mismatches=3, rna_bulges=0, dna_bulges=0, alt_pams=(),
                         start=False):
        """bulge-aware search (index.hpp:250-375) -> (offsets uint64[n+1], hits HIT_EX_DTYPE[])"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        n, L = seqs.shape
        pams = np.ascontiguousarray(pams, dtype=np.uint8)
        P = pams.shape[1] if pams.ndim == 2 else 0
        pams = pams.reshape(n, P)
        alt = b"".join(p.encode() for p in alt_pams)
        r = C.c_void_p()
        _check(lib().gs_enumerate_bulges(self._h, seqs.ctypes.data, n, L, pams.ctypes.data if P else None, P,
                                         alt if alt_pams else None, len(alt_pams), mismatches, rna_bulges,
                                         dna_bulges, GS_FLAG_PAM_AT_START if start else 0, C.byref(r)))
        try:
            ng, po, ph = C.c_uint64(), C.c_void_p(), C.c_void_p()
            _check(lib().gs_result_ex_get(r, C.byref(ng), C.byref(po), C.byref(ph)))
            offsets = np.frombuffer(C.string_at(po, 8 * (n + 1)), dtype=np.uint64).copy()
            nh = int(offsets[-1])
            hits = (np.frombuffer(C.string_at(ph, 48 * nh), dtype=HIT_EX_DTYPE).copy() if nh
                    else np.empty(0, dtype=HIT_EX_DTYPE))
        finally:
            lib().gs_result_ex_free(r)
        return offsets, hits

    def locked(self):
        """context manager: hold the handle across several device-pointer calls (gs_index_lock / gs_index_unlock);
        calls on this handle from other threads wait meanwhile"""
        import contextlib

        @contextlib.contextmanager
        def hold():
            _check(lib().gs_index_lock(self._h))
            try:
                yield self
            finally:
                _check(lib().gs_index_unlock(self._h))
        return hold()

    def prepare(self, n, L=20, pam="NGG", alt_pams=(), mismatches=3, start=False):
        """the first batch's one-off work (seed recipes, PAM-pair and deep tables, workspace for n guides) ahead of the first job
        (gs_index_prepare)"""
        alts = "".join(alt_pams).encode()
        _check(lib().gs_index_prepare(self._h, int(n), L, pam.encode(), len(pam), alts if alt_pams else None, len(alt_pams), mismatches,
                                      GS_FLAG_PAM_AT_START if start else 0))

    def set_option(self, key, value):
        """a switch of this handle (gs_index_set_option): value None removes it.  The environment's GS_* variables are
        read once, when the handle is made; afterwards this is the only way to change one."""
        _check(lib().gs_index_set_option(self._h, key.encode(), None if value is None else str(value).encode()))

    def set_options(self, **kv):
        for k, v in kv.items():
            self.set_option(k, v)

    def get_option(self, key):
        buf = C.create_string_buffer(256)
        rc = lib().gs_index_get_option(self._h, key.encode(), buf, 256)
        return buf.value.decode() if rc == 0 else None

    def last_sharing(self):
        """heavy items shared among waves in the last search launch (gs_index_last_sharing); form: 0 one launch, every item
        with its wave; 1 one launch that publishes heavy passes and helps; 2 two launches (publishing + helpers); 3 the two
        seeding launches of gs_seed.hip (no sharing); guides_with_heavy_kmer: what the form was chosen from"""
        out = (C.c_uint64 * 8)()
        _check(lib().gs_index_last_sharing(self._h, out))
        return dict(shared_items=int(out[0]), packages=int(out[1]), queue_packages=int(out[2]), tickets=int(out[3]),
                    guides_ordered_device_wide_alone=int(out[4]), launches_behind=int(out[5]), form=int(out[6]),
                    guides_with_heavy_kmer=int(out[7]))

    def last_counters(self):
        """k_search's counters of the last enumerate_device call (see gs_index_last_counters)"""
        out = (C.c_uint64 * 16)()
        _check(lib().gs_index_last_counters(self._h, out))
        v = list(out)
        return dict(n_ext=v[0], overflow_items=v[1], n_matches=v[2], items_two_sided=v[4], items_one_sided=v[5],
                    guides_redone=v[6], ordered_device_wide=bool(v[7] & 1), redo_ordered_device_wide=bool(v[7] & 2),
                    overflow_from_arena=bool(v[7] & 4),
                    ordered_by_one_composite_sort=bool(v[7] & 8), runs_turned_round=bool(v[7] & 16),
                    ordered_in_tiles=bool(v[7] & 32), tile_ordering_gave_up=bool(v[7] & 64),
                    items_pair_tables=v[7] >> 8, recipe_lines=v[3],
                    slots_per_item=v[13], matches_sum=v[14], matches_max_per_item=v[15],
                    table_lines=v[8], ctx16_lines=v[9], ctx_words=v[10], sa_isa_gathers=v[11], occ_lines=v[12])

    def enumerate_device(self, d_guides_ptr, n, L, d_pams_ptr, P, mismatches=3, alt_pams=(),
                         start=False, stream=None, faithful=False, count_requests=False):
        """Device-resident variant (what bench.py times): pointers are raw device addresses.
        Returns (d_offsets_ptr, d_hits_ptr, stats)."""
        alt = b"".join(p.encode() for p in alt_pams)
        flags = ((GS_FLAG_PAM_AT_START if start else 0) | (GS_FLAG_FAITHFUL_WALK if faithful else 0) |
                 (GS_FLAG_COUNT_REQUESTS if count_requests else 0))
        d_off, d_hits = C.c_void_p(), C.c_void_p()
        v = GsResultView()
        _check(lib().gs_enumerate_device(self._h, d_guides_ptr, n, L, d_pams_ptr, P,
                                         alt if alt_pams else None, len(alt_pams), mismatches, flags,
                                         stream, C.byref(d_off), C.byref(d_hits), C.byref(v)))
        stats = dict(n_ext=int(v.n_ext), n_matches=int(v.n_matches), n_hits=int(v.n_hits),
                     ms_search=float(v.ms_search), ms_total=float(v.ms_total))
        return d_off.value, d_hits.value, stats
