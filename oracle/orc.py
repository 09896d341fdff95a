"""ctypes binding of oracle/liborc.so.

ORACLE — TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under colorid_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build():
    """Compile the C restatement (gcc); no-op when liborc.so is newer than its sources."""
    if os.environ.get("ORC_LIB"):  # e.g. liborc_asan.so (make -C oracle asan) under LD_PRELOAD=libasan
        return os.environ["ORC_LIB"]
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_xxh3.c", "orc_colorid.c", "orc.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["make", "-C", _HERE, "-s", "liborc.so"])
    return so


class _Index(C.Structure):
    _fields_ = [("bloom_size", C.c_uint64), ("num_hash", C.c_uint64), ("k_size", C.c_uint64),
                ("n_colors", C.c_uint64), ("m_size", C.c_uint64), ("w32", C.c_uint32), ("rows", u32p),
                ("colors", C.POINTER(C.c_char_p)), ("n_ref_kmers", u64p)]


class _StrVec(C.Structure):
    _fields_ = [("n", C.c_size_t), ("s", C.POINTER(C.c_char_p)), ("len", C.POINTER(C.c_size_t))]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp = C.c_void_p
    ip = C.POINTER(_Index)
    sig = {
        "orc_xxh3_64_with_seed": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]),
        "orc_xxh3_published_64_with_seed": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]),
        "orc_xxh3_v07_64_with_seed": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]),
        "orc_set_hash_variant": (None, [C.c_int]),
        "orc_get_hash_variant": (C.c_int, []),
        "orc_is_good_base": (C.c_int, [C.c_uint8]),
        "orc_has_no_n": (C.c_int, [C.c_char_p, C.c_size_t]),
        "orc_qual_mask": (None, [C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint8, C.c_char_p]),
        "orc_revcomp": (None, [C.c_char_p, C.c_size_t, C.c_char_p]),
        "orc_kmers_new": (vp, [C.c_uint32]),
        "orc_kmers_free": (None, [vp]),
        "orc_kmers_len": (C.c_uint64, [vp]),
        "orc_kmers_keys": (u8p, [vp]),
        "orc_kmers_counts": (u64p, [vp]),
        "orc_kmers_insert": (None, [vp, C.c_char_p, C.c_uint64]),
        "orc_kmerize_vector": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.c_size_t]),
        "orc_kmerize_string": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
        "orc_kmerize_skip_n_set": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.c_size_t]),
        "orc_kmerize_fq_read": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint8]),
        "orc_clean_map": (vp, [vp, C.c_uint64]),
        "orc_auto_cutoff": (C.c_int64, [vp]),
        "orc_strvec_free": (None, [C.POINTER(_StrVec)]),
        "orc_read_fasta": (C.c_int, [C.c_char_p, C.POINTER(_StrVec)]),
        "orc_read_fasta_mf": (C.c_int, [C.c_char_p, C.POINTER(_StrVec), C.POINTER(_StrVec)]),
        "orc_kmers_from_fq_qual": (vp, [C.c_char_p, C.c_uint32, C.c_uint8]),
        "orc_kmers_fq_pe_qual": (vp, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint8]),
        "orc_index_new": (ip, [C.c_uint64] * 4),
        "orc_index_free": (None, [ip]),
        "orc_index_set_color": (None, [ip, C.c_uint64, C.c_char_p, C.c_uint64]),
        "orc_index_insert": (None, [ip, C.c_uint64, C.c_char_p]),
        "orc_index_contains": (C.c_int, [ip, C.c_uint64, C.c_char_p]),
        "orc_build_single": (ip, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8, C.c_int64]),
        "orc_build_single_mini": (ip, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8, C.c_int64]),
        "orc_find_minimizer": (None, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p]),
        "orc_minimerize_skip_n_set": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]),
        "orc_save_bigsi": (C.c_int, [C.c_char_p, ip]),
        "orc_read_bigsi": (ip, [C.c_char_p]),
        "orc_search_count": (None, [ip, vp, vp, C.c_uint64, vp, vp, vp, vp]),
        "orc_search_count_mt": (None, [ip, vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp]),
        "orc_sparse_build": (vp, [ip]),
        "orc_sparse_free": (None, [vp]),
        "orc_search_count_sparse": (None, [vp, ip, vp, vp, C.c_uint64, vp, vp, vp]),
        "orc_search_perfect": (None, [ip, vp, C.c_uint64, vp, C.POINTER(C.c_int)]),
        "orc_search_index_classic": (None, [ip, vp, C.c_uint64, vp]),
        "orc_search_index": (None, [ip, vp, C.c_uint64, C.c_uint64, vp]),
        "orc_readid_counts": (None, [ip, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp]),
        "orc_readid_counts_mt": (None, [ip, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, vp, vp, vp]),
        "orc_false_prob": (C.c_double, [C.c_double] * 3),
        "orc_not_fp_significant": (C.c_int, [C.c_uint64, C.c_double, C.c_double, C.c_uint64]),
        "orc_kmer_poll_plus": (C.c_int, [ip, vp, C.c_uint64, C.c_double, C.c_char_p, C.c_size_t,
                                         u64p, C.POINTER(C.c_int), u64p]),
        "orc_generate_report": (C.c_size_t, [ip, C.c_char_p, vp, vp, vp, vp, C.c_uint64, C.c_double,
                                             C.c_char_p, C.c_size_t]),
        "orc_generate_report_gene": (C.c_size_t, [ip, C.c_char_p, vp, C.c_uint64, C.c_double,
                                                  C.c_char_p, C.c_size_t]),
        "orc_unique_modes": (None, [vp, vp, C.c_uint64, C.c_uint64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def xxh3(b: bytes, seed: int = 0) -> int:
    return lib().orc_xxh3_64_with_seed(b, len(b), seed)


def xxh3_v07(b: bytes, seed: int = 0) -> int:
    return lib().orc_xxh3_v07_64_with_seed(b, len(b), seed)


class hash_variant:
    """`with orc.hash_variant(1): ...` — every oracle hash inside uses the v0.7 draft (candidate for crate xxh3 0.1.x)."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = lib().orc_get_hash_variant()
        lib().orc_set_hash_variant(self.v)

    def __exit__(self, *a):
        lib().orc_set_hash_variant(self.old)


def revcomp(b: bytes) -> bytes:
    out = C.create_string_buffer(len(b))
    lib().orc_revcomp(b, len(b), out)
    return out.raw


def qual_mask(seq: bytes, qual: bytes, q: int) -> bytes:
    out = C.create_string_buffer(len(seq))
    lib().orc_qual_mask(seq, qual, len(seq), q, out)
    return out.raw


class Kmers:
    """FnvHashMap<String,usize> / FnvHashSet<String> stand-in (first-occurrence iteration order)."""

    def __init__(self, k, handle=None):
        self.k = k
        self.h = handle if handle is not None else lib().orc_kmers_new(k)

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.orc_kmers_free(self.h)
            self.h = None

    def __len__(self):
        return lib().orc_kmers_len(self.h)

    def keys(self) -> np.ndarray:
        n = len(self)
        if n == 0:
            return np.zeros((0, self.k), np.uint8)
        return np.ctypeslib.as_array(lib().orc_kmers_keys(self.h), shape=(n * self.k,)).reshape(n, self.k).copy()

    def counts(self) -> np.ndarray:
        n = len(self)
        if n == 0:
            return np.zeros((0,), np.uint64)
        return np.ctypeslib.as_array(lib().orc_kmers_counts(self.h), shape=(n,)).copy()

    def as_dict(self):
        ks, cs = self.keys(), self.counts()
        return {ks[i].tobytes(): int(cs[i]) for i in range(len(cs))}

    def kmerize_vector(self, seq: bytes, d=1):
        return lib().orc_kmerize_vector(self.h, seq, len(seq), d)

    def kmerize_string(self, seq: bytes):
        return lib().orc_kmerize_string(self.h, seq, len(seq))

    def kmerize_skip_n_set(self, seq: bytes, d=1):
        return lib().orc_kmerize_skip_n_set(self.h, seq, len(seq), d)

    def kmerize_fq_read(self, seq: bytes, qual: bytes, q=15):
        return lib().orc_kmerize_fq_read(self.h, seq, qual, min(len(seq), len(qual)), q)

    def clean_map(self, t):
        return Kmers(self.k, lib().orc_clean_map(self.h, t))

    def auto_cutoff(self):
        return lib().orc_auto_cutoff(self.h)


def read_fasta(path):
    v = _StrVec()
    if lib().orc_read_fasta(path.encode(), C.byref(v)) != 0:
        raise IOError(path)
    out = [C.string_at(v.s[i], v.len[i]) for i in range(v.n)]
    lib().orc_strvec_free(C.byref(v))
    return out


def read_fasta_mf(path):
    lv, sv = _StrVec(), _StrVec()
    if lib().orc_read_fasta_mf(path.encode(), C.byref(lv), C.byref(sv)) != 0:
        raise IOError(path)
    labels = [C.string_at(lv.s[i], lv.len[i]) for i in range(lv.n)]
    seqs = [C.string_at(sv.s[i], sv.len[i]) for i in range(sv.n)]
    lib().orc_strvec_free(C.byref(lv))
    lib().orc_strvec_free(C.byref(sv))
    return labels, seqs


def kmers_from_fq_qual(path, k, q=15):
    h = lib().orc_kmers_from_fq_qual(path.encode(), k, q)
    if not h:
        raise IOError(path)
    return Kmers(k, h)


def kmers_fq_pe_qual(p1, p2, k, q=15):
    h = lib().orc_kmers_fq_pe_qual(p1.encode(), p2.encode(), k, q)
    if not h:
        raise IOError(p1)
    return Kmers(k, h)


class Index:
    """BigsyMapNew (bigsi.rs:19-27) held densely."""

    def __init__(self, m=None, n_hash=None, k=None, n_colors=None, handle=None):
        self.p = handle if handle is not None else lib().orc_index_new(m, n_hash, k, n_colors)
        if not self.p:
            raise RuntimeError("orc_index: NULL")

    def __del__(self):
        if getattr(self, "p", None) and _LIB is not None:
            _LIB.orc_index_free(self.p)
            self.p = None

    m = property(lambda s: s.p.contents.bloom_size)
    n_hash = property(lambda s: s.p.contents.num_hash)
    k = property(lambda s: s.p.contents.k_size)
    n_colors = property(lambda s: s.p.contents.n_colors)
    m_size = property(lambda s: s.p.contents.m_size)

    def set_minimizer(self, m_size):
        """turn the index into a BigsyMapMiniNew: keys are minimizers of length m_size (call before inserting)"""
        self.p.contents.m_size = m_size
    w32 = property(lambda s: s.p.contents.w32)

    def rows(self) -> np.ndarray:
        """Writable view m x w32 (uint32) of the dense BitVec storage."""
        c = self.p.contents
        return np.ctypeslib.as_array(c.rows, shape=(c.bloom_size * max(c.w32, 1),)).reshape(c.bloom_size, max(c.w32, 1))

    def colors(self):
        c = self.p.contents
        return [c.colors[i].decode() if c.colors[i] else "" for i in range(c.n_colors)]

    def n_ref_kmers(self):
        c = self.p.contents
        return [int(c.n_ref_kmers[i]) for i in range(c.n_colors)]

    def set_color(self, cid, name, n_ref):
        lib().orc_index_set_color(self.p, cid, name.encode(), n_ref)

    def insert(self, cid, kmer: bytes):
        lib().orc_index_insert(self.p, cid, kmer)

    def contains(self, cid, kmer: bytes):
        return bool(lib().orc_index_contains(self.p, cid, kmer))

    def save(self, path):
        if lib().orc_save_bigsi(path.encode(), self.p) != 0:
            raise IOError(path)

    @staticmethod
    def read(path):
        p = lib().orc_read_bigsi(path.encode())
        if not p:
            raise IOError(path)
        return Index(handle=p)

    @staticmethod
    def build_single_mini(ref_tsv, m, n_hash, k, m_size, quality=15, cutoff=-1):
        p = lib().orc_build_single_mini(ref_tsv.encode(), m, n_hash, k, m_size, quality, cutoff)
        if not p:
            raise IOError(ref_tsv)
        return Index(handle=p)

    @staticmethod
    def build_single(ref_tsv, m, n_hash, k, quality=15, cutoff=-1):
        p = lib().orc_build_single(ref_tsv.encode(), m, n_hash, k, quality, cutoff)
        if not p:
            raise IOError(ref_tsv)
        return Index(handle=p)

    # ---- search loops
    def search_count(self, kmers: np.ndarray, freq=None, want_unique=True):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        K = kmers.shape[0]
        Cn = self.n_colors
        freq64 = None if freq is None else np.ascontiguousarray(freq, np.uint64)
        hits = np.zeros(Cn, np.uint64)
        nu = np.zeros(Cn, np.uint64)
        sf = np.zeros(Cn, np.uint64)
        uc = np.zeros(K, np.uint32) if want_unique else None
        lib().orc_search_count(self.p, _ptr(kmers), _ptr(freq64), K, _ptr(hits), _ptr(nu), _ptr(sf), _ptr(uc))
        return hits, nu, sf, uc

    def search_count_mt(self, kmers: np.ndarray, freq, n_threads):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        K = kmers.shape[0]
        freq64 = None if freq is None else np.ascontiguousarray(freq, np.uint64)
        hits, nu, sf = (np.zeros(self.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(K, np.uint32)
        lib().orc_search_count_mt(self.p, _ptr(kmers), _ptr(freq64), K, n_threads, _ptr(hits), _ptr(nu), _ptr(sf), _ptr(uc))
        return hits, nu, sf, uc

    def sparse_map(self):
        """the reference's data structure (FNV-hashed map row -> bit vector) over this index; free with orc.sparse_free"""
        return lib().orc_sparse_build(self.p)

    def search_count_sparse(self, sparse, kmers: np.ndarray, freq):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        freq64 = None if freq is None else np.ascontiguousarray(freq, np.uint64)
        hits, nu, sf = (np.zeros(self.n_colors, np.uint64) for _ in range(3))
        lib().orc_search_count_sparse(sparse, self.p, _ptr(kmers), _ptr(freq64), kmers.shape[0], _ptr(hits), _ptr(nu), _ptr(sf))
        return hits, nu, sf

    def search_perfect(self, kmers: np.ndarray):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        words = np.zeros(max(self.w32, 1), np.uint32)
        missing = C.c_int(0)
        lib().orc_search_perfect(self.p, _ptr(kmers), kmers.shape[0], _ptr(words), C.byref(missing))
        return words, bool(missing.value)

    def search_index_classic(self, kmers: np.ndarray):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        rep = np.zeros(self.n_colors + 1, np.uint64)
        lib().orc_search_index_classic(self.p, _ptr(kmers), kmers.shape[0], _ptr(rep))
        return rep

    def search_index(self, kmers: np.ndarray, start_sample):
        kmers = np.ascontiguousarray(kmers, np.uint8)
        rep = np.zeros(self.n_colors + 1, np.uint64)
        lib().orc_search_index(self.p, _ptr(kmers), kmers.shape[0], start_sample, _ptr(rep))
        return rep

    def readid_counts(self, bases, seq_off, read_seq0, d=1, start_sample=3, n_threads=1):
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        n_reads = len(read_seq0) - 1
        rep = np.zeros((n_reads, self.n_colors + 1), np.uint32)
        nk = np.zeros(n_reads, np.uint32)
        st = np.zeros(n_reads, np.uint8)
        if n_threads > 1:   # the reference's rayon pool over the batch
            lib().orc_readid_counts_mt(self.p, _ptr(bases), _ptr(seq_off), _ptr(read_seq0), n_reads, d, start_sample, n_threads,
                                       _ptr(rep), _ptr(nk), _ptr(st))
        else:
            lib().orc_readid_counts(self.p, _ptr(bases), _ptr(seq_off), _ptr(read_seq0), n_reads, d, start_sample,
                                    _ptr(rep), _ptr(nk), _ptr(st))
        return rep, nk, st

    def kmer_poll_plus(self, report, kmer_length, fp_correct=1e-3):
        report = np.ascontiguousarray(report, np.uint64)
        buf = C.create_string_buffer(1 << 16)
        cnt, ntop, acc = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        lib().orc_kmer_poll_plus(self.p, _ptr(report), kmer_length, fp_correct, buf, len(buf),
                                 C.byref(cnt), C.byref(acc), C.byref(ntop))
        return buf.value.decode(), cnt.value, kmer_length, "accept" if acc.value else "reject", ntop.value

    def generate_report(self, query, hits, nu, sf, modes, num_kmers, cov=0.35):
        buf = C.create_string_buffer(1 << 20)
        n = lib().orc_generate_report(self.p, query.encode(), _ptr(hits), _ptr(nu), _ptr(sf), _ptr(modes),
                                      num_kmers, cov, buf, len(buf))
        return buf.raw[:n].decode()

    def generate_report_gene(self, query, hits, num_kmers, cov=0.35):
        buf = C.create_string_buffer(1 << 20)
        n = lib().orc_generate_report_gene(self.p, query.encode(), _ptr(hits), num_kmers, cov, buf, len(buf))
        return buf.raw[:n].decode()


def unique_modes(unique_colour, freq, n_colors):
    uc = np.ascontiguousarray(unique_colour, np.uint32)
    f = None if freq is None else np.ascontiguousarray(freq, np.uint64)
    modes = np.zeros(n_colors, np.uint64)
    lib().orc_unique_modes(_ptr(uc), _ptr(f), len(uc), n_colors, _ptr(modes))
    return modes


def find_minimizer(kmer: bytes, m: int) -> bytes:
    out = C.create_string_buffer(m)
    lib().orc_find_minimizer(kmer, len(kmer), m, out)
    return out.raw


def minimizer_set(seqs, k, m, d=1):
    """kmer.rs:363-394 over a read's mates -> Kmers of width m (first-occurrence order)"""
    km = Kmers(m)
    for s in seqs:
        lib().orc_minimerize_skip_n_set(km.h, s, len(s), k, d)
    return km


def sparse_free(sparse):
    lib().orc_sparse_free(sparse)


def false_prob(m, k, n):
    return lib().orc_false_prob(float(m), float(k), float(n))
