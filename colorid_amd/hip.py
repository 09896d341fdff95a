"""Thin object wrappers over the C ABI (host-pointer calls take numpy arrays, *_dev calls take ints)."""
import ctypes as C
import weakref

import numpy as np

from ._lib import check, load_library, vp

NOT_UNIQUE = 0xFFFFFFFF


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


class Context:
    def __init__(self, device_id=0, lib=None):
        self.lib = lib if lib is not None else load_library()   # lib: another build of the library (_lib.open_library), A/B tests
        h = vp()
        check(self.lib.cid_ctx_create(device_id, C.byref(h)))
        self.h = h
        self.device_id = device_id
        self.stream = None                   # a borrowed hipStream_t (set_stream), or None = the ctx's own stream
        self._children = weakref.WeakSet()   # indices / k-mer sets made from this ctx: they borrow its scratch, so they go first

    def set_stream(self, hip_stream):
        check(self.lib.cid_ctx_set_stream(self.h, vp(hip_stream) if hip_stream else None))
        self.stream = hip_stream or None

    def synchronize(self):
        check(self.lib.cid_ctx_synchronize(self.h))

    def tune(self, name, value):
        """cid_ctx_tune: a measurement / test switch of this context (include/colorid_hip.h)"""
        rc = self.lib.cid_ctx_tune(self.h, name.encode() if isinstance(name, str) else name, int(value))
        if rc != 0:
            from ._lib import CidError
            raise CidError(rc, self.lib.cid_last_error().decode(errors="replace"))

    def timer_start(self):
        check(self.lib.cid_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float(0)
        check(self.lib.cid_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.close()
            self.lib.cid_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Index:
    """Device-resident BIGSI matrix (BigsyMapNew.map, src/bigsi.rs:19-27)."""

    def __init__(self, ctx, bloom_size, num_hash, k_size, n_colors, hash_variant=0):
        self.ctx, self.lib = ctx, ctx.lib
        self.m, self.n_hash, self.k, self.n_colors = bloom_size, num_hash, k_size, n_colors
        self.w32 = (n_colors + 31) // 32
        self.m_size = 0
        h = vp()
        check(self.lib.cid_index_create(ctx.h, bloom_size, num_hash, k_size, n_colors, hash_variant, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def set_minimizer(self, m_size):
        check(self.lib.cid_index_set_minimizer(self.h, m_size))
        self.m_size = m_size
        return self

    def set_hash_variant(self, hash_variant):
        check(self.lib.cid_index_set_hash_variant(self.h, hash_variant))
        return self

    def put_rows(self, row_ids, words):
        row_ids = np.ascontiguousarray(row_ids, np.uint64)
        words = np.ascontiguousarray(words, np.uint32).reshape(len(row_ids), self.w32)
        check(self.lib.cid_index_put_rows(self.h, _p(row_ids), _p(words), len(row_ids)))

    def put_records(self, records: bytes):
        """rows as raw .bxi records ({u64 row; u64 W32; W32 x u32; u64 n_bits} each), parsed on the device"""
        rec = 24 + 4 * self.w32
        assert len(records) % rec == 0
        buf = np.frombuffer(records, np.uint8)
        check(self.lib.cid_index_put_records(self.h, _p(buf), len(records) // rec))

    def put_dense(self, rows_u32):
        """rows_u32: bloom_size x w32 dense BitVec storage; only non-zero rows are sent (as a .bxi holds them)."""
        rows_u32 = np.ascontiguousarray(rows_u32, np.uint32).reshape(self.m, self.w32)
        nz = np.flatnonzero(rows_u32.any(axis=1)).astype(np.uint64)
        self.put_rows(nz, rows_u32[nz])

    def device_matrix(self):
        ptr, rs = vp(), C.c_uint64(0)
        check(self.lib.cid_index_device_matrix(self.h, C.byref(ptr), C.byref(rs)))
        return ptr.value, rs.value

    def finalize(self):
        check(self.lib.cid_index_finalize(self.h))
        return self

    def get_rows(self, row_ids):
        row_ids = np.ascontiguousarray(row_ids, np.uint64)
        out = np.zeros((len(row_ids), self.w32), np.uint32)
        check(self.lib.cid_index_get_rows(self.h, _p(row_ids), _p(out), len(row_ids)))
        return out

    def get_records(self, row_begin, n_rows) -> bytes:
        """the non-zero rows of [row_begin, row_begin + n_rows) as raw .bxi records"""
        rec = 24 + 4 * self.w32
        buf = np.empty(max(1, n_rows * rec), np.uint8)
        n = C.c_uint64(0)
        check(self.lib.cid_index_get_records(self.h, row_begin, n_rows, _p(buf), C.byref(n)))
        return buf[:n.value * rec].tobytes()

    def insert_kmers_dev(self, d_kmers, d_colour_of_kmer, n_kmers):
        check(self.lib.cid_index_insert_kmers_dev(self.h, vp(d_kmers), vp(d_colour_of_kmer), n_kmers))

    # ---- a5
    def search_count(self, kmers, freq=None, want_unique=True, want_unique_colour=True):
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, self.k)
        K = kmers.shape[0]
        f = None if freq is None else np.ascontiguousarray(freq, np.uint32)
        hits = np.zeros(self.n_colors, np.uint64)
        nu = np.zeros(self.n_colors, np.uint64) if want_unique else None
        sf = np.zeros(self.n_colors, np.uint64) if want_unique else None
        uc = np.zeros(K, np.uint32) if want_unique_colour else None
        check(self.lib.cid_search_count(self.ctx.h, self.h, _p(kmers), _p(f), K, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_dev(self, d_kmers, d_freq, n_kmers, d_hits, d_n_unique=None, d_sum=None, d_uc=None):
        check(self.lib.cid_search_count_dev(self.ctx.h, self.h, vp(d_kmers), vp(d_freq) if d_freq else None, n_kmers,
                                            vp(d_hits), vp(d_n_unique) if d_n_unique else None,
                                            vp(d_sum) if d_sum else None, vp(d_uc) if d_uc else None))

    # ---- a4
    def search_perfect(self, kmers):
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, self.k)
        words = np.zeros(self.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_search_perfect(self.ctx.h, self.h, _p(kmers), kmers.shape[0], _p(words), C.byref(missing)))
        return words, bool(missing.value)

    # ---- a6/a7/a9/a10
    def readid_count(self, bases, seq_off, read_seq0, d=1, start_sample=3):
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        n_reads = len(read_seq0) - 1
        rep = np.zeros((n_reads, self.n_colors + 1), np.uint32)
        nk = np.zeros(n_reads, np.uint32)
        st = np.zeros(n_reads, np.uint8)
        check(self.lib.cid_readid_count(self.ctx.h, self.h, _p(bases), _p(seq_off), len(seq_off) - 1, _p(read_seq0),
                                        n_reads, d, start_sample, _p(rep), _p(nk), _p(st)))
        return rep, nk, st

    def readid_count_sparse(self, bases, seq_off, read_seq0, d=1, start_sample=3):
        """-> (row_start u64[n_reads+1], colours u32[E], counts u32[E], n_kmers, status)"""
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        n_reads = len(read_seq0) - 1
        nk = np.zeros(n_reads, np.uint32)
        st = np.zeros(n_reads, np.uint8)
        ne = C.c_uint64(0)
        check(self.lib.cid_readid_count_sparse(self.ctx.h, self.h, _p(bases), _p(seq_off), len(seq_off) - 1, _p(read_seq0), n_reads, d,
                                               start_sample, _p(nk), _p(st), C.byref(ne)))
        rs = np.zeros(n_reads + 1, np.uint64)
        col = np.zeros(ne.value, np.uint32)
        cnt = np.zeros(ne.value, np.uint32)
        check(self.lib.cid_readid_sparse_fetch(self.ctx.h, _p(rs), _p(col), _p(cnt)))
        return rs, col, cnt, nk, st

    def readid_count_dev(self, d_bases, d_seq_off, d_read_seq0, n_reads, d, start_sample, max_read_bytes, max_read_windows,
                         d_report, d_nk, d_status):
        check(self.lib.cid_readid_count_dev(self.ctx.h, self.h, vp(d_bases), vp(d_seq_off), vp(d_read_seq0), n_reads, d,
                                            start_sample, max_read_bytes, max_read_windows, vp(d_report), vp(d_nk), vp(d_status)))

    def readid_count_resident(self, d_bases, seq_off, read_seq0, d, start_sample, d_report, d_nk, d_status):
        """bases in HBM, offsets (numpy u64) on the host: reads of any length (long reads take the long-read path)"""
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        check(self.lib.cid_readid_count_resident(self.ctx.h, self.h, vp(d_bases), _p(seq_off), len(seq_off) - 1, _p(read_seq0),
                                                 len(read_seq0) - 1, d, start_sample, vp(d_report), vp(d_nk), vp(d_status)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cid_index_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class KmerSet:
    """Device-resident distinct canonical k-mers with multiplicities (cid_kmerset, k <= 32)."""

    MODE_FASTA, MODE_FASTQ = 0, 1

    def __init__(self, ctx, k):
        self.ctx, self.lib, self.k = ctx, ctx.lib, k
        h = vp()
        check(self.lib.cid_kmerset_create(ctx.h, k, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def add_seqs(self, seqs, mode=0):
        """seqs: list of bytes"""
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        bases = np.frombuffer(b"".join(seqs), np.uint8) if seqs else np.zeros(0, np.uint8)
        check(self.lib.cid_kmerset_add_seqs(self.h, _p(bases), _p(off), len(seqs), mode))

    def add_seqs_dev(self, d_bases, d_seq_off, n_seqs, max_len, mode=0):
        """reads already in HBM: device pointers to the bases and to n_seqs + 1 offsets into them"""
        check(self.lib.cid_kmerset_add_seqs_dev(self.h, vp(d_bases), vp(d_seq_off), n_seqs, max_len, mode))

    def finalize(self):
        n = C.c_uint64(0)
        check(self.lib.cid_kmerset_finalize(self.h, C.byref(n)))
        return n.value

    def __len__(self):
        n = C.c_uint64(0)
        check(self.lib.cid_kmerset_size(self.h, C.byref(n)))
        return n.value

    def histogram(self):
        nb = C.c_size_t(0)
        check(self.lib.cid_kmerset_count_histogram(self.h, None, None, 0, C.byref(nb)))
        vals = np.zeros(nb.value, np.uint32)
        cnts = np.zeros(nb.value, np.uint64)
        check(self.lib.cid_kmerset_count_histogram(self.h, _p(vals), _p(cnts), nb.value, C.byref(nb)))
        return vals, cnts

    def clean(self, t):
        check(self.lib.cid_kmerset_clean(self.h, t))

    def set_target_index(self, index):
        """before the first add_seqs: the set comes out ordered by (first row in `index`, code) — same contents, cheaper search"""
        check(self.lib.cid_kmerset_set_target_index(self.h, index.h))

    def order_for_index(self, index):
        check(self.lib.cid_kmerset_order_for_index(self.h, index.h))

    def device_ascii(self):
        """k > 32: (device pointer to the n x k ASCII k-mers, device pointer to the multiplicities, n)"""
        a, c, n = vp(), vp(), C.c_uint64(0)
        check(self.lib.cid_kmerset_device_ascii(self.h, C.byref(a), C.byref(c), C.byref(n)))
        return a.value, c.value, n.value

    def download(self):
        n = len(self)
        km = np.zeros((n, self.k), np.uint8)
        cnt = np.zeros(n, np.uint32)
        check(self.lib.cid_kmerset_download(self.h, _p(km), _p(cnt)))
        return km, cnt

    def as_dict(self):
        km, cnt = self.download()
        return {km[i].tobytes(): int(cnt[i]) for i in range(len(cnt))}

    def search_count(self, index):
        n = len(self)
        hits = np.zeros(index.n_colors, np.uint64)
        nu = np.zeros(index.n_colors, np.uint64)
        sf = np.zeros(index.n_colors, np.uint64)
        uc = np.zeros(n, np.uint32)
        check(self.lib.cid_search_count_set(self.ctx.h, index.h, self.h, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_report(self, index):
        """-> (hits, n_unique, sum_unique_freq, mode_unique_freq): what generate_report needs, nothing per k-mer"""
        out = [np.zeros(index.n_colors, np.uint64) for _ in range(4)]
        check(self.lib.cid_search_count_set_report(self.ctx.h, index.h, self.h, *[_p(o) for o in out]))
        return tuple(out)

    def search_perfect(self, index):
        words = np.zeros(index.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_search_perfect_set(self.ctx.h, index.h, self.h, _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cid_kmerset_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class _BorrowedContext(Context):
    """A rank's context owned by a Group (not destroyed on close)."""

    def __init__(self, lib, h, device_id):
        self.lib, self.h, self.device_id, self.stream = lib, h, device_id, None
        self._children = weakref.WeakSet()

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.close()
            self.h = None

    def __del__(self):
        self.close()


class Group:
    """Several GPUs of one node (cid_group): reads / k-mers sharded over the ranks, the index replicated, per-accession
    counters all-reduced (RCCL over xGMI; through the host when a device id repeats)."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = vp()
        check(self.lib.cid_group_create(ids, len(device_ids), C.byref(h)))
        self.h = h
        self.n = len(device_ids)
        self.ctxs = []
        for r in range(self.n):
            ch = vp()
            check(self.lib.cid_group_ctx(self.h, r, C.byref(ch)))
            self.ctxs.append(_BorrowedContext(self.lib, ch, device_ids[r]))
        self._replica_handles = None
        self._src = None

    @property
    def uses_rccl(self):
        y = C.c_int(0)
        check(self.lib.cid_group_uses_rccl(self.h, C.byref(y)))
        return bool(y.value)

    def replicate(self, index):
        """index: a finalized Index made from self.ctxs[r] for some r (normally 0)"""
        arr = (vp * self.n)()
        check(self.lib.cid_group_replicate_index(self.h, index.h, arr))
        self._replica_handles, self._src = arr, index
        return self

    def _idx(self):
        assert self._replica_handles is not None, "call replicate() first"
        return self._replica_handles, self._src

    def search_count(self, kmers, freq=None):
        arr, ix = self._idx()
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, ix.k)
        K = kmers.shape[0]
        f = None if freq is None else np.ascontiguousarray(freq, np.uint32)
        hits, nu, sf = (np.zeros(ix.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(K, np.uint32)
        check(self.lib.cid_group_search_count(self.h, arr, _p(kmers), _p(f), K, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_set(self, kmerset):
        arr, ix = self._idx()
        hits, nu, sf = (np.zeros(ix.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(len(kmerset), np.uint32)
        check(self.lib.cid_group_search_count_set(self.h, arr, kmerset.h, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_perfect(self, kmers):
        arr, ix = self._idx()
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, ix.k)
        words = np.zeros(ix.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_group_search_perfect(self.h, arr, _p(kmers), kmers.shape[0], _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def search_perfect_set(self, kmerset):
        arr, ix = self._idx()
        words = np.zeros(ix.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_group_search_perfect_set(self.h, arr, kmerset.h, _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def readid_count_sparse(self, bases, seq_off, read_seq0, d=1, start_sample=3):
        arr, _ = self._idx()
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        n_reads = len(read_seq0) - 1
        nk = np.zeros(n_reads, np.uint32)
        st = np.zeros(n_reads, np.uint8)
        ne = C.c_uint64(0)
        check(self.lib.cid_group_readid_count_sparse(self.h, arr, _p(bases), _p(seq_off), len(seq_off) - 1, _p(read_seq0), n_reads, d,
                                                     start_sample, _p(nk), _p(st), C.byref(ne)))
        rs = np.zeros(n_reads + 1, np.uint64)
        col = np.zeros(ne.value, np.uint32)
        cnt = np.zeros(ne.value, np.uint32)
        check(self.lib.cid_group_readid_sparse_fetch(self.h, _p(rs), _p(col), _p(cnt)))
        return rs, col, cnt, nk, st

    def kmerset(self, k):
        """A k-mer set counted over all ranks (cid_group_kmerset)."""
        return GroupKmerSet(self, k)

    def stripes(self, m, n_hash, k, n_colors_total, hash_variant=0):
        """A colour-striped index over the ranks (cid_group_stripes_*): rank r holds colours [base[r], base[r+1])."""
        return GroupStripes(self, m, n_hash, k, n_colors_total, hash_variant)

    def close(self):
        if getattr(self, "h", None):
            if self._replica_handles is not None:
                for r in range(self.n):
                    h = self._replica_handles[r]
                    if h and h != self._src.h.value:
                        self.lib.cid_index_destroy(vp(h))
                self._replica_handles = None
            for st in list(getattr(self, "_stripes", [])):
                st.close()
            for ks in list(getattr(self, "_ksets", [])):
                ks.close()
            for c in self.ctxs:
                c.close()          # closes the indices / k-mer sets made from the ranks' contexts
            self.lib.cid_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class GroupStripes:
    """One index cut into colour stripes, one per rank of a Group.  Outputs cover all colours."""

    def __init__(self, group, m, n_hash, k, n_colors_total, hash_variant=0):
        self.g, self.lib = group, group.lib
        self.m, self.n_hash, self.k, self.n_colors = m, n_hash, k, n_colors_total
        self.w32 = (n_colors_total + 31) // 32
        self.arr = (vp * group.n)()
        check(self.lib.cid_group_stripes_create(group.h, m, n_hash, k, n_colors_total, hash_variant, self.arr))
        base = np.zeros(group.n + 1, np.uint32)
        check(self.lib.cid_group_stripes_base(group.h, self.arr, _p(base)))
        self.base = base
        if not hasattr(group, "_stripes"):
            group._stripes = []
        group._stripes.append(self)

    def put_rows(self, row_ids, words):
        row_ids = np.ascontiguousarray(row_ids, np.uint64)
        words = np.ascontiguousarray(words, np.uint32).reshape(len(row_ids), self.w32)
        check(self.lib.cid_group_stripes_put_rows(self.g.h, self.arr, _p(row_ids), _p(words), len(row_ids)))

    def put_records(self, records, n_records):
        buf = np.frombuffer(records, np.uint8)
        check(self.lib.cid_group_stripes_put_records(self.g.h, self.arr, _p(buf), n_records))

    def finalize(self):
        for r in range(self.g.n):
            check(self.lib.cid_index_finalize(vp(self.arr[r])))
        return self

    def search_count(self, kmers, freq=None):
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, self.k)
        K = kmers.shape[0]
        f = None if freq is None else np.ascontiguousarray(freq, np.uint32)
        hits, nu, sf = (np.zeros(self.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(K, np.uint32)
        check(self.lib.cid_group_stripes_search_count(self.g.h, self.arr, _p(kmers), _p(f), K, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_set(self, kmerset):
        hits, nu, sf = (np.zeros(self.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(len(kmerset), np.uint32)
        check(self.lib.cid_group_stripes_search_count_set(self.g.h, self.arr, kmerset.h, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_set_report(self, kmerset):
        hits, nu, sf, md = (np.zeros(self.n_colors, np.uint64) for _ in range(4))
        check(self.lib.cid_group_stripes_search_count_set_report(self.g.h, self.arr, kmerset.h, _p(hits), _p(nu), _p(sf), _p(md)))
        return hits, nu, sf, md

    def search_perfect(self, kmers):
        kmers = np.ascontiguousarray(kmers, np.uint8).reshape(-1, self.k)
        words = np.zeros(self.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_group_stripes_search_perfect(self.g.h, self.arr, _p(kmers), kmers.shape[0], _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def search_perfect_set(self, kmerset):
        words = np.zeros(self.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_group_stripes_search_perfect_set(self.g.h, self.arr, kmerset.h, _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def readid_count_sparse(self, bases, seq_off, read_seq0, d=1, start_sample=3):
        bases = np.ascontiguousarray(bases, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, np.uint64)
        n_reads = len(read_seq0) - 1
        nk = np.zeros(n_reads, np.uint32)
        st = np.zeros(n_reads, np.uint8)
        ne = C.c_uint64(0)
        check(self.lib.cid_group_stripes_readid_count_sparse(self.g.h, self.arr, _p(bases), _p(seq_off), len(seq_off) - 1, _p(read_seq0), n_reads,
                                                             d, start_sample, _p(nk), _p(st), C.byref(ne)))
        rs = np.zeros(n_reads + 1, np.uint64)
        col = np.zeros(ne.value, np.uint32)
        cnt = np.zeros(ne.value, np.uint32)
        check(self.lib.cid_group_readid_sparse_fetch(self.g.h, _p(rs), _p(col), _p(cnt)))
        return rs, col, cnt, nk, st

    def close(self):
        if getattr(self, "arr", None) is not None:
            for r in range(self.g.n):
                if self.arr[r]:
                    self.lib.cid_index_destroy(vp(self.arr[r]))
            self.arr = None
            if self in getattr(self.g, "_stripes", []):
                self.g._stripes.remove(self)


class GroupKmerSet:
    """Distinct canonical k-mers of a query counted over all ranks of a Group (cid_group_kmerset): rank r ends up with the r-th code
    range; the parts laid end to end are the set in a KmerSet's order."""

    def __init__(self, group, k):
        self.g, self.lib, self.k = group, group.lib, k
        h = vp()
        check(self.lib.cid_group_kmerset_create(group.h, k, C.byref(h)))
        self.h = h
        if not hasattr(group, "_ksets"):
            group._ksets = []
        group._ksets.append(self)

    def add_seqs(self, seqs, mode=0):
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        bases = np.frombuffer(b"".join(seqs), np.uint8) if seqs else np.zeros(0, np.uint8)
        check(self.lib.cid_group_kmerset_add_seqs(self.h, _p(bases), _p(off), len(seqs), mode))

    def finalize(self):
        n = C.c_uint64(0)
        check(self.lib.cid_group_kmerset_finalize(self.h, C.byref(n)))
        return n.value

    def __len__(self):
        n = C.c_uint64(0)
        check(self.lib.cid_group_kmerset_size(self.h, C.byref(n)))
        return n.value

    def part_sizes(self):
        s = np.zeros(self.g.n, np.uint64)
        check(self.lib.cid_group_kmerset_part_sizes(self.h, _p(s)))
        return s

    def histogram(self):
        nb = C.c_size_t(0)
        check(self.lib.cid_group_kmerset_count_histogram(self.h, None, None, 0, C.byref(nb)))
        vals = np.zeros(nb.value, np.uint32)
        cnts = np.zeros(nb.value, np.uint64)
        check(self.lib.cid_group_kmerset_count_histogram(self.h, _p(vals), _p(cnts), nb.value, C.byref(nb)))
        return vals, cnts

    def clean(self, t):
        check(self.lib.cid_group_kmerset_clean(self.h, t))

    def download(self):
        n = len(self)
        km = np.zeros((n, self.k), np.uint8)
        cnt = np.zeros(n, np.uint32)
        check(self.lib.cid_group_kmerset_download(self.h, _p(km), _p(cnt)))
        return km, cnt

    def search_count(self):
        arr, ix = self.g._idx()
        hits, nu, sf = (np.zeros(ix.n_colors, np.uint64) for _ in range(3))
        uc = np.zeros(len(self), np.uint32)
        check(self.lib.cid_group_search_count_parts(self.g.h, arr, self.h, _p(hits), _p(nu), _p(sf), _p(uc)))
        return hits, nu, sf, uc

    def search_count_report(self):
        """hits, n_unique, sum_unique_freq, mode_unique_freq per colour; nothing per k-mer leaves the GPUs"""
        arr, ix = self.g._idx()
        hits, nu, sf, md = (np.zeros(ix.n_colors, np.uint64) for _ in range(4))
        check(self.lib.cid_group_search_count_parts_report(self.g.h, arr, self.h, _p(hits), _p(nu), _p(sf), _p(md)))
        return hits, nu, sf, md

    def search_perfect(self):
        arr, ix = self.g._idx()
        words = np.zeros(ix.w32, np.uint32)
        missing = C.c_int(0)
        check(self.lib.cid_group_search_perfect_parts(self.g.h, arr, self.h, _p(words), C.byref(missing)))
        return words, bool(missing.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cid_group_kmerset_destroy(self.h)
            self.h = None
            if self in getattr(self.g, "_ksets", []):
                self.g._ksets.remove(self)


class FastqReader:
    """cid_fastq: FASTQ text or block-gzip members -> records -> quality-masked reads -> classification counts, all on the device."""

    def __init__(self, ctx, n_files=1, quality=0):
        self.ctx, self.lib, self.n_files = ctx, ctx.lib, n_files
        h = vp()
        check(self.lib.cid_fastq_create(ctx.h, n_files, quality, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def push_text(self, file, text: bytes, last=False):
        buf = np.frombuffer(text, np.uint8) if len(text) else np.zeros(1, np.uint8)
        check(self.lib.cid_fastq_push_text(self.h, file, _p(buf), len(text), 1 if last else 0))   # CID_FASTQ_LAST

    def push_bgzf(self, file, members, text_lens, last=False):
        """members: list of whole BGZF members (bytes); text_lens: their ISIZE"""
        blob = np.frombuffer(b"".join(members) + b"\0", np.uint8)
        ln = np.array([len(m) for m in members], np.uint32)
        off = (np.cumsum(np.concatenate([[0], ln[:-1]])) if len(ln) else np.zeros(0)).astype(np.uint32)
        tl = np.array(text_lens, np.uint32)
        check(self.lib.cid_fastq_push_bgzf(self.h, file, _p(blob), len(blob) - 1, _p(off), _p(ln), _p(tl), len(members), 1 if last else 0))

    def classify(self, index, d=1, start_sample=3, max_pushes=0):
        """-> (ids [list of bytes], n_kmers, status, row_start, colours, counts) of every complete record held (after taking per file
        the oldest max_pushes waiting block-gzip pushes; 0 = all)"""
        n, ne, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self.lib.cid_fastq_classify(self.h, index.h, d, start_sample, max_pushes, C.byref(n), C.byref(ne), C.byref(nb)))
        return self._fetch(n, ne, nb)

    def classify_begin(self, index, d=1, start_sample=3, max_pushes=0):
        """first half of classify: records cut and packed, the classifier launched; the results of the step before stay fetchable"""
        check(self.lib.cid_fastq_classify_begin(self.h, index.h, d, start_sample, max_pushes))

    def classify_end(self):
        """second half: -> (n_reads, n_entries, id_bytes) of the step; fetch() then returns its results"""
        n, ne, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self.lib.cid_fastq_classify_end(self.h, C.byref(n), C.byref(ne), C.byref(nb)))
        return n, ne, nb

    def fetch(self, sizes):
        """the results of the last ended step (sizes: what classify_end returned), as classify returns them"""
        return self._fetch(*sizes)

    def _fetch(self, n, ne, nb):
        nk = np.zeros(n.value, np.uint32)
        st = np.zeros(n.value, np.uint8)
        rs = np.zeros(n.value + 1, np.uint64)
        col = np.zeros(ne.value, np.uint32)
        cnt = np.zeros(ne.value, np.uint32)
        io = np.zeros(n.value + 1, np.uint64)
        ids = np.zeros(max(nb.value, 1), np.uint8)
        check(self.lib.cid_fastq_fetch(self.h, _p(nk), _p(st), _p(rs), _p(col), _p(cnt), _p(io), _p(ids)))
        raw = ids.tobytes()
        names = [raw[int(io[r]):int(io[r + 1]) - 1] for r in range(n.value)]
        return names, nk, st, rs, col, cnt

    def count_kmers(self, kmerset, max_pushes=0):
        """every complete record held adds its reads' k-mers to `kmerset` (search's fastq producers); -> reads taken"""
        n = C.c_uint64(0)
        check(self.lib.cid_fastq_count_kmers(self.h, kmerset.h, max_pushes, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.cid_fastq_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
