"""cid_bgzf_inflate: block-gzip members inflated on the GPU (one wave per member, DEFLATE decoded out of LDS) against zlib — every block
type (stored, fixed and dynamic Huffman codes), every compression level and strategy, member sizes 0 … 65536, texts from
incompressible to one repeated byte, arbitrary destination alignment; and corrupt members (data, CRC-32, ISIZE, truncation) are
reported by index as the CPU path (zlib) reports them."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bgzf_member(text: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra_subfield=False) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body = co.compress(text) + co.flush()
    extra = (b"XY" + struct.pack("<H", 3) + b"abc" if extra_subfield else b"") + b"BC" + struct.pack("<H", 2)
    xlen = len(extra) + 2
    bsize = 12 + xlen + len(body) + 8 - 1
    assert bsize < 65536 + 30000
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", xlen) + extra + struct.pack("<H", bsize & 0xFFFF) + body +
            struct.pack("<II", zlib.crc32(text) & 0xFFFFFFFF, len(text)))


def inflate(lib, ctx, members, texts_len, text_off=None):
    blob = b"".join(members)
    off = np.cumsum([0] + [len(m) for m in members[:-1]]).astype(np.uint32) if members else np.zeros(0, np.uint32)
    ln = np.array([len(m) for m in members], np.uint32)
    tl = np.array(texts_len, np.uint32)
    to = np.cumsum(np.concatenate([[0], tl[:-1]])).astype(np.uint32) if text_off is None else np.array(text_off, np.uint32)
    total = int((to + tl).max()) if len(tl) else 0
    out = np.zeros(total + 1, np.uint8)
    buf = np.frombuffer(blob + b"\0", np.uint8)
    bad = C.c_size_t(0)
    rc = lib.cid_bgzf_inflate(ctx.h, buf.ctypes.data, len(blob), off.ctypes.data, ln.ctypes.data, to.ctypes.data, tl.ctypes.data, len(members),
                              out.ctypes.data, total, C.byref(bad))
    return rc, out[:total].tobytes(), bad.value, to


def fastq_text(rng, n):
    out = []
    for i in range(n):
        L = int(rng.integers(50, 151))
        out.append(b"@read%d/1\n" % i + bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8)) + b"\n+\n" +
                   bytes(rng.choice(list(b"FFFFFF:,#"), size=L).astype(np.uint8)) + b"\n")
    return b"".join(out)


def test_inflate_equals_zlib_over_block_types_and_shapes(hip_ctx):
    lib = hip_ctx.lib
    rng = np.random.default_rng(1)
    fq = fastq_text(rng, 3000)
    texts, members = [], []
    shapes = [(b"", 6), (b"A", 6), (b"ACGT" * 5, 1), (fq[:65280], 6), (fq[1000:66536], 9), (fq[:65280], 1), (fq[:40000], 0),
              (bytes(rng.integers(0, 256, 65536).astype(np.uint8)), 6),      # incompressible: stored blocks inside a level-6 stream
              (bytes(rng.integers(0, 256, 70).astype(np.uint8)), 0),
              (b"\n" * 65536, 6), (b"AC" * 30000, 9), (b"ACGTTGCA" * 8000 + fq[:1000], 4),   # long overlapping matches
              (bytes(rng.integers(0, 4, 65536).astype(np.uint8)), 6),        # 2-bit alphabet: short codes, deep trees for the rest
              (bytes(rng.choice([65, 67], size=50000, p=[0.999, 0.001]).astype(np.uint8)), 6)]
    for t, lv in shapes:
        texts.append(t); members.append(bgzf_member(t, lv))
    for strat in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        for t in (fq[:65280], fq[70000:70100], b"N" * 3000 + fq[:5000]):
            texts.append(t); members.append(bgzf_member(t, 6, strat))
    texts.append(fq[:12345]); members.append(bgzf_member(fq[:12345], 6, extra_subfield=True))     # another extra subfield before "BC"
    for _ in range(40):                                                                            # random cuts of the FASTQ text
        a = int(rng.integers(0, len(fq) - 65536)); n = int(rng.integers(1, 65537))
        texts.append(fq[a:a + n]); members.append(bgzf_member(fq[a:a + n], int(rng.integers(1, 10))))
    rc, out, bad, to = inflate(lib, hip_ctx, members, [len(t) for t in texts])
    assert rc == 0, lib.cid_last_error()
    assert out == b"".join(texts)
    # texts written at odd offsets (the kernel stores 16-byte pieces aligned to the destination), with gaps between them
    offs, pos = [], 3
    for t in texts:
        offs.append(pos); pos += len(t) + int(rng.integers(0, 40))
    rc, out, bad, to = inflate(lib, hip_ctx, members, [len(t) for t in texts], text_off=offs)
    assert rc == 0, lib.cid_last_error()
    for t, o in zip(texts, offs):
        assert out[o:o + len(t)] == t
    # an empty batch
    assert lib.cid_bgzf_inflate(hip_ctx.h, None, 0, None, None, None, None, 0, None, 0, None) == 0


def test_inflate_a_million_reads_worth_of_members(hip_ctx):
    lib = hip_ctx.lib
    rng = np.random.default_rng(2)
    fq = fastq_text(rng, 60_000)                       # ~ 18 MB of text = 280 members of 65280 bytes
    texts = [fq[i:i + 65280] for i in range(0, len(fq), 65280)]
    members = [bgzf_member(t, 1 + i % 9) for i, t in enumerate(texts)]
    rc, out, bad, to = inflate(lib, hip_ctx, members * 4, [len(t) for t in texts] * 4)
    assert rc == 0, lib.cid_last_error()
    assert out == fq * 4


def test_inflate_reports_corrupt_members(hip_ctx):
    lib = hip_ctx.lib
    rng = np.random.default_rng(3)
    fq = fastq_text(rng, 2000)
    texts = [fq[i:i + 30000] for i in range(0, 150000, 30000)]
    good = [bgzf_member(t, 6) for t in texts]

    def expect_bad(members, lens, which, what):
        rc, out, bad, to = inflate(lib, hip_ctx, members, lens)
        assert rc == -1 and bad == which and what in lib.cid_last_error(), (rc, bad, lib.cid_last_error())

    lens = [len(t) for t in texts]
    m = list(good); b = bytearray(m[2]); b[-6] ^= 0x01; m[2] = bytes(b)                      # CRC-32 field
    expect_bad(m, lens, 2, b"CRC-32")
    m = list(good); ln = list(lens); ln[3] -= 1                                               # the caller's length differs from ISIZE
    expect_bad(m, ln, 3, b"ISIZE")
    m = list(good); b = bytearray(m[1]); b[3] = 0; m[1] = bytes(b)                           # no FEXTRA flag: not a BGZF header
    expect_bad(m, lens, 1, b"header")
    hits = 0
    for pos in (40, 200, 1000, 3000, 5000):                                                  # a flipped bit inside the DEFLATE data: some check must fire
        m = list(good); b = bytearray(m[4]); b[pos] ^= 0x10; m[4] = bytes(b)
        rc, out, bad, to = inflate(lib, hip_ctx, m, lens)
        assert rc == -1 and bad == 4
        hits += 1
    assert hits == 5
    m = list(good); body = m[0][:-8]; m[0] = body[:len(body) // 2] + m[0][-8:]               # truncated DEFLATE data (header and trailer intact)
    rc, out, bad, to = inflate(lib, hip_ctx, m, lens)
    assert rc == -1 and bad == 0
    # everything still works afterwards
    rc, out, bad, to = inflate(lib, hip_ctx, good, lens)
    assert rc == 0 and out == b"".join(texts)


def test_inflate_in_two_halves_on_two_contexts(hip_ctx):
    """cid_bgzf_inflate_start / _finish: two batches in flight on two contexts (what the CLI's reader does), the caller's member buffer
    overwritten right after _start; misuse (finish without start, a second start, a text buffer of another size) is refused."""
    import colorid_amd
    lib = hip_ctx.lib
    other = colorid_amd.Context(0)
    rng = np.random.default_rng(21)
    fq = fastq_text(rng, 8000)
    batches = []
    for b in range(4):
        texts = [fq[i:i + 60000] for i in range(b * 1000, len(fq) - 60000, 240000)]
        members = [bgzf_member(t, 1 + b) for t in texts]
        batches.append((texts, members))
    ctxs = [hip_ctx, other]
    pend = []
    outs = []

    def finish(item):
        cx, texts, total = item
        out = np.zeros(total + 1, np.uint8)
        bad = C.c_size_t(0)
        assert lib.cid_bgzf_inflate_finish(cx.h, out.ctypes.data, total, C.byref(bad)) == 0, lib.cid_last_error()
        outs.append((out[:total].tobytes(), b"".join(texts)))

    for i, (texts, members) in enumerate(batches):
        blob = np.frombuffer(b"".join(members) + b"\0", np.uint8).copy()
        off = np.cumsum([0] + [len(m) for m in members[:-1]]).astype(np.uint32)
        ln = np.array([len(m) for m in members], np.uint32)
        tl = np.array([len(t) for t in texts], np.uint32)
        to = np.cumsum(np.concatenate([[0], tl[:-1]])).astype(np.uint32)
        cx = ctxs[i & 1]
        assert lib.cid_bgzf_inflate_start(cx.h, blob.ctypes.data, len(blob) - 1, off.ctypes.data, ln.ctypes.data, to.ctypes.data, tl.ctypes.data, len(members), int(tl.sum())) == 0
        blob[:] = 0xAA                                            # the members were taken before _start returned
        if i == 0:
            assert lib.cid_bgzf_inflate_start(cx.h, blob.ctypes.data, len(blob) - 1, off.ctypes.data, ln.ctypes.data, to.ctypes.data, tl.ctypes.data, len(members), int(tl.sum())) == -5   # CID_ERR_STATE
        pend.append((cx, texts, int(tl.sum())))
        if len(pend) == 2:
            finish(pend.pop(0))
    while pend:
        finish(pend.pop(0))
    assert len(outs) == 4 and all(a == b for a, b in outs)
    bad = C.c_size_t(0)
    z = np.zeros(8, np.uint8)
    assert lib.cid_bgzf_inflate_finish(other.h, z.ctypes.data, 8, C.byref(bad)) == -5          # no start (CID_ERR_STATE)
    texts, members = batches[0]
    blob = np.frombuffer(b"".join(members) + b"\0", np.uint8)
    off = np.cumsum([0] + [len(m) for m in members[:-1]]).astype(np.uint32); ln = np.array([len(m) for m in members], np.uint32)
    tl = np.array([len(t) for t in texts], np.uint32); to = np.cumsum(np.concatenate([[0], tl[:-1]])).astype(np.uint32)
    assert lib.cid_bgzf_inflate_start(other.h, blob.ctypes.data, len(blob) - 1, off.ctypes.data, ln.ctypes.data, to.ctypes.data, tl.ctypes.data, len(members), int(tl.sum())) == 0
    assert lib.cid_bgzf_inflate_finish(other.h, z.ctypes.data, 8, C.byref(bad)) == -1          # another text size than announced
    other.close()


def test_line_reader_with_gpu_inflate_reads_the_same_lines(tmp_path):
    """The CLI's reader with COLORID_GPU_INFLATE=1 (BGZF members decoded by cid_bgzf_inflate on a context of the reader thread) yields
    the lines of the plain file — members of 65280 and of 777 bytes, with and without the end-of-file marker — and dies on a corrupt member."""
    import os
    import subprocess
    from test_linereader_cpu import ROOT, HERE, write_bgzf, _fastq
    exe = str(tmp_path / "lr_shim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "cpu_shim", "linereader_shim.cpp"),
                    os.path.join(ROOT, "colorid_amd", "csrc", "host", "fastx_kmers.cpp"), "-L" + os.path.join(ROOT, "colorid_amd"),
                    "-lcolorid_hip", "-lz", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "colorid_amd"), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    rng = np.random.default_rng(11)
    text = _fastq(rng, 150_000)                  # ~ 50 MB: crosses the 64 MiB batch only with the small members below; several batches of 777-byte members
    plain = tmp_path / "a.fastq"
    plain.write_bytes(text)
    want = subprocess.run([exe, str(plain)], capture_output=True, text=True, check=True).stdout
    for block, eof in ((65280, True), (777, False), (4096, True)):
        p = tmp_path / f"b{block}.fastq.gz"
        write_bgzf(p, text if block != 777 else text[:3_000_000], block=block, level=1, eof_marker=eof)
        ref = want if block != 777 else subprocess.run([exe, str(tmp_path / "c.fastq")], capture_output=True, text=True,
                                                        check=(tmp_path / "c.fastq").write_bytes(text[:3_000_000]) > 0).stdout
        for mode in ("view", "prefetch"):
            r = subprocess.run([exe, str(p), mode], capture_output=True, text=True, env=dict(os.environ, COLORID_GPU_INFLATE="1"))
            assert r.returncode == 0, r.stderr[-300:]
            assert r.stdout == ref, (block, mode)
    bad = tmp_path / "bad.fastq.gz"
    raw = bytearray((tmp_path / "b65280.fastq.gz").read_bytes())
    raw[len(raw) // 2] ^= 0x55
    bad.write_bytes(bytes(raw))
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True, env=dict(os.environ, COLORID_GPU_INFLATE="1"))
    assert r.returncode == 101 and "corrupt gzip member" in r.stderr


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("INFLATE_FUZZ_SEED0", 0)), int(__import__("os").environ.get("INFLATE_FUZZ_SEED0", 0)) + int(__import__("os").environ.get("INFLATE_FUZZ_N", 6))))
def test_inflate_fuzz_valid_and_garbage_members(hip_ctx, seed):
    """Random texts (alphabets of 1 … 256 symbols, runs, repeats, FASTQ) at random sizes, levels and strategies decode to themselves; members
    with random bytes for their DEFLATE data, or valid ones with a few bytes overwritten, are either reported or — when the damage
    happens to decode and the CRC-32 still matches, which it practically never does — returned as zlib would return them.  The
    kernel always comes back: every step of its decoder consumes input or fails."""
    lib = hip_ctx.lib
    rng = np.random.default_rng(7000 + seed)
    texts, members = [], []
    for _ in range(48):
        kind = int(rng.integers(0, 5))
        n = int(rng.integers(0, 65537))
        if kind == 0:
            t = bytes(rng.integers(0, int(rng.integers(1, 257)), n).astype(np.uint8))
        elif kind == 1:
            t = (bytes(rng.integers(65, 91, int(rng.integers(1, 40))).astype(np.uint8)) * 70000)[:n]
        elif kind == 2:
            t = fastq_text(rng, 450)[:n]
        elif kind == 3:
            a = np.repeat(rng.integers(0, 256, n // 7 + 1).astype(np.uint8), rng.integers(1, 30, n // 7 + 1))[:n]
            t = bytes(a)
        else:
            t = bytes(rng.choice(np.frombuffer(b"ACGTN\n", np.uint8), n, p=[0.24, 0.24, 0.24, 0.24, 0.03, 0.01]))
        strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
        texts.append(t); members.append(bgzf_member(t, int(rng.integers(0, 10)), strat))
    rc, out, bad, to = inflate(lib, hip_ctx, members, [len(t) for t in texts])
    assert rc == 0, lib.cid_last_error()
    assert out == b"".join(texts)
    # damaged members: the call reports the first one zlib would also refuse
    dam = list(members)
    first_bad = None
    for i in range(0, len(dam), 5):
        m = bytearray(dam[i])
        body0, body1 = 18, len(m) - 8
        if body1 - body0 < 4:
            continue
        if i % 10 == 0:
            m[body0:body1] = bytes(rng.integers(0, 256, body1 - body0).astype(np.uint8))      # garbage instead of DEFLATE data
        else:
            for p in rng.integers(body0, body1, 3):
                m[int(p)] ^= int(rng.integers(1, 256))
        dam[i] = bytes(m)
        try:
            ok = zlib.decompress(dam[i], 31) == texts[i]
        except zlib.error:
            ok = False
        if not ok and first_bad is None:
            first_bad = i
    rc, out, bad, to = inflate(lib, hip_ctx, dam, [len(t) for t in texts])
    if first_bad is None:
        assert rc == 0
    else:
        assert rc == -1 and bad == first_bad, (rc, bad, first_bad, lib.cid_last_error())


class _Bits:
    """DEFLATE bit packing: header fields LSB first, Huffman codes MSB first."""
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF); self.acc >>= 8; self.n -= 8

    def code(self, code, nbits):
        for i in range(nbits - 1, -1, -1):
            self.put((code >> i) & 1, 1)

    def bytes(self):
        return bytes(self.out) + (bytes([self.acc & 0xFF]) if self.n else b"")


def _dynamic_block(lit_lens, dist_len, symbols):
    """One final dynamic-Huffman block: literal/length code lengths {symbol: len} (symbols <= 257), ONE distance code (symbol 0) of
    length dist_len, then `symbols` = list of ('lit', s) / ('match3', None: length 3 at distance 1) / ('eob', None)."""
    lens = [lit_lens.get(s, 0) for s in range(258)] + [dist_len]
    # canonical codes
    def canon(ls):
        codes, code = {}, 0
        for L in range(1, 16):
            for s, l in enumerate(ls):
                if l == L:
                    codes[s] = (code, L); code += 1
            code <<= 1
        return codes
    lit_codes = canon(lens[:258])
    # run-length the 259 lengths with symbols 0..15 and 18 (11..138 zeros) / 17 (3..10 zeros)
    seq, i = [], 0
    while i < len(lens):
        if lens[i] == 0:
            j = i
            while j < len(lens) and lens[j] == 0:
                j += 1
            run = j - i
            while run >= 11:
                r = min(run, 138); seq.append((18, r - 11, 7)); run -= r
            if run >= 3:
                seq.append((17, run - 3, 3)); run = 0
            seq += [(0, 0, 0)] * run
            i = j
        else:
            seq.append((lens[i], 0, 0)); i += 1
    used = sorted({s for s, _, _ in seq})
    # a complete code over the used code-length symbols: the first gets length 1 ... the last two share the longest
    cl = {}
    for idx, s in enumerate(used):
        cl[s] = min(idx + 1, len(used) - 1) if len(used) > 1 else 1
    cl_codes = canon([cl.get(s, 0) for s in range(19)])
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    hclen = max(i for i, s in enumerate(order) if cl.get(s, 0)) + 1
    hclen = max(hclen, 4)
    b = _Bits()
    b.put(1, 1); b.put(2, 2); b.put(258 - 257, 5); b.put(0, 5); b.put(hclen - 4, 4)
    for s in order[:hclen]:
        b.put(cl.get(s, 0), 3)
    for s, extra, nb in seq:
        b.code(*cl_codes[s])
        if nb:
            b.put(extra, nb)
    for kind, s in symbols:
        if kind == "lit":
            b.code(*lit_codes[s])
        elif kind == "eob":
            b.code(*lit_codes[256])
        else:   # length 3 (symbol 257, no extra bits), distance 1 (distance symbol 0, no extra bits): the single distance code is all zeros
            b.code(*lit_codes[257])
            b.code(0, dist_len)
    return b.bytes()


def _member_of_deflate(body, text):
    extra = b"BC" + struct.pack("<H", 2)
    bsize = 12 + len(extra) + 2 + len(body) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", len(extra) + 2) + extra + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(text) & 0xFFFFFFFF, len(text)))


def test_incomplete_code_sets_are_judged_as_zlib_judges_them(hip_ctx):
    """zlib's inflate_table accepts an incomplete Huffman set only when it is ONE code of length 1 (never for the code-length code):
    hand-assembled dynamic blocks with a single distance / literal code of length 1 (valid) and of length 2 (invalid)."""
    lib = hip_ctx.lib
    cases = []
    full = {65: 1, 256: 2, 257: 2}
    stream = [("lit", 65)] * 3 + [("match3", None), ("eob", None)]
    cases.append((_dynamic_block(full, 1, stream), b"AAAAAA"))            # one distance code of length 1: fine
    cases.append((_dynamic_block(full, 2, stream), b"AAAAAA"))            # one distance code of length 2: "invalid distances set"
    cases.append((_dynamic_block({256: 1}, 1, [("eob", None)]), b""))     # literal tree = the end-of-block code alone, length 1: fine
    cases.append((_dynamic_block({256: 2}, 1, [("eob", None)]), b""))     # ... of length 2: "invalid literal/lengths set"
    verdicts = []
    for body, text in cases:
        try:
            ok = zlib.decompress(body, -15) == text
        except zlib.error:
            ok = False
        verdicts.append(ok)
    assert verdicts == [True, False, True, False], verdicts               # (what this zlib says; the GPU must say the same)
    for (body, text), ok in zip(cases, verdicts):
        rc, out, bad, _ = inflate(lib, hip_ctx, [_member_of_deflate(body, text)], [len(text)])
        assert (rc == 0 and out == text) if ok else (rc == -1 and bad == 0), (ok, rc, out, lib.cid_last_error())
    # in a batch the first member zlib refuses is the one reported
    good = bgzf_member(b"ACGT" * 100)
    members = [good, _member_of_deflate(*cases[0]), _member_of_deflate(*cases[1]), good]
    rc, out, bad, _ = inflate(lib, hip_ctx, members, [400, 6, 6, 400])
    assert rc == -1 and bad == 2
