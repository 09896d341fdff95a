"""colorid_amd/csrc/host/par_gunzip.hpp — one DEFLATE stream decoded in chunks on several threads (the CLI's reader of single-stream gzip,
the commonest fastq.gz) — against zlib: the text, its CRC-32, where the stream ends.  The shim is built with ASan + UBSan.  FASTQ-like
text at several levels / strategies with flushes in between (empty stored blocks, windows reset), text with binary stretches (chunk starts
are refused there: serial), stored and fixed blocks, tiny and empty streams, chunks smaller than a block up to larger than the input,
input handed over in pieces of any size; damaged and truncated streams are refused as zlib refuses them."""
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pg") / "par_gunzip_shim")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(HERE, "cpu_shim", "par_gunzip_shim.cpp"), "-lz"], check=True)
    return exe


def fastq(rng, n, quals=(35, 74)):
    out = []
    for i in range(n):
        L = int(rng.integers(30, 151))
        s = bytes(rng.choice(list(b"ACGTN"), L, p=[.245, .245, .245, .245, .02]).astype(np.uint8))
        q = bytes(rng.integers(quals[0], quals[1], L).astype(np.uint8))
        out.append(b"@r%d/%d\n" % (i, i % 3) + s + b"\n+\n" + q + b"\n")
    return b"".join(out)


def run(shim, tmp_path, raw, chunk, n_chunks, threads, piece):
    src, dst = tmp_path / "in.deflate", tmp_path / "out.txt"
    src.write_bytes(raw)
    p = subprocess.run([shim, str(src), str(dst), str(chunk), str(n_chunks), str(threads), str(piece)], capture_output=True, text=True)
    return p, dst.read_bytes()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_parallel_inflate_equals_zlib(shim, tmp_path, seed):
    rng = np.random.default_rng(seed)
    accepted = 0
    for it in range(14):
        kind = it % 7
        if kind == 0: text = fastq(rng, int(rng.integers(1, 30000)))
        elif kind == 1: text = fastq(rng, int(rng.integers(1, 20000)), (73, 74))                                  # one quality letter: long matches
        elif kind == 2: text = bytes(rng.integers(0, 256, int(rng.integers(0, 400000))).astype(np.uint8))          # not text
        elif kind == 3: text = b"ACGT" * int(rng.integers(0, 300000))
        elif kind == 4: text = fastq(rng, int(rng.integers(1, 8000))) + bytes(rng.integers(0, 256, 5000).astype(np.uint8)) + fastq(rng, 5000)
        elif kind == 5: text = b""
        else: text = fastq(rng, int(rng.integers(1, 3)))
        level = int(rng.choice([0, 1, 1, 3, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        raw, pos = b"", 0
        while pos < len(text):
            n = int(rng.integers(1, max(2, len(text) // 3 + 1)))
            raw += co.compress(text[pos:pos + n]); pos += n
            fl = int(rng.integers(0, 6))
            if fl == 0: raw += co.flush(zlib.Z_SYNC_FLUSH)
            elif fl == 1: raw += co.flush(zlib.Z_FULL_FLUSH)
        raw += co.flush()
        tail = bytes(rng.integers(0, 256, int(rng.integers(0, 20))).astype(np.uint8))     # whatever follows the stream is handed back
        for chunk, nc, th, piece in ((65536, int(rng.integers(2, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 200000))),
                                     (int(rng.integers(65536, 400000)), 4, 3, 1 << 20)):
            p, got = run(shim, tmp_path, raw + tail, chunk, nc, th, piece)
            assert p.returncode == 0, (it, kind, level, strategy, p.stderr[-300:])
            assert got == text, (it, kind, level, strategy)
            f = p.stdout.split()
            assert int(f[1]) == zlib.crc32(text) and int(f[3]) == len(text) and int(f[5]) == len(tail)
            accepted += int(f[9])
    assert accepted > 20          # chunks that started in the middle of the stream and were taken


def test_parallel_inflate_refuses_what_zlib_refuses(shim, tmp_path):
    rng = np.random.default_rng(9)
    text = fastq(rng, 30000)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(text) + co.flush()
    p, got = run(shim, tmp_path, raw[:len(raw) * 2 // 3], 65536, 6, 3, 1 << 20)             # cut: the stream never ends
    assert p.returncode == 2 and "ends early" in p.stderr
    assert text.startswith(got)
    refused = 0   # (counted, not required)
    for k in range(12):
        bad = bytearray(raw)
        at = int(rng.integers(len(raw) // 10, len(raw) - 10))
        if k % 2: bad[at] ^= 1 << int(rng.integers(0, 8))
        else: bad[at:at + 64] = bytes(rng.integers(0, 256, 64).astype(np.uint8))
        try:
            ztext = zlib.decompress(bytes(bad), -15)
        except zlib.error:
            ztext = None
        p, got = run(shim, tmp_path, bytes(bad), 65536, 6, 3, 1 << 20)
        if ztext is None:
            assert p.returncode == 2, (k, at)
            refused += 1
        else:                       # a flip zlib decodes (another text, or a stream that ends elsewhere): the same text here
            assert p.returncode == 0 and got == ztext, (k, at)
    # (a complete Huffman code decodes any bits: most damage only changes the text, and zlib — like this decoder — goes on)
    for blob in (b"\x07", raw[:50] + b"\x07" + raw[50:60]):        # a block of type 3 at the start; a stream that runs into the end of the input
        with pytest.raises(zlib.error):
            zlib.decompress(blob, -15)
        p, got = run(shim, tmp_path, blob, 65536, 4, 2, 1 << 20)
        assert p.returncode == 2


def test_parallel_inflate_long_runs(shim, tmp_path):
    """Text that compresses a thousandfold: single blocks hold megabytes of text (a buffer sized by the compressed span does not take one —
    it grows), a round of chunks is decoded again serially many times over (more serial stretches than chunks)."""
    for text in ((b"ACGT" * 50 + b"\n") * 60_000, b"A" * 9_000_000, b"@r\n" + b"N" * 6_000_000 + b"\n+\n" + b"#" * 6_000_000 + b"\n"):
        for level in (1, 6):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            raw = co.compress(text) + co.flush()
            for chunk, nc, th in ((65536, 4, 3), (65536, 2, 2), (200000, 8, 4)):
                p, got = run(shim, tmp_path, raw + b"12345678", chunk, nc, th, 1 << 20)
                assert p.returncode == 0, (len(text), level, chunk, p.stderr[-300:])
                assert got == text and int(p.stdout.split()[5]) == 8, (len(text), level, chunk)
