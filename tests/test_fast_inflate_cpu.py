"""colorid_amd/csrc/host/fast_inflate.hpp — the CLI's own streaming DEFLATE decoder for single-stream gzip — against zlib: every block
type, level and strategy, texts from empty to megabytes (FASTQ-like, random bytes, long runs, long-distance repeats), input and output cut
into pieces of every awkward size (the decoder works in whole steps between them), and damaged streams, which must be refused or decode
to something else — never crash (the driver is built with AddressSanitizer and UBSan).  Host code only: runs without a GPU."""
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fi") / "inflate_shim")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(HERE, "cpu_shim", "inflate_shim.cpp")], check=True)
    return exe


def texts(rng):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    fq = b"".join(b"@r%d len\n" % i + acgt[rng.integers(0, 4, 150)].tobytes() + b"\n+\n" + np.frombuffer(b"FFFF:,#", np.uint8)[rng.integers(0, 7, 150)].tobytes() + b"\n"
                  for i in range(3000))
    genome = acgt[rng.integers(0, 4, 40_000)].tobytes()
    return {
        "empty": b"", "one": b"A", "fastq": fq, "random": rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),
        "runs": b"A" * 100_000 + b"CG" * 50_000 + b"ACGTACG" * 20_000,
        "far": genome + rng.integers(0, 256, 30_000, dtype=np.uint8).tobytes() + genome[:32_000] + genome[5000:37_000],   # matches 32 KiB back
        "binary_skew": np.minimum(rng.geometric(0.02, 300_000), 255).astype(np.uint8).tobytes(),                           # long codes (> 11 bits)
    }


def deflate(text, level, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    return co.compress(text) + co.flush()


def run(shim, tmp_path, raw, in_chunk, out_block, tail=b"TRAILER!"):
    src, dst = tmp_path / "x.deflate", tmp_path / "x.out"
    src.write_bytes(raw + tail)
    r = subprocess.run([shim, str(src), str(dst), str(in_chunk), str(out_block)], capture_output=True, text=True)
    return r, (dst.read_bytes() if dst.exists() else b"")


def test_every_block_type_level_and_cut(shim, tmp_path):
    rng = np.random.default_rng(4)
    T = texts(rng)
    cases = 0
    for name, text in T.items():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED)):
            raw = deflate(text, level, strategy)
            for in_chunk, out_block in ((1 << 20, 4 << 20), (1100, 33_000), (2049, 40_001)):
                r, got = run(shim, tmp_path, raw, in_chunk, out_block)
                assert r.returncode == 0, (name, level, strategy, in_chunk, r.stderr[-300:])
                assert got == text, (name, level, strategy, in_chunk, len(got), len(text))
                assert "unused 8" in r.stderr, (name, level, r.stderr)      # it stopped exactly at the end of the stream
                cases += 1
    assert cases == len(T) * 8 * 3
    # several flushed pieces in one stream (sync flush = empty stored blocks between Huffman blocks), small windows, memLevel 1 (tiny blocks)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = b"".join(co.compress(T["fastq"][i:i + 7001]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(T["fastq"]), 7001)) + co.flush()
    r, got = run(shim, tmp_path, raw, 1500, 33_000)
    assert r.returncode == 0 and got == T["fastq"]
    for wbits, mem in ((-9, 1), (-12, 3)):
        r, got = run(shim, tmp_path, deflate(T["far"], 6, wbits=wbits, mem=mem), 4096, 1 << 20)
        assert r.returncode == 0 and got == T["far"]


@pytest.mark.parametrize("seed", range(40))
def test_damaged_streams_never_crash(shim, tmp_path, seed):
    """bit flips, truncations and garbage: refused (exit 2 with a reason) or decoded to something else — the gzip reader's CRC-32 catches
    those — but never a crash, an out-of-bounds access (ASan) or a hang"""
    rng = np.random.default_rng(1000 + seed)
    text = texts(rng)[("fastq", "runs", "far", "binary_skew")[seed % 4]][: int(rng.integers(1, 150_000))]
    raw = bytearray(deflate(text, int(rng.choice([1, 6, 9])), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE]))))
    kind = seed % 5
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            raw[int(rng.integers(0, len(raw)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        raw = raw[: int(rng.integers(0, len(raw)))]
    elif kind == 2:
        raw = bytearray(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes())
    elif kind == 3:
        p = int(rng.integers(0, len(raw)))
        raw[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
    else:
        raw = raw[: max(1, len(raw) // 2)] + raw[len(raw) // 2 + int(rng.integers(1, 40)):]
    r, got = run(shim, tmp_path, bytes(raw), int(rng.choice([1100, 5000, 1 << 20])), int(rng.choice([33_000, 1 << 20])), tail=b"")
    assert r.returncode in (0, 2), (r.returncode, r.stderr[-400:])
    if r.returncode == 2:
        assert r.stderr.strip() != ""
    else:   # zlib may accept the damaged stream too; if it does, both decode the same text
        try:
            want = zlib.decompressobj(-15).decompress(bytes(raw))
        except zlib.error:
            want = None
        if want is not None and len(want) == len(got):
            assert got == want
