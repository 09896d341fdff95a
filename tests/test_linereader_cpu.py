"""The CLI's input reader (colorid_amd/csrc/host/fastx_kmers.cpp, LineReader) on plain text, single-stream gzip, multi-member gzip
and block gzip (BGZF: members inflated by several threads): the same lines in the same order.  Host code only — runs without a GPU."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def write_bgzf(path, data: bytes, block=65280, level=6, eof_marker=True):
    """Block gzip as bgzip / htslib write it (SAM spec §4.1): members of <= 64 KiB with their size in a 'BC' extra field."""
    with open(path, "wb") as f:
        chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
        for c in chunks:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            bsize = 12 + 6 + len(body) + 8 - 1
            f.write(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize))
            f.write(body + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    lib = os.path.join(ROOT, "colorid_amd", "libcolorid_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libcolorid_hip.so not built")
    exe = str(tmp_path_factory.mktemp("lr") / "lr_shim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "cpu_shim", "linereader_shim.cpp"),
                    os.path.join(ROOT, "colorid_amd", "csrc", "host", "fastx_kmers.cpp"), "-L" + os.path.join(ROOT, "colorid_amd"),
                    "-lcolorid_hip", "-lz", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "colorid_amd"), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def _fastq(rng, n_reads):
    out = []
    for i in range(n_reads):
        L = int(rng.integers(30, 251))
        s = bytes(rng.choice(list(b"ACGTN"), size=L).astype(np.uint8))
        out.append(b"@read%d some words\n" % i + s + b"\n+\n" + b"I" * L + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("n_reads", [0, 1, 3000, 120_000])
def test_linereader_same_lines_for_every_container(shim, tmp_path, n_reads):
    rng = np.random.default_rng(n_reads)
    text = _fastq(rng, n_reads)
    if n_reads == 3000:
        text = text[:-1]                      # no newline at the very end
    paths = {}
    paths["plain"] = tmp_path / "a.fastq"
    paths["plain"].write_bytes(text)
    paths["gz"] = tmp_path / "a.fastq.gz"
    with gzip.open(paths["gz"], "wb", compresslevel=4) as f:
        f.write(text)
    paths["multi"] = tmp_path / "m.fastq.gz"   # three concatenated members (MultiGzDecoder in the reference)
    third = len(text) // 3
    with open(paths["multi"], "wb") as f:
        for part in (text[:third], text[third:2 * third], text[2 * third:]):
            f.write(gzip.compress(part, 3))
    paths["bgzf"] = tmp_path / "b.fastq.gz"
    write_bgzf(paths["bgzf"], text)
    paths["bgzf_small"] = tmp_path / "c.fastq.gz"     # odd block size: lines and reads straddle members and batches
    write_bgzf(paths["bgzf_small"], text, block=777, eof_marker=False)
    outs = {}
    for name, p in paths.items():
        for threads in (("8",) if "bgzf" not in name else ("8", "3", "1")):
            # the copying call, the view call (lines as pointers into the decoded blocks), both alternating, and a stream that was
            # started ahead of its reader (LineReader::prefetch: what the CLI does while the index loads)
            for mode in ("copy", "view", "mixed", "prefetch"):
                r = subprocess.run([shim, str(p), mode], capture_output=True, text=True, env=dict(os.environ, COLORID_GZ_THREADS=threads))
                assert r.returncode == 0, (name, mode, r.stderr)
                outs[(name, threads, mode)] = r.stdout.split()
            if name in ("gz", "multi"):   # single-stream gzip: zlib's raw inflate + libdeflate's CRC-32 when the host has libdeflate, gzread otherwise
                r = subprocess.run([shim, str(p), "view"], capture_output=True, text=True, env=dict(os.environ, COLORID_LIBDEFLATE="0"))
                assert r.returncode == 0, (name, r.stderr)
                outs[(name, threads, "gzread")] = r.stdout.split()
                # ... and zlib's raw inflate instead of the CLI's own decoder (fast_inflate.hpp) inside that reader
                r = subprocess.run([shim, str(p), "view"], capture_output=True, text=True, env=dict(os.environ, COLORID_FAST_INFLATE="0"))
                assert r.returncode == 0, (name, r.stderr)
                outs[(name, threads, "zlib_raw")] = r.stdout.split()
            if "bgzf" in name:   # BGZF members go through libdeflate when the host has it: the same lines with zlib
                r = subprocess.run([shim, str(p), "view"], capture_output=True, text=True, env=dict(os.environ, COLORID_GZ_THREADS=threads, COLORID_LIBDEFLATE="0"))
                assert r.returncode == 0, (name, r.stderr)
                outs[(name, threads, "zlib")] = r.stdout.split()
    want = outs[("plain", "8", "copy")]
    assert int(want[0]) == text.count(b"\n") + (1 if text and not text.endswith(b"\n") else 0)
    assert int(want[1]) == len(text) - text.count(b"\n")
    for key, got in outs.items():
        assert got == want, key


def test_linereader_reports_a_corrupt_bgzf_member(shim, tmp_path):
    rng = np.random.default_rng(9)
    text = _fastq(rng, 5000)
    p = tmp_path / "bad.fastq.gz"
    write_bgzf(p, text)
    raw = bytearray(p.read_bytes())
    raw[len(raw) // 2] ^= 0x55                  # somewhere inside a member's deflate stream (or its CRC)
    p.write_bytes(bytes(raw))
    for env in ({}, {"COLORID_LIBDEFLATE": "0"}):     # libdeflate (when present) and zlib both check the member's CRC-32
        r = subprocess.run([shim, str(p)], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 101 and ("corrupt gzip member" in r.stderr or "BGZF" in r.stderr or "truncated" in r.stderr)


def test_linereader_single_stream_gzip_containers(shim, tmp_path):
    """Single-stream gzip through the container parser of the libdeflate path (and through gzread): header fields (extra, name,
    comment, header CRC), an empty member between two others, trailing bytes that are no member; and a corrupt stream, a wrong CRC-32
    and a truncated file are reported."""
    rng = np.random.default_rng(5)
    text = _fastq(rng, 20_000)
    plain = tmp_path / "p.fastq"
    plain.write_bytes(text)
    want = subprocess.run([shim, str(plain)], capture_output=True, text=True, check=True).stdout

    def member(data, flg=0, level=6):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(data) + co.flush()
        hdr = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\xff"
        if flg & 4:
            hdr += struct.pack("<H", 5) + b"hello"
        if flg & 8:
            hdr += b"name.fastq\0"
        if flg & 16:
            hdr += b"a comment\0"
        if flg & 2:
            hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
        return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)

    third = len(text) // 3
    good = member(text[:third], 4 | 8 | 16 | 2) + member(b"") + member(text[third:2 * third], 8, level=1) + member(text[2 * third:], 0, level=9)
    cases = {"fields": good, "trailing": good + b"\0" * 37 + b"not a member"}
    for tag, blob in cases.items():
        p = tmp_path / f"{tag}.fastq.gz"
        p.write_bytes(blob)
        for env in ({}, {"COLORID_LIBDEFLATE": "0"}, {"COLORID_FAST_INFLATE": "0"}):
            r = subprocess.run([shim, str(p), "view"], capture_output=True, text=True, env=dict(os.environ, **env))
            assert r.returncode == 0 and r.stdout == want, (tag, env, r.stderr[-200:])
    bad = bytearray(good); bad[len(good) // 2] ^= 0x40
    crc = bytearray(good); crc[-8] ^= 1
    for tag, blob in (("flip", bytes(bad)), ("crc", bytes(crc)), ("cut", good[:len(good) - 20])):
        p = tmp_path / f"bad_{tag}.fastq.gz"
        p.write_bytes(blob)
        for env in ({}, {"COLORID_FAST_INFLATE": "0"}):
            r = subprocess.run([shim, str(p), "view"], capture_output=True, text=True, env=dict(os.environ, **env))
            assert r.returncode == 101 and ("gzip member" in r.stderr), (tag, env, r.returncode, r.stderr[-200:])


def test_linereader_single_stream_gzip_on_several_threads(shim, tmp_path):
    """par_gunzip.hpp inside LineReader (COLORID_GZ_THREADS >= 3): a gzip stream of several rounds of chunks, at two levels; the same
    text as three members (one of them small); bytes behind the last member; against the serial decoder (COLORID_PAR_GZIP=0) and the
    plain text.  A flipped byte, a wrong CRC-32 and a cut file are reported as by the serial decoder."""
    rng = np.random.default_rng(21)
    n = 260_000
    L = 100
    seqs = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, L))
    quals = rng.integers(35, 74, (n, L)).astype(np.uint8)
    text = b"".join(b"@r%d\n" % i + seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(n))      # ~56 MB: ~30 MB compressed
    plain = tmp_path / "big.fastq"
    plain.write_bytes(text)
    want = subprocess.run([shim, str(plain)], capture_output=True, text=True, check=True).stdout

    def member(data, level):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        return b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + co.compress(data) + co.flush() + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)

    cut1, cut2 = len(text) // 2, len(text) // 2 + 70_000
    blobs = {"one_l1": member(text, 1), "one_l6": member(text, 6),
             "three": member(text[:cut1], 6) + member(text[cut1:cut2], 1) + member(text[cut2:], 6) + b"\0" * 11 + b"trailing"}
    par = dict(os.environ, COLORID_GZ_THREADS="4")
    for tag, blob in blobs.items():
        p = tmp_path / f"{tag}.fastq.gz"
        p.write_bytes(blob)
        for env in (par, dict(par, COLORID_PAR_GZIP="0")):
            r = subprocess.run([shim, str(p)], capture_output=True, text=True, env=dict(env, COLORID_TIMING="1"))
            assert r.returncode == 0 and r.stdout == want, (tag, env.get("COLORID_PAR_GZIP"), r.stderr[-200:])
            taken = [int(l.split(" rounds, ")[1].split()[0]) for l in r.stderr.splitlines() if "decoded on 4 threads" in l]
            assert (sum(taken) >= 10) == ("COLORID_PAR_GZIP" not in env), (tag, taken)
    good = blobs["one_l6"]
    bad = bytearray(good); bad[len(good) // 2] ^= 0x40
    crc = bytearray(good); crc[-8] ^= 1
    for tag, blob in (("flip", bytes(bad)), ("crc", bytes(crc)), ("cut", good[:len(good) - 20]), ("cut_half", good[:len(good) // 2])):
        p = tmp_path / f"bad_{tag}.fastq.gz"
        p.write_bytes(blob)
        for env in (par, dict(par, COLORID_PAR_GZIP="0")):
            r = subprocess.run([shim, str(p)], capture_output=True, text=True, env=env)
            assert r.returncode == 101 and ("gzip member" in r.stderr), (tag, env.get("COLORID_PAR_GZIP"), r.returncode, r.stderr[-200:])
