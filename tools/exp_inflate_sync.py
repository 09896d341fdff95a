#!/usr/bin/env python3
"""Feasibility of decoding ONE block-gzip member with all 64 lanes of a wave: lanes would start at guessed bit offsets
inside a DEFLATE block and rely on Huffman codes falling into step with the true symbol boundaries.  This probe measures, on the CPU,
how far a decoder started at an arbitrary bit of a dynamic-Huffman block of FASTQ text runs before it meets a true boundary (and stays
on it), in bits and in symbols, and how often it fails first (an invalid code, a distance beyond the window — detectable — or a false
end-of-block).  Input: synthetic FASTQ (150-bp reads, realistic quality strings) cut into 64 KiB members and deflated by zlib level 6,
as bgzip does.  Output: one JSON line -> profiles/r05_inflate_sync_probe.json"""
import json, random, sys, zlib

random.seed(5)
def fastq(n):
    out = []
    for i in range(n):
        seq = "".join(random.choice("ACGT") for _ in range(150))
        q = []
        cur = 38
        for _ in range(150):
            cur = max(2, min(40, cur + random.choice([-3, -1, 0, 0, 0, 0, 1, 1])))
            q.append(chr(33 + cur))
        out.append(f"@SRR548019.{i} HWI-ST{random.randint(100,999)}:{random.randint(1,8)}:{random.randint(1000,2000)}:{random.randint(1000,20000)}/1\n{seq}\n+\n{''.join(q)}\n")
    return "".join(out).encode()

class Bits:
    def __init__(self, data, pos=0): self.d, self.p = data, pos
    def get(self, n):
        v = 0
        for i in range(n):
            byte = self.d[self.p >> 3]
            v |= ((byte >> (self.p & 7)) & 1) << i
            self.p += 1
        return v

def build(lengths):
    """canonical Huffman: {(len, code): symbol}"""
    maxl = max(lengths) if lengths else 0
    bl = [0] * (maxl + 1)
    for l in lengths:
        if l: bl[l] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for b in range(1, maxl + 1):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    tab = {}
    for s, l in enumerate(lengths):
        if l:
            tab[(l, nxt[l])] = s
            nxt[l] += 1
    return tab, maxl

def decode_sym(b, tab, maxl):
    code = 0
    for l in range(1, maxl + 1):
        code = (code << 1) | b.get(1)
        s = tab.get((l, code))
        if s is not None: return s
    return None

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

def read_header(b):
    final, typ = b.get(1), b.get(2)
    if typ != 2: return final, typ, None, None
    hlit, hdist, hclen = b.get(5) + 257, b.get(5) + 1, b.get(4) + 4
    order = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
    cl = [0] * 19
    for i in range(hclen): cl[order[i]] = b.get(3)
    ctab, cmax = build(cl)
    lens = []
    while len(lens) < hlit + hdist:
        s = decode_sym(b, ctab, cmax)
        if s < 16: lens.append(s)
        elif s == 16: lens += [lens[-1]] * (3 + b.get(2))
        elif s == 17: lens += [0] * (3 + b.get(3))
        else: lens += [0] * (11 + b.get(7))
    return final, typ, build(lens[:hlit]), build(lens[hlit:])

def walk(data, pos, lt, dt, limit_bits, end_bit):
    """symbol boundaries visited from bit `pos` (a list), and why the walk ended"""
    b = Bits(data, pos)
    seen = []
    try:
        while b.p < limit_bits:
            seen.append(b.p)
            s = decode_sym(b, *lt)
            if s is None: return seen, "bad literal/length code"
            if s == 256: return seen, "end of block" if b.p == end_bit else "false end of block"
            if s > 256:
                if s > 285: return seen, "bad length symbol"
                b.get(LEXT[s - 257])
                d = decode_sym(b, *dt)
                if d is None or d > 29: return seen, "bad distance code"
                b.get(DEXT[d])
    except IndexError:
        return seen, "ran off the data"
    return seen, "limit"

text = fastq(4000)
stats = {"members": 0, "blocks": 0, "block_bits": [], "block_symbols": [], "sync_bits": [], "sync_symbols": [], "fail": {}, "never": 0, "trials": 0}
for m0 in range(0, min(len(text), 20 * 65280), 65280):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = co.compress(text[m0:m0 + 65280]) + co.flush()
    data += b"\0" * 8
    stats["members"] += 1
    b = Bits(data)
    while True:
        final, typ, lt, dt = read_header(b)
        if typ != 2: break
        start = b.p
        true_seen, why = walk(data, start, lt, dt, len(data) * 8, -1)
        # the true walk ends at its end-of-block: the boundary after it
        bb = Bits(data, true_seen[-1]); decode_sym(bb, *lt); end = bb.p
        true_set = set(true_seen)
        stats["blocks"] += 1; stats["block_bits"].append(end - start); stats["block_symbols"].append(len(true_seen))
        for _ in range(40):
            g = random.randrange(start + 64, max(start + 65, end - 3000))
            if g in true_set: continue
            seen, why = walk(data, g, lt, dt, min(end, g + 6000), end)
            stats["trials"] += 1
            hit = next((i for i, p in enumerate(seen) if p in true_set), None)
            if hit is None:
                stats["never"] += 1
                stats["fail"][why] = stats["fail"].get(why, 0) + 1
            else:
                stats["sync_bits"].append(seen[hit] - g); stats["sync_symbols"].append(hit)
        b.p = end
        if final: break
def q(v, f): v = sorted(v); return v[min(len(v) - 1, int(f * len(v)))] if v else None
out = {"members": stats["members"], "blocks": stats["blocks"], "block_bits_mean": sum(stats["block_bits"]) / len(stats["block_bits"]),
       "block_symbols_mean": sum(stats["block_symbols"]) / len(stats["block_symbols"]), "trials": stats["trials"],
       "synced": len(stats["sync_bits"]), "never_within_6000_bits": stats["never"], "fail_reasons": stats["fail"],
       "sync_bits": {"median": q(stats["sync_bits"], .5), "p90": q(stats["sync_bits"], .9), "p99": q(stats["sync_bits"], .99), "max": max(stats["sync_bits"])},
       "sync_symbols": {"median": q(stats["sync_symbols"], .5), "p90": q(stats["sync_symbols"], .9), "p99": q(stats["sync_symbols"], .99), "max": max(stats["sync_symbols"])}}
print(json.dumps(out))
