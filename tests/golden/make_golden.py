"""Regenerates tests/golden/xxh3_kat.json from python-xxhash (libxxhash XXH3_64bits_withSeed, v0.8.x).

These are the PUBLISHED-algorithm known answers the oracle's hash restatement is pinned to; the crate the
reference links (xxh3 ^0.1.1) is not available here, so no vector comes from the reference binary itself.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import random

import xxhash

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rnd = random.Random(20241008)
    cases = []
    fixed = [b"", b"A", b"AC", b"ACG", b"ATGC", b"ATGT", b"A" * 21, b"ACGTACGTACGTACGTACGTACGTACGTACG",
             b"TAATTAAATCTAACAATTTCGTTACAGATTT", b"acgtACGTacgtACGTacgtACGTacg"]
    for b in fixed:
        cases.append(b)
    for ln in list(range(1, 140)) + [200, 239, 240, 241, 255, 256, 500, 1024, 1025, 3000]:
        cases.append(bytes(rnd.choice(b"ACGTacgtN") for _ in range(ln)))
    out = []
    for b in cases:
        for seed in (0, 1, 2, 3, 7, 31, 2**32 + 5, 2**64 - 1):
            out.append({"hex": b.hex(), "seed": seed, "h": xxhash.xxh3_64_intdigest(b, seed=seed)})
    with open(os.path.join(HERE, "xxh3_kat.json"), "w") as f:
        json.dump({"source": f"python-xxhash {xxhash.VERSION} / libxxhash {xxhash.XXHASH_VERSION}", "vectors": out}, f)
    print(len(out), "vectors")


if __name__ == "__main__":
    main()
