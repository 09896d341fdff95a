"""The reference's own end-to-end check, restated: /root/reference/test.sh builds an index of four Listeria phage genomes
(`build -s 750000 -n 4 -k 27`, test.sh:3), classifies test_data/SRR548019.fastq.gz (`read_id -d 10`, test.sh:19) and searches it
(`search -f 1`, test.sh:31), then asserts ONE output row (test.sh:37):

    ./test_data/SRR548019.fastq.gz  187112  Listeria_phage_B056  1.00  185.95  8  26642

That row is the only output of the Rust program the reference tree holds, and so the only artefact that can tell whether this
build hashes like crate xxh3 0.1.x (the 26642 / 185.95 fields depend on which cross-phage Bloom false positives occur).  The
fastq is NOT in the reference tree (nor fetchable here).  So:
  * the flow itself always runs, on a synthetic stand-in for the run (reads drawn from phage B056), checked against the oracle;
  * the golden row is asserted as soon as the real file is supplied — COLORID_TEST_FASTQ=/path/SRR548019.fastq.gz or
    tests/golden/SRR548019.fastq.gz — under every available --hash; the variant(s) that reproduce it are reported, and the test
    fails if none does (then no available variant is the reference's hash)."""
import os
import subprocess

import numpy as np
import pytest

from util import synth_fastq_records, write_fastq_gz

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.environ.get("COLORID_BIN", os.path.join(ROOT, "colorid_amd", "bin", "colorid"))
REFS = os.path.join(HERE, "golden", "refs")
PHAGES = ["Listeria_phage_B021", "Listeria_phage_B051", "Listeria_phage_B056", "Listeria_phage_B545"]
BANNER = "\n ************** initializing logger *****************\n\n"
GOLDEN_ROW = ["187112", "Listeria_phage_B056", "1.00", "185.95", "8", "26642"]     # test.sh:37, fields 2..7
VARIANTS = ["xxh3_v08", "xxh3_v07"]


def real_fastq():
    for p in (os.environ.get("COLORID_TEST_FASTQ"), os.path.join(HERE, "golden", "SRR548019.fastq.gz")):
        if p and os.path.exists(p):
            return p
    return None


def cli(*args, cwd):
    p = subprocess.run([BIN, *args], capture_output=True, text=True, cwd=cwd)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith(BANNER)
    return p.stdout[len(BANNER):]


def test_sh_flow(cwd, fastq, variant):
    """test.sh:3-31 with --hash `variant`; returns (search rows, read_id rows)."""
    cli("build", "-s", "750000", "-n", "4", "-k", "27", "-b", "phage", "-r", "ref_file.txt", "--hash", variant, cwd=cwd)
    cli("build", "-s", "750000", "-n", "4", "-k", "27", "-b", "phage", "-r", "ref_file.txt", "-t", "2", "--hash", variant, cwd=cwd)   # test.sh:11
    cli("read_id", "-b", "phage.bxi", "-q", fastq, "-n", "test_read_id", "-d", "10", "--hash", variant, cwd=cwd)
    out = cli("search", "-b", "phage.bxi", "-q", fastq, "-f", "1", "--hash", variant, cwd=cwd)
    rows = [l.split("\t") for l in out.splitlines() if l.strip()]
    reads = [l.rstrip("\n").split("\t") for l in open(os.path.join(cwd, "test_read_id_reads.txt"))]
    return rows, reads


test_sh_flow.__test__ = False


@pytest.fixture()
def workdir(tmp_path):
    (tmp_path / "ref_file.txt").write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    return str(tmp_path)


def test_flow_on_a_synthetic_run_matches_the_oracle(orc, workdir):
    genome = b"".join(orc.read_fasta(os.path.join(REFS, "Listeria_phage_B056.fasta")))
    rng = np.random.default_rng(548019)
    recs = synth_fastq_records(rng, [genome], 20000, 100, lower_rate=0.0)
    fq = os.path.join(workdir, "synthetic_B056.fastq.gz")
    write_fastq_gz(fq, recs)
    oix = orc.Index.build_single(os.path.join(workdir, "ref_file.txt"), 750000, 4, 27)
    rows, reads = test_sh_flow(workdir, fq, "xxh3_v08")
    # search -f 1 (batch_search_pe.rs:26-84, reports.rs:8-48) from the oracle
    km = orc.kmers_from_fq_qual(fq, 27, 15).clean_map(1)
    keys, cnt = km.keys(), km.counts()
    hits, nu, sf, uc = oix.search_count(keys, cnt)
    modes = orc.unique_modes(uc, cnt, oix.n_colors)
    want = oix.generate_report(fq, hits, nu, sf, modes, len(km), 0.35)
    assert sorted("\t".join(r) for r in rows) == sorted(l for l in want.splitlines() if l)
    b056 = [r for r in rows if r[2] == "Listeria_phage_B056"]
    assert len(b056) == 1 and float(b056[0][3]) > 0.8                       # the phage the reads came from is covered
    assert len(reads) == 20000 and sum(1 for r in reads if r[1] == "Listeria_phage_B056") > 5000


def test_golden_row_of_test_sh(workdir):
    fq = real_fastq()
    if fq is None:
        pytest.skip("test_data/SRR548019.fastq.gz is not in the reference tree: set COLORID_TEST_FASTQ (or drop the file into "
                    "tests/golden/) to decide hash parity with the Rust binary")
    got, matching = {}, []
    for v in VARIANTS:
        rows, _ = test_sh_flow(workdir, fq, v)
        b056 = [r[1:] for r in rows if len(r) == 7 and r[2] == "Listeria_phage_B056"]
        got[v] = b056
        if b056 and b056[0] == GOLDEN_ROW:
            matching.append(v)
    print("hash variants reproducing test.sh:37:", matching)
    assert matching, f"no available hash variant reproduces test.sh:37 {GOLDEN_ROW}: got {got}"
