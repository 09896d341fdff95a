# wall-clock of whole CLI runs on the 256-genome demo (2.8 GB .bxi, 1 M reads in a single-stream fastq.gz and in a block-gzip file):
# read_id, default search, search -g; 3 runs each.  After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) + tools/exp_batch_id.sh.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
wall() { local s=$(date +%s.%N); "$@" > /dev/null 2>&1; local e=$(date +%s.%N); python3 -c "print('%.2f s' % ($e - $s))"; }
for f in reads.fastq.gz reads.bgzf.fastq.gz; do
  for rep in 1 2 3; do echo "read_id [$f]: $(wall $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_w)"; done
  for rep in 1 2 3; do echo "search  [$f]: $(wall $BIN search -b $W/idx.bxi -q $W/$f -f 0 -p 0.005)"; done
  for rep in 1 2 3; do echo "search -g [$f]: $(wall $BIN search -b $W/idx.bxi -q $W/$f -g -f 0 -p 0.005)"; done
done
