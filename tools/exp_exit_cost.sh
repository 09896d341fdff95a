# what the process pays AFTER its last useful instruction (run on the GPU box after tools/e2e_demo.py): timestamps of exit_group against the
# moment the shell gets the process back, for read_id with and without the index mapping and with smaller pieces of state
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
t() {
  local label=$1; shift
  for rep in 1 2; do
    local s=$(date +%s.%N)
    env "$@" > /dev/null 2> /tmp/cid_e2e/exit.err
    local e=$(date +%s.%N)
    local ret=$(grep -o "subcommand returned [0-9]* ms (at [0-9]* ms)" /tmp/cid_e2e/exit.err | grep -o "at [0-9]*" | grep -o "[0-9]*")
    python3 -c "print('$label: wall %.0f ms, subcommand returned at ${ret:-0} ms, rest %.0f ms' % (($e - $s) * 1e3, ($e - $s) * 1e3 - ${ret:-0}))"
  done
}
t "read_id bgzf               " COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_x
t "read_id bgzf no index mmap " COLORID_TIMING=1 COLORID_INDEX_MMAP=0 $BIN read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_x
t "read_id bgzf host front end" COLORID_TIMING=1 COLORID_DEVICE_FASTQ=0 $BIN read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_x
t "search -s one genome       " COLORID_TIMING=1 $BIN search -b $W/idx.bxi -q $W/g007.fasta -s
t "python: hipInit + 3 GB hipMalloc + exit" python3 -c "
import ctypes, os
h = ctypes.CDLL('libamdhip64.so'); p = ctypes.c_void_p()
h.hipMalloc(ctypes.byref(p), ctypes.c_size_t(3 << 30)); h.hipMemset(p, 0, ctypes.c_size_t(3 << 30)); h.hipDeviceSynchronize()
import sys; sys.stderr.write('timing: subcommand returned 0 ms (at 0 ms)\n'); os._exit(0)"
t "python: hipInit only + exit" python3 -c "
import ctypes, os
h = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(); h.hipGetDeviceCount(ctypes.byref(n)); h.hipSetDevice(0); p = ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20))
os._exit(0)"
t "python: nothing" python3 -c "import os; os._exit(0)"
