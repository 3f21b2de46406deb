#!/bin/bash
# AddressSanitizer + UBSan on the CPU builds (GPU sanitizers are not available on the pool): the oracle under the KAT and
# C-ABI tests, and the host CLI (TOML, FASTA/FASTQ ingest, config reader, drivers) under its CPU tests.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp $R/oracle/libibf_oracle.so $T/oracle.so; cp $R/readbouncer_amd/readbouncer_amd_cli $T/cli
trap 'cp $T/oracle.so $R/oracle/libibf_oracle.so; cp $T/cli $R/readbouncer_amd/readbouncer_amd_cli; touch $R/oracle/libibf_oracle.so' EXIT
gcc -O1 -g -std=c11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -D_POSIX_C_SOURCE=200809L \
    -shared -o $R/oracle/libibf_oracle.so $R/oracle/ibf_oracle.c -lm -lpthread
touch $R/oracle/libibf_oracle.so
(cd $R && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_oracle_kat.py tests/test_capi_cpu.py -x -q)
(cd $R/readbouncer_amd/host && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer rb_main.cpp \
    -o ../readbouncer_amd_cli -L.. -lreadbouncer_amd -Wl,-rpath,'$ORIGIN' -lpthread)
(cd $R && ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_cli.py -x -q -m "not gpu")
# ThreadSanitizer on the parallel ingest (parser threads, ordered hand-over, block pool): 60 segments, 4 threads
python3 - "$T/tsan.fq" <<'PY'
import sys, random
random.seed(1)
with open(sys.argv[1], "w") as fh:
    for i in range(30000):
        n = random.randint(1, 900)
        fh.write("@r%d\n%s\n+\n%s\n" % (i, "".join(random.choice("ACGTN") for _ in range(n)), "".join(random.choice("@>+I") for _ in range(n))))
PY
(cd $R/readbouncer_amd/host && g++ -O1 -g -std=c++17 -fsanitize=thread rb_main.cpp -o $T/cli_tsan -L.. -lreadbouncer_amd \
    -Wl,-rpath,$R/readbouncer_amd -lpthread)
TSAN_OPTIONS="halt_on_error=1" $T/cli_tsan --ingest-threads 4 --segment-bytes 250000 --parse-stats $T/tsan.fq
# the ingest's failure paths (refused / throwing page-locked allocator) under ASan+UBSan and under TSan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer $R/tests/cpp/test_seqio.cpp -o $T/seqio_asan -lpthread
$T/seqio_asan
g++ -O1 -g -std=c++17 -fsanitize=thread $R/tests/cpp/test_seqio.cpp -o $T/seqio_tsan -lpthread
TSAN_OPTIONS="halt_on_error=1" $T/seqio_tsan
# the pool's work queues (per-worker FIFOs, jobs on the callers' stacks, least-loaded pick) under TSan and ASan+UBSan
g++ -O1 -g -std=c++17 -fsanitize=thread $R/tests/cpp/test_workq.cpp -o $T/workq_tsan -lpthread
TSAN_OPTIONS="halt_on_error=1" $T/workq_tsan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer $R/tests/cpp/test_workq.cpp -o $T/workq_asan -lpthread
$T/workq_asan
echo "sanitizers: clean"
