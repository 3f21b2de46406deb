#!/bin/bash
# AddressSanitizer + UBSan (+ ThreadSanitizer where threads meet) on the CPU builds (GPU sanitizers are not available on the pool): the
# oracle AND the product library's own host code under the KAT, C-ABI and reference-threshold tests, the host CLI (TOML, FASTA/FASTQ
# ingest, config reader, drivers) under its CPU tests, and the threaded pieces -- ingest, work queues, the reader threads of a filter
# load, the live step -- in stand-alone harnesses.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp $R/oracle/libibf_oracle.so $T/oracle.so; cp $R/readbouncer_amd/readbouncer_amd_cli $T/cli
trap 'cp $T/oracle.so $R/oracle/libibf_oracle.so; cp $T/cli $R/readbouncer_amd/readbouncer_amd_cli; touch $R/oracle/libibf_oracle.so' EXIT
gcc -O1 -g -std=c11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -D_POSIX_C_SOURCE=200809L \
    -shared -o $R/oracle/libibf_oracle.so $R/oracle/ibf_oracle.c -lm -lpthread
touch $R/oracle/libibf_oracle.so
# ... and the PRODUCT's own host code: csrc/rb_host.cpp (with the reader threads of rb_io.h), rb_live.cpp and rb_pool.cpp instrumented,
# linked with the device objects as they are (hipcc-built; their host halves only run with a GPU) into a test library that the C-ABI
# tests load instead of the shipped one (RB_AMD_LIBRARY): file I/O against the oracle, the multi-threaded .ibf read, thresholds, build
# helpers, the null / zero contract, the no-device error paths
make -C $R/readbouncer_amd/csrc >/dev/null
for f in rb_host rb_live rb_pool; do
  g++ -O1 -g -std=c++17 -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer -c $R/readbouncer_amd/csrc/$f.cpp -o $T/$f.o
done
g++ -shared -fPIC -fsanitize=address,undefined $T/rb_host.o $T/rb_live.o $T/rb_pool.o $R/readbouncer_amd/csrc/rb_kernels.o $R/readbouncer_amd/csrc/rb_engine.o \
    $R/readbouncer_amd/csrc/rb_probe.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -o $T/libreadbouncer_amd_asan.so
(cd $R && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    RB_AMD_LIBRARY=$T/libreadbouncer_amd_asan.so python -m pytest tests/test_oracle_kat.py tests/test_capi_cpu.py tests/test_reference_ci.py -x -q)
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
# the reader threads of a filter load (csrc/rb_io.h: one gang per load, short and shrinking files, refused threads) and the live step's host
# logic (csrc/rb_live.cpp + rb_host.cpp, the engine call replaced by a stand-in at link time) under ASan+UBSan and under TSan
for t in io live; do
  src="$R/tests/cpp/test_$t.cpp"; [ $t = live ] && src="$src $R/readbouncer_amd/csrc/rb_live.cpp $R/readbouncer_amd/csrc/rb_host.cpp"
  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/readbouncer_amd/csrc $src -o $T/${t}_asan -lpthread
  $T/${t}_asan $T
  g++ -O1 -g -std=c++17 -fsanitize=thread -I$R/readbouncer_amd/csrc $src -o $T/${t}_tsan -lpthread
  TSAN_OPTIONS="halt_on_error=1" $T/${t}_tsan $T
done
echo "sanitizers: clean"
