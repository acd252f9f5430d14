#!/bin/bash
# tools/sanitize/run.sh -- one ASan + UBSan pass over the host-side code that parses untrusted bytes or runs the parity oracle
# (VERDICT r2 item 9).  Runs on any machine with g++ (no GPU): from the repo root
#     bash tools/sanitize/run.sh > profiles/r03_sanitizer.txt 2>&1
#   1. the plugin's `.spz` host parser (rcppml_amd/csrc/spz_parse.hpp) under a corpus of damaged files (spz_parse_fuzz.cpp)
#   2. the CPU oracle and the `.spz` oracle rebuilt with -fsanitize=address,undefined (make -C oracle sanitize) and the CPU test
#      files that exercise them run against those builds (RCPPML_ORACLE_DIR selects the directory; libasan is preloaded into python)
set -u
cd "$(dirname "$0")/../.."
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g -O1"
mkdir -p tools/sanitize/build
g++ -std=c++17 $SAN tools/sanitize/spz_parse_fuzz.cpp -o tools/sanitize/build/spz_parse_fuzz || exit 1
python - <<'PY' || exit 1
import numpy as np, os
d = "tools/sanitize/build/corpus"; os.makedirs(d, exist_ok=True)
z = np.load("tests/golden/spz_vectors.npz", allow_pickle=False)
n = 0
for k in z.files:
    a = z[k]
    if a.dtype == np.uint8 and a.ndim == 1 and a.size >= 128 and bytes(a[:4]) == b"SPRZ":
        a.tofile(os.path.join(d, k + ".spz")); n += 1
print("corpus: %d reference-written files from tests/golden/spz_vectors.npz + tests/golden/pbmc3k.spz" % n)
PY
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 tools/sanitize/build/spz_parse_fuzz tests/golden/pbmc3k.spz tools/sanitize/build/corpus/*.spz || { echo "FAILED: spz parser under sanitizers"; exit 1; }
make -C oracle sanitize || exit 1
ASAN_LIB=$(g++ -print-file-name=libasan.so)
RCPPML_ORACLE_DIR=$(pwd)/oracle/_san LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 OMP_NUM_THREADS=4 \
    python -m pytest tests/test_oracle.py tests/test_spz_cpu.py tests/test_data_cpu.py -q -x -p no:cacheprovider 2>&1 | tail -5
echo "sanitizer pass done (exit code of the pytest leg: ${PIPESTATUS[0]})"
