# The C++ host layer and its test under AddressSanitizer + UBSan on the CPU (no device: the no-device half of tests/cpp/host_test.cpp --
# worker pool, staging fallback, Function sampling, buffers, .flan files).  GPU sanitizers are not available on this pool.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/flan_host_test_asan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/flan_amd/host $R/flan_amd/host/*.cpp $R/tests/cpp/host_test.cpp \
    -o $OUT -L$R/flan_amd -lflanhip -Wl,-rpath,$R/flan_amd -lpthread
ASAN_OPTIONS=detect_leaks=0 $OUT --no-device | tail -3
