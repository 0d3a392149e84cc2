#!/bin/bash
# Sanitizer run of the HOST code: builds oracle/_asan and orb_slam2_amd/_asan (+ tests/cpp/_asan) with
# -fsanitize=address,undefined (ROCm's clang, one shared runtime) and runs the test suite on them.
#   tools/run_asan.sh            CPU tests (-m "not gpu"): the oracle, the C-ABI's host-only entry points, argument errors
#   tools/run_asan.sh --gpu      additionally, on a box with a GPU: the C++ drop-in classes through their drivers
#                                (tests/test_gpu_adapter.py, tests/test_matcher_class.py, the adapter half of
#                                tests/test_guided_search.py) -- host code instrumented, device code as shipped
# (ThreadSanitizer was tried for the concurrent ORBmatcher test and cannot be used: under TSAN's address-space layout the HIP
#  runtime finds no device -- hipGetDeviceCount() returns 0 in the instrumented driver, profiles/r3_asan.txt.)
# Leak checking is off: CPython and the HIP runtime keep allocations for the life of the process.
set -eo pipefail
cd "$(dirname "$0")/.."
make -C oracle asan
make -C orb_slam2_amd/csrc asan
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export ORBFE_LIB=$PWD/orb_slam2_amd/_asan/liborbfe.so ORB_ORACLE_LIB=$PWD/oracle/_asan/liborb_oracle.so
export ORBFE_TEST_BIN_DIR=$PWD/tests/cpp/_asan
if [ "$1" = "--gpu" ]; then
    # the drivers carry the sanitizer runtime themselves; python stays uninstrumented (it only prepares files and compares)
    unset ORBFE_LIB ORB_ORACLE_LIB
    # No quarantine: freed device allocations (the grid pool gives grids back with hipFree) must not sit in ASAN's quarantine
    # until exit -- ROCm's runtime then recycles them from inside its own teardown (__cxa_finalize -> libhsa -> operator delete)
    # after it has marked the device allocator unloaded, and aborts with "CHECK failed: sanitizer_allocator_device.h:125".
    export ASAN_OPTIONS=$ASAN_OPTIONS:quarantine_size_mb=0:thread_local_quarantine_size_kb=0
    export LD_LIBRARY_PATH=$(dirname "$RT")${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
    python -u -m pytest tests/test_gpu_adapter.py tests/test_matcher_class.py tests/test_guided_search.py \
        -x -q -m gpu --timeout 600 --timeout-method thread
else
    LD_PRELOAD=$RT python -u -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
fi
