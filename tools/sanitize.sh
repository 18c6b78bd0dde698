#!/bin/bash
# `make sanitize` (SURVEY.md section 5 "Race detection / sanitizers"; the reference has none, /root/reference/CMakeLists.txt:43-45):
# the library's HOST code (record_host.cpp, host_workers.h, the host half of selfplay_host.hip / train_capi.hip / mcts_capi.hip) and the
# pybind11 boundary (pybind_elf.cc) are rebuilt with -fsanitize=address,undefined and, separately, -fsanitize=thread (clang's runtime,
# preloaded into python), the CPU tests that exercise them are run (the plain-C link test is left out: gcc cannot link against a
# library that needs clang's sanitizer runtime), and the normal build is restored.  Device code is not instrumented:
# GPU ASan needs xnack+ code objects (HSA_XNACK=1), which this pool's GPU boxes refuse.  Output: profiles/r06_sanitizers.txt
set -u
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
OUT=${1:-profiles/r06_sanitizers.txt}
TESTS="tests/test_records.py tests/test_reader_queues.py tests/test_wire_formats.py tests/test_pybind_boundary.py tests/test_abi.py tests/test_compat.py tests/test_sgf.py tests/test_stl_emul.py"
: > $OUT
run() {   # $1 = -fsanitize list, $2 = runtime name, $3.. = env
  local san=$1 rt=$2; shift 2
  echo "==== -fsanitize=$san (host code of libelf_amd.so + pybind11 boundary, clang $($CLANG --version | head -1 | sed 's/.*version //'))" >> $OUT
  make -C elf_amd/csrc clean > /dev/null
  if ! make -C elf_amd/csrc -j8 SAN=$san CXX=$CLANG > /tmp/san_build.log 2>&1; then echo "BUILD FAILED" >> $OUT; tail -20 /tmp/san_build.log >> $OUT; return; fi
  local lib=$($CLANG -print-file-name=libclang_rt.$rt-x86_64.so)
  rm -f /tmp/elf_san_log.*
  env "$@" LD_PRELOAD=$lib timeout 3000 python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi.py::test_header_is_plain_c_and_links_from_c > /tmp/san_run.log 2>&1
  echo "exit code $?" >> $OUT
  tail -4 /tmp/san_run.log >> $OUT
  echo "-- sanitizer reports (ERROR / WARNING / runtime error lines, first 40):" >> $OUT
  cat /tmp/elf_san_log.* 2>/dev/null > /tmp/san_reports.log      # the runtimes write to log_path (pytest captures the tests' stderr)
  grep -E "ERROR: |WARNING: ThreadSanitizer|runtime error:|SUMMARY: " /tmp/san_run.log /tmp/san_reports.log | sed 's/^[^:]*://' | sort | uniq -c | sort -rn | head -40 >> $OUT
  echo "-- reports located in THIS library (libelf_amd.so / _elf*.so): $(grep -E "SUMMARY: " /tmp/san_reports.log | grep -cE "libelf_amd|_elf[a-z_]*\\.cpython")    in the uninstrumented reference build the tests load as their checker (oracle/_ref/*.so, the reference's own comm::CommInternalT): $(grep -E "SUMMARY: " /tmp/san_reports.log | grep -c "oracle/_ref")" >> $OUT
  echo "-- first report in full:" >> $OUT
  awk '/WARNING: ThreadSanitizer|ERROR: AddressSanitizer|runtime error:/{p=1} p{print} /SUMMARY:/{if(p) exit}' /tmp/san_reports.log | head -60 >> $OUT
  echo >> $OUT
}
run address,undefined asan ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path=/tmp/elf_san_log UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/elf_san_log
run thread tsan TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:second_deadlock_stack=1:log_path=/tmp/elf_san_log
make -C elf_amd/csrc clean > /dev/null
make -C elf_amd/csrc -j8 > /dev/null 2>&1 && echo "normal build restored" >> $OUT
cat $OUT
