#!/bin/bash
# tools/san_run.sh asan|tsan COMMAND...  -- runs COMMAND with the sanitizer runtime preloaded and STR_ER_LIB pointing at the sanitizer build
# (tools/san_build.sh).  Python itself is not instrumented: leak checking is off, everything else is reported for the library's own frames.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
kind=$1; shift
rt=$kind; [ $kind = ubsan ] && rt=ubsan_standalone
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.$rt-x86_64.so | head -1)
export STR_ER_LIB=$ROOT/scene-text-recognition_amd/lib/san/libstr_er_hip_$kind.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:log_path=${SAN_LOG:-/tmp/san_$kind}
export UBSAN_OPTIONS=print_stacktrace=1:log_path=${SAN_LOG:-/tmp/san_$kind}
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 log_path=${SAN_LOG:-/tmp/san_$kind} suppressions=$ROOT/tools/tsan.supp"
LD_PRELOAD=$RT${LD_PRELOAD:+:$LD_PRELOAD} "$@"
