#!/bin/bash
# Host-side ASan + UBSan run of the library (make -C exp-trmf-nips16_amd asan) on a GPU box: the ABI tests, the golden parity
# cases, the session life cycle (create / run / append_rows / destroy), and (round 6) the one-shot path with its pool / pinned ring / copy
# threads, the in-process session groups (TRMF_DEVICES), the split path of long rows and the multi-process tests, under the sanitizers.  usage: scripts/asan_gpu.sh <outfile>
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=${1:-/dev/stdout}
# SAN=undefined build (make -C exp-trmf-nips16_amd asan SAN=undefined): preload the UBSan runtime; SAN=address,undefined: the ASan one
if grep -q asan_init $R/exp-trmf-nips16_amd/build/asan/trmf_float32.so 2>/dev/null; then RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
else RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1); fi
cd $R
TRMF_CORELIB_DIR=$R/exp-trmf-nips16_amd/build/asan LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1 timeout 2400 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_python_frontend.py tests/test_gpu_oneshot.py tests/test_gpu_devices.py tests/test_gpu_split.py \
  tests/test_dist.py -m gpu -x -q -k "not full_size and not c3full and not config4 and not c5s and not forced_split_every" > $OUT 2>&1
echo "asan pytest exit $?" >> $OUT
grep -c "ERROR: AddressSanitizer\|runtime error:" $OUT
