#!/bin/bash
# AddressSanitizer over the HOST side of libmidenhip (the verifier, the blob and byte parsers, the statement layers, the kernel generator):
# every translation unit compiled --offload-host-only with -fsanitize=address (GPU ASan is not available; the device code is not built at
# all: the nine fat-binary symbols are stubbed), linked into /tmp/libmidenhip_asan.so, then the host-only tests and the hostile-input loop of
# tests/test_abi_robustness.py run against it (MIDENHIP_LIB) with the ASan runtime preloaded into python.  No GPU.  ~3 minutes.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/miden-vm_amd/csrc
B=/tmp/mh_build_asan; mkdir -p $B
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
for f in ctx.cpp air.cpp air_jit.cpp verifier.cpp miden.cpp precompile.cpp comm_rccl.cpp comm_local.cpp ntt.hip lmcs.hip quotient.hip logup.hip deep.hip fri.hip prover.hip capi.hip; do
  b=$(basename $f | sed 's/\..*//')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-host-only -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Wno-unused-result -x hip -c $f -o $B/$b.o &
done
wait
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -c p2_host_simd.cpp -o $B/p2_host_simd.o
rm -f $B/zz_fatbin_stubs.o
/opt/rocm/bin/hipcc --offload-host-only -shared -fPIC -fsanitize=address -shared-libasan -o /tmp/libmidenhip_asan_nostub.so $B/*.o -lhiprtc -ldl
nm -u /tmp/libmidenhip_asan_nostub.so | grep hip_fatbin | awk '{print "__attribute__((visibility(\"default\"))) const char " $2 "[64] = {0};"}' > $B/fatbin_stubs.c
gcc -fPIC -c $B/fatbin_stubs.c -o $B/zz_fatbin_stubs.o
/opt/rocm/bin/hipcc --offload-host-only -shared -fPIC -fsanitize=address -shared-libasan -o /tmp/libmidenhip_asan.so $B/*.o -lhiprtc -ldl
cd $ROOT
export MIDENHIP_LIB=/tmp/libmidenhip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
MH_ROBUST_N=${MH_ROBUST_N:-1500} python -m pytest -x -q tests/test_abi_robustness.py tests/test_fuzz_host.py tests/test_verifier_cpu.py tests/test_proof_structure.py tests/test_abi.py \
  tests/test_host_compress_simd.py tests/test_ref_lifted_stark.py tests/test_session_c_abi.py -p no:cacheprovider 2>&1 | tail -15
