#!/bin/bash
# AddressSanitizer + UBSan run of the HOST code of libprealps_hip.so (partitioner, nested dissection
# + multifrontal factorisation self-check, operator build / halo plans in plan-only mode) through the
# CPU test suite.  GPU sanitizers are not available on the pool; the device objects are linked in as
# they are.  Run from the repo root in the build container:  bash tools/asan_cpu.sh
set -e -o pipefail
D=$(mktemp -d)
cd prealps_amd/csrc
make -s
for f in context operator partition block_jacobi mpi_glue nd ecg dense_ops smalldense; do
  gcc -O1 -g -fPIC -std=gnu11 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I. -c $f.c -o $D/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libprealps_hip.so build/kernels.o build/spmm.o build/bj_g4.o build/bj_g4_nt14.o build/bj_g4_nt16.o build/nd_factor.o build/runtime.o build/comm_rccl.o $D/*.o \
  -fopenmp -lgomp -lm -ldl -fsanitize=address,undefined
cd ../..
cp prealps_amd/libprealps_hip.so $D/orig.so
trap 'cp $D/orig.so prealps_amd/libprealps_hip.so' EXIT
cp $D/libprealps_hip.so prealps_amd/libprealps_hip.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests/test_partition_cpu.py tests/test_nd_cpu.py tests/test_distributed_cpu.py tests/test_library_cpu.py -q \
  -k "not links_unchanged and not own_c_driver"
