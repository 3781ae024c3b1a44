#!/bin/bash
# Kernel-tuning builds: tools/build_variants.sh name "-DNB_THREADS=640 ..." [name flags]...
# -> ddcmd_amd/lib/variants/libddcmi_<name>.so ; select with DDCMI_LIB=<path>.
set -e
cd "$(dirname "$0")/../ddcmd_amd/csrc"
make -s
mkdir -p ../lib/variants build/var
while [ $# -ge 2 ]; do
   name=$1; flags=$2; shift 2
   for f in ddcmi scan bonded; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -I../../include -Ihip $flags -c hip/$f.hip -o build/var/${name}_$f.o &
   done
   wait
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/variants/libddcmi_$name.so build/host/*.o build/var/${name}_ddcmi.o build/var/${name}_scan.o build/var/${name}_bonded.o -L/opt/rocm/lib -lrccl -lm -Wl,-rpath,/opt/rocm/lib
   echo built $name
done
