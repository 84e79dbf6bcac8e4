#!/bin/bash
# register / scratch / LDS use of the kernels of one HIP source (gfx950):  tools/kres.sh obgc_ecosys.hip [name filter]
cd "$(dirname "$0")/../ocean-bgc_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -fno-fast-math -ffp-contract=off \
  -Xclang -target-feature -Xclang -packed-fp32-ops -mllvm -disable-machine-licm -Wall -Wno-unused-function $EXTRA -c "$1" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs Spill" \
  | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//; s/Function Name: //' | paste - - - - - - - | grep -E "${2:-.}" \
  | sed 's/_ZN4obgc12_GLOBAL__N_1[0-9]*//; s/ENS_13BgcKernelArgs.*Params[E]*//; s/EEvNS_13Bgc[^ \t]*//'
