#!/bin/bash
# Timing variants of the PEG-16000 fused kernel (results wrong): build_ablate/libldpc_hip_abl<N>.so
# with -DLDPC_FUSED_ABLATE=N in the translation unit that holds fused_q88_reg_kernel<2,16,...>.
set -e
cd "$(dirname "$0")/../../ldpc_amd/csrc"
mkdir -p ../../build_ablate
for N in "$@"; do
  # N = T: the timing build (cycle counters per phase), otherwise an ablation mask
  # N = P<k>: timing build with the issue-priority experiment k
  # N = D<defs>: plain build with the given -D definitions, comma separated (D LDPC_FUSED_CH=4,LDPC_FUSED_BCH=16)
  if [ "${N#D}" != "$N" ]; then DEF="-D$(echo ${N#D} | sed 's/,/ -D/g')"; elif [ "$N" = T ]; then DEF="-DLDPC_FUSED_TIMING=1"; elif [ "${N#P}" != "$N" ]; then DEF="-DLDPC_FUSED_TIMING=1 -DLDPC_FUSED_PRIO=${N#P}"; else DEF="-DLDPC_FUSED_ABLATE=$N"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -DFQ_PART=2 $DEF -c fused_inst.hip -o ../../build_ablate/fused_inst_2_abl$N.o &
done
wait
for N in "$@"; do
  OBJS=$(ls build/*.o | grep -v fused_inst_2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ablate/libldpc_hip_abl$N.so $OBJS ../../build_ablate/fused_inst_2_abl$N.o
done
ls -la ../../build_ablate/*.so
