#!/bin/bash
# Measurement aid (profiles/r06/flake.md): code objects of the FAILING build of tools/sweep_probe.hip (MODE 1, the compiler's vectorisers
# on) with chosen packed instructions rewritten as two plain ones (tools/depack_pk.py).  tools/build_hsaco.sh <name>="<depack args>" ...
# -> tools/sweep_hsaco/<name>.hsaco (git-ignored; they travel to the GPU box).  "orig" = the assembly as compiled.
set -euo pipefail
cd "$(dirname "$0")"; L=/opt/rocm/lib/llvm/bin; d=sweep_hsaco; mkdir -p $d
hipcc -O3 --offload-arch=gfx950 -DMODE=${MODE:-1} -S --cuda-device-only -o $d/base.s sweep_probe.hip 2>/dev/null
for spec in "$@"; do
  name=${spec%%=*}; args=${spec#*=}
  if [ "$name" = orig ]; then cp $d/base.s $d/$name.s; else python depack_pk.py $d/base.s $d/$name.s $args; fi
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $d/$name.s -o $d/$name.o && $L/ld.lld -shared $d/$name.o -o $d/$name.hsaco && rm $d/$name.o
done
