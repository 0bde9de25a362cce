#!/bin/bash
# SQ counter passes over one bench.py command line (run on the GPU box from the repo root):
#   tools/pmc_sq.sh <kernel name substring> <out dir> <bench.py arguments ...>
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
key=$1; out=$2; shift 2
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py "$@" --min-seconds 0 --profile-steps 2 > /dev/null 2> $out/p$i.err
done
python tools/pmc_sq.py "$key" $out/p1 $out/p2 $out/p3 $out/p4 | tee $out/summary.txt
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
