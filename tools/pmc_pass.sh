#!/bin/bash
# PMC passes (each counter group in its own run, --kernel-trace only) for a python tool.  usage: pmc_pass.sh <outdir> <script> [args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; pmc=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$name -o p -- python3 "$@" > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "$name done"; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
run fetch "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
run write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$@"
