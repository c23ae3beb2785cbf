#!/bin/bash
# Per-launch HBM traffic and MFMA pipe utilisation of the whole conv plan: four rocprofv3 PMC passes (one counter each, --kernel-trace only) over whole batches one at a time.
#   tools/pmc_steps.sh B N_SCALES PREC MODEL OUTFILE
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
B=$1; N=$2; P=$3; M=$4; OUT=$5
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/pmcs_$grp
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs_$grp -- python $R/tools/run_batches.py $B $N $P 6 $M > /tmp/pmcs_$grp.log 2>&1
  tail -1 /tmp/pmcs_$grp.log
done
( echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES (one per pass) -- python tools/run_batches.py $B $N $P 6 $M ; python tools/pmc_steps.py ..."
  python $R/tools/pmc_steps.py /tmp/pmcs_FETCH_SIZE /tmp/pmcs_WRITE_SIZE $B $N $P $M /tmp/pmcs_SQ_VALU_MFMA_BUSY_CYCLES /tmp/pmcs_SQ_BUSY_CYCLES ) > $R/$OUT 2>&1
head -3 $R/$OUT; tail -2 $R/$OUT
head -2 $(find /tmp/pmcs_FETCH_SIZE -name "*counter_collection.csv" | head -1)
