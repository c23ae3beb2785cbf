cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_b2.txt
echo "# rocprofv3 --kernel-trace --pmc <counter> -- python tools/prof_dominant.py fp16 20 2   (one counter per pass; batch_frames=2 -> tile 128x64)" > $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc; timeout 40 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py fp16 20 2 > /tmp/pmc.log 2>&1; rc=$?; echo "group '$grp' rc=$rc"
  if [ $rc -ne 0 ]; then echo "# '$grp': rocprofv3 rc=$rc (timeout 40 s)" >> $O; continue; fi
  timeout 20 python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $O
done
cat $O
