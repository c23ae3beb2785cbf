cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_b2.txt
echo "# RTP_DIAG_SKIP_POST=2 rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_dominant.py fp16 20 2   (one group per pass; batch_frames=2 -> tile 128x64)" > $O
export RTP_DIAG_SKIP_POST=2
for grp in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc; timeout 60 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc -- python $R/tools/prof_dominant.py fp16 20 2 > /tmp/pmc.log 2>&1; rc=$?; echo "group '$grp' rc=$rc"
  if [ $rc -ne 0 ]; then echo "# group '$grp': rocprofv3 rc=$rc" >> $O; [ $rc -eq 124 ] && break; continue; fi
  timeout 20 python $R/tools/pmc_summary.py /tmp/pmc conv_ring >> $O
done
cat $O
