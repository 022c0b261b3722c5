cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq$i -o pmc -- python $R/scripts/pmc_probe.py 40 > /dev/null 2> $R/gpurun_out/pmc_sq$i.err
  echo "set $i rc=$?"
done
