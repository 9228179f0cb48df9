# PMC passes over the wave-specialised convolution kernel on single ResBlocks (separate --pmc passes; kernel-trace only).
#   gpurun -- 'bash tools/pmc_ws.sh [tag]'      -> gpurun_out/<tag>/pmc_<shape>.csv
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_ws}; mkdir -p $O
for shape in "128 128 16000 64" "64 64 64000 64"; do
tag=$(echo $shape | tr ' ' '_')
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" "TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES TCP_TCP_LATENCY TCP_TA_TCP_STATE_READ" \
           "TCC_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o pmc -- python $R/tools/run_resblock.py $shape 3 > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
done
python $R/tools/pmc_summary.py /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 /tmp/pmc_4 /tmp/pmc_5 /tmp/pmc_6 /tmp/pmc_7 /tmp/pmc_8 > $O/pmc_$tag.csv
python - <<PY
import csv
rows=list(csv.reader(open("$O/pmc_$tag.csv")))
h=rows[0]
for r in rows[1:]:
    if r[0].startswith("conv_ws"):
        for k,v in zip(h,r): print(f"{k:36s} {v}")
PY
done
