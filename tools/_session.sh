set -o pipefail
cd $GRAFT_REPO_ROOT
python tools/bench_scales.py > /dev/null 2>&1
export PMC_SCRIPT=tools/bench_scales.py
bash tools/pmc_pass.sh v_sq1 real SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY > /dev/null
bash tools/pmc_pass.sh v_sq2 real SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH > /dev/null
bash tools/pmc_pass.sh v_tcp real TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum > /dev/null
bash tools/pmc_pass.sh v_fetch real FETCH_SIZE > /dev/null
bash tools/pmc_pass.sh v_write real WRITE_SIZE > /dev/null
bash tools/pmc_pass.sh v_grbm real GRBM_GUI_ACTIVE > /dev/null
echo done
