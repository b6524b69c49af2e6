cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc64; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
# SQ and GRBM passes only: a TA_* counter pass did not come back within 25 minutes on this pool
export SIZES=64000000 CONFIGS=1024:0:4
p() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 tools/kbench.py > $O/$n.log 2>&1; }
p sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
p sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
p sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU
p grbm GRBM_GUI_ACTIVE GRBM_COUNT
tail -2 $O/*.log
