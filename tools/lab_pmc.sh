#!/bin/bash
# usage (on the GPU box): tools/lab_pmc.sh <V> <variant substring>   -> SQ counters of one lab variant, four passes
V=$1; only=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LAB_LIB=$R/m4ri-rust_amd/lib/libm4ri_hip.so
export LAB_ONLY="$only" LAB_PROFILE=1
tools/pmc_run.sh lab_a "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES" $R/tools/lpn_lab $V
tools/pmc_run.sh lab_b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_ANY" $R/tools/lpn_lab $V
tools/pmc_run.sh lab_c "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" $R/tools/lpn_lab $V
tools/pmc_run.sh lab_d "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" $R/tools/lpn_lab $V
