#!/bin/bash
# SQ counters of the transposition kernel on 65536^2 (four passes of four counters; run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tools/pmc_run.sh tr_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" python3 $R/tools/transpose_pmc.py
tools/pmc_run.sh tr_b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" python3 $R/tools/transpose_pmc.py
tools/pmc_run.sh tr_c "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python3 $R/tools/transpose_pmc.py
tools/pmc_run.sh tr_d "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" python3 $R/tools/transpose_pmc.py
