#!/bin/bash
# SQ / GRBM counter passes on the isolated fused kernel (M = 2^20)
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out/pmc_sq1" "$R/gpurun_out/pmc_sq2"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u > "$R/gpurun_out/counters_list.txt"
wc -l "$R/gpurun_out/counters_list.txt"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq1" -- python3 "$R/tools/prof_fused.py" 4 > "$R/gpurun_out/pmc_sq1/out.log" 2>&1 || { tail -n 15 "$R/gpurun_out/pmc_sq1/out.log"; }
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq2" -- python3 "$R/tools/prof_fused.py" 4 > "$R/gpurun_out/pmc_sq2/out.log" 2>&1 || { tail -n 15 "$R/gpurun_out/pmc_sq2/out.log"; }
find "$R/gpurun_out/pmc_sq1" "$R/gpurun_out/pmc_sq2" -name "*counter_collection.csv" | head
