#!/bin/bash
# usage: tools/profile_r03.sh <outdir>   -- round-3 evidence: kernel-trace stats of bench.py, HBM traffic counters (separate passes), and
# the SQ counters (MFMA busy / stalls / LDS) of the step with out_costs.0 in polyphase form and in the round-2 form (MVSGI_POLY=0).
# Every pass under its own timeout; stops at the first failure.
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- $B --steps 5 --warmup 2 > $R/$OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_write.log 2>&1 || exit 1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU"
timeout -k 10 300 rocprofv3 --pmc $SQ1 --output-format csv -d $R/$OUT/sq1_poly -- $B --steps 2 --warmup 1 > $R/$OUT/sq1_poly.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc $SQ2 --output-format csv -d $R/$OUT/sq2_poly -- $B --steps 2 --warmup 1 > $R/$OUT/sq2_poly.log 2>&1 || exit 1
export MVSGI_POLY=0
timeout -k 10 300 rocprofv3 --pmc $SQ1 --output-format csv -d $R/$OUT/sq1_r02 -- $B --steps 2 --warmup 1 > $R/$OUT/sq1_r02.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc $SQ2 --output-format csv -d $R/$OUT/sq2_r02 -- $B --steps 2 --warmup 1 > $R/$OUT/sq2_r02.log 2>&1 || exit 1
