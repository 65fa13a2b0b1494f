#!/bin/bash
# usage: tools/profile_r06.sh <outdir>   -- round-6 evidence (the round-4 passes in the library's default arithmetic -- the fp16 split since round 5 -- plus the same G16V passes in the bf16 split).  Every pass under its own timeout; stops at the first failure.
#  1. the default bench command (2 parts of 64 frames on two streams, one hipGraph): the line, rocprofv3 kernel-trace stats
#  2. the same kernels alone on the chip (--streams 1 --batch 128): kernel-trace stats (what the line's per-kernel attribution times), HBM
#     traffic counters (separate passes), SQ counters of every step kernel
#  3. the other BASELINE configurations: bench line + kernel-trace stats + HBM traffic counters each
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
S="python3 $R/tools/summarize_rocprof.py"
# PART=1: the G16V passes only; PART=2: the other configurations only (a gpurun call is at most 20 minutes)
if [ "$PART" != 2 ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_default -- $B --steps 5 --warmup 2 > $R/$OUT/stats_default.log 2>&1 || exit 1
$S stats $R/$OUT/stats_default $R/$OUT/bench_default_kernel_stats.txt > /dev/null || exit 1
rm -rf $R/$OUT/stats_default
B1="$B --streams 1 --batch 128"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_s1 -- $B1 --steps 5 --warmup 2 > $R/$OUT/stats_s1.log 2>&1 || exit 1
$S stats $R/$OUT/stats_s1 $R/$OUT/bench_streams1_b128_kernel_stats.txt > /dev/null || exit 1
rm -rf $R/$OUT/stats_s1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- $B1 --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- $B1 --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/pmc_write.log 2>&1 || exit 1
PMC_FRAMES_PER_LAUNCH=128 $S pmc $R/$OUT/pmc_traffic_G16V.json $R/$OUT/pmc_fetch $R/$OUT/pmc_write > /dev/null || exit 1
rm -rf $R/$OUT/pmc_fetch $R/$OUT/pmc_write
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU"
timeout -k 10 300 rocprofv3 --pmc $SQ1 --output-format csv -d $R/$OUT/sq1 -- $B1 --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc $SQ2 --output-format csv -d $R/$OUT/sq2 -- $B1 --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/sq2.log 2>&1 || exit 1
python3 $R/tools/summarize_sq.py $R/$OUT/pmc_step_kernels.txt "B=64 step, one stream" "conv3d;sweep;softargmin;up2" $R/$OUT/sq1 $R/$OUT/sq2 > /dev/null || exit 1
rm -rf $R/$OUT/sq1 $R/$OUT/sq2
# the bf16 split of the same step (MVSGI_CONV_MODE=bf16x3, the default of rounds 1-4): line, kernel-trace stats alone on the chip
BF="$B --mode bf16x3"
timeout -k 10 280 python3 $R/bench.py --mode bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/$OUT/bench_G16V_bf16x3.json 2> $R/$OUT/bench_G16V_bf16x3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_bf16 -- $BF --streams 1 --batch 128 --steps 5 --warmup 2 > $R/$OUT/stats_bf16.log 2>&1 || exit 1
$S stats $R/$OUT/stats_bf16 $R/$OUT/bench_streams1_b128_bf16x3_kernel_stats.txt > /dev/null || exit 1
rm -rf $R/$OUT/stats_bf16
fi
[ "$PART" = 1 ] && exit 0
for T in G16VV E8 4cam-32 E16-48-96; do
  PB=32; [ $T = E8 ] && PB=64; [ $T = 4cam-32 ] && PB=16
  C="$B --config $T --streams 1 --batch $PB"
  timeout -k 10 280 python3 $R/bench.py --config $T --batch $((2 * PB)) --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$OUT/bench_$T.json 2> $R/$OUT/bench_$T.err || exit 1
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -- $C --steps 4 --warmup 2 > $R/$OUT/stats_$T.log 2>&1 || exit 1
  $S stats $R/$OUT/stats_$T $R/$OUT/${T}_kernel_stats.txt > /dev/null || exit 1
  rm -rf $R/$OUT/stats_$T
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pf_$T -- $C --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/pf_$T.log 2>&1 || exit 1
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pw_$T -- $C --steps 2 --warmup 1 --settle-seconds 0 > $R/$OUT/pw_$T.log 2>&1 || exit 1
  PMC_FRAMES_PER_LAUNCH=$PB $S pmc $R/$OUT/pmc_traffic_$T.json $R/$OUT/pf_$T $R/$OUT/pw_$T > /dev/null || exit 1
  rm -rf $R/$OUT/pf_$T $R/$OUT/pw_$T
done
