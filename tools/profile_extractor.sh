R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof_ext
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ext/stats -- python3 $R/tools/extractor_probe.py 32 5 > $R/gpurun_out/prof_ext/stats.log 2>&1
