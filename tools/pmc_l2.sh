#!/bin/bash
# usage: tools/pmc_l2.sh <outdir> -- <abs python script and args>   (L2 / fabric counters only; stops at the first failure)
OUT=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 60 rocprofv3 --pmc TCC_HIT TCC_MISS TCC_REQ --output-format csv -d $R/$OUT/p1 -- python3 "$@" > $R/$OUT/p1.log 2>&1 || exit 1
timeout -k 5 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/p2 -- python3 "$@" > $R/$OUT/p2.log 2>&1 || exit 1
