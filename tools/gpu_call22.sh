#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r03_list_avail.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/r03_list_avail.txt
