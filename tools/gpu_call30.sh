#!/bin/bash
# the same memory-side counters for the expansion kernel ALONE (build/expand_ctx_lab launches the product's launch_expand without anything beside it)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03mem_alone
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for p in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$i -o pmc -- $R/build/expand_ctx_lab > $O/pmc_$i.txt 2> $O/pmc_$i.err || { tail -5 $O/pmc_$i.err; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $O/pmc_1 $O/pmc_2 $O/pmc_3 > $O/pmc_mem_alone.txt
grep -E "k_sha_expand<384, 8, 16, 0>" $O/pmc_mem_alone.txt | cut -c1-150
head -12 $O/pmc_1.txt
