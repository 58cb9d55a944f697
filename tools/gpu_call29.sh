#!/bin/bash
# memory-side counters of the expansion kernel in the pipeline: does the L2 stall on DRAM write credits, and do the translations miss?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03mem
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for p in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum" \
         "TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$i -o pmc -- python3 $R/bench.py --steps 48 --warmup 16 --no-cpu-baseline --consumer-shard 0 > $O/pmc_$i.json 2> $O/pmc_$i.err || { tail -5 $O/pmc_$i.err; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 > $O/pmc_mem.txt
grep -E "k_sha_expand|k_place_field|k_pairing_team |k_cofactor " $O/pmc_mem.txt | cut -c1-150
