#!/bin/bash
# SQ-side counters of the chain kernels running WITHOUT the HBM streams (tools/chains_only.py), one pass per counter set
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03chains
mkdir -p $O
cd $R && timeout -k 10 200 python3 tools/chains_only.py > $O/chains_only.json 2> $O/chains_only.err || exit 1
cat $O/chains_only.json
cd /tmp && export TMPDIR=/tmp
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$i -o pmc -- python3 $R/tools/chains_only.py --steps 30 --warmup 30 > $O/pmc_$i.json 2> $O/pmc_$i.err || { tail -5 $O/pmc_$i.err; exit 1; }
done
cd $R
python3 tools/pmc_summary.py $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5 > $O/pmc_chains.txt
grep -c . $O/pmc_chains.txt
