#!/bin/bash
# consumer-mode shard legs under chain_variant 0 (default: long chains out of line) and 2 (everything inlined), one call
mkdir -p gpurun_out
: > gpurun_out/r04_consumer_chainvar.txt
for rep in 1 2; do
for cv in 0 2; do
  echo "BLSW_CHAIN_VARIANT=$cv" >> gpurun_out/r04_consumer_chainvar.txt
  BLSW_CHAIN_VARIANT=$cv timeout -k 10 300 python tools/experiments/r04_consumer_probe.py "[8192, null, false]" "[8192, null, false]" "[32768, null, false]" >> gpurun_out/r04_consumer_chainvar.txt 2> gpurun_out/r04_consumer_chainvar.err || { tail -5 gpurun_out/r04_consumer_chainvar.err; exit 1; }
done
done
cat gpurun_out/r04_consumer_chainvar.txt
