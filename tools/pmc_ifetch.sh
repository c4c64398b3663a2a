#!/bin/bash
# instruction-fetch counters of ONE layer / variant (tools/ws_one.py):  bash tools/pmc_ifetch.sh <tag> <variant> <cin> <cout> <h> <w> <n>
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_if_$i -o ${tag} -- python3 $root/tools/ws_one.py "$@" > /dev/null 2> $out/${tag}_if_$i.err || echo "pass $i failed"
done
cd $root
python3 tools/pmc_waits.py ${tag}_if $out/${tag}_if_1 $out/${tag}_if_2 > /dev/null
rm -rf $out/${tag}_if_1 $out/${tag}_if_2
python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_if_sq.json'))
for k,v in d.items():
    if 'conv' in k: print(k, json.dumps(v))
"
