#!/bin/bash
# Round artefacts on the GPU box: kernel-trace stats, PMC traffic passes (separate runs), config times, bench line.
# usage: bash tools/round_artifacts.sh r1     (writes under gpurun_out/<tag>_*)
set -u
tag=${1:-r1}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -o ${tag} -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $o/${tag}_trace_bench.json 2> $o/${tag}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_fetch -o ${tag} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $o/${tag}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_write -o ${tag} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $o/${tag}_pmc_write.err
f=$(find $o/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $o/${tag}_pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py gram_ring_kernel $o/${tag}_pmc_gram.json "$f" "$w"
s=$(find $o/${tag}_trace -name "*kernel_stats.csv" | head -1)
cp "$s" $o/${tag}_kernel_stats.csv
python3 tools/config_times.py > $o/${tag}_config_times.json 2> $o/${tag}_config_times.err
python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err
tail -c 300 $o/${tag}_bench.err
head -c 1500 $o/${tag}_kernel_stats.csv
cat $o/${tag}_config_times.json | head -c 2500
