#!/bin/bash
# tools/collect_round.sh <tag>: everything a round commits under profiles/ in ONE gpurun call -- the GPU test suite with
# durations, tools/collect_profiles.sh (kernel stats, PMC traffic and SQ counters, micro-benchmarks, the default bench
# line), the kernel sequences of one C5 step and of one replayed tracking / mapping iteration, C4 on one GPU, the eager
# host-path profile.  Output in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
# (a gpurun call lasts 20 minutes at most: `tools/collect_round.sh r03 a` = suite + collect_profiles, `... b` = the rest)
tag=${1:-r03}; part=${2:-ab}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
if [[ $part == *a* ]]; then
python -m pytest tests -q -m gpu --durations=12 > $O/${tag}_gpu_suite.log 2>&1; tail -1 $O/${tag}_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/${tag}_gpu_suite.log 2>&1; tail -1 $O/${tag}_gpu_suite.log
tools/collect_profiles.sh $tag | cut -c1-600
fi
if [[ $part == *b* ]]; then
tools/c5_sequence.sh $tag > /dev/null 2>&1; tail -1 $O/${tag}_c5_step_kernel_sequence.txt
tools/replay_sequence.sh $tag > /dev/null 2>&1; tail -1 $O/${tag}_tracking_replay_kernel_sequence.txt
tools/mapping_sequence.sh ${tag}w8 41 > /dev/null 2>&1; cp $O/${tag}w8_mapping_replay_kernel_sequence.txt $O/${tag}_mapping_replay_kernel_sequence_window8.txt; tail -1 $O/${tag}_mapping_replay_kernel_sequence_window8.txt
python bench.py --workload c4 --steps 200 > $O/${tag}_c4_n1.log 2>&1; tail -1 $O/${tag}_c4_n1.log | cut -c1-400
python tools/host_overhead.py 300 > $O/${tag}_host_overhead.txt 2>&1; head -3 $O/${tag}_host_overhead.txt
fi
