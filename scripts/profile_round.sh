#!/bin/bash
# Round-end profiling on the GPU box (run through gpurun): the driver-style bench line, a kernel trace of the shipped
# launch shapes (launch sets of four frames) run one set at a time on one stream (clean per-kernel durations), a kernel
# trace of the shipped schedule (three streams x four frames per launch set: overlap analysis), and the two PMC traffic
# passes (one launch set at a time as well).  Usage: scripts/profile_round.sh <tag>
#   -> gpurun_out/prof_<tag>/{bench.json, trace_seq, trace_pipe, fetch, write}
# (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; no other trace domains)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UOC_CONV_TUNE_CACHE=/tmp/uoc_tune_$1.txt
QUIET="--cpu-frames 0 --sustained-seconds 0 --skip-pcie --skip-latency"
UOC_BENCH_FULL=$O/bench_full.json timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
export UOC_BENCH_FULL=$O/bench_scratch.json
timeout 300 rocprofv3 --kernel-trace -d $O/trace_seq -o t -- python $R/bench.py --steps 12 --warmup 4 --inflight 1 $QUIET > $O/bench_seq_under_rocprof.json 2> $O/trace_seq.err
timeout 300 rocprofv3 --kernel-trace -d $O/trace_pipe -o t -- python $R/bench.py --steps 24 --warmup 3 $QUIET --profile-steps 0 > $O/bench_pipe_under_rocprof.json 2> $O/trace_pipe.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 8 --warmup 4 --inflight 1 $QUIET --profile-steps 0 > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python $R/bench.py --steps 8 --warmup 4 --inflight 1 $QUIET --profile-steps 0 > /dev/null 2> $O/write.err
cd $R
python scripts/rocpd_stats.py $(find $O/trace_seq -name "*.db" | head -1) 0.5 > $O/kernel_stats_seq.md
python scripts/rocpd_stats.py $(find $O/trace_pipe -name "*.db" | head -1) > $O/kernel_stats_pipe.md
python scripts/rocpd_overlap.py $(find $O/trace_pipe -name "*.db" | head -1) 0.3 > $O/overlap_pipe.md
python scripts/pmc_traffic.py $O $O/pmc_traffic.md 0.5 > /dev/null 2>&1
find $O -name "*.db" -size +40M -delete; find $O -name "*counter_collection.csv" -size +30M -delete
cat $O/bench.json; echo; head -12 $O/kernel_stats_seq.md; head -16 $O/overlap_pipe.md
