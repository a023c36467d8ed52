#!/bin/bash
# Round-end profiling on the GPU box (run through gpurun): kernel trace + the two PMC traffic passes of the
# same bench command.  Usage: scripts/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{trace,fetch,write}
# (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; no other trace domains)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UOC_CONV_TUNE_CACHE=/tmp/uoc_tune_$1.txt
timeout 300 python $R/bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o bench -- python $R/bench.py --steps 6 --warmup 2 --cpu-frames 0 --sustained-seconds 0 > $O/bench_under_rocprof.json 2> $O/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python $R/bench.py --steps 4 --warmup 1 --cpu-frames 0 --profile-steps 0 --sustained-seconds 0 > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python $R/bench.py --steps 4 --warmup 1 --cpu-frames 0 --profile-steps 0 --sustained-seconds 0 > /dev/null 2> $O/write.err
find $O -name "*.db" -o -name "*counter_collection.csv" | head; tail -c 400 $O/bench.json
