#!/bin/bash
# Streams x frames-per-launch sweep on the GPU box (sustained rate over ~6 s per configuration)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/inflight; mkdir -p $O
cd $R
for cfg in "$@"; do set -- $cfg
  timeout 240 python bench.py --steps 48 --warmup 6 --inflight $1 --frames-per-launch $2 --cpu-frames 0 --profile-steps 0 --sustained-seconds 6 --skip-pcie \
     > $O/s$1_g$2.json 2> $O/s$1_g$2.err
  echo "streams=$1 frames/launch=$2: $(python -c "import json;d=json.load(open('$O/s$1_g$2.json'));print(d['value'],'fps timed;',d['sustained_frames_per_s'],'fps sustained')" 2>&1 | tail -1)"
done
