#!/bin/bash
# The multi-rank programs at the world size the driver will use (8), on ONE GPU: eight processes share cuda:0 (--oversubscribe, gloo
# exchange -- RCCL cannot place two ranks on one device), so the numbers are INVALID as scaling points; what this checks is the
# 8-rank plumbing end to end: rendezvous, dealing (strong split 4 per rank, config 4 = 8 x 128 = 1024 utterances, config 5 LPT over 8),
# the exchange with the range-flag row, rank-0's line.   Usage: gpurun --timeout 420 -- bash tools/rehearsal8.sh r04_rehearsal8
out=gpurun_out/${1:-rehearsal8}; mkdir -p $out
python __graft_entry__.py > $out/build.log 2>&1 || { echo BUILD FAILED; tail $out/build.log; exit 1; }
common="--gpus 8 --oversubscribe --no-profile --no-power --cpu-utts 0"
run() { name=$1; shift; ( time timeout 150 python bench.py $common "$@" ) 2> $out/$name.err | grep -a '^{' > $out/$name.json; echo "$name rc=${PIPESTATUS[0]} lines=$(wc -l < $out/$name.json)"; }
run strong  --scaling strong --steps 3 --warmup 1
run weak    --scaling weak --steps 2 --warmup 1
run config4 --config 4 --steps 1 --warmup 1
run config5 --config 5 --steps 1 --warmup 1
