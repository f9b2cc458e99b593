#!/bin/bash
# Flakiness soak of the final library on one MI355X: the whole GPU suite twice more and the contention tests (three handles /
# three threads, cooperative RNN-T clusters, repair pass) eight times.  Usage: gpurun --timeout 900 -- bash tools/soak.sh r04_soak
out=gpurun_out/${1:-soak}; mkdir -p $out
python __graft_entry__.py > $out/build.log 2>&1 || { echo BUILD FAILED; tail $out/build.log; exit 1; }
fail=0
for i in 1 2; do
  timeout 420 python -m pytest tests -q -x -m gpu -p no:cacheprovider > $out/full_$i.log 2>&1 || fail=1
  tail -1 $out/full_$i.log
done
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python -m pytest tests/test_hip_hardening.py tests/test_hip_range.py -q -x -p no:cacheprovider -k "threads or rnnt or range" > $out/contend_$i.log 2>&1 || fail=1
  tail -1 $out/contend_$i.log
done
echo "soak fail=$fail" | tee $out/verdict.txt
exit $fail
