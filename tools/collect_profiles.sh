#!/bin/bash
# Copy the artefacts of a tools/gpu_full.sh visit (gpurun_out/TAG/) into profiles/ under the round's prefix.
#   bash tools/collect_profiles.sh r06_final r06
TAG=$1; P=$2
S=gpurun_out/$TAG; D=profiles
cp_if() { [ -s "$1" ] && cp "$1" "$2"; }
grep -a '^{' $S/bench.log > $D/${P}_bench_line.log
grep -a '^{' $S/bench_torchrun1.log > $D/${P}_bench_torchrun1_line.log
cp_if $S/configs.jsonl $D/${P}_configs.jsonl
cp_if $S/strong_proxy.jsonl $D/${P}_strong_proxy.jsonl
cp_if $S/trace_summary.txt $D/${P}_trace_summary.txt
cp_if $S/trace_b4_summary.txt $D/${P}_trace_batch4_summary.txt
cp_if $S/trace_c1_summary.txt $D/${P}_trace_config1_summary.txt
cp_if $S/trace_c3_summary.txt $D/${P}_trace_config3_summary.txt
cp_if $S/fetch_summary.txt $D/${P}_f16x3_fetch.txt
cp_if $S/write_summary.txt $D/${P}_f16x3_write.txt
cp_if $S/sq_summary.txt $D/${P}_f16x3_sq.txt
cp_if $S/pmc_traffic_f16x3.json $D/pmc_traffic_f16x3.json
cp_if $S/measured_errors.jsonl $D/${P}_measured_errors.jsonl
cp_if $S/pytest_gpu.log $D/${P}_pytest_gpu.log
cp_if $S/smoke.log $D/${P}_smoke.log
cp_if $S/published_shapes.jsonl $D/${P}_published_shapes.jsonl
cp_if $S/hbm_kernels.txt $D/${P}_hbm_kernels.txt
cp_if $S/trace_ragged_summary.txt $D/${P}_trace_ragged_packed_summary.txt
cp_if $S/extra_lines.jsonl $D/${P}_extra_lines.jsonl
cp_if $S/repro.txt $D/${P}_repro_product.txt
cp_if $S/pkfma_rule.txt $D/${P}_pkfma_rule_final.txt
cp_if $S/scale8.log $D/${P}_scale8_n1.log
for f in $S/rehearsal_*.json $S/rehearsal_refused.err; do [ -s "$f" ] && cp "$f" $D/${P}_$(basename $f); done
sed -i "s#/tmp/code/[^ ]*/repo/##g; s#/root/repo/##g" $D/${P}_*summary*.txt $D/${P}_f16x3_*.txt $D/${P}_hbm_kernels.txt 2>/dev/null
ls $D | grep "^${P}_" | wc -l
