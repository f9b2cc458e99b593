#!/bin/bash
# 8-GPU day-one kit.  NOTHING here has run on more than one GPU -- the builder's boxes have one; this is
# the script for the first visit to an 8-GPU MI355X node.  It runs the driver's own command at N = 1, 2, 4, 8 (weak scaling:
# 32 utterances per GPU), then the strong split of one 32-utterance batch, then configs 4 and 5 at N = 8, and for every N > 1
# REQUIRES the exchange to be gam_gather_ids -- RCCL behind the C ABI (gam_comm.h), which has only ever run with a world of
# one -- instead of bench.py's quiet fall-back to torch.distributed: --strict-gather makes a failed gam_comm_create a hard
# error carrying gam_comm_last_error().  Set GAM_COMM_TIMEOUT_S (default 90) if ncclCommInitRank is slow on the node.
#   usage:  bash tools/scale8.sh [outdir]           (one JSON line per run in <outdir>/scale8.jsonl, a table at the end)
set -u
out=${1:-gpurun_out/scale8}; mkdir -p "$out"; : > "$out/scale8.jsonl"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py > "$out/build.log" 2>&1 || { echo BUILD FAILED; tail "$out/build.log"; exit 1; }
ngpu=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "visible GPUs: $ngpu"
fail=0
# FIRST: the communicator behind the C ABI at the largest world this node offers -- gam_comm_create (ncclCommInitRank through dlopen) has never
# run with more than one rank -- and one tiny gam_gather_ids, checked on every rank, before anything is built or timed.
if [ "$ngpu" -ge 2 ]; then
  nn=$ngpu; [ "$nn" -gt 8 ] && nn=8
  ( timeout 300 python bench.py --gpus $nn --comm-selftest ) 2> "$out/comm_selftest.err" | grep -a '^{' | tail -1 | tee "$out/comm_selftest.json"
  grep -q '"comm_selftest": "ok"' "$out/comm_selftest.json" || { echo "COMMUNICATOR SELF-TEST FAILED at world $nn:"; tail -20 "$out/comm_selftest.err"; exit 1; }
  grep -a -i "nccl\|rccl" "$out/comm_selftest.err" | head -5
else
  ( timeout 300 python bench.py --gpus 1 --comm-selftest ) 2> "$out/comm_selftest.err" | grep -a '^{' | tail -1 | tee "$out/comm_selftest.json"
fi
run() {   # run <name> <bench args...>
  name=$1; shift
  ( time timeout 900 python bench.py "$@" ) 2> "$out/$name.err" | grep -a '^{' | tail -1 > "$out/$name.json"
  if [ ! -s "$out/$name.json" ]; then echo "$name: NO LINE (see $out/$name.err)"; tail -5 "$out/$name.err"; fail=1; return; fi
  cat "$out/$name.json" >> "$out/scale8.jsonl"
  python - "$out/$name.json" "$name" <<'PY' || fail=1
import json, sys
j = json.load(open(sys.argv[1])); n = j["n_gpus"]
gp = j.get("gather_path", "")
print(f"{sys.argv[2]:>14}: n_gpus {n}  {j['value']:>10.1f} {j['unit']}  {j['ms_per_step']:.2f} ms/step  scaling {j['scaling']}  gather: {gp}")
if n > 1 and not gp.startswith("gam_gather_ids"):
    print(f"  !! {sys.argv[2]}: the exchange did NOT run behind the C ABI ({gp})"); sys.exit(1)
if "INVALID" in j:
    print("  !! INVALID:", j["INVALID"]); sys.exit(1)
PY
}
for n in 1 2 4 8; do
  [ "$n" -le "$ngpu" ] || { echo "skipping N=$n ($ngpu GPUs visible)"; continue; }
  run weak_n$n --gpus $n --steps 20 --warmup 3 --strict-gather
done
for n in 2 4 8; do
  [ "$n" -le "$ngpu" ] || continue
  run strong_n$n --gpus $n --scaling strong --steps 20 --warmup 3 --strict-gather --cpu-utts 0
done
if [ "$ngpu" -ge 8 ]; then
  run config3_n8 --gpus 8 --config 3 --steps 10 --warmup 2 --strict-gather --cpu-utts 0
  run config4_n8 --gpus 8 --config 4 --steps 3 --warmup 1 --strict-gather --cpu-utts 0
  run config4_strong_n8 --gpus 8 --config 4 --scaling strong --steps 3 --warmup 1 --strict-gather --cpu-utts 0
  run config5_n8 --gpus 8 --config 5 --steps 3 --warmup 1 --strict-gather --cpu-utts 0
fi
python - "$out/scale8.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = {}
for j in rows:
    key = (j["config"]["baseline_config"], j["scaling"])
    if j["n_gpus"] == 1:
        base[j["config"]["baseline_config"]] = j
print("\nscaling table (speed-up vs the N = 1 line of the same config; weak: value ratio, strong: time ratio)")
for j in rows:
    b = base.get(j["config"]["baseline_config"])
    if not b:
        continue
    sp = j["value"] / b["value"]
    print(f"  config {j['config']['baseline_config']} {j['scaling']:>6} N={j['n_gpus']}: {j['value']:.0f} RTFx  x{sp:.2f}  (efficiency {sp / j['n_gpus']:.2f})")
PY
[ $fail -eq 0 ] && echo "scale8: all lines produced, every N > 1 exchange ran through gam_gather_ids" || { echo "scale8: FAILURES above"; exit 1; }
