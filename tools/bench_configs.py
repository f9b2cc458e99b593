#!/usr/bin/env python3
"""The five BASELINE.json configurations on ONE MI355X, one JSON line each (bench.py --config N; the multi-GPU
forms of the same commands are `torchrun --nproc-per-node N bench.py --gpus N --config N [--scaling strong]`).

    python tools/bench_configs.py [--only 1,3,4,5] [--out gpurun_out/configs.jsonl]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXTRA = {
    1: ["--config", "1", "--steps", "20", "--warmup", "5", "--no-profile"],
    2: ["--config", "2", "--steps", "10", "--warmup", "3"],
    3: ["--config", "3", "--steps", "10", "--warmup", "3"],
    4: ["--config", "4", "--steps", "4", "--warmup", "1"],      # (its first timed step is profiled: the line carries `roofline`)
    5: ["--config", "5", "--steps", "4", "--warmup", "1"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="1,2,3,4,5")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "configs.jsonl"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for c in [int(x) for x in args.only.split(",")]:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + EXTRA[c], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"config {c} failed:\n{r.stderr[-2000:]}")
                continue
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not js:
                print(f"config {c}: no JSON line\n{r.stdout[-1000:]}\n{r.stderr[-1000:]}")
                continue
            d = json.loads(js[-1])
            f.write(json.dumps(d, ensure_ascii=False) + "\n")
            f.flush()
            keys = ("metric", "value", "ms_per_step", "tokens_decoded_per_step")
            print(json.dumps({"config": c, **{k: d[k] for k in keys if k in d},
                              "cpu": (d.get("cpu_baseline") or {}).get("gpu_ids_identical")}, ensure_ascii=False))


if __name__ == "__main__":
    main()
