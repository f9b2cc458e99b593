"""r06: do two HIP streams always get two hardware queues?  (No: profiles/experiments/r06_session_scripts/two_chain_probe.py found two torch streams on ONE queue, where their
kernels serialise.  The overlapped RNN-T decode -- engine.rnnt_greedy(overlap=True) -- relies on its side stream running BESIDE the launch stream.)

For K = 0 .. 11 streams created (and used once) before it, a side stream of normal / high priority runs a one-thread spin kernel
(torch.cuda._sleep) while the launch stream -- the null stream, or a non-default torch stream -- runs another.  Printed: the time of both together
over the time of one alone (1.0 = the two streams run side by side, 2.0 = they share a hardware queue and serialise).
    python tools/queue_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
CYC = 20_000_000       # ~10 ms


def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for main_kind in ("null", "stream"):
    made = []
    for k in range(12):
        if k:
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                torch.cuda._sleep(1000)
            made.append(s)
        main = torch.cuda.default_stream(dev) if main_kind == "null" else torch.cuda.Stream(dev)
        res = []
        for prio in (0, -1):
            side = torch.cuda.Stream(dev, priority=prio)
            run([main]); run([main, side])
            t1 = min(run([main]) for _ in range(2))
            t2 = min(run([main, side]) for _ in range(2))
            res.append(f"side priority {prio:2d}: both / one = {t2 / t1:.2f}")
            made.append(side)
        if main_kind != "null":
            made.append(main)
        print(f"QUEUE launch stream = {main_kind:6s} other streams alive: {len(made) - 2:2d}   " + "   ".join(res), flush=True)

# two HIGH-priority streams against each other (the decode side stream and the collect stream must not share a queue either), and the range
print("QUEUE priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
his = [torch.cuda.Stream(dev, priority=-1) for _ in range(6)]
for i in range(1, 6):
    run([his[0]]); run([his[0], his[i]])
    t1 = min(run([his[0]]) for _ in range(2))
    t2 = min(run([his[0], his[i]]) for _ in range(2))
    print(f"QUEUE high-priority stream 0 beside high-priority stream {i}: both / one = {t2 / t1:.2f}", flush=True)
nulls = torch.cuda.default_stream(dev)
t1 = min(run([nulls]) for _ in range(2))
t3 = min(run([nulls, his[0], his[1]]) for _ in range(2))
print(f"QUEUE null + two high-priority streams: all / one = {t3 / t1:.2f}", flush=True)
