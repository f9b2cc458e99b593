"""Debug 4 (r05): WHAT differs in a mismatching overlapped decode?  Dump every joint evaluation's log-probs and compare."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gigaam_amd
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine

REPS = int(sys.argv[1])
ck = synth.make_checkpoint("v2_rnnt", seed=1, n_layers=2, rnnt_blank_bias=13.5)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
ms = 10
def mk(k, b):
    lens = [int(16000 * (1.0 + 0.37 * ((3 * i + k) % 11))) for i in range(b)]
    w, l = synth.synth_audio(b, max(lens) / 16000.0, seed=300 + k, lengths=lens)
    return w.cuda(), l.cuda()
w1, l1 = mk(1, 32)
w2, l2 = mk(2, 5)
enc, elen = eng.encode(*eng.frontend(w1, l1))
enc, elen = enc.clone(), elen.clone()
CAP = 1400
C = int(os.environ.get("GAM_DEBUG_SIDE_CLUSTER", "2"))
eng.set_rnnt_cluster(C)
ref = eng.rnnt_greedy(enc, elen, ms, dump_cap=CAP)
ref_rows = HipEngine.collect(ref)[0]
ref_dump, ref_cnt = ref[3].clone(), ref[4].clone()
eng.set_rnnt_cluster(-1)
torch.cuda.synchronize()
print("steps per utterance (max):", int(ref_cnt.max()), "tokens", sum(len(i) for i, _ in ref_rows))
nbad = 0
for rep in range(REPS):
    dec = eng.rnnt_greedy(enc, elen, ms, dump_cap=CAP, overlap=True)
    eng.encode(*eng.frontend(w2, l2))
    rows = HipEngine.collect(dec)[0]
    torch.cuda.synchronize()
    if rows == ref_rows:
        continue
    nbad += 1
    if nbad > 4:
        continue
    d, c = dec[3], dec[4]
    for b in range(32):
        if rows[b] == ref_rows[b]:
            continue
        n = min(int(c[b]), int(ref_cnt[b]))
        diff = (d[b, :n] - ref_dump[b, :n]).abs()
        steps = (diff.amax(dim=1) > 0).nonzero().flatten()
        s0 = int(steps[0])
        row = diff[s0]
        cls = (row > 0).nonzero().flatten().tolist()
        print(f"rep {rep} utt {b}: first differing joint evaluation {s0} of {n}; classes differing {len(cls)}/34 {cls[:12]}; max abs diff {float(row.max()):.3e};"
              f" ref top {int(ref_dump[b, s0].argmax())} got top {int(d[b, s0].argmax())}; ref row {ref_dump[b, s0, :6].tolist()} got {d[b, s0, :6].tolist()}")
        # is the wrong row equal to some OTHER step's reference row (stale / misplaced data)?
        eq = ((ref_dump[b, :n] - d[b, s0]).abs().amax(dim=1) == 0).nonzero().flatten().tolist()
        print("     the wrong row equals the reference row of step(s):", eq[:5])
        if int(os.environ.get("GAM_RNNT_DBG", "0")) & 64:
            names = ["sum_h", "sum_pp", "sum_zenc", "sum_z", "label", "frame", "sum_c", "tabv0"]
            for st in range(max(0, s0 - 2), min(n, s0 + 3)):
                print("     step", st, "ref", dict(zip(names, [round(x, 5) for x in ref_dump[b, st, :8].tolist()])))
                print("     step", st, "got", dict(zip(names, [round(x, 5) for x in d[b, st, :8].tolist()])))
        break
print("mismatching runs:", nbad, "of", REPS)
