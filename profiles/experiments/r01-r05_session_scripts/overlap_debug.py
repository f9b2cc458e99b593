"""Debug: is the overlapped RNN-T decode deterministic w.r.t. the serial path at the same cluster size?  (r05)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gigaam_amd
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine

model_name, bias = sys.argv[1], float(sys.argv[2])
ck = synth.make_checkpoint(model_name, seed=1, n_layers=2, rnnt_blank_bias=bias)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
ms = ck["cfg"]["decoding"].get("max_symbols_per_step", 10)
batches = []
for k in range(6):
    b = [9, 32, 5, 17, 33, 8][k]
    lens = [int(16000 * (1.0 + 0.37 * ((3 * i + k) % 11))) for i in range(b)]
    batches.append(synth.synth_audio(b, max(lens) / 16000.0, seed=300 + k, lengths=lens))

def serial_with(cus):
    out = []
    for wav, wlen in batches:
        enc, elen = eng.encode(*eng.frontend(wav, wlen))
        eng.set_rnnt_cluster(HipEngine.side_cluster(wav.shape[0], cus))
        out.append(HipEngine.collect(eng.rnnt_greedy(enc, elen, ms))[0])
    eng.set_rnnt_cluster(-1)
    return out

def overlapped(cus, via_model):
    pend, got = None, []
    for wav, wlen in batches:
        if via_model:
            dec = model.launch_batch(wav, wlen, overlap=True)[0]
        else:
            enc, elen = eng.encode(*eng.frontend(wav, wlen))
            dec = eng.rnnt_greedy(enc, elen, ms, overlap=True, side_cus=cus)
        if pend is not None:
            got.append(HipEngine.collect(pend)[0])
        pend = dec
    got.append(HipEngine.collect(pend)[0])
    return got

s1, s2 = serial_with(64), serial_with(64)
print("serial deterministic:", s1 == s2)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nbad = 0
for rep in range(REPS):
    for via in (False, True):
        g = overlapped(64, via)
        bad = [(k, i) for k, (ra, rb) in enumerate(zip(g, s1)) for i, (a, b) in enumerate(zip(ra, rb)) if a != b]
        nbad += bool(bad)
        if bad or rep == 0:
            print("rep", rep, "via_model" if via else "engine", "mismatches:", bad[:10], flush=True)
        for k, i in bad[:2]:
            a, b = g[k][i], s1[k][i]
            j = next((t for t, (x, y) in enumerate(zip(a[0], b[0])) if x != y), min(len(a[0]), len(b[0])))
            print("   first diff at token", j, "of", len(a[0]), len(b[0]), a[0][max(0, j - 3): j + 4], b[0][max(0, j - 3): j + 4], a[1][max(0, j-3): j + 4], b[1][max(0, j-3): j+4])
print("runs with a mismatch:", nbad, "of", 2 * REPS)
