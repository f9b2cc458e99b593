"""Debug 5 (r05): does an ENCODER run write into the decode's scratch buffers (an out-of-bounds store that is harmless while
the decode runs in front of the next encoder, and fatal beside it)?  Hash tok / encp before and after encoders of other batches."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gigaam_amd
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine

ck = synth.make_checkpoint(sys.argv[1] if len(sys.argv) > 1 else "v2_rnnt", seed=1, n_layers=2, rnnt_blank_bias=13.5)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
def mk(k, b):
    lens = [int(16000 * (1.0 + 0.37 * ((3 * i + k) % 11))) for i in range(b)]
    w, l = synth.synth_audio(b, max(lens) / 16000.0, seed=300 + k, lengths=lens)
    return w.cuda(), l.cuda()
bs = [9, 32, 5, 17, 33, 8]
batches = [mk(k, b) for k, b in enumerate(bs)]
def hashes():
    out = []
    for which in (0, 1, 2):
        h, n = C.c_uint64(0), C.c_int64(0)
        assert eng.lib.gam_debug_buffer_hash(eng._h, which, C.byref(h), C.byref(n)) == 0
        out.append((hex(h.value), n.value))
    return out
for w, l in batches:      # grow every buffer first
    enc, elen = eng.encode(*eng.frontend(w, l)); HipEngine.collect(eng.rnnt_greedy(enc, elen, 10))
for k, (w, l) in enumerate(batches):
    enc, elen = eng.encode(*eng.frontend(w, l))
    HipEngine.collect(eng.rnnt_greedy(enc, elen, 10))
    h0 = hashes()
    for k2, (w2, l2) in enumerate(batches):
        eng.encode(*eng.frontend(w2, l2))
        h1 = hashes()
        if h1 != h0:
            print(f"decode of batch {k} (b={bs[k]}) done; ENCODER of batch {k2} (b={bs[k2]}) CHANGED decode scratch: {h0} -> {h1}")
            h0 = h1
print("done")
