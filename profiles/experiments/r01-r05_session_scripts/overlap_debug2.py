"""Debug 2 (r05): which side is non-deterministic under overlap -- the encoder output of batch n+1, or the decode of batch n?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gigaam_amd
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine

model_name, bias, REPS = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
ck = synth.make_checkpoint(model_name, seed=1, n_layers=2, rnnt_blank_bias=bias)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
ms = ck["cfg"]["decoding"].get("max_symbols_per_step", 10)
batches = []
for k in range(6):
    b = [9, 32, 5, 17, 33, 8][k]
    lens = [int(16000 * (1.0 + 0.37 * ((3 * i + k) % 11))) for i in range(b)]
    w, l = synth.synth_audio(b, max(lens) / 16000.0, seed=300 + k, lengths=lens)
    batches.append((w.cuda(), l.cuda()))
xa_small = torch.randn(640, 768, device="cuda"); xa_big = torch.randn(16064, 768, device="cuda"); wa = torch.randn(768, 768, device="cuda") * 0.03
qa = torch.randn(5, 120, 768, device="cuda"); la = torch.tensor([120, 100, 90, 77, 50], device="cuda")
canary = None
sq_stream = None
if os.environ.get("BESIDE", "").startswith("canary"):
    import ctypes as C
    canary = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_canary.so"))
    canary.lds_canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
eng2 = None
if os.environ.get("BESIDE") == "other_handle":
    model2 = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
    eng2 = model2.encoder.engine
    for w, l in batches:
        eng2.encode(*eng2.frontend(w, l)); eng2.encode(*eng2.frontend(w, l))
encs = []
serial = []
for wav, wlen in batches:
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    encs.append((enc.clone(), elen.clone()))
    eng.set_rnnt_cluster(HipEngine.side_cluster(wav.shape[0], 64))
    serial.append(HipEngine.collect(eng.rnnt_greedy(enc, elen, ms))[0])
eng.set_rnnt_cluster(-1)
torch.cuda.synchronize()
enc_bad = dec_bad_A = dec_bad_B = dec_bad_C = 0
for rep in range(REPS):
    if os.environ.get("ONLY_B"):
        pend, got = None, []
        for k, (enc, elen) in enumerate(encs):
            if os.environ.get("BESIDE") == "squeeze":
                # spinner workgroups holding the WHOLE LDS of most CUs for 25 ms, launched first: the decode's workgroups must pack two per CU
                # onto the few CUs left (its C <= 4 variants have an occupancy of two) with NOTHING else running
                import ctypes as C
                if canary is None:
                    canary = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_canary.so"))
                    canary.lds_canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
                out_c = torch.zeros((256, 18), dtype=torch.int32, device="cuda")
                if sq_stream is None:
                    sq_stream = torch.cuda.Stream()
                sq_stream.wait_stream(torch.cuda.current_stream())
                canary.lds_canary_launch(C.c_void_p(out_c.data_ptr()), int(os.environ.get("SQUEEZE_WGS", "200")), 159 * 1024, 25000.0,
                                         C.c_void_p(sq_stream.cuda_stream))
            dec = eng.rnnt_greedy(enc, elen, ms, overlap=True, side_cus=int(os.environ.get("SIDE_CUS", "64")))
            w2, l2 = batches[(k + 1) % 6]
            if os.environ.get("BESIDE") == "matmul":
                x = torch.empty((8192, 8192), device="cuda"); y = x @ x
            elif os.environ.get("BESIDE") == "other_handle":
                eng2.encode(*eng2.frontend(w2, l2))
            elif os.environ.get("BESIDE") == "frontend_only":
                eng.frontend(w2, l2)
            elif os.environ.get("BESIDE") == "big_encoder":
                eng.encode(*eng.frontend(*batches[1]))
            elif os.environ.get("BESIDE", "").startswith("canary"):
                # a neighbour that only holds LDS and re-reads it (tools/lds_canary.hip): does ANY co-resident workgroup perturb the decode?
                import ctypes as C
                kb = int(os.environ["BESIDE"].split(":")[1]) if ":" in os.environ["BESIDE"] else 90
                out_c = torch.zeros((256, 18), dtype=torch.int32, device="cuda")
                canary.lds_canary_launch(C.c_void_p(out_c.data_ptr()), 256, kb * 1024, 4000.0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            elif os.environ.get("BESIDE") == "squeeze":
                pass      # (the spinner was launched BEFORE the decode: see below)
            elif os.environ.get("BESIDE") == "op_gemm_small":
                for _ in range(40): eng.op_gemm(xa_small, wa)
            elif os.environ.get("BESIDE") == "op_gemm_big":
                for _ in range(6): eng.op_gemm(xa_big, wa)
            elif os.environ.get("BESIDE") == "op_attention":
                for _ in range(40): eng.op_attention(qa, qa, qa, la)
            elif os.environ.get("BESIDE") == "stem_only":
                eng.encode(*eng.frontend(w2, l2), n_layers_run=0)
            elif os.environ.get("BESIDE") == "layers1":
                eng.encode(*eng.frontend(w2, l2), n_layers_run=1)
            elif os.environ.get("BESIDE") == "nograph":
                eng.encode(*eng.frontend(w2, l2))
            else:
                eng.encode(*eng.frontend(w2, l2))
            if pend is not None:
                got.append(HipEngine.collect(pend)[0])
            pend = dec
        got.append(HipEngine.collect(pend)[0])
        torch.cuda.synchronize()
        db = [k for k, (a, b) in enumerate(zip(got, serial)) if a != b]
        dec_bad_B += bool(db)
        continue
    # A: the product's overlap; keep the encoder outputs
    pend, got, kept = None, [], []
    for wav, wlen in batches:
        enc, elen = eng.encode(*eng.frontend(wav, wlen))
        kept.append(enc)
        dec = eng.rnnt_greedy(enc, elen, ms, overlap=True, side_cus=64)
        if pend is not None:
            got.append(HipEngine.collect(pend)[0])
        pend = dec
    got.append(HipEngine.collect(pend)[0])
    torch.cuda.synchronize()
    eb = [k for k, (e, (e0, _)) in enumerate(zip(kept, encs)) if not torch.equal(e, e0)]
    db = [k for k, (a, b) in enumerate(zip(got, serial)) if a != b]
    enc_bad += bool(eb); dec_bad_A += bool(db)
    if eb or db:
        print("A rep", rep, "encoder outputs differing:", eb, "decodes differing:", db, flush=True)
    # B: decode-only overlap on FIXED encoder outputs, an encoder of another batch running beside it (its output unused)
    pend, got = None, []
    for k, (enc, elen) in enumerate(encs):
        dec = eng.rnnt_greedy(enc, elen, ms, overlap=True, side_cus=64)
        w2, l2 = batches[(k + 1) % 6]
        eng.encode(*eng.frontend(w2, l2))
        if pend is not None:
            got.append(HipEngine.collect(pend)[0])
        pend = dec
    got.append(HipEngine.collect(pend)[0])
    torch.cuda.synchronize()
    db = [k for k, (a, b) in enumerate(zip(got, serial)) if a != b]
    dec_bad_B += bool(db)
    if db:
        print("B rep", rep, "decodes differing:", db, flush=True)
    # C: decode on the side stream with NOTHING beside it
    pend, got = None, []
    for k, (enc, elen) in enumerate(encs):
        dec = eng.rnnt_greedy(enc, elen, ms, overlap=True, side_cus=64)
        if pend is not None:
            got.append(HipEngine.collect(pend)[0])
        pend = dec
    got.append(HipEngine.collect(pend)[0])
    torch.cuda.synchronize()
    db = [k for k, (a, b) in enumerate(zip(got, serial)) if a != b]
    dec_bad_C += bool(db)
    if db:
        print("C rep", rep, "decodes differing:", db, flush=True)
print(f"reps {REPS}: A encoder-mismatch runs {enc_bad}, A decode-mismatch runs {dec_bad_A}; B (fixed enc, encoder beside) {dec_bad_B}; C (side stream alone) {dec_bad_C}")
