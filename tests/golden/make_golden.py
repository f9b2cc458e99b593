"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

For each case it instantiates the reference's own, unmodified modules
(gigaam/encoder.py, decoder.py, decoding.py via oracle/ref_shim.py) on a seeded
synthetic checkpoint, runs them on seeded synthetic audio features, checks that
oracle/gigaam_oracle.py agrees, and stores the REFERENCE outputs in
tests/golden/<case>.npz.  Weights/audio are regenerated from seeds at test time
(numpy PCG64), so fixtures hold outputs only.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from oracle import gigaam_oracle as O  # noqa: E402
from oracle.ref_shim import import_reference  # noqa: E402

# name -> (model, ckpt seed, n_layers, audio: (batch, seconds, seed, lengths))
CASES = {
    "v2_ctc_l2": ("v2_ctc", 1, 2, (3, 4.0, 11, [64000, 50000, 33333])),
    "v2_ctc_l2_b1": ("v2_ctc", 1, 2, (1, 2.5, 12, None)),
    "v2_rnnt_l2": ("v2_rnnt", 1, 2, (3, 4.0, 13, [64000, 41234, 57000])),
    "v3_ctc_l2": ("v3_ctc", 1, 2, (3, 4.0, 14, [64000, 50000, 33333])),
    "v3_e2e_rnnt_l2": ("v3_e2e_rnnt", 1, 2, (2, 3.0, 15, [48000, 30011])),
    "v1_ctc_l2": ("v1_ctc", 1, 2, (3, 3.0, 16, [48000, 40000, 20000])),
    "v2_ctc_l2_short": ("v2_ctc", 1, 2, (2, 0.3125, 17, [5000, 3200])),  # reference tests/test_batching.py:125-140
    "v3_e2e_ctc_l2": ("v3_e2e_ctc", 1, 2, (2, 3.0, 19, [48000, 35000])),
    "v1_rnnt_l2": ("v1_rnnt", 1, 2, (2, 2.5, 20, [40000, 26000])),
    "emo_l2": ("emo", 1, 2, (2, 3.0, 18, [48000, 36000])),
}


def strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def kw(d):
    return {k: v for k, v in d.items() if k != "_target_"}


def run_case(ref, case):
    model, seed, nl, (b, secs, aseed, lens) = CASES[case]
    ck = synth.make_checkpoint(model, seed=seed, n_layers=nl)
    cfg, sd = ck["cfg"], ck["state_dict"]
    wav, wlen = synth.synth_audio(b, secs, seed=aseed, lengths=lens)
    out = {}
    with torch.no_grad():
        feat, flen = O.log_mel(wav, wlen, cfg["preprocessor"],
                               sd["preprocessor.featurizer.0.spectrogram.window"],
                               sd["preprocessor.featurizer.0.mel_scale.fb"])
        enc = ref.encoder.ConformerEncoder(**kw(cfg["encoder"])).eval()
        enc.load_state_dict(strip(sd, "encoder."))
        pre_ref, _ = enc.pre_encode(x=feat.transpose(1, 2), lengths=flen)
        y_ref, l_ref = enc(feat, flen)
        stages = {}
        y_or, l_or = O.encoder_forward(sd, cfg["encoder"], feat, flen, stages=stages)
        assert l_ref.tolist() == l_or.tolist() and l_ref.dtype == l_or.dtype == torch.int32
        valid = (torch.arange(y_ref.shape[2])[None, :] < l_ref[:, None])[:, None, :]
        d_enc = float(((y_ref - y_or) * valid).abs().max())
        d_pre = float(((pre_ref - stages["pre_encode"]) * valid.transpose(1, 2)).abs().max())
        assert d_enc < 2e-5 and d_pre < 2e-5, (case, d_enc, d_pre)
        out.update(feat_probe=feat[:, ::7, ::13].numpy(), feat_len=flen.numpy(),
                   pre_encode=pre_ref.numpy(), encoded=y_ref.numpy(), enc_len=l_ref.numpy())
        stats = {"enc_absdiff_oracle_vs_ref": d_enc, "pre_absdiff": d_pre}

        head_sd = strip(sd, "head.")
        if cfg["head"]["_target_"].endswith("Linear"):
            # the reference's own GigaAMEmo methods (gigaam/model.py:272-293), run unbound on a stand-in
            # `self` whose encoder / head are the reference encoder above and a torch Linear
            import importlib
            import types
            sys.modules["hydra"].utils.instantiate = lambda *a, **k: None
            ref_model = importlib.import_module("gigaam.model")
            lin = torch.nn.Linear(cfg["head"]["in_features"], cfg["head"]["out_features"])
            lin.load_state_dict(head_sd)
            fake = types.SimpleNamespace(encoder=enc, head=lin, id2name=cfg["id2name"])
            p_export = ref_model.GigaAMEmo.forward_for_export(fake, feat, flen)          # whole-axis mean, batch
            single = []
            for i in range(b):   # get_probs: one unpadded file at a time (model.py:276-283)
                f1, l1 = O.log_mel(wav[i:i + 1, : int(wlen[i])], wlen[i:i + 1], cfg["preprocessor"],
                                   sd["preprocessor.featurizer.0.spectrogram.window"], sd["preprocessor.featurizer.0.mel_scale.fb"])
                fake.prepare_wav = lambda _f, f1=f1, l1=l1: (f1, l1)
                fake.forward = lambda x, l: enc(x, l)
                d = ref_model.GigaAMEmo.get_probs(fake, "unused.wav")
                single.append([d[cfg["id2name"][k]] for k in range(len(d))])
            p_single = torch.tensor(single)
            assert float((O.emo_probs(sd, y_ref) - p_export).abs().max()) < 1e-6
            y1 = [enc(*O.log_mel(wav[i:i + 1, : int(wlen[i])], wlen[i:i + 1], cfg["preprocessor"],
                                 sd["preprocessor.featurizer.0.spectrogram.window"], sd["preprocessor.featurizer.0.mel_scale.fb"]))[0]
                  for i in range(b)]
            p_or_single = torch.cat([O.emo_probs(sd, y) for y in y1])
            assert float((p_or_single - p_single).abs().max()) < 1e-6
            out.update(probs_export=p_export.numpy(), probs_single=p_single.numpy())
            stats["probs_single"] = p_single.tolist()
            return out, stats
        if cfg["head"]["_target_"].endswith("CTCHead"):
            head = ref.decoder.CTCHead(**kw(cfg["head"])).eval()
            head.load_state_dict(head_sd)
            dec = ref.decoding.CTCGreedyDecoding(cfg["decoding"]["vocabulary"])
            lp_ref = head(y_ref)
            r = dec.decode(head, y_ref, l_ref)
            lp_or = O.ctc_log_probs(sd, y_ref)
            o = O.ctc_greedy(lp_or, l_ref)
            assert float((lp_ref - lp_or).abs().max()) < 2e-5
            out["log_probs"] = lp_ref.numpy()
            top2 = lp_ref.topk(2, dim=-1).values
            marg = (top2[..., 0] - top2[..., 1])
            stats["min_margin"] = float(marg[valid[:, 0, :]].min())
            stats["blank_frac"] = float((lp_ref.argmax(-1) == lp_ref.shape[-1] - 1).float().mean())
        else:
            head = ref.decoder.RNNTHead(cfg["head"]["decoder"], cfg["head"]["joint"]).eval()
            head.load_state_dict(head_sd)
            ms = cfg["decoding"]["max_symbols_per_step"]
            dec = ref.decoding.RNNTGreedyDecoding(cfg["decoding"]["vocabulary"], max_symbols_per_step=ms)
            r = dec.decode(head, y_ref, l_ref)
            trace = []
            o = O.rnnt_greedy(sd, y_ref, l_ref, ms, trace=trace)
            stats["joint_steps"] = len(trace)
            marg = [float(t[2].topk(2).values[0] - t[2].topk(2).values[1]) for t in trace]
            stats["min_margin"] = min(marg)
            # pin predict/joint against the reference modules on the first steps
            g_ref, (h_ref, c_ref) = head.decoder.predict(None, None, batch_size=1)
            g_or, (h_or, c_or) = O.rnnt_predict(sd, None, None)
            assert float((g_ref[0, 0] - g_or).abs().max()) < 1e-6
            lab = torch.tensor([[3]])
            g2_ref, (h2, c2) = head.decoder.predict(lab, (h_ref, c_ref), batch_size=1)
            g2_or, (h2o, c2o) = O.rnnt_predict(sd, 3, (h_or, c_or))
            assert float((g2_ref[0, 0] - g2_or).abs().max()) < 1e-6 and float((c2[:, 0] - c2o).abs().max()) < 1e-6
            f = y_ref.transpose(1, 2)[0:1, 5:6]
            j_ref = head.joint.joint(f, g2_ref)[0, 0, 0]
            j_or = O.rnnt_joint(sd, f[0, 0], g2_or)
            assert float((j_ref - j_or).abs().max()) < 2e-5
            out["joint_probe"] = j_ref.numpy()
            out["trace_first"] = torch.stack([t[2] for t in trace[:64]]).numpy()
        ids_flat, frames_flat, counts = [], [], []
        for (txt, ids, fr), (oi, of) in zip(r, o):
            assert ids == oi and fr == of, (case, ids[:10], oi[:10])
            ids_flat += ids
            frames_flat += fr
            counts.append(len(ids))
        out.update(ids=np.asarray(ids_flat, np.int32), frames=np.asarray(frames_flat, np.int32),
                   counts=np.asarray(counts, np.int32))
        stats["counts"] = counts
        stats["distinct_labels"] = len(set(ids_flat))
        assert sum(counts) > 0, f"{case}: degenerate decode (nothing emitted)"
    return out, stats


def main():
    torch.set_num_threads(os.cpu_count())
    ref = import_reference()
    here = os.path.dirname(os.path.abspath(__file__))
    all_stats = {}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for case in CASES:
        if only and case not in only:
            continue
        out, stats = run_case(ref, case)
        np.savez_compressed(os.path.join(here, case + ".npz"), **out)
        all_stats[case] = stats
        print(case, json.dumps(stats))
    meta = os.path.join(here, "golden_stats.json")
    old = json.load(open(meta)) if os.path.exists(meta) else {}
    old.update(all_stats)
    json.dump(old, open(meta, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
