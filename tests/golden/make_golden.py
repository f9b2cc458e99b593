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

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import CASES, EMO_CASE, RNNT_MIN_MARGIN, make_case_checkpoint  # noqa: E402

ALL_CASES = dict(CASES, emo_l2=EMO_CASE)


def strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def kw(d):
    return {k: v for k, v in d.items() if k != "_target_"}


def run_case(ref, case):
    ck, wav, wlen = make_case_checkpoint(ALL_CASES[case])
    cfg, sd = ck["cfg"], ck["state_dict"]
    b = wav.shape[0]
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
            o = O.rnnt_greedy(sd, y_ref, l_ref, ms, cfg["head"]["decoder"]["pred_rnn_layers"], trace=trace)
            stats["joint_steps"] = len(trace)
            marg = [float(t[2].topk(2).values[0] - t[2].topk(2).values[1]) for t in trace]
            stats["min_margin"] = min(marg)
            assert min(marg) > RNNT_MIN_MARGIN, (case, min(marg))
            # The REFERENCE's joint log-probs of every step, per sample in decode order: the reference's own
            # RNNTGreedyDecoding.decode run one sample at a time with RNNTJoint.joint wrapped by a recorder
            # (decoder.py:41-47 untouched).  This is what the HIP kernel's logits dump is compared with.
            ref_trace, ref_counts = [], []
            orig_joint = head.joint.joint
            for i in range(b):
                rec = []
                head.joint.joint = lambda f, g, rec=rec: (rec.append(orig_joint(f, g)), rec[-1])[1]
                r1 = dec.decode(head, y_ref[i:i + 1, :, : int(l_ref[i])].contiguous(), l_ref[i:i + 1])
                assert r1[0][1] == r[i][1] and r1[0][2] == r[i][2], (case, i)
                steps = torch.cat([x.reshape(-1, x.shape[-1]) for x in rec])
                per = torch.stack([t[2] for t in trace if t[0] == i])
                assert steps.shape == per.shape, (steps.shape, per.shape)
                assert float((steps - per).abs().max()) < 2e-5, (case, i, float((steps - per).abs().max()))
                ref_trace.append(steps)
                ref_counts.append(steps.shape[0])
            head.joint.joint = orig_joint
            out["trace"] = torch.cat(ref_trace).numpy()
            out["trace_counts"] = np.asarray(ref_counts, np.int32)
            n_frames = int(l_ref.sum())
            n_tok = sum(len(x[1]) for x in r)
            stats["symbols_per_frame"] = round(n_tok / n_frames, 3)
            from collections import Counter
            stats["max_symbols_on_a_frame"] = max(max(Counter(x[2]).values()) if x[2] else 0 for x in r)
            stats["max_symbols_per_step"] = ms
            stats["blank_step_frac"] = round(1.0 - n_tok / len(trace), 3)
            # pin predict/joint against the reference modules on the first steps
            g_ref, (h_ref, c_ref) = head.decoder.predict(None, None, batch_size=1)
            nlp = cfg["head"]["decoder"]["pred_rnn_layers"]
            g_or, (h_or, c_or) = O.rnnt_predict(sd, None, None, nlp)
            assert float((g_ref[0, 0] - g_or).abs().max()) < 1e-6
            lab = torch.tensor([[3]])
            g2_ref, (h2, c2) = head.decoder.predict(lab, (h_ref, c_ref), batch_size=1)
            g2_or, (h2o, c2o) = O.rnnt_predict(sd, 3, (h_or, c_or), nlp)
            assert float((g2_ref[0, 0] - g2_or).abs().max()) < 1e-6 and float((c2[:, 0] - c2o).abs().max()) < 1e-6
            f = y_ref.transpose(1, 2)[0:1, 5:6]
            j_ref = head.joint.joint(f, g2_ref)[0, 0, 0]
            j_or = O.rnnt_joint(sd, f[0, 0], g2_or)
            assert float((j_ref - j_or).abs().max()) < 2e-5
            out["joint_probe"] = j_ref.numpy()
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
    for case in ALL_CASES:
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
