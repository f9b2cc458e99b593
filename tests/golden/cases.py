"""Golden-case table shared by tests/golden/make_golden.py (which writes the fixtures from the
reference's own modules) and the tests (which regenerate the seeded weights/audio and compare).

name -> (model, ckpt seed, n_layers, audio: (batch, seconds, seed, lengths), options)
options: blank_bias  -- RNN-T: blank bias of the synthetic joint (synth.make_state_dict)
         max_symbols -- RNN-T: max_symbols_per_step of the decoding config (reference default 10)
         pred_rnn_layers -- RNN-T: depth of the predictor's nn.LSTM (reference decoder.py:78-83; published checkpoints: 1)

RNN-T cases come in two regimes:
  * blank-dominant (0.1-1 symbols per frame, what reference gigaam/decoding.py:162-205 sees on trained
    models): v2_rnnt_l2, v1_rnnt_l2, v3_rnnt_l2, v3_e2e_rnnt_l2 (V = 1025)
  * emission-heavy (several symbols per frame, exercises the max_symbols cap): *_dense
Every RNN-T case has an oracle top-1/top-2 margin > 2e-3 on every joint step (asserted by the
generator), so ids/frames/log-probs are compared unconditionally.
"""

CASES = {
    "v2_ctc_l2": ("v2_ctc", 1, 2, (3, 4.0, 11, [64000, 50000, 33333]), {}),
    "v2_ctc_l2_b1": ("v2_ctc", 1, 2, (1, 2.5, 12, None), {}),
    "v3_ctc_l2": ("v3_ctc", 1, 2, (3, 4.0, 14, [64000, 50000, 33333]), {}),
    "v1_ctc_l2": ("v1_ctc", 1, 2, (3, 3.0, 16, [48000, 40000, 20000]), {}),
    "v2_ctc_l2_short": ("v2_ctc", 1, 2, (2, 0.3125, 17, [5000, 3200]), {}),  # reference tests/test_batching.py:125-140
    "v3_e2e_ctc_l2": ("v3_e2e_ctc", 1, 2, (2, 3.0, 19, [48000, 35000]), {}),
    # blank-dominant RNN-T
    "v2_rnnt_l2": ("v2_rnnt", 1, 2, (3, 4.0, 23, [64000, 41234, 57000]), {"blank_bias": 13.5}),
    "v1_rnnt_l2": ("v1_rnnt", 1, 2, (2, 2.5, 20, [40000, 26000]), {"blank_bias": 14.0}),
    "v3_rnnt_l2": ("v3_rnnt", 1, 2, (2, 3.0, 21, [48000, 37000]), {"blank_bias": 15.0}),
    "v3_e2e_rnnt_l2": ("v3_e2e_rnnt", 1, 2, (2, 3.0, 35, [48000, 30011]), {"blank_bias": 14.0}),
    # a TWO-layer predictor LSTM (reference decoder.py:78-83 builds nn.LSTM(.., pred_rnn_layers); every published checkpoint
    # has 1): the one-workgroup decode kernel runs it (gam_decode.h), the cluster kernel is for L = 1
    "v2_rnnt_l2_lstm2": ("v2_rnnt", 1, 2, (3, 4.0, 62, [64000, 45000, 52000]), {"blank_bias": 15.0, "pred_rnn_layers": 2}),
    # emission-heavy RNN-T (max_symbols cap reached on most frames)
    "v2_rnnt_l2_dense": ("v2_rnnt", 1, 2, (3, 4.0, 13, [64000, 41234, 57000]), {"blank_bias": 8.0}),
    "v3_e2e_rnnt_l2_dense": ("v3_e2e_rnnt", 1, 2, (2, 3.0, 45, [48000, 30011]), {"blank_bias": 7.0, "max_symbols": 3}),
}
EMO_CASE = ("emo", 1, 2, (2, 3.0, 18, [48000, 36000]), {})   # tests/golden/emo_l2.npz

RNNT_MIN_MARGIN = 2e-3


def make_case_checkpoint(name_or_case):
    """(checkpoint, wav, wav_len) of a case, regenerated from its seeds (numpy PCG64: identical on every box)."""
    from gigaam_amd import synth
    case = CASES[name_or_case] if isinstance(name_or_case, str) else name_or_case
    model, seed, nl, (b, secs, aseed, lens), opt = case
    ck = synth.make_checkpoint(model, seed=seed, n_layers=nl, rnnt_blank_bias=opt.get("blank_bias"),
                               pred_rnn_layers=opt.get("pred_rnn_layers"))
    if "max_symbols" in opt:
        ck["cfg"]["decoding"]["max_symbols_per_step"] = opt["max_symbols"]
    wav, wlen = synth.synth_audio(b, secs, seed=aseed, lengths=lens)
    return ck, wav, wlen
