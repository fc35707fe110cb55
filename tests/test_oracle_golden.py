"""Pin the CPU oracle (oracle/fcl_oracle.py) against outputs of the REAL reference (tests/golden/*.npz,
made by oracle/gen_golden.py in the survey container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import TINY_S, TINY_T, max_abs, torch_state_dict
from conftest import GOLDEN
from fcl_taco2_amd import hparams as HP
from fcl_taco2_amd import synthetic as SYN
from oracle import fcl_oracle as O

TOL_STAGE = 1e-5  # per-stage fp32 tolerance on tiny dims (SURVEY.md §8c recommends <= 1e-4)
TOL_MEL = 2e-4    # full-dims mel vs the reference (north_star bar is 1e-3)


def test_manifest_matches_reference():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    S, T = HP.student_hparams(), HP.teacher_hparams()
    for tag, spec in [("student_share", HP.param_spec(S, T, True)), ("student_noshare", HP.param_spec(S, T, False)),
                      ("teacher", HP.param_spec(T)), ("kd_teacher", HP.param_spec(T))]:
        assert {k: list(v) for k, v in spec.items()} == man[tag]
    assert man["student_share_nparams"] == 6466595 and man["teacher_nparams"] == 28972579


@pytest.mark.parametrize("tag,hp,thp,share", [("student_share", TINY_S, TINY_T, True),
                                              ("student_noshare", TINY_S, TINY_T, False),
                                              ("teacher", TINY_T, None, True)])
def test_g1_inference_every_stage(golden, tag, hp, thp, share):
    g = golden("g1_infer_" + tag)
    sd = torch_state_dict(hp, thp, share)
    x, dur = torch.from_numpy(g["x"]), torch.from_numpy(g["dur"])
    with torch.no_grad():
        enc, taps = O.encoder_forward(sd, hp, x.unsqueeze(0), [x.numel()])
        assert max_abs(taps[0][0], g["embed"]) == 0.0
        for i in range(3):
            assert max_abs(taps[1 + i][0], g["conv%d" % i]) < TOL_STAGE
        assert max_abs(enc[0], g["h"]) < TOL_STAGE
        pad = O.make_pad_mask([x.numel()])
        assert max_abs(O.duration_predictor(sd, hp, enc, pad)[0], g["d_log"]) < TOL_STAGE
        assert np.array_equal(O.duration_predictor(sd, hp, enc, pad, inference=True)[0].numpy(), g["d_int"])
        r = O.inference(sd, hp, x, dur=dur)
    for k in ("p_outs", "e_outs", "p_embs", "e_embs", "before", "after"):
        assert max_abs(r[k], g[k]) < TOL_STAGE, k


def test_g2_full_student_mel(golden):
    g = golden("g2_student_c1")
    hp = HP.student_hparams(dropout_rate=0.0)
    sd = torch_state_dict(hp, HP.teacher_hparams())
    x, d = SYN.utterance_c1(hp.idim)
    assert np.array_equal(x, g["x"]) and np.array_equal(d, g["dur"])
    with torch.no_grad():
        r = O.inference(sd, hp, torch.from_numpy(x), dur=torch.from_numpy(d))
    assert max_abs(r["h"], g["h"]) < TOL_MEL
    assert max_abs(r["before"], g["before"]) < TOL_MEL
    assert max_abs(r["after"], g["after"]) < TOL_MEL
    assert r["after"].shape == (int(d.sum()), 80)


def test_g2b_batched_extension(golden):
    g = golden("g2b_student_batch3")
    hp = HP.student_hparams(dropout_rate=0.0)
    sd = torch_state_dict(hp, HP.teacher_hparams())
    xs = [torch.from_numpy(g["x%d" % i]) for i in range(3)]
    ds = [torch.from_numpy(g["dur%d" % i]) for i in range(3)]
    with torch.no_grad():
        mels = O.synthesize_batch(sd, hp, xs, ds)
    for i in range(3):
        assert max_abs(mels[i], g["after%d" % i]) < TOL_MEL


def test_g3_injected_prenet_dropout(golden):
    g = golden("g3_student_c1_masked")
    hp = HP.student_hparams()
    sd = torch_state_dict(hp, HP.teacher_hparams())
    x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["dur"])
    keep = torch.from_numpy(SYN.closed_form_keep_mask((int(d.max()), 2, int((d > 0).sum()), hp.prenet_units), int(g["keep_seed"])))
    with torch.no_grad():
        r = O.inference(sd, hp, x, dur=d, prenet_keep=keep)
    assert max_abs(r["after"], g["after"]) < TOL_MEL


def test_g2t_full_teacher_mel(golden):
    g = golden("g2t_teacher_c1")
    hp = HP.teacher_hparams(dropout_rate=0.0)
    sd = torch_state_dict(hp)
    with torch.no_grad():
        r = O.inference(sd, hp, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
    assert max_abs(r["after"], g["after"]) < TOL_MEL


def test_g4_duration_rounding_bit_exact(golden):
    g = golden("g4_integer")
    lin = torch.from_numpy(g["lin"])
    assert np.array_equal(torch.clamp(torch.round(lin), min=0).long().numpy(), g["lin_round"])
    # half-to-even on exact ties, clamp of negatives
    assert g["lin_round"].tolist()[:11] == [0, 0, 0, 0, 0, 0, 1, 2, 2, 4, 4]
    assert np.array_equal(O.duration_round(torch.from_numpy(g["logits"])).numpy(), g["logits_round"])


def _raw_batch(g, n, pre="in_"):
    return ([g[pre + "xs%d" % i] for i in range(n)], [g[pre + "ys%d" % i] for i in range(n)],
            [g[pre + "ds%d" % i] for i in range(n)], [g[pre + "f0%d" % i] for i in range(n)],
            [g[pre + "en%d" % i] for i in range(n)])


def test_g4_converter_layout_bit_exact(golden):
    g = golden("g4_integer")
    b = O.convert_batch(*_raw_batch(g, 4))
    for k in ("xs", "ilens", "ys", "olens", "extras", "new_ys", "non_zero_lens_mask", "ds_nonzeros",
              "output_masks", "position", "f0", "energy"):
        assert np.array_equal(b[k].numpy(), g["out_" + k]), k
    assert (g["out_non_zero_lens_mask"] == 0).sum() > (g["out_xs"] == 0).sum() - 1  # has zero-duration phonemes


def _know(g):
    return (torch.from_numpy(g["t_after"]), torch.from_numpy(g["t_before"]),
            [torch.from_numpy(g["t_enc%d" % i]) for i in range(5)],
            [torch.from_numpy(g["t_dec%d" % i]) for i in range(8)],
            [torch.from_numpy(g["t_pro%d" % i]) for i in range(5)])


def test_g1_forward_teacher_student_losses(golden):
    g4, g = golden("g4_integer"), golden("g1_forward")
    b = O.convert_batch(*_raw_batch(g4, 4))
    with torch.no_grad():
        kt = O.model_forward(torch_state_dict(TINY_T), TINY_T, b, "kd_teacher")
    know = _know(g)
    assert max_abs(kt[0], know[0]) < TOL_STAGE and max_abs(kt[1], know[1]) < TOL_STAGE
    for mine, ref in zip(kt[2] + kt[3] + kt[4], know[2] + know[3] + know[4]):
        assert max_abs(mine, ref) < TOL_STAGE
    for share in (True, False):
        tag = "student_%s_" % ("share" if share else "noshare")
        with torch.no_grad():
            rep = O.model_forward(torch_state_dict(TINY_S, TINY_T, share), TINY_S, b, "student", TINY_T, share, know)
        for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss", "output_l1_loss",
                  "output_mse_loss", "encoder_loss", "decoder_loss", "prosody_loss"):
            assert abs(float(rep[k]) - float(g[tag + k])) < 1e-4 * max(1.0, abs(float(g[tag + k]))), (k, float(rep[k]), float(g[tag + k]))
    with torch.no_grad():
        rep = O.model_forward(torch_state_dict(TINY_T), TINY_T, b, "teacher")
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss"):
        assert abs(float(rep[k]) - float(g["teacher_" + k])) < 1e-4 * max(1.0, abs(float(g["teacher_" + k]))), k


def test_g5_training_gradients(golden):
    g4, g = golden("g4_integer"), golden("g5_teacher_train")
    b = O.convert_batch(*_raw_batch(g4, 4))
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in torch_state_dict(TINY_T).items()}
    rep = O.model_forward(sd, TINY_T, b, "teacher")
    rep["loss"].backward()
    assert abs(float(rep["loss"]) - float(g["loss"])) < 1e-4
    for k in g:
        if k.startswith("grad:"):
            ref = g[k]
            assert max_abs(sd[k[5:]].grad, ref) < 1e-4 * max(1.0, float(np.abs(ref).max())), k
    params = [v for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in params if p.grad is not None))
    assert abs(float(gn) - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])


def _grad_sd(hp, thp=None, share=True):
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
            for k, v in torch_state_dict(hp, thp, share).items()}


def _check_grads(sd, g, tol=1e-4):
    n = 0
    for k in g:
        if k.startswith("grad:"):
            n += 1
            ref = g[k]
            assert max_abs(sd[k[5:]].grad, ref) < tol * max(1.0, float(np.abs(ref).max())), k
    params = [v for k, v in sd.items() if v.dtype.is_floating_point and v.requires_grad and v.grad is not None]
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in params))
    assert abs(float(gn) - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    return n


def test_g7_teacher_train_mode_with_injected_masks(golden):
    """The oracle's TRAIN form (batch-statistics BatchNorm, every dropout / zoneout draw injected) vs the real reference in model.train()."""
    from helpers import TINY_T7

    g4, g = golden("g4_integer"), golden("g7_teacher_train_mode")
    b = O.convert_batch(*_raw_batch(g4, 4))
    masks = O.masks_from_sequence([g["mask%03d" % i] for i in range(int(g["n_masks"]))], TINY_T7)
    sd = _grad_sd(TINY_T7)
    rep = O.model_forward(sd, TINY_T7, b, "teacher", bn_train=True, masks=masks)
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), k
    assert _check_grads(sd, g) >= 12


def test_g8_student_kd_gradients_eval_form(golden):
    g4, g1, g = golden("g4_integer"), golden("g1_forward"), golden("g8_student_kd_eval")
    b = O.convert_batch(*_raw_batch(g4, 4))
    know = (torch.from_numpy(g1["t_after"]), torch.from_numpy(g1["t_before"]), [torch.from_numpy(g1["t_enc%d" % i]) for i in range(5)],
            [torch.from_numpy(g1["t_dec%d" % i]) for i in range(8)], [torch.from_numpy(g1["t_pro%d" % i]) for i in range(5)])
    sd = _grad_sd(TINY_S, TINY_T, True)
    rep = O.model_forward(sd, TINY_S, b, "student", TINY_T, True, know)
    rep["loss"].backward()
    for k in ("loss", "output_l1_loss", "output_mse_loss", "encoder_loss", "decoder_loss", "prosody_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), k
    assert _check_grads(sd, g) >= 20


def test_g9_kd_step_train_mode(golden):
    """tts_distill.py:159-161 in train mode: frozen train-mode teacher (its own masks) -> student forward/backward (its masks)."""
    from helpers import TINY_S7, TINY_T7

    g4, g = golden("g4_integer"), golden("g9_kd_step_train_mode")
    b = O.convert_batch(*_raw_batch(g4, 4))
    tm = O.masks_from_sequence([g["tmask%03d" % i] for i in range(int(g["n_tmasks"]))], TINY_T7)
    sm = O.masks_from_sequence([g["smask%03d" % i] for i in range(int(g["n_smasks"]))], TINY_S7)
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_T7), TINY_T7, b, "kd_teacher", bn_train=True, masks=tm)
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[3][1], g["t_dec1"]) < TOL_STAGE and max_abs(know[4][3], g["t_pro3"]) < TOL_STAGE
    sd = _grad_sd(TINY_S7, TINY_T7, True)
    rep = O.model_forward(sd, TINY_S7, b, "student", TINY_T7, True, know, bn_train=True, masks=sm)
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "output_l1_loss", "output_mse_loss", "encoder_loss", "decoder_loss", "prosody_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), k
    assert _check_grads(sd, g) >= 20


def test_g10_unmasked_loss_variant(golden):
    """`--use-masking False` (the reference's argparse default): mel / prosody / output-KD means over the padded tensors; duration loss and the
    encoder / decoder / prosody KD terms masked as ever.  Teacher step and student KD step vs the real reference (losses + gradients)."""
    from helpers import TINY_SU, TINY_TU

    g4, g1 = golden("g4_integer"), golden("g1_forward")
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g10_teacher_unmasked")
    sd = _grad_sd(TINY_TU)
    rep = O.model_forward(sd, TINY_TU, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 12
    assert abs(float(g["l1_loss"]) - float(golden("g5_teacher_train")["loss"])) > 1e-2  # (it IS a different objective than the masked one)
    g = golden("g10_student_kd_unmasked")
    know = (torch.from_numpy(g1["t_after"]), torch.from_numpy(g1["t_before"]), [torch.from_numpy(g1["t_enc%d" % i]) for i in range(5)],
            [torch.from_numpy(g1["t_dec%d" % i]) for i in range(8)], [torch.from_numpy(g1["t_pro%d" % i]) for i in range(5)])
    sd = _grad_sd(TINY_SU, TINY_TU, True)
    rep = O.model_forward(sd, TINY_SU, b, "student", TINY_TU, True, know)
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss", "output_l1_loss", "output_mse_loss", "encoder_loss", "decoder_loss",
              "prosody_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20
    import json

    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "records.json")))
    assert rec["use_weighted_masking"].startswith("RuntimeError")  # the reference itself cannot run that variant: refusing it is parity


def test_g11_use_residual_variant(golden):
    """`--use-residual True` (the reference's argparse default): encoder `convs[i](xs) + xs`.  Inference mel (teacher class) and the student KD
    step against a use_residual KD teacher vs the real reference.  (The plain teacher class cannot TRAIN with it in the reference: its encoder
    adds in place and autograd raises -- recorded in records.json.)"""
    import json

    from helpers import TINY_SR, TINY_TR

    g4, g = golden("g4_integer"), golden("g11_teacher_residual")
    sd0 = torch_state_dict(TINY_TR)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        out = O.inference(sd0, TINY_TR, x, dur=torch.from_numpy(g["dur"]))
        h = O.encoder_inference(sd0, TINY_TR, x)
    assert max_abs(out["after"], g["after"]) < TOL_STAGE and max_abs(h, g["h"]) < TOL_STAGE
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g11_student_kd_residual")
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TR), TINY_TR, b, "kd_teacher")
    assert max_abs(know[2][1], g["t_enc1"]) < TOL_STAGE and max_abs(know[2][4], g["t_enc4"]) < TOL_STAGE
    sd = _grad_sd(TINY_SR, TINY_TR, True)
    rep = O.model_forward(sd, TINY_SR, b, "student", TINY_TR, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "records.json")))
    assert rec["use_residual_teacher_training"].startswith("RuntimeError")


def test_g12_output_activation_variant(golden):
    """`--output-activation sigmoid`: the activated frame is fed back in the free-running loop, the final output is activated, forward() activates
    both outputs behind the postnet.  Inference mel, teacher step and student KD step vs the real reference."""
    from helpers import TINY_SA, TINY_TA

    g4, g = golden("g4_integer"), golden("g12_teacher_sigmoid_inference")
    sd0 = torch_state_dict(TINY_TA)
    with torch.no_grad():
        out = O.inference(sd0, TINY_TA, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
    assert max_abs(out["after"], g["after"]) < TOL_STAGE and float(out["after"].min()) > 0.0 and float(out["after"].max()) < 1.0
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g12_teacher_sigmoid")
    sd = _grad_sd(TINY_TA)
    rep = O.model_forward(sd, TINY_TA, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 10
    g = golden("g12_student_kd_sigmoid")
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TA), TINY_TA, b, "kd_teacher")
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[1], g["t_before"]) < TOL_STAGE
    sd = _grad_sd(TINY_SA, TINY_TA, True)
    rep = O.model_forward(sd, TINY_SA, b, "student", TINY_TA, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss", "output_mse_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20


def test_g14_decoder_options(golden):
    """zoneout_rate 0 + use_concate False + append_position False (teacher: inference mel, training step; student: inference mel) and the KD step on
    the two options the reference's KD decoder can run -- all against the real reference; its failure with use_concate False is on record."""
    from helpers import TINY_SO, TINY_SOK, TINY_TO, TINY_TOK

    g4 = golden("g4_integer")
    for hp, name in ((TINY_TO, "g14_teacher_options_inference"), (TINY_SO, "g14_student_options_inference")):
        g = golden(name)
        thp = TINY_TO if hp is TINY_SO else None
        sd0 = torch_state_dict(hp, thp, True) if thp is not None else torch_state_dict(hp)
        assert "dec.lstm.0.weight_ih" in sd0 and sd0["dec.feat_out.weight"].shape[1] == hp.dunits and sd0["dec.lstm.0.weight_ih"].shape[1] == hp.adim + hp.prenet_units
        with torch.no_grad():
            out = O.inference(sd0, hp, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
        assert max_abs(out["after"], g["after"]) < TOL_STAGE, name
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g14_teacher_options")
    sd = _grad_sd(TINY_TO)
    rep = O.model_forward(sd, TINY_TO, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 10
    g = golden("g14_student_kd_options")
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TOK), TINY_TOK, b, "kd_teacher")
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[1], g["t_before"]) < TOL_STAGE
    sd = _grad_sd(TINY_SOK, TINY_TOK, True)
    rep = O.model_forward(sd, TINY_SOK, b, "student", TINY_TOK, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss", "output_mse_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "records.json")))
    assert rec["use_concate_false_kd_forward"].startswith("TypeError")
    # what the reference does with the options the HIP path still refuses (hparams.check_supported / TrainEngine): it fails on them itself, or only
    # its KD classes do
    assert rec["use_fe_condition_false_teacher"].startswith("AttributeError") and rec["use_fe_condition_false_kd"].startswith("AttributeError")
    assert rec["prenet_layers_0_teacher"].startswith("AttributeError") and rec["postnet_layers_0_teacher"].startswith("TypeError")
    assert rec["postnet_layers_3_teacher"] == "runs" and rec["postnet_layers_3_kd"].startswith("IndexError")
    assert rec["econv_layers_2_teacher"] == "runs" and rec["econv_layers_2_kd"].startswith("IndexError")


def test_g15_no_batch_norm(golden):
    """`--use-batch-norm false`: inference mels (teacher, student), teacher step, student KD step vs the real reference."""
    from helpers import TINY_SN, TINY_TN

    g4 = golden("g4_integer")
    for hp, thp, name in ((TINY_TN, None, "g15_teacher_nobn_inference"), (TINY_SN, TINY_TN, "g15_student_nobn_inference")):
        g = golden(name)
        sd0 = torch_state_dict(hp, thp, True) if thp is not None else torch_state_dict(hp)
        assert not any(".1.running_mean" in k for k in sd0)
        with torch.no_grad():
            out = O.inference(sd0, hp, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
        assert max_abs(out["after"], g["after"]) < TOL_STAGE, name
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g15_teacher_nobn")
    sd = _grad_sd(TINY_TN)
    rep = O.model_forward(sd, TINY_TN, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 10
    g = golden("g15_student_kd_nobn")
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TN), TINY_TN, b, "kd_teacher")
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[1], g["t_before"]) < TOL_STAGE and max_abs(know[2][1], g["t_enc1"]) < TOL_STAGE
    sd = _grad_sd(TINY_SN, TINY_TN, True)
    rep = O.model_forward(sd, TINY_SN, b, "student", TINY_TN, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss", "output_mse_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20


def test_g16_encoder_widths_differ(golden):
    """embed_dim != econv_chans != eunits (teacher 24 / 32 / 40, student 12 / 16 / 24): inference mels, teacher step, student KD step vs the real reference."""
    from helpers import TINY_SW, TINY_TW

    g4 = golden("g4_integer")
    for hp, thp, name in ((TINY_TW, None, "g16_teacher_widths_inference"), (TINY_SW, TINY_TW, "g16_student_widths_inference")):
        g = golden(name)
        sd0 = torch_state_dict(hp, thp, True) if thp is not None else torch_state_dict(hp)
        with torch.no_grad():
            out = O.inference(sd0, hp, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
        assert max_abs(out["after"], g["after"]) < TOL_STAGE, name
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g16_teacher_widths")
    sd = _grad_sd(TINY_TW)
    rep = O.model_forward(sd, TINY_TW, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 12
    g = golden("g16_student_kd_widths")
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TW), TINY_TW, b, "kd_teacher")
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[2][0], g["t_enc0"]) < TOL_STAGE and max_abs(know[2][4], g["t_enc4"]) < TOL_STAGE
    sd = _grad_sd(TINY_SW, TINY_TW, True)
    rep = O.model_forward(sd, TINY_SW, b, "student", TINY_TW, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss", "output_mse_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 20


def test_g17_layer_counts(golden):
    """econv_layers 2, postnet_layers 3 (teacher class): inference mel and training step vs the real reference."""
    from helpers import TINY_TL

    g4, g = golden("g4_integer"), golden("g17_teacher_layers_inference")
    with torch.no_grad():
        out = O.inference(torch_state_dict(TINY_TL), TINY_TL, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
    assert max_abs(out["after"], g["after"]) < TOL_STAGE
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden("g17_teacher_layers")
    sd = _grad_sd(TINY_TL)
    rep = O.model_forward(sd, TINY_TL, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 12


@pytest.mark.parametrize("name", ["g18_teacher_dlayers1", "g18_teacher_dlayers3", "g19_teacher_prenet1", "g19_teacher_prenet3", "g20_teacher_elayers2"])
def test_g18_g19_g20_structure_options(golden, name):
    """`dlayers` 1 / 3 (decoder_sa.py:357-369, 500-504), `prenet_layers` 1 / 3 (decoder_sa.py:119-158), `elayers` 2 (encoder_sa.py:96-100) on the
    teacher class: inference mel and training step (losses, gradients) vs the real reference."""
    from helpers import TINY_VARIANTS

    hp = TINY_VARIANTS[name]
    g4, g = golden("g4_integer"), golden(name + "_inference")
    with torch.no_grad():
        out = O.inference(torch_state_dict(hp), hp, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]))
    assert max_abs(out["after"], g["after"]) < TOL_STAGE
    b = O.convert_batch(*_raw_batch(g4, 4))
    g = golden(name)
    sd = _grad_sd(hp)
    rep = O.model_forward(sd, hp, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 10


def test_g22_kd_classes_with_structure_options(golden):
    """KD teacher with `prenet_layers` 3 / `elayers` 2, student with `prenet_layers` 1 / `elayers` 2: the teacher's 5-tuple and the student's KD step
    (losses, gradients incl. the projections and the second BiLSTM layer) vs the real reference."""
    from helpers import TINY_SQ, TINY_TQ

    g4, g = golden("g4_integer"), golden("g22_student_kd_structure")
    b = O.convert_batch(*_raw_batch(g4, 4))
    with torch.no_grad():
        know = O.model_forward(torch_state_dict(TINY_TQ), TINY_TQ, b, "kd_teacher")
    assert max_abs(know[0], g["t_after"]) < TOL_STAGE and max_abs(know[2][4], g["t_enc4"]) < TOL_STAGE
    assert max_abs(know[3][0], g["t_dec0"]) < TOL_STAGE and max_abs(know[3][2], g["t_dec2"]) < TOL_STAGE
    sd = _grad_sd(TINY_SQ, TINY_TQ, True)
    rep = O.model_forward(sd, TINY_SQ, b, "student", TINY_TQ, True, know)
    rep["loss"].backward()
    for k in ("loss", "encoder_loss", "decoder_loss", "prosody_loss", "output_l1_loss", "output_mse_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 18


def test_g21_reduction_factor_2(golden):
    """`reduction_factor` 2 (teacher class): the converter's layout in frames (tts.py:250-258) bit-exact, inference mel (position t / d in steps,
    r frames per step), training step (every r-th target frame teacher-forced, position t / (r d) from the converter, targets cut to whole groups)."""
    from helpers import TINY_R2

    g = golden("g21_teacher_r2")
    b = O.convert_batch(*_raw_batch(g, 4), reduction_factor=2)
    for k in ("xs", "ilens", "ys", "olens", "extras", "new_ys", "non_zero_lens_mask", "ds_nonzeros", "output_masks", "position", "f0", "energy"):
        assert np.array_equal(b[k].numpy(), g["out_" + k]), k
    gi = golden("g21_teacher_r2_inference")
    with torch.no_grad():
        out = O.inference(torch_state_dict(TINY_R2), TINY_R2, torch.from_numpy(gi["x"]), dur=torch.from_numpy(gi["dur"]))
    assert out["after"].shape == gi["after"].shape and max_abs(out["after"], gi["after"]) < TOL_STAGE
    sd = _grad_sd(TINY_R2)
    rep = O.model_forward(sd, TINY_R2, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 10


def test_g13_speaker_embeddings(golden):
    """`spk_embed_dim`: F.normalize(spemb) appended to every encoder state (..._sa.py:555-557, 636-638).  Inference mel, the teacher step and the KD
    teacher's 5-tuple vs the real reference (the KD student cannot run with speaker embeddings in the reference: records.json)."""
    from helpers import TINY_TK

    g4, g = golden("g4_integer"), golden("g13_teacher_spk_inference")
    sd0 = torch_state_dict(TINY_TK)
    with torch.no_grad():
        out = O.inference(sd0, TINY_TK, torch.from_numpy(g["x"]), dur=torch.from_numpy(g["dur"]), spemb=torch.from_numpy(g["spemb"]))
    assert max_abs(out["after"], g["after"]) < TOL_STAGE
    g = golden("g13_teacher_spk")
    b = O.convert_batch(*_raw_batch(g4, 4))
    b["spembs"] = torch.from_numpy(g["spembs"])
    sd = _grad_sd(TINY_TK)
    rep = O.model_forward(sd, TINY_TK, b, "teacher")
    rep["loss"].backward()
    for k in ("loss", "l1_loss", "mse_loss", "dur_loss", "pitch_loss", "energy_loss"):
        assert abs(float(rep[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, float(rep[k]), float(g[k]))
    assert _check_grads(sd, g) >= 12
    gk = golden("g13_kd_teacher_spk")
    with torch.no_grad():
        know = O.model_forward(sd0, TINY_TK, b, "kd_teacher")
    assert know[2][4].shape[-1] == TINY_TK.eunits and know[4][3].shape[-1] == TINY_TK.adim
    for got, key in ((know[0], "after"), (know[2][4], "enc4"), (know[3][1], "dec1"), (know[4][3], "p_embs"), (know[4][0], "d_outs")):
        assert max_abs(got, gk[key]) < TOL_STAGE, key


def test_g6_padding_leak_and_zero_duration(golden):
    g = golden("g6_padding_leak")
    rec = json.load(open(os.path.join(GOLDEN, "records.json")))
    sd = torch_state_dict(TINY_T)
    with torch.no_grad():
        hs, _ = O.encoder_forward(sd, TINY_T, torch.from_numpy(g["xs"]), g["ilens"].tolist())
        h1 = O.encoder_inference(sd, TINY_T, torch.from_numpy(g["xs"][1, :5]))
    assert max_abs(hs, g["hs_batched"]) < TOL_STAGE
    assert max_abs(h1, g["h1_single"]) < TOL_STAGE
    assert max_abs(hs[1, :5], h1) > 1e-3  # the leak is real and reproduced
    assert rec["zero_duration"] == "AssertionError"
    with pytest.raises(AssertionError):
        O.inference(sd, TINY_T, torch.from_numpy(g["xs"][1, :5]), dur=torch.tensor([2, 0, 1, 3, 1]))
