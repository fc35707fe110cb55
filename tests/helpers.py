"""Shared test helpers: closed-form state_dicts as torch tensors, tiny hparams used by the goldens."""
import numpy as np
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import hparams as HP
from fcl_taco2_amd import synthetic as SYN

TINY_S = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                            postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.0)
TINY_T = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                            postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0)
# the train-mode goldens (G7, G9) use the same shapes (hence the same closed-form weights) with the shipped dropout rate
TINY_S7 = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                             postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.5)
TINY_T7 = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                             postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.5)


def np_state_dict(hp, thp=None, share_proj=True):
    return SYN.closed_form_state_dict(HP.param_spec(hp, thp, share_proj))


def torch_state_dict(hp, thp=None, share_proj=True):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np_state_dict(hp, thp, share_proj).items()}


def max_abs(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0

# the `--use-masking False` loss variant (G10): same shapes / weights as TINY_S / TINY_T
TINY_SU = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                             postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.0, use_masking=False)
TINY_TU = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                             postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0, use_masking=False)

# the `--use-residual True` variant (G11): same shapes / weights, encoder convs with skip connections
TINY_SR = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                             postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.0, use_residual=True)
TINY_TR = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                             postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0, use_residual=True)

# the `--output-activation sigmoid` variant (G12): same shapes / weights, outputs through torch.nn.functional.sigmoid
TINY_SA = HP.student_hparams(idim=12, odim=8, embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20,
                             postnet_chans=12, duration_predictor_chans=20, dropout_rate=0.0, output_activation="sigmoid")
TINY_TA = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                             postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0, output_activation="sigmoid")

# decoder options outside the shipped recipes (G14): zoneout_rate 0 (plain LSTMCell, no `.cell` key level), use_concate False, append_position False;
# the KD pair keeps use_concate (the reference's KD decoder cannot run forward() without it: tests/golden/records.json)
_OPT = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, zoneout_rate=0.0, append_position=False)
TINY_TO = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, use_concate=False, **_OPT)
TINY_SO = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, use_concate=False, **_OPT)
TINY_TOK = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **_OPT)
TINY_SOK = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **_OPT)

# `--use-batch-norm false` (G15): encoder / postnet blocks without a normalisation layer
_NOBN = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, use_batch_norm=False)
TINY_TN = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, **_NOBN)
TINY_SN = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, **_NOBN)

# encoder widths that all differ (G16; every shipped recipe sets embed_dim == econv_chans == eunits)
_WID = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0)
TINY_TW = HP.teacher_hparams(embed_dim=24, econv_chans=32, eunits=40, dunits=40, prenet_units=28, postnet_chans=20, **_WID)
TINY_SW = HP.student_hparams(embed_dim=12, econv_chans=16, eunits=24, dunits=24, prenet_units=20, postnet_chans=12, **_WID)

# layer counts outside the shipped recipes that the reference's teacher class runs (G17): two encoder blocks, three postnet blocks
TINY_TL = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20,
                             duration_predictor_chans=20, dropout_rate=0.0, econv_layers=2, postnet_layers=3)

# speaker embeddings (G13): F.normalize(spemb) appended to the encoder states; predictors / embeddings / decoder on eunits + 8 channels
TINY_TK = HP.teacher_hparams(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28,
                             postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0, spk_embed_dim=8)

# structure options of the reference's teacher class beyond the shipped recipes (G18 - G20, round 5): decoder cell count, prenet block count,
# stacked encoder BiLSTM
_OPT = dict(idim=12, odim=8, embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, duration_predictor_chans=20, dropout_rate=0.0)
TINY_VARIANTS = {
    "g18_teacher_dlayers1": HP.teacher_hparams(dlayers=1, **_OPT),
    "g18_teacher_dlayers3": HP.teacher_hparams(dlayers=3, **_OPT),
    "g19_teacher_prenet1": HP.teacher_hparams(prenet_layers=1, **_OPT),
    "g19_teacher_prenet3": HP.teacher_hparams(prenet_layers=3, **_OPT),
    "g20_teacher_elayers2": HP.teacher_hparams(elayers=2, **_OPT),
}

# reduction_factor 2 on the teacher class (G21)
TINY_R2 = HP.teacher_hparams(reduction_factor=2, **_OPT)

# the KD classes with the structure options their tap lists allow (G22): teacher with three prenet blocks, student with one; two BiLSTM layers each
_KDS = dict(idim=12, odim=8, duration_predictor_chans=20, dropout_rate=0.0, elayers=2)
TINY_TQ = HP.teacher_hparams(embed_dim=32, eunits=32, econv_chans=32, dunits=40, prenet_units=28, postnet_chans=20, prenet_layers=3, **_KDS)
TINY_SQ = HP.student_hparams(embed_dim=16, eunits=16, econv_chans=16, dunits=24, prenet_units=20, postnet_chans=12, prenet_layers=1, **_KDS)
