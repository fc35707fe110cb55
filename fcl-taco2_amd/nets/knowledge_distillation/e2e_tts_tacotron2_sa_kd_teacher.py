"""Frozen KD teacher plug-in class — mirrors reference
nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_teacher.py:Tacotron2_sa (no inference() there either)."""
from ..base import Tacotron2Base


class Tacotron2_sa(Tacotron2Base):
    role = "kd_teacher"

    def __init__(self, idim, odim, args=None, com_args=None):
        self._setup(idim, odim, args, com_args, None)
