"""FCL-taco2-T plug-in class — mirrors reference nets/teacher_training/e2e_tts_tacotron2_sa.py:Tacotron2_sa."""
from ..base import Tacotron2Base


class Tacotron2_sa(Tacotron2Base):
    role = "teacher"

    def __init__(self, idim, odim, args=None, com_args=None):
        self._setup(idim, odim, args, com_args, None)
