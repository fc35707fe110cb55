"""FCL-taco2-S plug-in class — mirrors reference
nets/knowledge_distillation/e2e_tts_tacotron2_sa_kd_student.py:Tacotron2_sa."""
from ..base import Tacotron2Base


class Tacotron2_sa(Tacotron2Base):
    role = "student"

    def __init__(self, idim, odim, args=None, com_args=None, teacher_args=None):
        self._setup(idim, odim, args, com_args, teacher_args)
