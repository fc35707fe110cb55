"""Native KD step vs per-launch engine, repeated in one process: which engine is the odd one out when they disagree (developer aid, round 6)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_train_native as T
from fcl_taco2_amd import hparams as HP, synthetic as SYN
from fcl_taco2_amd.training import TrainEngine

DEV = "cuda:0"
SYNC = os.environ.get("DIAG_SYNC", "0") == "1"
share, flags, masking = True, (True, False, True, True), True
S, Th = HP.student_hparams(use_masking=masking), HP.teacher_hparams()
batch = T._batch(21)


def student(native):
    m = SYN.build_model("student", S, Th, DEV, share_proj=share, weights="init", seed=3)
    m.distill_output_knowledge, m.distill_encoder_knowledge, m.distill_decoder_knowledge, m.distill_prosody_knowledge = flags
    return TrainEngine(m, seed=5, native=native)


for rep in range(8):
    t_ref, t_nat = T._engines("kd_teacher", False, seed=11), T._engines("kd_teacher", True, seed=11)
    k_ref = t_ref.knowledge(batch, mode="train")
    k_nat = t_nat.knowledge(batch, mode="train", native=True)
    engs = [student(False), student(True), student(True), student(False)]
    for e in engs:
        e.zero_grad()
    know = [k_ref, k_nat, k_nat, k_ref]
    for step in range(int(os.environ.get("DIAG_STEPS", "1"))):
        for e in engs:
            e.zero_grad()
        for e, k in zip(engs, know):
            e.forward_backward(batch, k, mode="train")
            if SYNC:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    names = ["dec.lstm.0.cell.weight_hh", "dec.postnet.postnet.3.0.weight", "dec.feat_out.weight", "enc.embed.weight"]
    worst = lambda a, b: max(T._rel(a.G[k], b.G[k]) for k in a.G)
    print("rep %d  ref0~ref1 %.1e  nat0~nat1 %.1e  nat0~ref0 %.1e  nat1~ref0 %.1e" % (rep, worst(engs[0], engs[3]), worst(engs[1], engs[2]), worst(engs[1], engs[0]), worst(engs[2], engs[0])))
    for i in (1, 2):
        if worst(engs[i], engs[0]) > 1e-4:
            bad = {k: T._rel(engs[i].G[k], engs[0].G[k]) for k in engs[0].G}
            print("   nat%d bad tensors: %s" % (i - 1, ", ".join("%s %.0e" % (k.replace("dec.", "d.").replace("postnet.postnet", "post").replace("weight", "w"), v) for k, v in bad.items() if v > 2e-5)))
