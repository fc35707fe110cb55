"""Diagnostic: is the full-size gradient error run-to-run stable, and does it depend on the side-stream weight gradients / the cooperating BiLSTM?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import fcl_oracle as O  # noqa: E402
import test_gpu_training_fullsize as TF  # noqa: E402
from fcl_taco2_amd import hparams as HP, synthetic as SYN  # noqa: E402
from fcl_taco2_amd.training import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TF._threads()
T = HP.teacher_hparams()
batch = TF._batch(B, 41, T.idim)
masks = TF.random_masks(T, batch, 7)
model = SYN.build_model("teacher", T, None, "cuda:0", weights="init", seed=1)
sd = {k: (v.detach().cpu().double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else
          (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu().clone())) for k, v in model.state_dict().items()}
b64 = {k: (v.cpu().double() if torch.is_tensor(v) and v.dtype.is_floating_point else (v.cpu() if torch.is_tensor(v) else v)) for k, v in batch.items()}
r = O.model_forward(sd, T, b64, "teacher", bn_train=True, masks=masks)
r["loss"].backward()
keys = ["enc.convs.0.0.weight", "enc.convs.2.0.weight", "enc.blstm.weight_ih_l0", "enc.blstm.weight_hh_l0", "pitch_predictor.conv.0.0.weight",
        "duration_predictor.conv.0.0.weight", "dec.lstm.0.cell.weight_ih", "dec.lstm.1.cell.weight_hh", "dec.feat_out.weight", "dec.postnet.postnet.0.0.weight",
        "dec.prenet.prenet.0.0.weight", "pitch_embed.0.weight"]


def run(tag, **kw):
    m = SYN.build_model("teacher", T, None, "cuda:0", weights="init", seed=1)
    eng = TrainEngine(m, **kw)
    eng.forward_backward(batch, mode="train", masks=masks)
    torch.cuda.synchronize()
    g = {k: eng.G[k].detach().cpu().double().clone() for k in keys}
    print(tag, " ".join("%.1e" % (float((g[k] - sd[k].grad).norm()) / float(sd[k].grad.norm())) for k in keys))
    return g


print("keys:", " ".join(k.replace("predictor", "p").replace("weight", "w") for k in keys))
g1 = run("run1        ")
g2 = run("run2        ")
print("run1 vs run2", " ".join("%.1e" % (float((g1[k] - g2[k]).norm()) / float(g1[k].norm())) for k in keys))
g3 = run("no side dW  ", overlap_dw=False)
os.environ["FCL_BILSTM_TRAIN_STEPS"] = "1"
