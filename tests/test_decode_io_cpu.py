"""N1/N3 host glue: Kaldi ark/scp round trip and model.json / checkpoint parsing (CPU)."""
import json

import numpy as np
import pytest
import torch

import fcl_taco2_amd  # noqa: F401
from fcl_taco2_amd import decode as D
from fcl_taco2_amd.kaldi_io import ArkScpWriter, read_scp


def test_ark_scp_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    mats = {"LJ001-0001": rng.randn(7, 80).astype(np.float32), "utt_b": rng.randn(1, 80).astype(np.float32), "c": np.zeros((0, 80), np.float32)}
    with ArkScpWriter(str(tmp_path / "feats")) as w:
        for k, m in mats.items():
            w[k] = m
    back = read_scp(str(tmp_path / "feats.scp"))
    assert list(back) == list(mats)
    for k in mats:
        assert back[k].shape == mats[k].shape and np.array_equal(back[k], mats[k])
    raw = open(tmp_path / "feats.ark", "rb").read()
    assert raw.startswith(b"LJ001-0001 \0BFM \x04\x07\x00\x00\x00\x04\x50\x00\x00\x00")  # Kaldi binary FloatMatrix header


def test_native_batch_writer_produces_the_same_files(tmp_path):
    """ArkScpWriter.write_batch (fcl_kaldi_ark_append: one writev per batch straight from the landing buffer) == one `w[key] = mat` per utterance:
    byte-identical ark and scp, also when the two forms are mixed in one file and for empty matrices."""
    rng = np.random.RandomState(1)
    counts = [5, 0, 13, 1, 700]
    keys = ["utt%03d" % i for i in range(len(counts))]
    big = rng.randn(sum(counts), 80).astype(np.float32)
    with ArkScpWriter(str(tmp_path / "a")) as w:
        w["first"] = big[:3]
        s = 0
        for k, c in zip(keys, counts):
            w[k] = big[s : s + c]
            s += c
        w["last"] = big[:2]
    with ArkScpWriter(str(tmp_path / "b")) as w:
        w["first"] = big[:3]
        w.write_batch(keys, big, counts)
        w["last"] = big[:2]
    assert open(tmp_path / "a.ark", "rb").read() == open(tmp_path / "b.ark", "rb").read()
    assert open(tmp_path / "a.scp").read().replace("/a.ark", "/x.ark") == open(tmp_path / "b.scp").read().replace("/b.ark", "/x.ark")
    back = read_scp(str(tmp_path / "b.scp"))
    assert list(back) == ["first"] + keys + ["last"] and np.array_equal(back["utt004"], big[-700:]) and back["utt001"].shape == (0, 80)
    with pytest.raises(Exception):
        with ArkScpWriter(str(tmp_path / "c")) as w:
            w.write_batch(["bad key"], big[:1], [1])


def test_speaker_embedding_vectors_in_the_manifest(tmp_path):
    """data.json of a multi-speaker recipe: input[1].feat = "<ark>:<offset>" of a Kaldi FloatVector (x-vector) per utterance (tts.py:327-332)."""
    import struct

    from fcl_taco2_amd.decode import read_manifest
    from fcl_taco2_amd.kaldi_io import read_vec

    v = np.arange(8, dtype=np.float32) - 3.0
    ark = tmp_path / "xvector.ark"
    with open(ark, "wb") as f:
        f.write(b"spk1 ")
        off = f.tell()
        f.write(b"\0BFV \x04" + struct.pack("<i", 8) + v.tobytes())
    assert np.array_equal(read_vec(str(ark), off), v)
    js = {"utts": {"u1": {"input": [{"name": "input1"}, {"name": "input2", "feat": "%s:%d" % (ark, off), "shape": [8]}],
                          "output": [{"tokenid": "3 4 5"}]},
                   "u2": {"input": [{"name": "input1"}], "output": [{"tokenid": "7"}]}}}
    p = tmp_path / "data.json"
    p.write_text(json.dumps(js))
    utts = read_manifest(str(p))
    assert utts[0][0] == "u1" and np.array_equal(utts[0][1], [3, 4, 5]) and np.array_equal(utts[0][2], v) and len(utts[1]) == 2


def test_model_conf_manifest_and_checkpoint_formats(tmp_path):
    conf = tmp_path / "model.json"
    args = dict(model_module="nets.knowledge_distillation.e2e_tts_tacotron2_sa_kd_student:Tacotron2_sa", embed_dim=256, eunits=256,
                share_proj=True)
    conf.write_text(json.dumps([80, 80, args]))
    idim, odim, ns = D.get_model_conf(str(conf))
    assert (idim, odim, ns.embed_dim) == (80, 80, 256)
    cls = D.dynamic_import(ns.model_module)  # reference class path -> this package's class
    assert cls.__module__.startswith("fcl_taco2_amd.nets.knowledge_distillation") and cls.role == "student"
    sd = {"enc.embed.weight": torch.zeros(2, 2)}
    for name, obj in (("snapshot.ep.1", {"model": sd, "optimizer": {}}), ("amp_checkpoint_10.pt", {"model": sd, "optimizer": {}, "amp": {}}),
                      ("model.loss.best", sd), ("dp.pt", {"module.enc.embed.weight": torch.zeros(2, 2)})):
        torch.save(obj, tmp_path / name)
        assert list(D.load_state_dict(str(tmp_path / name))) == ["enc.embed.weight"]
    man = tmp_path / "data.json"
    man.write_text(json.dumps({"utts": {"a": {"output": [{"tokenid": "3 4 5"}]}, "b": {"output": [{"tokenid": "7"}]}}}))
    utts = D.read_manifest(str(man))
    assert utts[0][0] == "a" and utts[0][1].tolist() == [3, 4, 5] and utts[1][1].tolist() == [7]


def test_vocoder_driver_host_side(tmp_path):
    """parallel-wavegan-decode replacement, host parts: batch packing, config.yml -> generator geometry, 16-bit PCM writer, checkpoint nesting."""
    import wave

    from fcl_taco2_amd import vocoder_decode as VD

    assert VD.make_batches([5, 9, 1, 7, 3], 10) == [[1], [3], [0, 4, 2]]  # longest first, at most 10 frames per batch
    assert VD.make_batches([50], 10) == [[0]] and VD.make_batches([], 10) == []
    ck = tmp_path / "ck.pkl"
    torch.save({"model": {"generator": {"first_conv.bias": torch.ones(3)}, "discriminator": {"x": torch.zeros(1)}}, "optimizer": {}}, ck)
    sd = VD.load_checkpoint(str(ck))
    assert list(sd["model"]) == ["generator"] and isinstance(sd["model"]["generator"]["first_conv.bias"], np.ndarray)
    assert VD.generator_config(str(ck)) == ({}, 22050)  # no config.yml beside the checkpoint: the published v1 geometry
    (tmp_path / "config.yml").write_text("sampling_rate: 16000\ngenerator_params:\n  layers: 4\n  stacks: 2\n  aux_channels: 20\n"
                                         "  upsample_params:\n    upsample_scales: [2, 3]\n")
    cfg, rate = VD.generator_config(str(ck))
    assert cfg == dict(layers=4, stacks=2, aux_channels=20, upsample_scales=(2, 3)) and rate == 16000
    (tmp_path / "bad.yml").write_text("generator_params:\n  use_causal_conv: true\n")
    try:
        VD.generator_config(str(ck), str(tmp_path / "bad.yml"))
        assert False
    except NotImplementedError:
        pass
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 1e-5], dtype=np.float32)
    VD.write_wav(str(tmp_path / "a.wav"), x, 22050)
    with wave.open(str(tmp_path / "a.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 8)
        pcm = np.frombuffer(w.readframes(8), dtype="<i2")
    assert pcm.tolist() == [0, 16384, -16384, 32767, -32767, 32767, -32768, 0]
