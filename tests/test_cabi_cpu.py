"""CPU-only checks of the drop-in boundary: libfcl_hip.so loads, exports every symbol include/fcl_hip.h
declares (and nothing undeclared), and rejects bad arguments with an error code + message — all without
touching a GPU (argument validation returns before any HIP call)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "fcl_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from fcl_taco2_amd import _lib

    return _lib.load()


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fcl_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from fcl_taco2_amd import _lib

    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names  # the ctypes table mirrors the header one to one
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\bT (fcl_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == names  # nothing undeclared leaks out of the C ABI


def test_version_and_error_string(lib):
    from fcl_taco2_amd import _lib as L

    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fcl_hip.h")).read()
    assert "#define FCL_ABI_VERSION %d" % L.ABI_VERSION in header and lib.fcl_version() == L.ABI_VERSION
    rc = lib.fcl_linear_fwd(None, 4, None, 4, None, None, 4, 1, 4, 4, 0, None)
    assert rc == -1 and b"null" in lib.fcl_last_error()
    rc = lib.fcl_conv1d_fwd(1, 1, None, 1, 1, None, 2, 4, 4, 4, 4, 0, None)  # even kernel size
    assert rc == -2 and b"odd" in lib.fcl_last_error()
    assert lib.fcl_bilstm_workspace_bytes(32, 100, 128) >= 4 * (2 * 32 * 100 * 512 + 6 * 32 * 128)


def test_decoder_loop_argument_validation(lib):
    from fcl_taco2_amd import _lib

    w = _lib.DecoderWeights()
    io = _lib.DecoderIO()
    w.c, w.p, w.u, w.odim = 256, 256, 256, 81  # odim not a multiple of 4
    assert lib.fcl_decoder_loop_fwd(C.byref(w), C.byref(io), None) == -2
    w.odim = 80
    assert lib.fcl_decoder_loop_fwd(C.byref(w), C.byref(io), None) == -1  # null weights
    assert lib.fcl_decoder_loop_workspace_bytes(C.byref(w), 2500) > 2500 * 4 * (1024 + 80 + 512 + 6 * 256 + 80)
    # the weight stream of the persistent row-tile kernel: 8 + 2 * 3 + 16 + 256 slots of 16 KB for FCL-taco2-S; other widths are not covered
    assert lib.fcl_decoder_stream_bytes(C.byref(w)) == (8 + 6 + 16 + 256) * 16384
    assert lib.fcl_decoder_stream_pack(C.byref(w), None, 0, None) == -1
    buf = (C.c_char * 64)()
    assert lib.fcl_decoder_stream_pack(C.byref(w), C.cast(buf, C.c_void_p), 64, None) == -5  # FCL_ERR_WORKSPACE: too small
    w.u = 1024
    assert lib.fcl_decoder_stream_bytes(C.byref(w)) == 0 and lib.fcl_decoder_stream_pack(C.byref(w), C.cast(buf, C.c_void_p), 64, None) == -2


def test_missing_library_fails_loudly(monkeypatch):
    from fcl_taco2_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfcl_hip.so")
    with pytest.raises(_lib.FclError, match="no CPU fallback"):
        _lib.load()


def test_plan_refuses_cpu_device():
    import numpy as np
    from fcl_taco2_amd import _lib, hparams as HP, synthetic as SYN
    from fcl_taco2_amd.plan import SynthesisPlan

    hp = HP.student_hparams()
    with pytest.raises(_lib.FclError, match="GPU"):
        SynthesisPlan(SYN.closed_form_state_dict(HP.param_spec(hp)), hp, "cpu")
    HP.student_hparams(reduction_factor=2, dlayers=3, prenet_layers=1, elayers=2).check_supported()  # round 5: the teacher class's structure options
    for bad in (dict(reduction_factor=0), dict(dlayers=4), dict(prenet_layers=0), dict(elayers=0), dict(use_fe_condition=False)):
        with pytest.raises(NotImplementedError):
            HP.student_hparams(**bad).check_supported()


def test_round2_entry_points_validate_arguments_without_a_gpu(lib):
    """Argument validation of the round-2 entry points returns before any HIP call: GEMM arithmetic mode, batched operand forms, weight-gradient
    GEMM on transposed planes, the vocoder block."""
    from fcl_taco2_amd import _lib

    assert lib.fcl_get_gemm_mode() == _lib.GEMM_F32
    assert lib.fcl_set_gemm_mode(7) == -1 and b"unknown mode" in lib.fcl_last_error()
    assert lib.fcl_set_gemm_mode(_lib.GEMM_BF16) == 0 and lib.fcl_get_gemm_mode() == _lib.GEMM_BF16
    assert lib.fcl_set_gemm_mode(_lib.GEMM_F32) == 0
    assert lib.fcl_derive_blocks(5, 40, 24) == ((5 * 40 + 31) // 32) * 1 and lib.fcl_derive_blocks(0, 4, 4) == 0
    assert lib.fcl_derive_batch(None, 3, 10, None) == -1  # a table is required
    assert lib.fcl_derive_batch(None, 0, 0, None) == 0    # nothing to do
    assert lib.fcl_pack_planes_t(None, 4, 4, 4, 1, 0, None, None, None, None) == -1
    assert lib.fcl_gemm_tn_planes(128, 128, 128, 8, 100, 8, 12, 5, 0, None) == -2 and b"nblk" in lib.fcl_last_error()  # nblk must be a multiple of 4 dividing k
    a = _lib.PwgLayer()
    assert lib.fcl_pwg_layer_fwd(C.byref(a), None) == -1
    assert lib.fcl_pwg_noise(None, 16, 1, None) == -1
    assert lib.fcl_pwg_first_conv(1, 1, 1, None, 128, 16, 48, 0, None) == -1 and b"multiple of 32" in lib.fcl_last_error()
    # the frame-rate form of the auxiliary term: coefficient lines need both buffers; the block needs the one-launch form, both window alignments
    # and a hop that keeps a 128-sample tile inside one frame
    assert lib.fcl_pwg_aux_coeff(None, 256, 256, 1, 128, None) == -1
    assert lib.fcl_pwg_aux_coeff(16, 512, 256, 1, 128, None) == -1  # more samples than frames * hop
    a = _lib.PwgLayer()
    a.m, a.r, a.aux, a.ksize, a.dilation = 256, 64, 80, 3, 1
    for f in ("seg_lo", "seg_hi", "xp", "w_conv_p", "b_conv", "w_os_p", "b_os", "skips", "kp", "pt_a", "pt_b"):
        setattr(a, f, 128)
    a.x, a.ld_pt, a.hop = 128, 1, 256
    assert lib.fcl_pwg_layer_fwd(C.byref(a), None) == -1 and b"one-launch form" in lib.fcl_last_error()  # xp_out missing
    a.xp_out, a.hop = 256, 192
    assert lib.fcl_pwg_layer_fwd(C.byref(a), None) == -1 and b"multiple of 128" in lib.fcl_last_error()
    a.hop, a.kp = 256, 130
    assert lib.fcl_pwg_layer_fwd(C.byref(a), None) == -3 and b"128-byte aligned" in lib.fcl_last_error()
    a.kp, a.ld_pt, a.m = 128, 1, 256 * 40
    assert lib.fcl_pwg_layer_fwd(C.byref(a), None) == -2 and b"ld_pt" in lib.fcl_last_error()  # 40 frames + 16 > 32 columns
    assert lib.fcl_debug_ptr() is None


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors in _lib.py are laid out exactly like the structs of include/fcl_hip.h (sizeof and the offset of every mirrored field's
    last member): a C program that includes the header prints them, gcc compiles it here."""
    from fcl_taco2_amd import _lib

    pairs = [("fcl_decoder_weights_t", _lib.DecoderWeights), ("fcl_decoder_io_t", _lib.DecoderIO), ("fcl_gemm_term_t", _lib.GemmTerm),
             ("fcl_lstm_step_t", _lib.LstmStep), ("fcl_decoder_train_t", _lib.DecoderTrain), ("fcl_decoder_bptt_t", _lib.DecoderBptt),
             ("fcl_bilstm_train_t", _lib.BilstmTrain), ("fcl_bilstm_bptt_t", _lib.BilstmBptt), ("fcl_derive_t", _lib.Derive),
             ("fcl_pwg_layer_t", _lib.PwgLayer), ("fcl_prof_entry_t", _lib.ProfEntry), ("fcl_row_maps_t", _lib.RowMaps),
             ("fcl_te_config_t", _lib.TeConfig), ("fcl_te_batch_t", _lib.TeBatch), ("fcl_te_knowledge_t", _lib.TeKnowledge),
             ("fcl_loss_term_t", _lib.LossTerm)]
    body = ['#include <stdio.h>', '#include <stddef.h>', '#include "fcl_hip.h"', "int main(void) {"]
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        body.append('  printf("%s %%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (cname, cname, cname, last))
    body += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict((ln.split()[0], tuple(int(v) for v in ln.split()[1:])) for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        assert out[cname] == (C.sizeof(cls), getattr(cls, last).offset), (cname, out[cname], C.sizeof(cls), getattr(cls, last).offset)


def test_row_maps_argument_validation(lib):
    from fcl_taco2_amd import _lib

    a = _lib.RowMaps()
    assert lib.fcl_row_maps_build(None, None) == -1
    assert lib.fcl_row_maps_build(C.byref(a), None) == -2  # sizes
    a.b, a.n, a.lmax_cap, a.frames_cap = 2, 10, 8, 64
    assert lib.fcl_row_maps_build(C.byref(a), None) == -1 and b"exactly one" in lib.fcl_last_error()
    a.dur_i32 = 128
    assert lib.fcl_row_maps_build(C.byref(a), None) == -2 and b"padded" in lib.fcl_last_error()  # no utt_row0 and n != b * t_max
    a.t_max = 5
    assert lib.fcl_row_maps_build(C.byref(a), None) == -1 and b"null pointer" in lib.fcl_last_error()


def test_round3_entry_points_validate_arguments_without_a_gpu(lib):
    """The entry points added at the end of round 3 return FCL_ERR_* before any HIP call on bad arguments: the in-graph input feed, the BatchNorm
    backward's paired accumulators, the second column sum, the BiLSTM's optional row maps, the Kaldi batch writer."""
    from fcl_taco2_amd import _lib

    assert lib.fcl_feed_copy(None, 128, 64, 128, 128, None, None) == -1
    assert lib.fcl_feed_copy(256, 128, 60, 128, 128, None, None) == -1 and b"feed_copy" in lib.fcl_last_error()   # bytes % 16
    assert lib.fcl_feed_copy(264, 128, 64, 128, 128, None, None) == -3 and b"16-byte" in lib.fcl_last_error()      # misaligned destination
    assert lib.fcl_host_device_ptr(None) is None and b"host_device_ptr" in lib.fcl_last_error()
    assert lib.fcl_colsum2_fwd(None, None, None, None, None, None, 4, 4, 0, None) == -1
    assert lib.fcl_colsum2_fwd(128, None, None, None, 128, 128, 4, 4, 3, None) == -1 and b"mode needs y" in lib.fcl_last_error()
    assert lib.fcl_bn_bwd(128, 128, 128, 128, 128, 128, 128, 128, None, 4, 32, 128, None, None) == -1 and b"pairs" in lib.fcl_last_error()
    # fcl_bilstm_fwd: a row-map request is validated with the same rules as fcl_row_maps_build, before anything is launched
    a = _lib.RowMaps()
    ws = lib.fcl_bilstm_workspace_bytes(2, 5, 128)
    assert ws > 0
    rc = lib.fcl_bilstm_fwd(128, 128, 128, 128, 128, 128, 128, 128, 256, None, None, None, None, 2, 5, 32, 128, 0, 256, ws, None, C.byref(a), None)
    assert rc == -2 and lib.fcl_last_error()  # (empty request: sizes)
    assert lib.fcl_kaldi_ark_append(-1, 0, 0, None, None, None, 80, None) == -1


def test_native_training_engine_validates_its_configuration_and_bindings_without_a_gpu(lib):
    """fcl_te_* (round 4: the training step orchestrated in C++): creation checks the configuration, finalize checks that every parameter the
    configuration needs is bound with the right size -- all before any HIP call; the name tables are exported for the Python side."""
    from fcl_taco2_amd import _lib

    sites = [lib.fcl_te_site_name(i) for i in range(_lib.TE_MAX_SITES)]
    assert sites[0] == b"enc.convs/0" and b"zoneout/1/1" in sites and sites[-1] is None
    losses = [lib.fcl_te_loss_name(i) for i in range(_lib.TE_MAX_LOSSES)]
    assert losses[:5] == [b"after", b"before", b"dur", b"pitch", b"energy"] and b"pro4" in losses and losses[-1] is None
    h = C.c_void_p()
    assert lib.fcl_te_create(None, C.byref(h)) == -1
    cfg = _lib.TeConfig()
    cfg.role = 7
    assert lib.fcl_te_create(C.byref(cfg), C.byref(h)) == -1 and b"role" in lib.fcl_last_error()
    cfg = _lib.TeConfig(role=_lib.TE_TEACHER, idim=40, odim=80, embed_dim=48, econv_layers=3, econv_chans=48, econv_filts=5, eunits=48, dunits=64, prenet_units=32,
                        postnet_layers=5, postnet_chans=32, postnet_filts=5, dp_layers=2, dp_chans=32, dp_kernel=3, vp_layers=2, vp_chans=32, vp_kernel=3,
                        ve_kernel=9, dropout_rate=0.5, zoneout_rate=0.1, dp_dropout=0.1, vp_dropout=0.5, ve_dropout=0.5, use_masking=1, accum_grad=1)
    assert lib.fcl_te_create(C.byref(cfg), C.byref(h)) == -2 and b"multiples of 32" in lib.fcl_last_error()
    cfg.embed_dim = cfg.econv_chans = cfg.eunits = 64
    assert lib.fcl_te_create(C.byref(cfg), C.byref(h)) == 0 and h.value
    assert lib.fcl_te_finalize(h, 128) == -1 and b"enc.embed.weight is not bound" in lib.fcl_last_error()
    assert lib.fcl_te_bind_param(h, b"enc.embed.weight", 130, 256, 40 * 64) == -3  # misaligned
    assert lib.fcl_te_bind_param(h, b"enc.embed.weight", 256, 512, 40 * 63) == 0
    assert lib.fcl_te_finalize(h, 128) == -2 and b"expected 2560" in lib.fcl_last_error()
    assert lib.fcl_te_backward_stage(h, 1, None) == -1
    assert lib.fcl_te_last_launches(h) == 0 and lib.fcl_te_arena_bytes(h) == 0 and lib.fcl_te_side_stream(h) is None
    lib.fcl_te_destroy(h)


def test_the_library_contains_no_packed_fp32_instruction(tmp_path):
    """Round 6: on gfx950 a wave's packed-FP32 VALU results (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32; hipcc forms them from plain float4 code) come out stale
    in lanes 48 - 63 when another wave on the same SIMD issues MFMAs -- 1e-2 errors of the lane-split BiLSTM kernels beside any GEMM of another stream, wrong loss
    gradients in fresh engines' first KD update.  The library is therefore built with the feature taken away from the compiler (csrc/Makefile NOPK); this test
    disassembles the gfx950 code object of EVERY translation unit and fails on any packed-FP32 instruction, so a toolchain or flag change that brings them back fails
    HERE and not as silently wrong values on a busy GPU.  (The concurrency stress tests of tests/test_gpu_backward.py are the functional half.)"""
    import glob
    import re

    llvm = "/opt/rocm/lib/llvm/bin"
    objs = sorted(glob.glob(os.path.join(ROOT, "fcl-taco2_amd", "csrc", "*.o")))
    if not objs or not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("csrc/*.o or the ROCm LLVM tools are not here")
    pat = re.compile(r"\bv_pk_(?:[a-z0-9_]+_f32|mov_b32)\b")
    bad, kernels = {}, 0
    for obj in objs:
        base = os.path.basename(obj)[:-2]
        fat, co = str(tmp_path / (base + ".bin")), str(tmp_path / (base + ".co"))
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            continue  # a host-only translation unit
        r = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co,
                            "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co):
            continue
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
        kernels += dis.count("s_endpgm")
        hits = pat.findall(dis)
        if hits:
            bad[base] = len(hits)
    assert kernels > 100, kernels  # the disassembly really covered the library (hundreds of kernels)
    assert not bad, "packed-FP32 instructions in %s (build with csrc/Makefile's NOPK flags)" % bad
