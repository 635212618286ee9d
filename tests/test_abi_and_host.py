"""CPU: the C-ABI library loads and exports every symbol include/fs2.h declares; host-side module logic
(state-dict layout, error behaviour without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from fastspeech2_amd import _lib
    return _lib.build()      # hipcc cross-compiles gfx950 without a GPU (no-op when up to date)


def test_header_symbols_exported(libpath):
    from fastspeech2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fs2.h")).read()
    declared = sorted(set(re.findall(r"\b(fs2_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(libpath)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared


def _c_probe(tmp_path, structs):
    """sizeof / offsetof of every field of `structs` = [(c_name, [field, ...])] as gcc sees include/fs2.h."""
    import subprocess
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "fs2.h"', "int main(void) {",
           '  printf("abi %d\\n", FS2_ABI_VERSION);']
    for cname, fields in structs:
        src.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in fields:
            src.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    src += ["  return 0;", "}"]
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "probe")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (line.split() for line in out.splitlines())}


def _header_structs():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import gen_binding_doc
    finally:
        sys.path.pop(0)
    return gen_binding_doc, gen_binding_doc.parse_structs()


def _assert_mirror(probe, cname, fields, mirror):
    assert ctypes.sizeof(mirror) == probe[cname], (cname, ctypes.sizeof(mirror), probe[cname])
    assert [f for f, _ in fields] == [f[0] for f in mirror._fields_], cname          # same fields, same order
    for f, _ in fields:
        assert getattr(mirror, f).offset == probe["%s.%s" % (cname, f)], (cname, f)


def test_ctypes_mirrors_match_compiled_header(tmp_path):
    """Every struct of include/fs2.h, compiled by gcc: sizeof and the offset of every field equal those of the ctypes mirrors in
    fastspeech2_amd/_lib.py (a reordered, retyped, added or dropped field fails here, not at run time on the GPU)."""
    from fastspeech2_amd import _lib
    gen, structs = _header_structs()
    probe = _c_probe(tmp_path, [(c, [f for f, _ in fl]) for c, fl in structs])
    assert probe["abi"] == _lib.ABI_VERSION == gen.abi_version()
    mirrors = {"fs2_config": _lib.Config, "fs2_tensor_desc": _lib.TensorDesc, "fs2_batch": _lib.Batch, "fs2_encode_io": _lib.EncodeIO,
               "fs2_decode_io": _lib.DecodeIO, "fs2_op_gemm_args": _lib.OpGemmArgs}
    assert sorted(mirrors) == sorted(c for c, _ in structs)
    for cname, fields in structs:
        _assert_mirror(probe, cname, fields, mirrors[cname])
    for m in (_lib.Config, _lib.EncodeIO, _lib.DecodeIO, _lib.OpGemmArgs):      # struct_size is filled in by the mirror itself
        assert m().struct_size == ctypes.sizeof(m)


def _integration_blocks():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    gen, _ = _header_structs()
    i, j = doc.index(gen.BEGIN), doc.index(gen.END) + len(gen.END)
    generated = doc[i:j]
    u0, u1 = doc.index("<!-- BEGIN usage -->"), doc.index("<!-- END usage -->")
    code = lambda block: re.search(r"```python\n(.*?)```", block, re.S).group(1)
    return gen, generated, code(generated), code(doc[u0:u1])


def test_integration_md_binding_is_generated_from_the_header_and_executes(tmp_path, libpath):
    """the struct mirrors are byte-for-byte what tools/gen_binding_doc.py derives from include/fs2.h, they
    match the compiled header field by field, and the documented usage runs against the built library -- here (no GPU) up to
    fs2_create's "no HIP device" error; tests/test_gpu_parity.py runs it to the end on the GPU."""
    gen, generated, mirrors_src, usage_src = _integration_blocks()
    assert generated == gen.doc_block(), "INTEGRATION.md is stale: run `python tools/gen_binding_doc.py --write`"
    ns = {}
    exec(compile(mirrors_src, "INTEGRATION.md[mirrors]", "exec"), ns)
    exec(compile(usage_src, "INTEGRATION.md[usage]", "exec"), ns)
    _, structs = _header_structs()
    probe = _c_probe(tmp_path, [(c, [f for f, _ in fl]) for c, fl in structs])
    for cname, fields in structs:
        _assert_mirror(probe, cname, fields, ns[gen.PY_NAMES[cname]])
    if torch.cuda.is_available():
        pytest.skip("GPU present: the full run is in tests/test_gpu_parity.py")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ns["fs2_synthesize"](libpath, {}, torch.ones(5, dtype=torch.int64))


def test_struct_size_mismatch_is_rejected(libpath):
    """A binding written against an older header (round 2's 25-field fs2_config, no struct_size) is refused with FS2_ERR_ARG
    before any field is interpreted."""
    from fastspeech2_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()

    class OldCfg(ctypes.Structure):
        _fields_ = [("f%d" % i, ctypes.c_int32) for i in range(25)]
    old = OldCfg(68, 80, 256, 2, 4, 1024, 384, 4, 1024, 9, 2, 256, 3, 2, 256, 3, 256, 5, 256, 5, 1, 1, 1, 0, 1)
    fn = ctypes.CDLL(libpath).fs2_create          # untyped handle: argtypes of the typed one would refuse OldCfg in Python already
    assert fn(ctypes.byref(old), ctypes.byref(h)) == -1 and b"struct_size" in L.fs2_last_error(None)
    short = _lib.Config(68, 80, 256, 2, 4, 1024, 384, 4, 1024, 9, 2, 256, 3, 2, 256, 3, 256, 5, 256, 5, 1, 1, 1, 0, 1)
    short.struct_size -= 16
    assert L.fs2_create(ctypes.byref(short), ctypes.byref(h)) == -1 and b"struct_size" in L.fs2_last_error(None)
    assert L.fs2_abi_version() == _lib.ABI_VERSION
    # the operator entry point checks its argument struct before touching any pointer
    a = _lib.OpGemmArgs()
    a.struct_size = 8
    assert L.fs2_op_conv_gemm(None, ctypes.byref(a)) == -1 and b"struct_size" in L.fs2_last_error(None)


def test_create_without_gpu_fails_loudly(libpath):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fastspeech2_amd import _lib
    L = _lib.lib()
    cfg = _lib.Config(68, 80, 256, 2, 4, 1024, 384, 4, 1024, 9, 2, 256, 3, 2, 256, 3, 256, 5, 256, 5, 1, 1, 1, 0, 1)
    h = ctypes.c_void_p()
    rc = L.fs2_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in L.fs2_last_error(None)


def _model():
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    hp = default_hparams()
    return FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp), hp


def test_state_dict_layout_matches_reference():
    """Key names / shapes of SURVEY.md Appendix C (captured from the reference's state_dict())."""
    model, _ = _model()
    sd = model.state_dict()
    assert len(sd) == 225
    assert sum(p.numel() for p in model.parameters()) == 34015605
    expect = {
        "encoder.embed.0.weight": (68, 256), "encoder.embed.1.alpha": (), "encoder.embed.1.pe": (1, 5000, 256),
        "encoder.after_norm.weight": (256,), "encoder.encoders_.3.self_attn.linear_q.weight": (256, 256),
        "encoder.encoders_.0.feed_forward.w_1.weight": (1024, 256, 9), "encoder.encoders_.0.feed_forward.w_2.weight": (256, 1024, 1),
        "encoder.encoders_.0.concat_linear.weight": (256, 512), "duration_predictor.conv.1.0.weight": (256, 256, 3),
        "duration_predictor.conv.0.2.layer_norm.bias": (256,), "duration_predictor.linear.weight": (1, 256),
        "energy_predictor.energy_bins": (255,), "energy_predictor.predictor.conv.0.0.weight": (256, 256, 3),
        "pitch_predictor.pitch_bins": (255,), "pitch_embed.weight": (256, 256), "energy_embed.bias": (256,),
        "decoder.embed.0.weight": (384, 256), "decoder.embed.1.bias": (384,), "decoder.embed.4.alpha": (),
        "decoder.embed.4.pe": (1, 5000, 384), "decoder.encoders_.2.feed_forward.w_1.weight": (1024, 384, 9),
        "decoder.encoders_.0.concat_linear.weight": (384, 768), "decoder.after_norm.bias": (384,),
        "postnet.postnet.0.0.weight": (256, 80, 5), "postnet.postnet.4.0.weight": (80, 256, 5),
        "postnet.postnet.2.1.running_var": (256,), "postnet.postnet.4.1.num_batches_tracked": (), "feat_out.weight": (80, 384),
    }
    for k, shp in expect.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    bins = sd["pitch_predictor.pitch_bins"]
    assert abs(float(bins[0]) - 71.0) < 1e-4 and abs(float(bins[-1]) - 676.2261) < 1e-2
    assert float(sd["encoder.embed.0.weight"][0].abs().max()) == 0.0      # padding_idx row


def test_grown_positional_table_changes_the_weight_fingerprint():
    """extend_pe (reference core/embedding.py:50-56): a longer table is a NEW tensor; the cached fingerprint must notice, or the
    library keeps the 5000-row table and an utterance beyond it fails (round-2 advisor finding)."""
    model, _ = _model()
    fp0 = model._weights_fingerprint()
    assert model._weights_fingerprint() == fp0
    assert model.decoder.embed[-1].ensure(6000) and not model.decoder.embed[-1].ensure(5500)
    fp1 = model._weights_fingerprint()
    assert fp1 != fp0 and model.state_dict()["decoder.embed.4.pe"].shape[1] == 6000
    assert model.encoder.embed[-1].ensure(5001)
    assert model._weights_fingerprint() != fp1
    with pytest.raises(ValueError, match="alpha"):
        model.inference_batch(torch.ones(1, 5, dtype=torch.int64), [5], alpha=0.0)


def test_cpu_inputs_raise_not_fallback():
    model, _ = _model()
    model.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.inference(torch.ones(5, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model._forward(torch.ones(1, 5, dtype=torch.int64), torch.tensor([5]), is_inference=True)


def test_unsupported_configs_raise():
    from fastspeech2_amd import FeedForwardTransformer
    _, hp = _model()
    hp.model.reduction_factor = 2                     # implemented (fixture G8): feat_out is Linear(ddim, 2 * odim)
    m2 = FeedForwardTransformer(68, 80, hp)
    assert m2._cfg["reduction_factor"] == 2 and m2.feat_out.out_features == 160
    hp.model.reduction_factor = 9
    with pytest.raises(NotImplementedError):
        FeedForwardTransformer(68, 80, hp)
    hp.model.reduction_factor = 1
    hp.model.encoder_normalize_before = True          # pre-LN / concat_after blocks are implemented (fixture G7)
    hp.model.decoder_concat_after = True
    assert FeedForwardTransformer(68, 80, hp)._cfg["enc_normalize_before"] == 1
    hp.model.positionwise_layer_type = "conv2d"
    with pytest.raises(NotImplementedError):
        FeedForwardTransformer(68, 80, hp)


def test_portable_weights_are_deterministic():
    from fastspeech2_amd.synthetic import portable_state_dict, make_batch
    model, _ = _model()
    a = portable_state_dict(model.state_dict(), 0)
    b = portable_state_dict(model.state_dict(), 0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert abs(float(a["feat_out.weight"].abs().max()) - 1 / 384 ** 0.5) < 1e-3
    c3 = make_batch("c3")
    assert c3["xs"].shape[0] == 64 and int(c3["olens"].sum()) > 30000
    assert torch.equal(c3["ds"].sum(1), c3["olens"])


def test_script_twin_is_scriptable_without_a_gpu(tmp_path):
    """export_torchscript.py's flow (reference export_torchscript.py:46-49): torch.jit.script(model) + save; running the
    archive needs the GPU (tests/test_gpu_parity.py)."""
    from fastspeech2_amd import default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.fastspeech2_script import FeedForwardTransformer as Twin
    twin = Twin(N_PHONEME_SYMBOLS, 80, default_hparams()).eval()
    scripted = torch.jit.script(twin)
    assert "fs2::twin_inference" in str(scripted.graph)
    path = str(tmp_path / "fs2_twin.pt")
    scripted.save(path)
    again = torch.jit.load(path)
    assert again.flat_weights.numel() == twin.flat_weights.numel() > 26e6
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        again(torch.ones(5, dtype=torch.int64))


def test_load_checkpoint_forms(tmp_path):
    """inference.py:161-166 / train_fastspeech.py:235-244: {"model": ...} checkpoints, bare --old_model state dicts
    (strict=False), DataParallel prefixes, embedded hp_str."""
    from fastspeech2_amd import load_checkpoint, hparams_from_str
    from fastspeech2_amd.synthetic import portable_state_dict
    model, hp = _model()
    sd = portable_state_dict(model.state_dict(), 7)
    path = str(tmp_path / "chk.pt")
    torch.save({"model": sd, "optim": {}, "step": 58000, "hp_str": open(os.path.join(ROOT, "configs", "default.yaml")).read(), "githash": "abc"}, path)
    extras = load_checkpoint(model, path)
    assert extras["step"] == 58000 and torch.equal(model.state_dict()["feat_out.weight"], sd["feat_out.weight"])
    assert hparams_from_str(extras["hp_str"]).model.adim == 256
    bare = {("module." + k): v for k, v in sd.items() if "concat_linear" not in k}          # old checkpoint without unused keys
    with pytest.raises(RuntimeError):
        load_checkpoint(model, dict(bare))
    extras = load_checkpoint(model, dict(bare), old_model=True)
    assert any("concat_linear" in k for k in extras["missing_keys"]) and not extras["unexpected_keys"]


def test_module_pickles_and_deep_copies_without_its_runtime_state():
    """`copy.deepcopy(model)` / `torch.save(model)` carry parameters and knobs, never the per-process runtime state (ctypes handle, HIP streams,
    pinned staging slots, calls in flight: round-5 advisor finding -- a model that had run once in overlap_encoder mode could not be copied)."""
    import copy
    import ctypes as C
    import io
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    m = FeedForwardTransformer(N_PHONEME_SYMBOLS, 80, default_hparams()).eval()
    m.precision, m.overlap_encoder = "mix_mx", True
    m.__dict__["_handle"] = None
    m._enc_streams = {"key": C.c_void_p(5)}          # stand-ins for what a used model holds: none of them can be pickled
    m._pin_ring = [[1, 2, C.c_void_p(3)]]
    m._pending = [C.c_void_p(7)]
    m2 = copy.deepcopy(m)
    assert m2._enc_streams == {} and m2._pin_ring == [] and m2._pending == [] and m2._handle is None and m2._fingerprint is None
    assert m2.precision == "mix_mx" and m2.overlap_encoder is True
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert list(m3.state_dict()) == list(m.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    m._enc_streams, m._pin_ring, m._pending = {}, [], []
