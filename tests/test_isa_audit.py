"""The ISA audit behind the kernels with literal-register accumulators (fastspeech2_amd/_audit.py; round-4 advisor finding): the checks
themselves on hand-written listings, and the record `_lib.build()` leaves next to the library it ships."""
import os

import pytest

from fastspeech2_amd import _audit, _lib

ROW4 = "_ZN3fs214gemm_row4_bf16ILi3ELi3ELi5ELi0ELi2ELi0EEEvNS_8GemmArgsE"
W32 = "_ZN3fs28attn_w32ILi192EEEvNS_11AttnB16ArgsE"


def _listing(name, body):
    return "%s: ; @%s\n; %%bb.0:\n%s\n.Lfunc_end0:\n" % (name, name, "\n".join("\t" + l for l in body))


ASM = lambda *ins: [";;#ASMSTART"] + list(ins) + [";;#ASMEND"]          # noqa: E731
CLEAN_ROW4 = (["ds_read_b128 v[2:5], v164", "ds_read_b128 v[6:9], v165", "s_waitcnt lgkmcnt(0)"]
              + ASM("v_mfma_f32_16x16x32_bf16 a[0:3], v[2:5], v[6:9], a[0:3]") + ASM("v_mfma_f32_16x16x32_bf16 a[236:239], v[2:5], v[6:9], a[236:239]")
              + ASM("s_nop 7", "s_nop 7", "s_nop 3") + ASM("v_accvgpr_read_b32 v10, a0") + ["global_store_dword v[20:21], v10, off"])


def test_a_clean_row4_listing_passes():
    rec = _audit.audit_text(_listing(ROW4, CLEAN_ROW4))[ROW4]
    assert rec["violations"] == [] and rec["asm_mfma"] == 2 and rec["accumulators"] == 240 and (rec["nb"], rec["mt"]) == (3, 5)


@pytest.mark.parametrize("extra, tag", [
    (["v_accvgpr_write_b32 a8, v3"], "1:"),                                     # the compiler parks a value in an accumulator register
    (["v_accvgpr_read_b32 v3, a250"], "1:"),                                    # ... or uses any AGPR at all
    (["scratch_store_dwordx4 off, v[2:5], off offset:16"], "2:"),               # a spill
])
def test_row4_violations_are_found(extra, tag):
    rec = _audit.audit_text(_listing(ROW4, CLEAN_ROW4[:3] + extra + CLEAN_ROW4[3:]))[ROW4]
    assert any(v.startswith(tag) for v in rec["violations"]), rec["violations"]


def test_row4_mfma_beyond_the_accumulators_and_short_drain_are_found():
    body = CLEAN_ROW4[:3] + ASM("v_mfma_f32_16x16x32_bf16 a[240:243], v[2:5], v[6:9], a[240:243]") + ASM("s_nop 3") + ASM("v_accvgpr_read_b32 v10, a0")
    v = _audit.audit_text(_listing(ROW4, body))[ROW4]["violations"]
    assert any("beyond the 240 accumulators" in x for x in v) and any(x.startswith("5:") for x in v), v


def test_w32_valu_write_right_ahead_of_an_asm_mfma_is_found():
    body = ["v_mov_b32_e32 v7, v1"] + ASM("v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]")
    v = _audit.audit_text(_listing(W32, body))[W32]["violations"]
    assert any(x.startswith("3:") for x in v), v
    body = ["v_mov_b32_e32 v7, v1"] + ASM("s_nop 1", "v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]")
    assert _audit.audit_text(_listing(W32, body))[W32]["violations"] == []
    body = ["v_accvgpr_write_b32 a3, v1"] + ASM("v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]")       # compiler copy into O^T
    assert any(x.startswith("1:") for x in _audit.audit_text(_listing(W32, body))[W32]["violations"])


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libfs2_hip.so not built")
def test_the_shipped_library_carries_a_clean_audit_record_of_its_own_binary():
    rec = _lib.audit_record()          # None unless the record's hash is the hash of the .so on disk
    assert rec is not None, "no audit record for this binary: build it with __graft_entry__.build()"
    assert rec["clean"] and rec["violations"] == 0, rec
    kinds = [v["kind"] for v in rec["kernels"].values()]
    assert {k: kinds.count(k) for k in _audit.EXPECTED_KERNELS} == _audit.EXPECTED_KERNELS, kinds      # every instantiation the launchers use was found AND audited
    assert all(v["asm_mfma"] > 0 for v in rec["kernels"].values())


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libfs2_hip.so not built")
def test_the_library_itself_gates_the_literal_register_kernels_on_its_audit_record(tmp_path):
    """The gate lives in libfs2_hip.so (round-5 advisor finding: it used to live in the ctypes binding only, so libfs2_torch.so and C programs got
    the kernels unaudited): the library hashes ITSELF when it is first used and looks for the record next to it.  The shipped binary reports a
    clean audit and all three kernels allowed; a byte-identical copy WITHOUT the record, and one with a record of another hash, report 0 and
    start with the kernels off.  (No GPU needed: fs2_get_option touches no device.)"""
    import shutil
    import subprocess
    import sys
    probe = ("import ctypes as C, sys; L = C.CDLL(sys.argv[1]); v = C.c_int32(-7); out = []\n"
             "for n in (b'FS2_AUDIT_CLEAN', b'attn_w32_active', b'row4_active', b'qkv4_active'):\n"
             "    assert L.fs2_get_option(n, C.byref(v)) == 0; out.append(v.value)\n"
             "assert L.fs2_get_option(b'no_such_option', C.byref(v)) != 0\n"
             "print(out)")
    run = lambda so: subprocess.run([sys.executable, "-c", probe, so], capture_output=True, text=True, timeout=120)
    r = run(_lib.LIB_PATH)
    assert r.returncode == 0 and r.stdout.strip() == "[1, 1, 1, 1]", (r.stdout, r.stderr[-500:])
    d1 = tmp_path / "bare"
    d1.mkdir()
    shutil.copy(_lib.LIB_PATH, d1 / "libfs2_hip.so")
    r = run(str(d1 / "libfs2_hip.so"))
    assert r.returncode == 0 and r.stdout.strip() == "[0, 0, 0, 0]" and "no clean ISA-audit record" in r.stderr, (r.stdout, r.stderr[-500:])
    d2 = tmp_path / "stale"
    d2.mkdir()
    shutil.copy(_lib.LIB_PATH, d2 / "libfs2_hip.so")
    rec = open(_lib.AUDIT_PATH).read().replace(_lib.audit_record()["so_sha16"], "0123456789abcdef")
    (d2 / "libfs2_hip.audit.json").write_text(rec)
    r = run(str(d2 / "libfs2_hip.so"))
    assert r.returncode == 0 and r.stdout.strip() == "[0, 0, 0, 0]", (r.stdout, r.stderr[-500:])


def test_a_valu_written_dma_base_too_close_is_found():
    body = ["v_readfirstlane_b32 s12, v3", "v_readfirstlane_b32 s13, v4"] + ASM("s_mov_b32 m0, s20", "s_nop 0", "global_load_lds_dwordx4 v9, s[12:13]")
    v = _audit.audit_text(_listing(ROW4, body))[ROW4]["violations"]
    assert any(x.startswith("6:") for x in v), v
    body = ["v_readfirstlane_b32 s12, v3", "v_readfirstlane_b32 s13, v4"] + ASM("s_nop 4") + ASM("s_mov_b32 m0, s20", "s_nop 0", "global_load_lds_dwordx4 v9, s[12:13]")
    assert _audit.audit_text(_listing(ROW4, body))[ROW4]["violations"] == []
