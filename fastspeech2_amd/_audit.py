"""ISA audit of the kernels whose accumulators are LITERAL registers (csrc/attn_w32.h: O^T in a[0 : 16 NT); csrc/gemm_row4.h: the GEMM
accumulators in a[0 : 4 MT NT)).  hipcc does not know that these registers are live between the asm statements that name them -- the clobber
lists only say "destroyed" -- and pads no hazards around inline asm (cdna guide section 5.7), so a different compiler version could place a
temporary or an AGPR spill there, or drop a pad, and corrupt results silently.  `_lib.build()` therefore compiles with -save-temps, runs this
audit on the device assembly of the very library it ships and records the outcome next to the .so (libfs2_hip.audit.json, tied to the
binary's hash); the LIBRARY itself looks for that record of its own hash when it is first used (fs2_runtime.hip: audit_clean) and starts with
both kernels off (FS2_ATTN_W32 = 0, FS2_ROW4 = 0: the compiler-scheduled kernels run instead) when the record is missing, stale or not clean --
for every consumer of the C ABI, not only the ctypes binding.  Round-4 / round-5 advisor findings.

Checks per attn_w32<DK> kernel (tools/probes/audit_w32.py, rounds 4-5):
  1. no compiler-generated v_accvgpr_* touches a0 .. a(16 NT - 1) outside ASMSTART / ASMEND;
  3. every asm MFMA whose VGPR operand was written by a VALU instruction has >= 2 issue states between that write and itself;
  4. every read of an asm MFMA's VGPR result by a non-MFMA instruction sits >= 12 issue states or >= 4 MFMAs behind it.
Checks per gemm_row4_bf16<.., NB, MT, ..> kernel:
  1. no compiler-generated v_accvgpr_* and no compiler use of any AGPR at all (the accumulators are a[0 : 16 MT NB): every one of the
     kernel's AGPRs is ours);
  2. no scratch access;
  3. as 3. above (the fragments come from ds_read, so none is expected);
  5. an asm v_accvgpr_read of an accumulator sits >= 18 issue states behind the last asm MFMA (the drain in front of the epilogue).
Both: 6. the SGPR base of an asm LDS-DMA instruction was not written by a VALU instruction (v_readfirstlane) within the 5 preceding issue states.
"""
import re

# What a clean record must hold (round-5 advisor finding: "clean" used to require one attn_w32 record only, so a symbol regex that stopped
# matching would have let the row kernels through unaudited): the library instantiates attn_w32 for two head dims and gemm_row4_bf16 for
# {MT 4, 5} x {EPI 0 in split-bf16 with the residual from split-bf16 / from mx planes, EPI 0 in the mx arithmetic, EPI 1, EPI 2, EPI 3 (the QKV passes), EPI 4 (mx4 planes)}.  fs2_runtime.hip: launch_attn_w32_t / launch_row4_t / launch_qkv4_t.
EXPECTED_KERNELS = {"attn_w32": 2, "gemm_row4_bf16": 14}

_REG = re.compile(r'([va])\[(\d+):(\d+)\]|([va])(\d+)$')


def _regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r'([va])(\d+)$', tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def _instructions(lines):
    """[(text, inside an asm statement)] of a function body; labels kept as 'LABEL ...'."""
    ins, in_asm = [], False
    for l in lines:
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            if t.startswith('.LBB') or t.startswith('; %bb'):
                ins.append(('LABEL ' + t, False))
            continue
        ins.append((t, in_asm))
    return ins


def _states(t):
    if t.startswith('s_nop'):
        return int(t.split()[1]) + 1
    if t.startswith('LABEL'):
        return 0
    return 1


def _valu_write_ahead_of_mfma(ins):
    """check 3: [(valu instruction, mfma, states between)]"""
    out = []
    for k, (t, a) in enumerate(ins):
        if not (a and t.startswith('v_mfma')):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        srcs = set()
        for o in ops[1:]:
            f, r = _regs(o)
            if f == 'v':
                srcs |= r
        st, kk = 0, k - 1
        while kk >= 0 and st < 2:
            tt, _ = ins[kk]
            if tt.startswith('LABEL'):
                break
            if tt.startswith('v_') and not tt.startswith('v_mfma'):
                dst = tt.split(None, 1)[1].split(',')[0]
                f, r = _regs(dst)
                if f == 'v' and (r & srcs):
                    out.append((tt, t, st))
            st += _states(tt)
            kk -= 1
    return out


def _valu_sgpr_ahead_of_dma(ins):
    """check 6: an LDS-DMA instruction inside an asm statement takes its base from an SGPR pair; a VALU write of that pair (v_readfirstlane)
    needs 5 issue states before a VMEM instruction reads it, and hipcc pads nothing around asm.  -> [(writer, dma, states between)]"""
    out = []
    for k, (t, a) in enumerate(ins):
        if not (a and t.startswith('global_load_lds')):
            continue
        m = re.search(r's\[(\d+):(\d+)\]', t)
        if not m:
            continue
        base = {int(m.group(1)), int(m.group(2))}
        st, kk = 0, k - 1
        while kk >= 0 and st < 5:
            tt, _ = ins[kk]
            if tt.startswith('LABEL'):
                break
            if tt.startswith('v_readfirstlane'):
                d = re.match(r'v_readfirstlane_b32\s+s(\d+)', tt)
                if d and int(d.group(1)) in base:
                    out.append((tt, t, st))
            st += _states(tt)
            kk -= 1
    return out


def audit_w32(lines, dk):
    """-> dict(kernel facts, violations=[...]) for one attn_w32<DK> body."""
    nt = dk // 32
    ins = _instructions(lines)
    bad = []
    for t, a in ins:
        if t.startswith('v_accvgpr') and not a:
            for o in t.split(None, 1)[1].split(','):
                f, r = _regs(o)
                if f == 'a' and any(x < 16 * nt for x in r):
                    bad.append('1: compiler touches O^T registers: ' + t)
    for tt, t, st in _valu_write_ahead_of_mfma(ins):
        bad.append('3: VALU write %d state(s) ahead of an MFMA operand: %s | %s' % (st, tt, t))
    for tt, t, st in _valu_sgpr_ahead_of_dma(ins):
        bad.append('6: VALU-written SGPR base %d state(s) ahead of an LDS-DMA instruction: %s | %s' % (st, tt, t))
    nmf = 0
    for k, (t, a) in enumerate(ins):
        if not (a and t.startswith('v_mfma')):
            continue
        nmf += 1
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        f, r = _regs(ops[0])
        if f != 'v':
            continue
        st = m = 0
        kk = k + 1
        while kk < len(ins) and st < 12 and m < 4:
            tt, _ = ins[kk]
            if tt.startswith('LABEL') or tt.startswith('s_cbranch') or tt.startswith('s_branch'):
                break
            if tt.startswith('v_mfma'):
                m += 1
            elif tt.startswith(('v_', 'ds_', 'global_', 'scratch_')):
                body = tt.split(None, 1)[1] if ' ' in tt else ''
                toks = re.findall(r'[va]\[\d+:\d+\]|[va]\d+', body)
                for o in (toks[1:] if tt.startswith('v_') else toks):
                    ff, rr = _regs(o)
                    if ff == 'v' and (rr & r):
                        bad.append('4: MFMA result read %d state(s) / %d MFMA(s) behind it: %s | %s' % (st, m, t, tt))
                        break
            st += _states(tt)
            kk += 1
    return dict(kind='attn_w32', dk=dk, instructions=len(ins), asm_mfma=nmf, scratch=sum(1 for t, _ in ins if t.startswith('scratch_')), violations=bad)


def audit_row4(lines, nb, mt):
    """-> dict(kernel facts, violations=[...]) for one gemm_row4_bf16 body (accumulators a[0 : 16 MT NB))."""
    nacc = 16 * mt * nb
    ins = _instructions(lines)
    bad = []
    for t, a in ins:
        if a:
            continue
        if t.startswith('v_accvgpr'):
            bad.append('1: compiler-generated ' + t)
        elif t.startswith('scratch_'):
            bad.append('2: scratch access ' + t)
        elif not t.startswith('LABEL') and re.search(r'(?<![\w.])a(\[\d+:\d+\]|\d+)\b', t.split(None, 1)[1] if ' ' in t else ''):
            bad.append('1: compiler instruction names an AGPR: ' + t)
    for tt, t, st in _valu_write_ahead_of_mfma(ins):
        bad.append('3: VALU write %d state(s) ahead of an MFMA operand: %s | %s' % (st, tt, t))
    for tt, t, st in _valu_sgpr_ahead_of_dma(ins):
        bad.append('6: VALU-written SGPR base %d state(s) ahead of an LDS-DMA instruction: %s | %s' % (st, tt, t))
    nmf, top = 0, -1
    last_mfma = None
    for k, (t, a) in enumerate(ins):
        if a and t.startswith('v_mfma'):
            nmf += 1
            last_mfma = k
            f, r = _regs(t.split(None, 1)[1].split(',')[0])
            if f == 'a':
                top = max(top, max(r))
            else:
                bad.append('1: an MFMA accumulates outside the AGPR file: ' + t)
        elif a and t.startswith('v_accvgpr_read') and last_mfma is not None:
            st = sum(_states(x) for x, _ in ins[last_mfma + 1:k])
            if st < 18:
                bad.append('5: accumulator read %d state(s) behind the last MFMA: %s' % (st, t))
            last_mfma = None          # only the first read behind the loop matters
    if top >= nacc:
        bad.append('1: an MFMA writes a%d, beyond the %d accumulators' % (top, nacc))
    return dict(kind='gemm_row4_bf16', nb=nb, mt=mt, instructions=len(ins), asm_mfma=nmf, accumulators=nacc, violations=bad)


_W32 = re.compile(r'^(_ZN3fs28attn_w32ILi(\d+)EEEvNS_11AttnB16ArgsE):')
_ROW4 = re.compile(r'^(_ZN3fs214gemm_row4_bf16I((?:Li\d+E)+)EEvNS_8GemmArgsE):')      # template arguments: NSPLIT, NB, MT, EPI, SCHED, ARITH


def audit_text(text):
    """Audit every attn_w32 / gemm_row4_bf16 kernel found in a device assembly listing.  -> {kernel name: record}."""
    lines = text.split('\n')
    out = {}
    i = 0
    n = len(lines)
    while i < n:
        l = lines[i]
        m1 = _W32.match(l) if l.startswith('_ZN3fs28attn_w32') else None
        m2 = _ROW4.match(l) if l.startswith('_ZN3fs214gemm_row4') else None
        if not (m1 or m2):
            i += 1
            continue
        j = i + 1
        while j < n and not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = lines[i + 1:j]
        if m1:
            out[m1.group(1)] = audit_w32(body, int(m1.group(2)))
        else:
            targs = [int(x) for x in re.findall(r'Li(\d+)E', m2.group(2))]
            out[m2.group(1)] = audit_row4(body, targs[1], targs[2])
        i = j
    return out


def audit_file(path):
    with open(path) as f:
        return audit_text(f.read())
