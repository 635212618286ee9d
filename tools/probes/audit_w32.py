#!/usr/bin/env python3
"""Audit of the attn_w32 ISA (hipcc pads nothing around inline asm, and the O^T accumulators are literal registers):

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastspeech2_amd/csrc tools/probes/attn_w32_probe.hip -save-temps -o /tmp/x
  python3 tools/probes/audit_w32.py attn_w32_probe-hip-amdgcn-amd-amdhsa-gfx950.s [dk=192]

Checks, per attn_w32<DK> kernel:
  1. no compiler-generated v_accvgpr_* touches a0 .. a(16 NT - 1) outside ASMSTART/ASMEND (O^T lives there unseen by the compiler);
  2. no scratch access and no compiler v_accvgpr_* at all inside the tile loop's MFMA blocks;
  3. every asm MFMA whose VGPR operand (A, B or C) was written by a VALU instruction has >= 2 issue states between that write and itself
     (cdna guide 5.7: `s_nop 1`), counting instructions and s_nop states;
  4. every read of an asm MFMA's VGPR result by a non-MFMA instruction sits >= 12 issue states or >= 4 MFMAs behind it.
Exit code 1 on a violation.
"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r'([va])(\d+)$', tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def main():
    path = sys.argv[1]
    dk = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    nt = dk // 32
    s = open(path).read()
    name = '_ZN3fs28attn_w32ILi%dEEEvNS_11AttnB16ArgsE' % dk
    i = s.index(name + ':')
    j = s.index('.Lfunc_end', i)
    lines = s[i:j].split('\n')
    ins = []          # (text, in_asm)
    in_asm = False
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
    bad = 0
    # 1 / 2
    for k, (t, a) in enumerate(ins):
        if t.startswith('v_accvgpr') and not a:
            ops = t.split(None, 1)[1].split(',')
            for o in ops:
                f, r = regs(o)
                if f == 'a' and any(x < 16 * nt for x in r):
                    print('VIOLATION 1: compiler touches O^T registers:', t)
                    bad += 1
    # states helper
    def states(t):
        if t.startswith('s_nop'):
            return int(t.split()[1]) + 1
        if t.startswith('LABEL'):
            return 0
        return 1
    nmf = 0
    for k, (t, a) in enumerate(ins):
        if not (a and t.startswith('v_mfma')):
            continue
        nmf += 1
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        srcs = set()
        for o in ops[1:]:
            f, r = regs(o)
            if f == 'v':
                srcs |= r
        # 3: look back for VALU writers of srcs
        st = 0
        kk = k - 1
        while kk >= 0 and st < 2:
            tt, aa = ins[kk]
            if tt.startswith('LABEL'):
                break
            if tt.startswith('v_') and not tt.startswith('v_mfma'):
                dst = tt.split(None, 1)[1].split(',')[0]
                f, r = regs(dst)
                if f == 'v' and (r & srcs):
                    print('VIOLATION 3: VALU write %d state(s) ahead of an MFMA operand:\n    %s\n    %s' % (st, tt, t))
                    bad += 1
            st += states(tt)
            kk -= 1
        # 4: readers of a VGPR result
        f, r = regs(ops[0])
        if f == 'v':
            st = 0
            m = 0
            kk = k + 1
            while kk < len(ins) and st < 12 and m < 4:
                tt, aa = ins[kk]
                if tt.startswith('LABEL') or tt.startswith('s_cbranch') or tt.startswith('s_branch'):
                    break
                if tt.startswith('v_mfma'):
                    m += 1
                elif tt.startswith('v_') or tt.startswith('ds_') or tt.startswith('global_') or tt.startswith('scratch_'):
                    body = tt.split(None, 1)[1] if ' ' in tt else ''
                    toks = re.findall(r'[va]\[\d+:\d+\]|[va]\d+', body)
                    for o in toks[1:] if tt.startswith('v_') else toks:
                        ff, rr = regs(o)
                        if ff == 'v' and (rr & r):
                            print('VIOLATION 4: MFMA result read %d state(s) / %d MFMA(s) behind it:\n    %s\n    %s' % (st, m, t, tt))
                            bad += 1
                            break
                st += states(tt)
                kk += 1
    sc = sum(1 for t, a in ins if t.startswith('scratch_'))
    print('attn_w32<%d>: %d instructions, %d asm MFMAs, %d scratch accesses (slow-path call frame only is fine), %d violation(s)' % (dk, len(ins), nmf, sc, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
