#!/usr/bin/env python3
"""Command-line form of the ISA audit (fastspeech2_amd/_audit.py, which `_lib.build()` runs on every library it ships): every attn_w32<DK> and
gemm_row4_bf16 kernel found in a gfx950 assembly listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastspeech2_amd/csrc tools/probes/attn_w32_probe.hip -save-temps -o /tmp/x
  python3 tools/probes/audit_w32.py attn_w32_probe-hip-amdgcn-amd-amdhsa-gfx950.s
Exit code 1 on a violation.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from fastspeech2_amd import _audit  # noqa: E402


def main():
    res = _audit.audit_file(sys.argv[1])
    bad = 0
    for name, rec in res.items():
        for v in rec["violations"]:
            print("VIOLATION", v)
        bad += len(rec["violations"])
        print(name, {k: v for k, v in rec.items() if k != "violations"}, "%d violation(s)" % len(rec["violations"]))
    if not res:
        print("no attn_w32 / gemm_row4_bf16 kernel in", sys.argv[1])
    sys.exit(1 if bad or not res else 0)


if __name__ == "__main__":
    main()
