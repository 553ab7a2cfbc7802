#!/usr/bin/env python3
"""Build-time audit of csrc/attn160.hip's inline-asm register loads (cdna_hip_programming.md 5.7, item 1): the parked self output comes
back by `buffer_load_dwordx4 ... sc1` statements hipcc does not track, and is valid only behind the `s_waitcnt vmcnt(6)` statement that
names its registers.  Between the two hipcc must not read, copy, spill or overwrite those registers (it once placed v_mov copies in
FRONT of the wait).  Compiles the file with -save-temps for both 16-bit types and checks every instruction in between; also requires
zero spills and no scratch.  Exit code 1 on a violation.   python tools/audit_attn160.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diffsim_amd", "csrc", "attn160.hip")


def regs_of(tok):
    out = set()
    for lo, hi in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(lo), int(hi) + 1))
    for n in re.findall(r"(?<![\w\[:])v(\d+)\b", tok):
        out.add(int(n))
    return out


def audit(flags):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", SRC, "-o", os.path.join(d, "a.s")] + flags
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(os.path.join(d, "a.s")).read().split("\n")
    text = "\n".join(lines)
    m = re.search(r"\.vgpr_spill_count:\s*(\d+)", text[text.find("pair_tail160"):] if "pair_tail160" in text else text)
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text)]
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)]
    bad = []
    if any(spills) or any(scratch):
        bad.append(f"spills {spills} scratch {scratch}")
    loads = [i for i, l in enumerate(lines) if "buffer_load_dwordx4" in l and "sc1" in l]
    if len(loads) != 10:
        bad.append(f"expected 10 park loads, found {len(loads)}")
        return bad
    dest = set()
    for i in loads:
        dest |= regs_of(lines[i].split(",")[0])
    end = next((i for i in range(loads[-1], len(lines)) if "s_waitcnt vmcnt(6)" in lines[i]), None)
    if end is None:
        return bad + ["no s_waitcnt vmcnt(6) behind the park loads"]
    for i in range(loads[0] + 1, end):
        l = lines[i].strip()
        if not l or l.startswith(";") or l.startswith(".") or i in loads:
            continue
        hit = regs_of(l) & dest
        if hit:
            bad.append(f"line {i + 1}: '{l}' touches parked-output registers {sorted(hit)[:6]} before their wait")
    return bad


def main():
    rc = 0
    for name, flags in (("bf16", []), ("fp16", ["-DDSIM_H16_IS_F16"])):
        bad = audit(flags)
        print(f"attn160 asm-load audit [{name}]:", "ok" if not bad else "FAILED")
        for b in bad[:20]:
            print("   ", b)
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())
