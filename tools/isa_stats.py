"""Development aid: per-kernel register / scratch / instruction-mix summary of a hipcc -save-temps .s file.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c x.hip -save-temps ; python tools/isa_stats.py x-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.strip().split("\n")
    except FileNotFoundError:
        return names


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else None
    md = s[s.find("amdhsa.kernels:"):]
    ents = md.split("  - .agpr_count:")[1:]
    rows = []
    for e in ents:
        g = lambda k: re.search(r"\." + k + r":\s*(\S+)", e).group(1)
        rows.append((g("name"), e.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
    dn = demangle([r[0] for r in rows])
    for r, d in zip(rows, dn):
        d = d.replace("dsim::(anonymous namespace)::", "").replace("void ", "").replace("_ZN4dsim12_GLOBAL__N_1", "").replace("EEvNS_8GemmArgsEi", "")
        if flt and flt not in d:
            continue
        # instruction mix of the kernel body
        m = re.search(r"^" + re.escape(r[0]) + r":[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
        body = m.group(1) if m else ""
        cnt = lambda pat: len(re.findall(pat, body, re.M))
        mix = dict(mfma=cnt(r"^\s+v_mfma"), valu=cnt(r"^\s+v_(?!mfma)"), ds=cnt(r"^\s+ds_"), vmem=cnt(r"^\s+(buffer|global|scratch)_"),
                   salu=cnt(r"^\s+s_(?!waitcnt|barrier|nop)"), wait=cnt(r"^\s+s_waitcnt"), bar=cnt(r"^\s+s_barrier"))
        print(f"{d[:90]:90s} agpr {r[1]:>3s} vgpr {r[2]:>3s} sgpr {r[3]:>3s} spill {r[4]:>3s} scratch {r[5]:>4s} | "
              + " ".join(f"{k} {v}" for k, v in mix.items()))

main()
