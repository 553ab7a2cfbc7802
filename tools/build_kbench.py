"""Build tools/kbench (kernel micro-benchmark; development aid): the library's sources compiled with -DDSIM_DEVTOOLS
(tile overrides, environment switches, ablation instantiations -- none of which exist in the product library)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffsim_amd import build as B  # noqa: E402

EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]          # e.g. -DDSIM_EXP_LATE -> tools/kbench_EXP_LATE
TAG = "".join(a[2:].replace("DSIM", "") for a in EXTRA)
OBJ = os.path.join(B.CSRC, "_obj_dev" + TAG)
os.makedirs(OBJ, exist_ok=True)


def cc(src):
    path = os.path.join(ROOT, "tools", src) if src == "kbench.hip" else os.path.join(B.CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    if B._stale(obj, [path] + B.HEADERS):
        subprocess.run([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DDSIM_DEVTOOLS"] + EXTRA + ["-c", path, "-o", obj], check=True)
    return obj


only = [s for s in B.SOURCES if s not in ("unet.hip", "vae.hip", "dit.hip")]        # kbench calls the kernels directly
with ThreadPoolExecutor(max_workers=4) as ex:
    objs = list(ex.map(cc, only + ["kbench.hip"]))
out = os.path.join(ROOT, "tools", "kbench" + TAG)
subprocess.run([B.HIPCC, "--offload-arch=gfx950"] + objs + ["-o", out], check=True)
print("built", out)
