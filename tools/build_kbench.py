"""Build tools/kbench (kernel micro-benchmark; development aid) against the library's object files."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffsim_amd import build as B  # noqa: E402

B.build()
objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B.SOURCES]
out = os.path.join(ROOT, "tools", "kbench")
ko = os.path.join(B.OBJ, "kbench.o")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(ROOT, "tools", "kbench.hip"), "-o", ko],
               check=True)
subprocess.run([B.HIPCC, "--offload-arch=gfx950", ko] + objs + ["-o", out], check=True)
print("built", out)
