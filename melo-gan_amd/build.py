"""Builds libmelogan_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the
library is a plain C-ABI HIP shared object (include/melo_gan_hip.h).

    python melo-gan_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmelogan_hip.so")
SOURCES = ["runtime.hip", "conv_mfma.hip", "conv16_mfma.hip", "conv_thin.hip", "conv_bf16.hip", "conv_wino.hip", "linear_skinny.hip", "wgrad_mfma.hip", "small_kernels.hip", "row_chain.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
# conv_mfma.hip stores transposed weight quads as scalar pairs on purpose (see store_pair there)
EXTRA_FLAGS = {"conv_mfma.hip": ["-fno-slp-vectorize"]}


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    deps = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "melo_gan_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(d, op) for d in deps):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(OUT) or any(_newer(o, OUT) for o in objs):      # also after a hand-compiled object
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
