"""Build libvoicemap_hip.so for gfx950 in-tree (voicemap_amd/lib/).

hipcc cross-compiles without a GPU; objects are rebuilt only when a source or header is newer.
``python -m voicemap_amd.build [--force]``.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvoicemap_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"] + os.environ.get("VM_EXTRA_HIPCC_FLAGS", "").split()
# conv1_fused: its kernels read every MFMA result in a VALU epilogue and are VALU-bound; with the accumulators in VGPRs
# (instead of AGPRs) the epilogue needs no v_accvgpr_read copies and the kernels fit one more wave per SIMD.
FILE_FLAGS = {"conv1_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "voicemap_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    s = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(s)
            and os.path.getmtime(obj) >= _headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB):
            return LIB  # GPU box without a toolchain: use the prebuilt library that travelled with the tree
        raise RuntimeError("hipcc not found at %s and no prebuilt %s" % (HIPCC, LIB))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
