"""Build libfedrann_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "fedrann_hip.hip")
OUT = os.path.join(HERE, "libfedrann_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result", "-Wno-inline-asm", "-pthread"]


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))]
    deps.append(os.path.join(HERE, "..", "include", "fedrann_hip.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build_library(force=False, verbose=False):
    """Compile fedrann_amd/csrc/fedrann_hip.hip -> fedrann_amd/libfedrann_hip.so."""
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path()] + FLAGS + [SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
