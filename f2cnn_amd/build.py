"""Build libf2cnn_hip.so (hand-written HIP for gfx950) in-tree with hipcc. No GPU is needed to build."""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libf2cnn_hip.so")
ARCH = "gfx950"
# The FFT butterflies are scalar f32 code on register arrays; SLP-packing them into v_pk_*_f32 costs more
# v_mov than it saves (packed f32 VALU has no rate advantage at >= 2 waves/SIMD on gfx950).
PER_FILE_FLAGS = {"f2_envelope.hip": ("-fno-slp-vectorize",),
                  "f2_envelope_large.hip": ("-fno-slp-vectorize",), "f2_envelope_split.hip": ("-fno-slp-vectorize",),
                  "f2_envelope_pair.hip": ("-fno-slp-vectorize",), "f2_envelope_p3.hip": ("-fno-slp-vectorize",),
                  "f2_envelope_flagged.hip": ("-fno-slp-vectorize",), "f2_spectral.hip": ("-fno-slp-vectorize",)}


def hipcc_path():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libf2cnn_hip.so cannot be built")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    """Compile every csrc/*.hip for gfx950 into lib/libf2cnn_hip.so. Returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc_path(), "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
               *PER_FILE_FLAGS.get(os.path.basename(src), ()), *extra_flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc_path(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
