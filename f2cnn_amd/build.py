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


# Kernels whose inline-asm loads are waited for by hand (s_waitcnt counts written for ONE register allocation): a spill or
# scratch slot in them means the compiler moved registers the waits do not cover. The build reports it; at run time
# f2_cnn_create's self-check decides whether they are used (csrc/f2_cnn.hip: cnn_ws_selfcheck).
NO_SCRATCH_KERNELS = {"f2_cnn_ws.hip": ("k_conv12_ws", "k_conv34_ws", "k_dense1_ws")}
RESOURCE_REPORT = os.path.join(LIB_DIR, "kernel_resources.txt")


def parse_resource_remarks(text):
    """{mangled kernel name: {"VGPRs": n, "ScratchSize": bytes per lane, ...}} from -Rpass-analysis=kernel-resource-usage"""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


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
        if os.path.basename(src) in NO_SCRATCH_KERNELS:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    report = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        wanted = NO_SCRATCH_KERNELS.get(os.path.basename(src))
        if wanted:
            res = parse_resource_remarks(out)
            for kern in wanted:
                hits = {k: v for k, v in res.items() if kern in k}
                if not hits:
                    report.append(f"{kern}: no resource remark from this hipcc (not checked)")
                for k, v in hits.items():
                    bad = v.get("ScratchSize", 0) or v.get("VGPRs Spill", 0)
                    report.append(f"{kern}: VGPRs {v.get('VGPRs')} scratch {v.get('ScratchSize')} B/lane "
                                  f"VGPR spills {v.get('VGPRs Spill')}" + ("  <-- HAND-PLACED WAITS NOT VALID FOR THIS CODE" if bad else ""))
                    if bad:
                        print(f"[f2cnn_amd.build] WARNING: {kern} uses scratch memory with this hipcc; its hand-counted "
                              "s_waitcnt no longer cover every register move. The library's self-check "
                              "(f2_cnn_create) will compare it with the per-tile kernels before using it.", file=sys.stderr)
            out = "\n".join(l for l in out.splitlines() if "kernel-resource-usage" not in l)
        if verbose and out.strip():
            print(out)
    if report:
        ver = subprocess.run([hipcc_path(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        with open(RESOURCE_REPORT, "w") as f:
            f.write(ver.strip().splitlines()[0] + "\n" + "\n".join(report) + "\n")
    cmd = [hipcc_path(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
