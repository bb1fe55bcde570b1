"""Build liblatte_amd.so (hand-written HIP for gfx950) in-tree: latte_amd/lib/liblatte_amd.so.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEBUG_BUILD = bool(os.environ.get("LATTE_DEBUG_BUILD"))
# the measurement build (ablation instantiations, the kernels that were measured and not kept) lives beside the product library and is only
# ever loaded when LATTE_AMD_LIB names it (tools/, never tests or bench defaults)
OBJ = os.path.join(HERE, "build_dbg" if DEBUG_BUILD else "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblatte_amd_dbg.so" if DEBUG_BUILD else "liblatte_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["gemm.hip", "gemm_pw.hip", "gemm_tn.hip", "train.hip", "train_fin.hip", "train_attn.hip", "attention.hip", "qkv_attn.hip", "pointwise.hip", "debug.hip", "vae.hip", "engine.cpp", "train_engine.cpp", "vae_engine.cpp", "t2v_engine.cpp", "schedule.cpp"]
COMMON = ["--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if DEBUG_BUILD:   # measurement build: main-loop / epilogue ablation instantiations of the GEMM
    COMMON.append("-DLATTE_GEMM_ABLATE")
# compiler-flag A/B builds (measurement only): LATTE_BUILD_TAG=<tag> LATTE_EXTRA_HIPFLAGS="<flags>" builds lib/liblatte_amd_<tag>.so
# from objects under build_<tag>/ with the extra flags on the .hip files; loaded through LATTE_AMD_LIB like the measurement build
BUILD_TAG = os.environ.get("LATTE_BUILD_TAG", "")
EXTRA_HIP = os.environ.get("LATTE_EXTRA_HIPFLAGS", "").split() if BUILD_TAG else []
if BUILD_TAG:
    OBJ = os.path.join(HERE, "build_" + BUILD_TAG)
    LIB = os.path.join(LIBDIR, "liblatte_amd_%s.so" % BUILD_TAG)
FLAGS = {
    ".hip": COMMON + ["-O3"] + EXTRA_HIP,
    # host logic reproduces fp64/fp32 reference arithmetic: no FMA contraction
    ".cpp": COMMON + ["-O2", "-ffp-contract=off"],
}


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return hdrs


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_digest = _digest(_deps())
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        stamp = obj + ".stamp"
        want = _digest([src]) + hdr_digest + " ".join(FLAGS[os.path.splitext(s)[1]])
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            continue
        jobs.append((src, obj, stamp, want, FLAGS[os.path.splitext(s)[1]]))

    def run(job):
        src, obj, stamp, want, flags = job
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(want)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
