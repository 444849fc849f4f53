"""Build the native artefacts in-tree with hipcc for gfx950 (no JIT cache: the .so travels with
the repo snapshot to the GPU box)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
ROOT = os.path.dirname(_HERE)
ARCH = "gfx950"


def lib_path():
    # SBMBP_LIB selects an alternative build of the same library (kernel tuning A/B runs)
    return os.environ.get("SBMBP_LIB") or os.path.join(CSRC, "libsbmbp_hip.so")


def build_variant(name, defines):
    """compile a tuning variant csrc/variants/libsbmbp_<name>.so with extra -D flags"""
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "dist.hip", "host_graph.cpp")]
    vdir = os.path.join(CSRC, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, "libsbmbp_%s.so" % name)
    subprocess.check_call([_hipcc(), "-std=c++14", "-O3", "--offload-arch=" + ARCH, "-fPIC", "-shared", "-pthread", "-o", out] +
                          ["-D" + d for d in defines] + srcs + ["-lrccl"])
    return out


def bin_path():
    return os.path.join(ROOT, "bin", "bp")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "dist.hip", "host_graph.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kernels.h", "kernels_wide.h", "host_graph.h")] + [os.path.join(ROOT, "include", "sbmbp.h")]
    out = os.path.join(CSRC, "libsbmbp_hip.so")
    if force or _newer(out, deps):
        cmd = [_hipcc(), "-std=c++14", "-O3", "--offload-arch=" + ARCH, "-fPIC", "-shared", "-pthread", "-o", out] + srcs + ["-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_cli(force=False, verbose=False):
    src = os.path.join(CSRC, "bp_main.cpp")
    if not os.path.exists(src):
        return None
    out = bin_path()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or _newer(out, [src, lib_path(), os.path.join(CSRC, "host_graph.h")]):
        cmd = [_hipcc(), "-std=c++14", "-O2", src, "-o", out, "-L" + CSRC, "-lsbmbp_hip",
               "-Wl,-rpath," + CSRC, "-Wl,-rpath,$ORIGIN/../sbm-bp_amd/csrc"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    cli = build_cli(force, verbose)
    return lib, cli
