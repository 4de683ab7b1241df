"""Build libepipolar_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libepipolar_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every csrc/*.hip to an object and link the shared library.  Returns the library path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "epipolar_hip.h"))
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


GLUE_NAME = "epi_torch_glue"


def glue_path():
    import sysconfig
    return os.path.join(LIB_DIR, GLUE_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build_glue(force=False, verbose=True):
    """Compile csrc/torch_glue.cpp (C++ autograd glue over the C ABI) with g++ against the installed torch headers and
    link it to libepipolar_hip.so (same directory, rpath $ORIGIN).  Returns the extension module's path."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    out = glue_path()
    src = os.path.join(CSRC, "torch_glue.cpp")
    deps = [src, os.path.join(os.path.dirname(HERE), "include", "epipolar_hip.h"), LIB_PATH]
    if not force and not _stale(out, deps):
        return out
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-sign-compare",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + GLUE_NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in ce.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]:
        cmd += ["-I", inc]
    cmd += [src, "-o", out, "-L", torch_lib, "-L", LIB_DIR, "-L", os.path.join(rocm, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch", "-ltorch_python", "-lepipolar_hip", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_glue(force="--force" in sys.argv)
