"""Builds tests/hostcheck/_build/libhostcheck.so (test infrastructure: device math compiled for the host)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libhostcheck.so")


def build(force=False):
    src = os.path.join(HERE, "hostcheck.hip")
    csrc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "epipolarpose_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("selfsup.hip", "fundamental.hip", "linalg3.h", "common.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", src, "-o", OUT],
                   check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
