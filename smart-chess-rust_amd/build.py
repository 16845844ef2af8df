"""Builds the in-tree native libraries (gfx950 only):

  lib/libsc_engine.so      HIP kernels + C ABI (include/sc_engine.h)          hipcc --offload-arch=gfx950
  lib/libsc_rules_host.so  host build of the device rules code for CPU tests   g++
  lib/sc-selfplay          CLI mirroring the reference's `selfplay` flags      g++
  lib/sc-play              CLI mirroring the reference's `play` flags (matches) g++
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
BUILD = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build failed: " + " ".join(cmd[:4]))
    return r.stdout


def build(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(BUILD, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "sc_engine.h"))
    common = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    objs = []
    for src, extra in (("mcts_kernels.hip", ["-ffp-contract=off"]), ("nn_kernels.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
                       ("step_kernels.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]), ("engine.hip", [])):
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + hdrs):
            out = _run(common + extra + ["-c", s, "-o", o])
            if verbose and out.strip():
                print(out)
        objs.append(o)
    so = os.path.join(LIB, "libsc_engine.so")
    if force or _newer(so, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", so] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
    host = os.path.join(LIB, "libsc_rules_host.so")
    hs = os.path.join(CSRC, "rules_host_api.cpp")
    if force or _newer(host, [hs] + hdrs):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fPIC", "-shared", "-ffp-contract=off", "-o", host, hs])
    cli_src = os.path.join(CSRC, "selfplay_main.cpp")
    cli = os.path.join(LIB, "sc-selfplay")
    if os.path.exists(cli_src) and (force or _newer(cli, [cli_src, so] + hdrs)):
        _run(["g++", "-O2", "-std=c++17", "-pthread", cli_src, "-o", cli, "-L" + LIB, "-lsc_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib",
              "-Wl,-rpath,/opt/rocm/lib"])
    play_src = os.path.join(CSRC, "play_main.cpp")
    play = os.path.join(LIB, "sc-play")
    if os.path.exists(play_src) and (force or _newer(play, [play_src, so] + hdrs)):
        _run(["g++", "-O2", "-std=c++17", play_src, "-o", play, "-L" + LIB, "-lsc_engine", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib",
              "-Wl,-rpath,/opt/rocm/lib"])
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
