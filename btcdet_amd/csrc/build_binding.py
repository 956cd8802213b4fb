"""Builds btcdet_amd/_btcfast<EXT_SUFFIX> (csrc/binding.cpp): the compiled PyTorch binding of libbtcdet_hip.so.
Plain g++ against the torch headers of the running interpreter; links ../libbtcdet_hip.so with an $ORIGIN rpath.
    python btcdet_amd/csrc/build_binding.py [--force]"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)


def target():
    return os.path.join(PKG, "_btcfast" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    import torch
    from torch.utils import cpp_extension as ce
    src, out = os.path.join(HERE, "binding.cpp"), target()
    deps = [src, os.path.join(PKG, "..", "include", "btcdet_hip.h"), os.path.join(PKG, "libbtcdet_hip.so")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps[:2]):
        return out
    if not os.path.exists(deps[2]):
        raise RuntimeError("build libbtcdet_hip.so first (make -C btcdet_amd/csrc)")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out,
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-DTORCH_EXTENSION_NAME=_btcfast", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-Wno-attributes", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM", "-I/opt/rocm/include"]
    cmd += ["-I" + i for i in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    cmd += ["-L" + l for l in ce.library_paths()] + ["-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_python", "-lamdhip64"]
    cmd += ["-L" + PKG, "-lbtcdet_hip", "-Wl,-rpath,$ORIGIN"] + ["-Wl,-rpath," + l for l in ce.library_paths()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed for binding.cpp:\n" + r.stderr[-4000:])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
