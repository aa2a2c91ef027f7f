"""Build oracle/_ref/me_maps.so: MinkowskiEngine's CPU coordinate-map manager, compiled from the reference's OWN source file where it
lies (/root/reference/MinkowskiEngine/src/coordinate_map_manager.cpp, -DCPU_ONLY) plus the driver oracle/me_ref/me_maps_driver.cpp,
with g++ against the installed torch / pybind11 headers.  TEST INFRASTRUCTURE: only oracle/gen_golden_me.py and tests use it.

Not the reference's build system (setup.py compiles ~40 translation units and needs a BLAS with cblas.h for the convolution
arithmetic, which this image lacks - that part stays unbuilt, see the driver's header); no reference source is copied.
Needs /root/reference (build container only); the GPU box uses the prebuilt .so, and the tests run from tests/golden/me_maps.npz.

    python oracle/build_me_ref.py [--force]
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ME = "/root/reference/MinkowskiEngine"
OUT = os.path.join(HERE, "_ref", "me_maps.so")
DRIVER = os.path.join(HERE, "me_ref", "me_maps_driver.cpp")


def build(force=False, verbose=True):
    src = os.path.join(ME, "src", "coordinate_map_manager.cpp")
    if not os.path.exists(src):
        return None                                   # no reference tree here (GPU box): use the prebuilt file if there is one
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(DRIVER), os.path.getmtime(src)):
        return OUT
    import torch
    T = os.path.dirname(torch.__file__)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-w", "-DCPU_ONLY", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-DTORCH_EXTENSION_NAME=me_maps", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{T}/include", f"-I{T}/include/torch/csrc/api/include", f"-I{sysconfig.get_paths()['include']}",
           f"-I{ME}/src", f"-I{ME}/src/3rdparty", DRIVER, src,
           f"-L{T}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", f"-Wl,-rpath,{T}/lib", "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def load():
    """import the built module (None if it does not exist)"""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location("me_maps", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print("built", build(force="--force" in sys.argv))
