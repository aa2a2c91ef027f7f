"""Build libyoho_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m yoho_amd.build            # in-tree: yoho_amd/lib/libyoho_hip.so

The library is linked against the HIP runtime only (libamdhip64.so.7 by SONAME); inside a
PyTorch-ROCm process the already-loaded runtime of torch is the one that gets used, so device
pointers from torch tensors are valid in the library.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
# YOHO_EXPERIMENTS=1: a second library beside the shipped one, with the timing-experiment switches compiled in (YOHO_PARTI_DEBUG /
# YOHO_FGEMM_DEBUG, csrc/common.h experiment_env); tools select it with YOHO_LIB=exp, nothing else ever loads it
EXPERIMENTS = os.environ.get("YOHO_EXPERIMENTS") == "1"
OBJDIR = os.path.join(LIBDIR, "exp") if EXPERIMENTS else LIBDIR
LIB = os.path.join(LIBDIR, "libyoho_hip_exp.so" if EXPERIMENTS else "libyoho_hip.so")
SOURCES = ["api.hip", "gconv.hip", "gconv16.hip", "fourier.hip", "gemmf.hip", "gemmf2.hip", "gft16.hip", "cone1.hip", "sparse.hip", "train.hip", "layout.hip", "match.hip", "matchf.hip", "gridnn.hip", "estim.hip", "pair.hip"]
# Kernels whose results must be bit-exact against numpy / torch-CPU arithmetic are compiled without
# FMA contraction (hipcc defaults to -ffp-contract=fast and __fmul_rn/__fadd_rn are plain operators
# in this ROCm, so they would fuse); explicit fma()/fmaf() calls are unaffected.
EXTRA = {"layout.hip": ["-ffp-contract=off"], "match.hip": ["-ffp-contract=off"], "matchf.hip": ["-ffp-contract=off"], "gridnn.hip": ["-ffp-contract=off"], "estim.hip": ["-ffp-contract=off"], "pair.hip": ["-ffp-contract=off"]}
if os.environ.get("YOHO_SPCONV_ABLATE"):          # timing experiments: compile-time ablations of the fine-level sparse conv (YOHO_SPCONV_DEBUG)
    EXTRA["sparse.hip"] = ["-DYOHO_SPCONV_ABLATE"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
if EXPERIMENTS:
    FLAGS.append("-DYOHO_EXPERIMENTS")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "yoho_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def asm_command(src, out, flags=None):
    """the device assembly of one source with the flags its object is compiled with (yoho_amd.isa_audit, tools/check_isa.py)"""
    return [_hipcc()] + list(FLAGS if flags is None else flags) + EXTRA.get(src, []) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    # the ISA audit of gft16x_kernel's hidden ticket request runs on assembly made with THESE flags, as part of the build
    asm = os.path.join(OBJDIR, "gft16.s")
    procs.append(("gft16.hip (assembly for the ISA audit)", subprocess.Popen(asm_command("gft16.hip", asm))))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    # The audit matches mnemonics of THIS toolchain's output.  A positive violation (the ticket register referenced elsewhere, scratch,
    # AGPR copies) always fails the build.  "Pattern not found" - another hipcc spelling or scheduling the request differently - fails it
    # too unless YOHO_ISA_AUDIT=warn says the builder has looked (then the library is linked with static striding advised:
    # YOHO_XF_STEAL=0 avoids the audited code path at run time); YOHO_ISA_AUDIT=skip skips the audit.
    policy = os.environ.get("YOHO_ISA_AUDIT", "strict")
    if policy != "skip":
        from .isa_audit import audit_file
        try:
            ok, msg = audit_file(asm)
            found = not msg.startswith(("expected exactly one hidden ticket request", "the ticket register")) or ok
        except RuntimeError as e:                # the kernel itself was not found in the assembly
            ok, msg, found = False, str(e), False
        if verbose:
            print("ISA audit:", msg, flush=True)
        if not ok:
            if policy == "warn" and not found:
                print("WARNING: the ISA audit did not find the pattern it checks (another toolchain?); linking anyway because YOHO_ISA_AUDIT=warn. "
                      "Run contexts with YOHO_XF_STEAL=0 unless the assembly has been inspected by hand.", file=sys.stderr, flush=True)
            else:
                raise RuntimeError("ISA audit of gft16x_kernel failed (the library was NOT linked; YOHO_ISA_AUDIT=warn links when only the "
                                   "pattern is missing): " + msg)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
