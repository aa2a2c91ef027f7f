"""ISA audit of gft16x_kernel's hidden ticket request (see yoho_amd/isa_audit.py) by hand:

    python tools/check_isa.py            # exit code 0 = the invariant holds, for the shipped flags AND the YOHO_EXPERIMENTS ones

Compiles csrc/gft16.hip to assembly with the flag lists of yoho_amd.build (the build itself runs the same audit on every build).
"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from yoho_amd import build as B          # noqa: E402
from yoho_amd.isa_audit import audit     # noqa: E402


def check(experiments=None):
    """experiments: None = the flags of the library yoho_amd.build is configured for, True / False = with / without -DYOHO_EXPERIMENTS"""
    flags = [f for f in B.FLAGS if f != "-DYOHO_EXPERIMENTS"]
    if experiments or (experiments is None and B.EXPERIMENTS):
        flags.append("-DYOHO_EXPERIMENTS")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gft16.s")
        subprocess.check_call(B.asm_command("gft16.hip", out, flags), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return audit(open(out).read())


if __name__ == "__main__":
    rc = 0
    for exp in (False, True):
        ok, msg = check(exp)
        print(("OK" if ok else "FAILED") + (" (experiments build): " if exp else " (shipped build): ") + msg)
        rc |= 0 if ok else 1
    sys.exit(rc)
