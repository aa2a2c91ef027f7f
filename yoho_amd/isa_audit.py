"""ISA audit of the one place where a kernel hides a returning memory operation from the compiler.

gft16x_kernel (csrc/gft16.hip) draws its chunk tickets with a `global_atomic_add ... sc0` inside an asm block: the value comes back
a chunk later, under the counted vmcnt wait at the top of the next iteration, and is published from the same asm block.  Between two
executions of that block the compiler believes the register already holds the value - so the register must not be spilled, copied or
otherwise referenced anywhere else in the kernel, or a stale value would travel.  `audit` checks exactly that on gfx950 assembly (plus:
no scratch, no AGPR copies in the kernel).

yoho_amd.build runs it as part of EVERY build, on assembly produced with the very flags the object is compiled with (the shipped library
and the YOHO_EXPERIMENTS one alike), and fails the build when the invariant does not hold; tools/check_isa.py and tests/test_abi.py run the
same check by hand / in the CPU suite.
"""
import re

KERNEL = "_ZN4yoho13gft16x_kernelENS_9Gft16ArgsE"


def _kernel(txt):
    m = re.search(r"^%s:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(KERNEL), txt, flags=re.S | re.M)
    if not m:
        raise RuntimeError("gft16x_kernel not found in the assembly")
    meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(KERNEL), txt, flags=re.S)
    return m.group(1).splitlines(), (meta.group(1) if meta else "")


def registers_of(operand_text):
    """VGPR numbers an instruction's operand text touches: v12, v[4:7]"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", operand_text))
    return regs


def audit(asm_text):
    """(ok, message) for the gfx950 assembly text of csrc/gft16.hip"""
    lines, meta = _kernel(asm_text)
    code = [(i, ln.strip()) for i, ln in enumerate(lines) if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.strip().endswith(":")]
    hidden = [(i, ln) for i, ln in code if re.match(r"global_atomic_add\s+v\d+,\s*v\[\d+:\d+\],\s*v\d+,\s*off\s+sc0", ln)]
    if len(hidden) != 1:
        return False, f"expected exactly one hidden ticket request in gft16x_kernel, found {len(hidden)}"
    idx, ln = hidden[0]
    t = int(re.match(r"global_atomic_add\s+v(\d+)", ln).group(1))
    prev = [c for c in code if c[0] < idx][-1][1]
    if not re.match(r"ds_write_b32\s+v\d+,\s*v%d\b" % t, prev):
        return False, f"the ticket register v{t} is not published by the ds_write_b32 right in front of the request: {prev!r}"
    others = []
    for i, c in code:
        if i == idx or c == prev and i == [x for x in code if x[0] < idx][-1][0]:
            continue
        ops = c.split(None, 1)[1] if " " in c else ""
        if t in registers_of(ops):
            others.append(c)
    # one initialisation in front of the loop (v_mov) is the only other reference allowed
    inits = [c for c in others if re.match(r"v_mov_b32(_e32)?\s+v%d\b" % t, c)]
    rest = [c for c in others if c not in inits]
    if len(inits) > 1 or rest:
        return False, f"ticket register v{t} is referenced outside the asm block: {rest or inits}"
    if any("v_accvgpr" in c for _, c in code):
        return False, "gft16x_kernel copies registers through AGPRs (a spill in disguise)"
    if re.search(r"\.amdhsa_private_segment_fixed_size\s+[1-9]", meta):
        return False, "gft16x_kernel uses scratch memory"
    return True, f"ticket register v{t}: written by the hidden request, read by the publishing ds_write_b32, initialised once, nothing else"



def audit_file(path):
    with open(path) as f:
        return audit(f.read())
