"""Scan the shipped gfx950 code object for the hazard the compiler cannot see: the 64-bit DPP FMAs of the resident round kernel are
inline asm (`v_fmac_f64_dpp ... row_newbcast`, frx_round_kernel.hpp rk_fma_row), so LLVM's hazard recognizer never inserts the wait states a DPP
operand needs after a VALU write of the same register ("VALU writes VGPR -> DPP reads that VGPR: 2 wait states", CDNA3/4 ISA guide 4.5).  The source
ties an `s_nop 1` to the operands (rk_dpp_settle); this script checks the RESULT: for every DPP instruction of the code object, no VALU instruction
within the two preceding wait states (an `s_nop N` counts N + 1; a branch target or a branch in between ends the search conservatively as "unknown
predecessor" only when fewer than two wait states separate it from the DPP instruction - reported separately) writes a register the DPP operand reads.

  python scripts/check_dpp_hazards.py [path/to/libfrx.so]        -> JSON {dpp_instructions, hazards: [...], unknown_predecessor: n}; exit code 1 on a hazard
"""
import json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    m = REG.search(tok)
    if not m: return set()
    if m.group(1) is not None: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def disassemble(lib):
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(lib, os.path.join(td, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(td) if "gfx950" in f]
        if not co: raise RuntimeError("no gfx950 code object in " + lib)
        out = []                                                # one code object per device translation unit (three since round 6): all of them
        for c in sorted(co):
            out += subprocess.run([OBJDUMP, "-d", "--symbolize-operands", c], cwd=td, check=True, capture_output=True, text=True).stdout.split("\n")
            out.append("<end of code object>:")
        return out


def scan(lines):
    insts = []          # (mnemonic, operand string) or ("<label>", "")
    for l in lines:
        if re.match(r"^\S.*:$", l) or re.match(r"^<.*>:$", l.strip()):
            insts.append(("<label>", "")); continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", l)
        if m: insts.append((m.group(1), m.group(2)))
    n_dpp, hazards, unknown, n_asm, unknown_asm = 0, [], 0, 0, 0
    for i, (mn, ops) in enumerate(insts):
        if "_dpp" not in mn and " row_" not in ops and "quad_perm" not in ops and "wave_sh" not in ops: continue
        if not mn.startswith("v_"): continue
        n_dpp += 1
        is_asm = mn.startswith("v_fmac_f64")                   # the inline-asm kind (rk_fma_row): the only DPP instructions the compiler does not guard itself
        n_asm += is_asm
        parts = [p.strip() for p in ops.split(",")]
        if len(parts) < 2: continue
        src = regs(parts[1])                                   # src0 is the DPP operand (VOP2: dst, src0, src1; VOP1: dst, src0)
        if not src: continue
        states, j = 0, i - 1
        while j >= 0 and states < 2:
            pm, po = insts[j]
            if pm == "<label>" or pm.startswith("s_cbranch") or pm == "s_branch" or pm == "s_setpc_b64":
                unknown += 1; unknown_asm += is_asm; break
            if pm == "s_nop":
                try: states += int(po.split()[0], 0) + 1
                except Exception: states += 1
            else:
                if pm.startswith("v_") and not pm.startswith("v_cmp") and not pm.startswith("v_nop"):
                    dst = regs(po.split(",")[0])
                    if dst & src: hazards.append({"at": i, "dpp": mn + " " + ops, "writer": pm + " " + po, "wait_states_between": states})
                states += 1
            j -= 1
    return {"dpp_instructions": n_dpp, "inline_asm_dpp_fma": n_asm, "hazards": hazards, "unknown_predecessor": unknown, "unknown_predecessor_inline_asm": unknown_asm}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fast-racing_amd", "libfrx.so")
    res = scan(disassemble(lib))
    res["library"] = os.path.relpath(lib, ROOT)
    print(json.dumps({**res, "hazards": res["hazards"][:10], "n_hazards": len(res["hazards"])}))
    sys.exit(1 if res["hazards"] else 0)
