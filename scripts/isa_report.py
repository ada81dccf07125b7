"""Register pressure of the shipped kernels where it shows: in the gfx950 ISA.

Per kernel (code object metadata): VGPRs, AGPRs, SGPRs, SGPR / VGPR spill counts, scratch bytes, LDS, occupancy; and from the instruction stream the counts of
what a spill or an over-long live range costs on a lone wave's dependent chain:
    v_readlane / v_writelane   SGPR spills live in VGPR lanes: every use of a spilled scalar is a v_readlane_b32 (+ its wait states) in front of it
    accvgpr                    v_accvgpr_read / _write: VGPR values parked in AGPRs
    s_nop                      wait states the hazard recognizer (or rk_dpp_settle) inserted
    scratch                    scratch_load / scratch_store: real spills to memory
    v_mov                      register shuffles (v_mov_b32 / v_mov_b64 / v_dual_mov)
and the same counts per LOOP of the kernel (the compiler's own loop annotations: header label, depth) - the leader's round loop and the history loops are the
deepest, longest ones - so that a lever (kernarg reloads instead of live pointers, noinline bodies, 32-bit offsets) can be judged block by block before a GPU
is involved.  Static counts: a block's count says what ONE pass through it issues.  The dynamic side is scripts/r06/gpu_pmc_eval.sh (SQ_INSTS_VALU / SALU / the
F64 counters of k_eval_cluster against the stage kernels that run the same bodies without a spill).

Usage: python scripts/isa_report.py [--kernels k_eval_cluster,k_round,...] [--loops N] [--json out.json] [--asm file.s] [-D...]
Compiles the three device translation units of fast-racing_amd/csrc device-only to assembly, side by side (cached in /tmp by source content; ~1.5 minutes the first time).
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fast-racing_amd", "csrc")

CLASSES = [
    ("v_readlane", re.compile(r"^v_readlane_b32|^v_readfirstlane_b32")),
    ("v_writelane", re.compile(r"^v_writelane_b32")),
    ("accvgpr", re.compile(r"^v_accvgpr_(read|write|mov)")),
    ("s_nop", re.compile(r"^s_nop")),
    ("scratch", re.compile(r"^scratch_(load|store)")),
    ("v_mov", re.compile(r"^v_mov_b(32|64)|^v_dual_mov")),
    ("valu_f64", re.compile(r"^v_(add|mul|fma|fmac|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|max|min|ldexp|trunc|fract|frexp)\w*_f64")),
    ("valu", re.compile(r"^v_")),
    ("salu", re.compile(r"^s_(?!waitcnt|nop|barrier|sleep|endpgm|branch|cbranch|setprio|sethalt|trap|inst_prefetch|clause|delay_alu|waitcnt_depctr)")),
    ("s_load", re.compile(r"^s_(load|buffer_load)")),
    ("lds", re.compile(r"^ds_")),
    ("vmem", re.compile(r"^(global|flat|buffer)_")),
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("branch", re.compile(r"^s_(c?branch)")),
]


TUS = ["frx_device.hip", "frx_device_round.hip", "frx_device_eval.hip"]


def asm_path(defs, tus=None):
    """Assembly of the device translation units, concatenated (each compiled device-only, side by side; cached in /tmp by the sources' content)."""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    h = hashlib.sha1()
    for s in srcs:
        h.update(s.encode()); h.update(open(s, "rb").read())
    h.update(" ".join(defs).encode())
    tus = tus or TUS
    outs = [f"/tmp/{os.path.splitext(tu)[0]}_{h.hexdigest()[:12]}.s" for tu in tus]
    procs = [subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *defs, os.path.join(CSRC, tu), "-o", o], stderr=subprocess.DEVNULL)
             for tu, o in zip(tus, outs) if not os.path.exists(o)]
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    cat = f"/tmp/frx_all_{h.hexdigest()[:12]}_{len(tus)}.s"
    with open(cat, "w") as f:
        for o in outs:
            f.write(open(o).read()); f.write("\n")
    return cat


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def count(lines):
    c = {k: 0 for k, _ in CLASSES}
    c["total"] = 0
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c["total"] += 1
        for k, rx in CLASSES:
            if rx.match(op):
                c[k] += 1
    c["non_f64_valu"] = c["valu"] - c["valu_f64"]
    return c


def parse(path):
    lines = open(path).read().split("\n")
    kernels = {}
    # kernel bodies: "<sym>:" ... "s_endpgm"-terminated function followed by ".amdhsa_kernel <sym>"
    hsa = [(i, l.split()[1]) for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel ")]
    for i_hsa, sym in hsa:
        start = next(i for i in range(i_hsa, -1, -1) if lines[i].startswith(sym + ":"))
        body = lines[start:i_hsa]
        info = {}
        for l in lines[i_hsa:i_hsa + 90]:                   # the compiler's summary comments follow .end_amdhsa_kernel
            m = re.match(r"^;\s*(NumSgprs|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte):\s*(\d+)", l.strip())
            if m:
                info[m.group(1)] = int(m.group(2))
        kernels[sym] = {"meta": info, "body": body}
    # spill counts: metadata yaml
    txt = "\n".join(lines)
    for blk in txt.split("  - .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or nm.group(1) not in kernels:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        kernels[nm.group(1)]["meta"].update(sgpr_spill=g("sgpr_spill_count"), vgpr_spill=g("vgpr_spill_count"), scratch_bytes=g("private_segment_fixed_size"))
    return kernels


def loops_of(body):
    """Basic blocks grouped by the innermost loop the compiler says they belong to."""
    blocks, cur = [], {"label": "entry", "hdr": None, "depth": 0, "lines": []}
    blocks.append(cur)
    for i, l in enumerate(body):
        m = re.match(r"^\.L(BB\d+_\d+):\s*(?:;\s*(.*))?$", l) or re.match(r"^; %bb\.(\d+):\s*(?:;\s*(.*))?$", l)
        if m:
            note = m.group(2) or ""
            j = i + 1
            while j < len(body) and body[j].strip().startswith(";") and ("Loop" in body[j] or "Parent" in body[j] or "Child" in body[j]):
                note += " " + body[j].strip(); j += 1
            hdr, depth = None, 0
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if mm:
                hdr, depth = mm.group(1), int(mm.group(2))
            mm = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", note)
            if mm and l.startswith(".L"):
                hdr, depth = m.group(1), int(mm.group(1))
            cur = {"label": m.group(1), "hdr": hdr, "depth": depth, "lines": []}
            blocks.append(cur)
        else:
            cur["lines"].append(l)
    loops = {}
    for b in blocks:
        key = b["hdr"] or "(straight-line code outside loops)"
        e = loops.setdefault(key, {"depth": b["depth"], "blocks": 0, "lines": []})
        e["blocks"] += 1; e["lines"] += b["lines"]
    return {k: dict(depth=v["depth"], blocks=v["blocks"], **count(v["lines"])) for k, v in loops.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernels", default="k_eval_cluster,k_eval_service,k_round,k_forward_knot64,k_backward_knot64,k_penalty_lat,k_penalty_lat2")
    ap.add_argument("--loops", type=int, default=6, help="loops listed per kernel (longest first)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--asm", default=None)
    ap.add_argument("--tu", default=None, help="comma-separated translation units (default: all three)")
    args, defs = ap.parse_known_args()
    path = args.asm or asm_path(defs, args.tu.split(",") if args.tu else None)
    kernels = parse(path)
    names = demangle(list(kernels))
    want = [w for w in args.kernels.split(",") if w]
    report = {}
    for sym, k in kernels.items():
        nice = names[sym]
        short = re.sub(r"^void ", "", nice)
        short = re.sub(r"\(.*$", "", short).replace("frx::", "")
        if want and not any(short == w or short.startswith(w + "<") for w in want):
            continue
        c = count(k["body"])
        lp = loops_of(k["body"])
        top = sorted(lp.items(), key=lambda kv: -kv[1]["total"])[:args.loops]
        report[short] = {"meta": k["meta"], "instructions": c, "loops": {h: v for h, v in top}}
    cols = ["total", "valu", "valu_f64", "non_f64_valu", "v_mov", "v_readlane", "v_writelane", "accvgpr", "s_nop", "scratch", "salu", "s_load", "lds", "vmem", "s_waitcnt", "branch"]
    for name, r in report.items():
        m = r["meta"]
        print(f"== {name}")
        print("   VGPR %s AGPR %s SGPR %s | SGPR spills %s VGPR spills %s scratch %s B | LDS %s B occupancy %s | code %s B" % (
            m.get("NumVgprs"), m.get("NumAgprs"), m.get("NumSgprs"), m.get("sgpr_spill"), m.get("vgpr_spill"), m.get("scratch_bytes"), m.get("LDSByteSize"), m.get("Occupancy"), m.get("codeLenInByte")))
        print("   " + " ".join(f"{c}={r['instructions'][c]}" for c in cols))
        for h, v in r["loops"].items():
            print(f"     loop {h:<40s} depth {v['depth']} blocks {v['blocks']:>3d}: " + " ".join(f"{c}={v[c]}" for c in cols if v[c]))
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)
        print("wrote", args.json)


if __name__ == "__main__":
    main()
