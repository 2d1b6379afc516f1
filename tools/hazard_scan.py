"""Static check of the built gfx950 code objects for one hazard the assembler does not police: an MFMA that reads a VGPR
less than two wait states after a vector instruction wrote it.

The compiler's hazard recogniser inserts the `s_nop`s for the instructions it schedules itself, but it does not look inside
inline asm -- an asm VALU instruction placed right in front of the MFMA that consumes its result hands the matrix core the
register's stale contents (this was the round-2 "load-dependent NaN" of the FiLM-SIREN split-bf16 kernel; see
csrc/bf16x3_core.h split_pair).  The scan therefore needs no knowledge of where an instruction came from: any hit in the
disassembly is a bug.

usage: python tools/hazard_scan.py [object-or-code-object ...]     (default: the objects of the current product build)
exit status 1 when a hazard is found.
"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEED = 2                                     # wait states between a VALU write of a VGPR and an MFMA read of it


def _regs(tok: str):
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def _parse(text: str):
    """-> list of kernels, each a list of ('label', name) / ('ins', op, operands, text) in address order
    (llvm-objdump --symbolize-operands: branch targets are printed as L<n> and defined by '<L<n>>:' lines)."""
    kernels, cur = [], None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if re.fullmatch(r"L\d+", m.group(1)):
                if cur is not None:
                    cur[1].append(("label", m.group(1)))
            else:
                cur = (m.group(1), [])
                kernels.append(cur)
            continue
        ins = line.split("//")[0].strip()
        if not ins or cur is None:
            continue
        op, _, rest = ins.partition(" ")
        cur[1].append(("ins", op, [o.strip() for o in rest.split(",")] if rest else [], ins))
    return kernels


def _walk(items, incoming, record):
    """One linear pass over a kernel.  last_def: VGPR -> (slot after which it is readable - NEED bookkeeping, writer).
    A CONDITIONAL branch falls through, so the state survives it; only an unconditional branch / s_setpc ends the straight
    line.  At a label the tails of ALL predecessors meet: every branch to the label hands over the registers it saw written
    within the last NEED slots (`incoming`, filled when record=True), merged conservatively (latest write wins).
    -> (instructions, mfmas, hits)"""
    last_def, t, n_ins, n_mfma, hits = {}, 0, 0, 0, []
    for it in items:
        if it[0] == "label":
            for elapsed, r, writer in incoming.get(it[1], ()):
                cand = t - elapsed
                if r not in last_def or cand > last_def[r][0]:
                    last_def[r] = (cand, writer)
            continue
        _, op, ops, ins = it
        if op == "s_nop":
            t += int(ops[0], 0) + 1
            continue
        n_ins += 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            n_mfma += 1
            for o in ops[1:4]:
                for r in _regs(o.split(" ")[0]):
                    d = last_def.get(r)
                    if d is not None and t - d[0] < NEED:
                        hits.append((t - d[0], d[1], ins))
            t += 1
            continue
        if op.startswith("v_") and ops:
            for r in _regs(ops[0].split(" ")[0]):
                last_def[r] = (t + 1, ins)          # the next instruction issues 0 wait states after this one
        t += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            target = ops[0] if ops else ""
            if record and re.fullmatch(r"L\d+", target):
                tail = [(t - d[0], r, d[1]) for r, d in last_def.items() if t - d[0] < NEED]
                if tail:
                    incoming.setdefault(target, []).extend(tail)
        if op == "s_branch" or op.startswith("s_setpc"):
            last_def = {}                             # nothing falls through an unconditional branch
    return n_ins, n_mfma, hits


def scan_disassembly(text: str):
    """-> (kernels, instructions, mfmas, hits); a hit = (kernel, wait_states, writer, mfma)."""
    n_ins = n_mfma = 0
    hits = []
    kernels = _parse(text)
    for name, items in kernels:
        incoming = {}
        _walk(items, incoming, record=True)          # pass 1: what every branch carries to its target
        a, b, h = _walk(items, incoming, record=False)
        n_ins += a
        n_mfma += b
        hits += [(name,) + x for x in h]
    return len(kernels), n_ins, n_mfma, hits


def code_objects(path: str, tmp: str):
    """The gfx950 code object(s) inside a host object / shared library with ONE offload bundle, or the file itself."""
    with open(path, "rb") as f:
        head = f.read(64)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":          # e_machine = EM_AMDGPU
        return [path]
    fat = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", path], check=True, capture_output=True, text=True).stdout
    if ".hip_fatbin" not in sections:                # a host-only translation unit (csrc/render.hip): no device code
        return []
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = os.path.join(tmp, os.path.basename(path) + ".co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [out]


def scan_file(path: str):
    with tempfile.TemporaryDirectory() as tmp:
        res = []
        for co in code_objects(path, tmp):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", co], check=True, capture_output=True,
                                 text=True).stdout
            res.append(scan_disassembly(dis))
        return res


def product_objects():
    sys.path.insert(0, ROOT)
    from nerf_siren_amd import build
    return sorted(glob.glob(os.path.join(build._objdir(), "*.o")))


def main(argv):
    paths = argv or product_objects()
    if not paths:
        print("no objects (build first)")
        return 2
    bad = 0
    for p in paths:
        for kernels, n_ins, n_mfma, hits in scan_file(p):
            print(f"{os.path.basename(p)}: {kernels} kernels, {n_ins} instructions, {n_mfma} MFMAs, {len(hits)} hazards")
            for h in hits[:10]:
                print("   ", h)
            bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
