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


def scan_disassembly(text: str):
    """-> (kernels, instructions, mfmas, hits); a hit = (kernel, wait_states, writer, mfma)."""
    kern, hits, n_ins, n_mfma, kernels = None, [], 0, 0, 0
    last_def, t = {}, 0
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kern, last_def, t = m.group(1), {}, 0
            kernels += 1
            continue
        ins = line.split("//")[0].strip()
        if not ins or kern is None:
            continue
        op, _, rest = ins.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
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
                        hits.append((kern, t - d[0], d[1], ins))
            t += 1
            continue
        if op.startswith("v_") and ops:
            for r in _regs(ops[0].split(" ")[0]):
                last_def[r] = (t + 1, ins)          # the next instruction issues 0 wait states after this one
        if op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc"):
            last_def = {}                             # straight-line check only
        t += 1
    return kernels, n_ins, n_mfma, hits


def code_objects(path: str, tmp: str):
    """The gfx950 code object(s) inside a host object / shared library with ONE offload bundle, or the file itself."""
    with open(path, "rb") as f:
        head = f.read(64)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":          # e_machine = EM_AMDGPU
        return [path]
    fat = os.path.join(tmp, os.path.basename(path) + ".fatbin")
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
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
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
