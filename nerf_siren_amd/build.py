"""Build libnerfmi.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# experiment builds (tools/exp_*.py): NERFMI_LIB_OUT=<path> writes another library, NERFMI_SOURCES="a.hip b.hip" restricts
# the translation units, NERFMI_EXTRA_FLAGS="-D..." adds switches (objects of a different flag set get their own directory)
LIB = os.environ.get("NERFMI_LIB_OUT") or os.path.join(HERE, "lib", "libnerfmi.so")
SOURCES = ["rays.hip", "mlp.hip", "mlp_bwd.hip", "siren.hip", "siren_bwd.hip", "eg3d.hip", "eg3d_bwd.hip", "mlp_bf16x3.hip", "train_step.hip", "raygen.hip"]
# -ffp-contract=off: the per-ray kernels reproduce torch's op-by-op fp32 rounding
# (oracle/nerf_oracle.py); fused multiply-adds are written explicitly where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _deps(src: str, seen=None) -> set:
    """src plus every header it includes with quotes, recursively (the translation unit's rebuild set)."""
    seen = set() if seen is None else seen
    src = os.path.normpath(src)
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    with open(src) as f:
        for line in f:
            m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
            if m:
                _deps(os.path.join(os.path.dirname(src), m.group(1)), seen)
    return seen


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _cflags():
    extra = os.environ.get("NERFMI_EXTRA_FLAGS", "").split()       # experiment builds (-D switches in csrc/)
    return [f for f in FLAGS if f != "-shared"] + extra


def _objdir():
    # objects are kept between builds (one per translation unit, rebuilt when the unit or a header it includes changed);
    # a different flag set gets its own directory so experiment builds never mix with the product objects
    tag = hashlib.sha1(" ".join(_cflags()).encode()).hexdigest()[:10]
    return os.path.join(HERE, "lib", "obj-" + tag)


def _jobs():
    names = os.environ.get("NERFMI_SOURCES", "").split() or SOURCES
    srcs = [os.path.join(CSRC, s) for s in names]
    missing = [s for s in srcs if not os.path.exists(s)]
    if missing:
        raise RuntimeError(f"missing sources: {missing}")
    me = os.path.abspath(__file__)
    return [(src, os.path.join(_objdir(), os.path.basename(src) + ".o"), _deps(src) | {me}) for src in srcs]


def needs_build() -> bool:
    jobs = _jobs()
    return any(_stale(obj, deps) for _, obj, deps in jobs) or _stale(LIB, [obj for _, obj, _ in jobs if os.path.exists(obj)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(_objdir(), exist_ok=True)
    jobs = [(src, obj, [hipcc()] + _cflags() + ["-c", src, "-o", obj]) for src, obj, deps in _jobs()
            if force or _stale(obj, deps)]

    def run(job):
        src, obj, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and os.path.exists(obj):
            os.remove(obj)
        return src, r

    # the fully unrolled MLP kernels take minutes each: compile the translation units side by side
    from concurrent.futures import ThreadPoolExecutor
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for src, r in ex.map(run, jobs):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj, _ in _jobs()] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
