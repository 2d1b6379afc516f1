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
SOURCES = ["rays.hip", "mlp.hip", "mlp_bwd.hip", "siren.hip", "siren_bwd.hip", "eg3d.hip", "eg3d_bwd.hip", "mlp_bf16x3.hip", "train_step.hip", "raygen.hip", "render.hip"]
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


def _digest(paths, extra=()) -> str:
    """sha256 over the CONTENT of `paths` (sorted) and the strings in `extra`: what "up to date" is judged by, so that a
    fresh checkout (all mtimes = now) and the tree the library was built in agree."""
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _cflags():
    extra = os.environ.get("NERFMI_EXTRA_FLAGS", "").split()       # experiment builds (-D switches in csrc/)
    return [f for f in FLAGS if f != "-shared"] + extra


def _objdir():
    # objects are kept between builds (one per translation unit, rebuilt when the CONTENT of the unit or of a header it
    # includes changed); a different flag set gets its own directory so experiment builds never mix with the product objects
    tag = hashlib.sha1(" ".join(_cflags()).encode()).hexdigest()[:10]
    return os.path.join(HERE, "lib", "obj-" + tag)


def _jobs():
    names = os.environ.get("NERFMI_SOURCES", "").split() or SOURCES
    srcs = [os.path.join(CSRC, s) for s in names]
    missing = [s for s in srcs if not os.path.exists(s)]
    if missing:
        raise RuntimeError(f"missing sources: {missing}")
    out = []
    for src in srcs:
        obj = os.path.join(_objdir(), os.path.basename(src) + ".o")
        out.append((src, obj, _digest(_deps(src), _cflags())))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _obj_current(obj, digest) -> bool:
    return os.path.exists(obj) and _read(obj + ".sha256") == digest


def _lib_digest(jobs) -> str:
    return hashlib.sha256("\n".join(d for _, _, d in jobs).encode()).hexdigest()


def needs_build() -> bool:
    jobs = _jobs()
    return not (os.path.exists(LIB) and _read(LIB + ".sha256") == _lib_digest(jobs))


# what the last build() call in this process did: {"compiled": [...], "reused": [...], "linked": bool}
LAST_BUILD = {"compiled": [], "reused": [], "linked": False}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every translation unit whose content digest (sources + quoted headers + flags) differs from the one its
    object was built from, link to a temporary name once ALL objects exist, then move the library into place."""
    jobs = _jobs()
    LAST_BUILD.update(compiled=[], reused=[], linked=False)
    if not force and not needs_build():
        LAST_BUILD["reused"] = [os.path.basename(s) for s, _, _ in jobs]
        if verbose:
            print(f"libnerfmi.so is up to date (content digest of {len(jobs)} translation units matches): compiled 0, "
                  f"reused {len(jobs)}", flush=True)
        return LIB
    os.makedirs(_objdir(), exist_ok=True)
    todo = [(src, obj, dig, [hipcc()] + _cflags() + ["-c", src, "-o", obj]) for src, obj, dig in jobs
            if force or not _obj_current(obj, dig)]
    LAST_BUILD["reused"] = [os.path.basename(s) for s, o, d in jobs if not any(s == t[0] for t in todo)]

    def run(job):
        src, obj, dig, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        for stale in (obj, obj + ".sha256"):
            if os.path.exists(stale):
                os.remove(stale)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(obj + ".sha256", "w") as f:
                f.write(dig)
        elif os.path.exists(obj):
            os.remove(obj)
        return src, r

    # the fully unrolled MLP kernels take minutes each: compile the translation units side by side
    from concurrent.futures import ThreadPoolExecutor
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            for src, r in ex.map(run, todo):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
                LAST_BUILD["compiled"].append(os.path.basename(src))
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj, _ in jobs] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    with open(LIB + ".sha256", "w") as f:
        f.write(_lib_digest(jobs))
    LAST_BUILD["linked"] = True
    if verbose:
        print(f"compiled {len(LAST_BUILD['compiled'])} translation unit(s) {LAST_BUILD['compiled']}, reused "
              f"{len(LAST_BUILD['reused'])}, linked {os.path.basename(LIB)}", flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
