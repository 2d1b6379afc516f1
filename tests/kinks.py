"""Shared by the CPU and GPU gradient-parity tests: the ReLU sign pattern ("masks") an implementation used.

A pre-activation within rounding of 0 lands on either side of the ReLU kink under a different fp32 summation order and
switches a whole gradient path on or off.  Rounds 1-2 absorbed that in 5e-3 / 2e-2 tolerances.  Round 3 compares like with
like instead: every gradient comparison evaluates both sides with the SAME masks, and the units where two implementations
took different sides are counted, checked to lie within 1e-4 of the kink, and exempted -- exactly those, nothing else.

  * the reference's masks: fixtures g7_* store the reference's own bit for every unit the oracle sees within 1e-4 of the
    kink (tools/make_golden.py: forward hooks on the reference's ReLU blocks; outside that band the generator asserts that
    the reference's pattern equals the oracle's);
  * the HIP path's masks: read out of the saved-activation image (csrc/mlp_layout.h S_H / S_DIRH rows, post-ReLU values).
"""
import numpy as np

BAND = 1e-4                                            # |pre-activation| below this = "at the kink"
SAVED_ROWS, S_H, S_DIRH = 2604, 64, 64 + 8 * 256 + 256 + 32     # csrc/mlp_layout.h


def oracle_masks(cache):
    return {"h": [p > 0 for p in cache["pres"]], "dir": (cache["dir_pre"] > 0) if "dir_pre" in cache else None}


def reference_masks(g, mi, cache):
    """The sign pattern the REFERENCE used for model `mi` of fixture g: the oracle's, with the stored reference bits at the
    units within BAND of the kink.  -> (masks, number of units where the reference and the oracle took different sides)"""
    m = oracle_masks(cache)
    m = {"h": [a.copy() for a in m["h"]], "dir": None if m["dir"] is None else m["dir"].copy()}
    lay, pt, un, bit = (g[f"kink{mi}_{k}"] for k in ("layer", "point", "unit", "refbit"))
    flips = 0
    for li in range(9):
        sel = lay == li
        arr = m["h"][li] if li < 8 else m["dir"]
        if arr is None:
            continue
        flips += int((arr[pt[sel], un[sel]] != bit[sel]).sum())
        arr[pt[sel], un[sel]] = bit[sel]
    return m, flips


def hip_masks(saved, n_points, order=None):
    """ReLU masks of the HIP forward from its saved-activation image (tile-major: one tile per 32 points).  Inside a tile
    the fp32 path keeps the "x4" element order (row r, point p at ((r >> 2) * 32 + p) * 4 + (r & 3); csrc/mlp_core.h at4),
    the split-bf16 path [row][32 points].  `saved`: the tensor the ops layer returned (it carries its order as
    `_nerfmi_math`) or an array with order = "x4" | "rows"."""
    if order is None:
        order = {"f32": "x4", "bf16x3": "rows", None: "x4"}[getattr(saved, "_nerfmi_math", None)]
    arr = saved.detach().cpu().numpy() if hasattr(saved, "detach") else np.asarray(saved)
    ld = (n_points + 31) // 32 * 32
    tiles = arr[: SAVED_ROWS * ld].reshape(ld // 32, SAVED_ROWS * 32)
    if order == "x4":
        tiles = tiles.reshape(ld // 32, SAVED_ROWS // 4, 32, 4).transpose(0, 1, 3, 2)
    img = tiles.reshape(ld // 32, SAVED_ROWS, 32).transpose(1, 0, 2).reshape(SAVED_ROWS, ld)
    img = img[:, :n_points]
    return {"h": [img[S_H + 256 * l: S_H + 256 * (l + 1)].T > 0 for l in range(8)], "dir": img[S_DIRH: S_DIRH + 128].T > 0}


def count_flips(masks, cache, band=BAND):
    """Units where `masks` differs from the oracle's own pattern -> count; every one must be within `band` of the kink."""
    own = oracle_masks(cache)
    pres = list(cache["pres"]) + [cache.get("dir_pre")]
    n = 0
    for li in range(9):
        a = masks["h"][li] if li < 8 else masks["dir"]
        b = own["h"][li] if li < 8 else own["dir"]
        if a is None or b is None:
            continue
        diff = a != b
        n += int(diff.sum())
        assert np.all(np.abs(pres[li][diff]) < band), (li, float(np.abs(pres[li][diff]).max()))
    return n


def rel(a, b):
    return float(np.linalg.norm((np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel())
                 / max(np.linalg.norm(np.asarray(b, np.float64).ravel()), 1e-30))
