"""CPU-side checks of the drop-in boundary: libnerfmi.so builds/loads without a
GPU and exports every symbol include/nerfmi.h declares; the Python mirror keeps
the reference's signatures; there is no CPU fallback."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from nerf_siren_amd import build
    return build.build()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nerfmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerfmi_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nerfmi.h but not exported"
    from nerf_siren_amd import _lib
    assert sorted(_lib.SIGNATURES) == names          # the ctypes binding covers the whole header
    lib.nerfmi_version.restype = ctypes.c_int
    assert lib.nerfmi_version() >= 100               # no GPU needed
    lib.nerfmi_nerf_packed_floats.restype = ctypes.c_size_t
    assert lib.nerfmi_nerf_packed_floats() == 1162248   # csrc/mlp_layout.h PACKED_FLOATS


def test_argument_validation_without_gpu(libpath):
    """Invalid arguments are rejected before any launch (safe on a CPU-only box)."""
    from nerf_siren_amd import _lib
    l = _lib.lib()
    assert l.nerfmi_sample_stratified(None, None, 4, 64, 0, 0.0, None, None) == -1
    assert b"null" in l.nerfmi_last_error()
    assert l.nerfmi_composite(None, 0, None, None, None, 0.0, 1, 5000, 0, None, None, None, None, None) == -1
    assert l.nerfmi_searchsorted(None, None, 3, 2, 5, 5, 0, None, None) == -1
    assert b"broadcast" in l.nerfmi_last_error()
    assert l.nerfmi_sample_stratified(None, None, 0, 64, 0, 0.0, None, None) == 0      # empty batch is a no-op
    # the entry points added around the path (f1 / f2) and the opt-in math validate the same way
    assert l.nerfmi_mse_loss(None, None, None, 12, 1.0, None, None, None, None) == -1 and b"null" in l.nerfmi_last_error()
    assert l.nerfmi_mse_loss(None, None, None, 0, 1.0, None, None, None, None) == -1
    assert l.nerfmi_adam_step(None, None, None, None, 8, 5e-4, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, None) == -1   # step >= 1
    assert l.nerfmi_adam_step(None, None, None, None, 8, 5e-4, 0.4, 0.999, 1e-8, 0.0, 1, 1.0, None) == -1   # null, beta1
    assert l.nerfmi_adam_step(None, None, None, None, 0, 5e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == 0    # empty
    assert l.nerfmi_generate_rays(None, 1, 4, 4, 5.0, None, 7, 0, 2.0, 6.0, None, None) == -1
    assert b"n_images*H*W" in l.nerfmi_last_error()
    assert l.nerfmi_generate_rays(None, 1, 4, 4, -1.0, None, 16, 0, 2.0, 6.0, None, None) == -1
    assert l.nerfmi_ray_directions(0, 4, 5.0, None, None) == -1
    assert l.nerfmi_get_rays(None, None, 0, None, None, None) == 0
    # round 3: the one-call render, the conditioning-gradient backward, the split-bf16 SIREN training entries, the in-kernel
    # EG3D draws and the profiler validate the same way
    a24 = [None] * 24
    assert l.nerfmi_render_rays_fused(2, *a24[:5], 4, 64, 64, 0, 0.0, 0.0, 1, 0, 1, 2, *a24[:8]) == -1 and b"field_kind" in l.nerfmi_last_error()
    assert l.nerfmi_render_rays_fused(0, *a24[:5], 4, 64, 64, 0, 0.0, 0.0, 1, 0, 1, 2, *a24[:8]) == -1 and b"null" in l.nerfmi_last_error()
    assert l.nerfmi_render_rays_fused(0, *a24[:5], 0, 64, 64, 0, 0.0, 0.0, 1, 0, 1, 2, *a24[:8]) == 0          # empty batch
    assert l.nerfmi_render_rays_workspace_floats(1024, 64, 64, 0) == 1024 * (64 + 64 + 256 + 128 + 512)
    assert l.nerfmi_render_rays_workspace_floats(1024, 64, 0, 1) == 1024 * (64 + 64 + 64)
    import ctypes as C
    gp = (C.c_void_p * 22)()
    assert l.nerfmi_siren_backward_cond(None, None, None, None, 64, gp, None, None, None, None) == -1
    assert b"conditioning-gradient" in l.nerfmi_last_error()
    assert l.nerfmi_siren_backward_fast(None, None, None, None, None, 64, 64, gp, None, None, None, None) == -1
    assert l.nerfmi_siren_forward_rays_train_fast(None, None, None, None, None, None, 4, 64, 4, None, None, None) == -1
    assert l.nerfmi_siren_forward_rays_train_fast(None, None, None, None, None, None, 0, 64, 1, None, None, None) == 0
    assert l.nerfmi_eg3d_sample_stratified_philox(None, None, 0.0, 1.0, 1, 2, 8, 48, 0, None, None) == -1
    assert l.nerfmi_eg3d_sample_importance_philox(None, None, 1, 2, 8, 48, 48, None, None) == -1
    assert l.nerfmi_eg3d_sample_stratified(None, None, 0.0, 1.0, None, 8, 48, 0, None, None) == -1
    # the profiler is inert without launches: start / report (empty) / stop
    assert l.nerfmi_profile_start() == 0 and l.nerfmi_profile_report(None, 0) == 0 and l.nerfmi_profile_stop() == 0
    assert l.nerfmi_nerf_forward_rays_fast(None, None, None, None, 4, 8, 0, None, None, None) == -1
    assert l.nerfmi_nerf_backward_rays_fast(None, None, 4, 8, None, None, None, None, None) == -1
    assert l.nerfmi_siren_forward_rays_fast(None, None, None, None, None, None, 0, 8, 1, 0, None, None) == 0
    assert l.nerfmi_nerf_fast_bytes() > 3 * 2 * 1100000 and l.nerfmi_siren_fast_bytes() > 3 * 2 * 500000
    # round 2: FiLM-SIREN training path, module-API training forward, in-kernel draws, EG3D sample grid
    assert l.nerfmi_siren_forward_rays_train(None, None, None, None, None, 4, 8, 4, None, None, None) == -1
    assert b"null" in l.nerfmi_last_error()
    assert l.nerfmi_siren_forward_rays_train(None, None, None, None, None, 0, 8, 1, None, None, None) == 0   # empty
    assert l.nerfmi_siren_forward_points_train(None, None, None, None, None, 5, 0, None, None, None) == -1   # per_cond >= 1
    assert l.nerfmi_siren_backward(None, None, None, None, 0, 1, None, None, None) == -1                      # n_points >= 1
    assert l.nerfmi_siren_backward(None, None, None, None, 64, 64, None, None, None) == -1
    assert l.nerfmi_siren_saved_floats(64) == 4676 * (64 + 32)                     # csrc/siren_core.h SIREN_SAVED_ROWS + dump tile
    assert l.nerfmi_siren_backward_workspace_floats(64) > 2312 * (64 + 32)
    assert l.nerfmi_siren_packed_floats() == 1076736                               # forward + small + transposed images + stream tail (siren_core.h)
    assert l.nerfmi_nerf_forward_embedded_train(None, None, 8, None, None, None) == -1
    assert l.nerfmi_nerf_forward_embedded_train(None, None, 0, None, None, None) == 0
    assert l.nerfmi_render_draws(1, 2, 16, None, 0, None, 0, None, 0, None, None) == -1 and b"segment 0" in l.nerfmi_last_error()
    assert l.nerfmi_render_draws(1, 2, 0, None, 0, None, 0, None, 0, None, None) == 0
    assert l.nerfmi_sample_stratified_philox(None, 1, 2, 4, 64, 0, 1.0, None, None) == -1
    assert l.nerfmi_composite_philox(None, 0, None, None, 1, 2, 2, 1.0, 4, 64, 0, None, None, None, None, None) == -1
    assert b"segment" in l.nerfmi_last_error()                                     # 1 (coarse) or 3 (fine) only
    assert l.nerfmi_composite_backward_philox(None, None, None, 1, 2, 3, 1.0, 0, 64, 0, None, None, None, None, None) == 0
    assert l.nerfmi_importance_resample_philox(None, None, 1, 2, 4, 2, 64, None, None, None) == -1          # S >= 3
    assert l.nerfmi_create_samples(1, 0.0, 0.0, 0.0, 1.0, None, None) == -1 and l.nerfmi_create_samples(8, 0.0, 0.0, 0.0, 1.0, None, None) == -1


def test_python_api_mirrors_reference_signature():
    from nerf_siren_amd import rendering, nerf
    sig = inspect.signature(rendering.render_rays)
    names = list(sig.parameters)
    # models/rendering.py:70-83
    assert names[:13] == ["models", "embeddings", "rays", "N_samples", "use_disp", "perturb", "noise_std",
                          "N_importance", "chunk", "white_back", "test_time", "_cls_num", "network"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["N_samples"], d["use_disp"], d["perturb"], d["noise_std"], d["N_importance"], d["chunk"],
            d["white_back"], d["test_time"]) == (64, False, 0, 1, 0, 1024 * 32, False, False)
    assert list(inspect.signature(rendering.sample_pdf).parameters)[:5] == ["bins", "weights", "N_importance", "det", "eps"]
    m = nerf.NeRF()
    keys = list(m.state_dict().keys())
    assert keys[0] == "xyz_encoding_1.0.weight" and "xyz_encoding_final.bias" in keys and len(keys) == 24
    assert sum(p.numel() for p in m.parameters()) == 595844
    with pytest.raises(NotImplementedError):
        nerf.NeRF(D=4)


def test_no_cpu_fallback():
    import torch
    from nerf_siren_amd import ops, nerf
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.sample_stratified(torch.zeros(4, 8), 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nerf.Embedding(3, 10)(torch.zeros(4, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nerf_siren_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def _hazard_scan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "tools", "hazard_scan.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_hazard_scanner_recognises_the_pattern():
    hs = _hazard_scan()
    bad = """
0000000000001000 <k>:
\tv_cvt_pk_bf16_f32 v137, v28, v38
\ts_waitcnt lgkmcnt(3)
\tv_mfma_f32_32x32x16_bf16 a[240:255], v[30:33], v[136:139], a[240:255]
"""
    ok = bad.replace("s_waitcnt lgkmcnt(3)", "s_nop 1")
    far = bad.replace("s_waitcnt lgkmcnt(3)", "s_waitcnt lgkmcnt(3)\n\tv_add_f32_e32 v1, v2, v3")
    assert len(hs.scan_disassembly(bad)[3]) == 1 and hs.scan_disassembly(bad)[3][0][1] == 1
    assert hs.scan_disassembly(ok)[3] == [] and hs.scan_disassembly(far)[3] == []
    # advisor (round 2): a conditional branch that is NOT taken falls straight through -- the write in front of it still
    # counts; an unconditional branch ends the straight line; a write in front of a (back-)branch reaches the MFMA at its
    # target label
    fall = bad.replace("s_waitcnt lgkmcnt(3)", "s_cbranch_scc1 L9")
    assert len(hs.scan_disassembly(fall)[3]) == 1
    uncond = bad.replace("s_waitcnt lgkmcnt(3)", "s_branch L9")
    assert hs.scan_disassembly(uncond)[3] == []
    loop = """
0000000000001000 <k>:
\tv_mov_b32_e32 v1, 0
\ts_nop 4
0000000000001010 <L0>:
\tv_mfma_f32_32x32x16_bf16 a[240:255], v[30:33], v[136:139], a[240:255]
\ts_nop 7
\tv_cvt_pk_bf16_f32 v137, v28, v38
\ts_cbranch_scc1 L0
\ts_endpgm
"""
    h = hs.scan_disassembly(loop)[3]
    assert len(h) == 1 and h[0][1] == 1, h           # write, branch (one wait state), MFMA at the label
    assert hs.scan_disassembly(loop.replace("s_cbranch_scc1 L0", "s_nop 0\n\ts_cbranch_scc1 L0"))[3] == []


def test_no_valu_write_to_mfma_read_hazard_in_the_built_kernels(libpath):
    """gfx950 needs two wait states between a vector instruction's write of a VGPR and an MFMA's read of it.  The
    compiler pads its own instructions but not inline asm (csrc/bf16x3_core.h split_pair used to convert through asm and
    one kernel instance read a stale B operand): scan every built code object, any hit is a bug."""
    hs = _hazard_scan()
    objs = hs.product_objects()
    assert len(objs) >= 10
    mfmas = 0
    for o in objs:
        for kernels, n_ins, n_mfma, hits in hs.scan_file(o):
            assert not hits, (os.path.basename(o), hits[:3])
            mfmas += n_mfma
    assert mfmas > 100000                      # the MLP kernels were really disassembled
