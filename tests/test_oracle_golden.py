"""Pin the CPU oracle (oracle/nerf_oracle.py) to the reference's own outputs
(tests/golden/*.npz, produced by tools/make_golden.py from /root/reference).

Tolerances: bit-exact where the arithmetic is specified (linspace, z values,
cumsum/cumprod given equal inputs, searchsorted indices given equal (cdf,u));
a few ulp where torch's CPU kernels have an unspecified order or libm (row
sums, torch.norm, exp/sin/cos); 1e-4 abs end to end (north_star)."""
import numpy as np
import pytest

from nerf_siren_amd import synth
from oracle import nerf_oracle as O


def test_linspace(golden):
    g = golden("g1_sampler")
    for n in (2, 3, 64, 65, 128, 192):
        assert np.array_equal(O.linspace01(n), g[f"linspace_{n}"])


@pytest.mark.parametrize("S", [64, 17])
@pytest.mark.parametrize("disp", [0, 1])
@pytest.mark.parametrize("pert", [0.0, 1.0, 0.5])
def test_sampler_bit_exact(golden, S, disp, pert):
    g = golden("g1_sampler")
    tag = f"S{S}_disp{disp}_p{pert}"
    rays = g["rays_" + tag]
    z = O.sample_z(rays, S, bool(disp), pert, g["prand_" + tag])
    xyz = O.points(rays, z)
    assert np.array_equal(xyz, g["xyz_" + tag])


def test_embedding(golden):
    g = golden("g2_embedding")
    assert np.abs(O.embed(g["x"], 10) - g["emb10"]).max() <= 2.4e-7   # 2 ulp of 1.0
    assert np.abs(O.embed(g["x"], 4) - g["emb4"]).max() <= 2.4e-7
    assert np.array_equal(O.embed(g["x"], 10)[:, :3], g["x"])


def test_nerf_mlp(golden):
    g = golden("g3_nerf")
    p = synth.nerf_params(1)
    out = O.nerf_forward(p, g["x"])
    assert np.abs(out[:, :3] - g["out"][:, :3]).max() < 2e-6
    np.testing.assert_allclose(out[:, 3], g["out"][:, 3], rtol=2e-5, atol=2e-5)
    sig = O.nerf_forward(p, g["x"][:, :63], sigma_only=True)
    np.testing.assert_allclose(sig, g["sigma"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_composite(golden, tag):
    g = golden("g4_composite")
    r = O.composite(g[tag + "_sigma"], g[tag + "_rgb"], _z(g[tag + "_rays"], g[tag + "_sigma"].shape[1]),
                    g[tag + "_rays"][:, 3:6], g[tag + "_noise"], float(g[tag + "_noise_std"]),
                    bool(g[tag + "_white_back"]))
    np.testing.assert_allclose(r["rgb"], g[tag + "_out_rgb"], atol=5e-7, rtol=0)
    np.testing.assert_allclose(r["opacity"], g[tag + "_out_opacity"], atol=5e-7, rtol=0)
    np.testing.assert_allclose(r["depth"], g[tag + "_out_depth"], atol=3e-6, rtol=0)
    np.testing.assert_allclose(r["weights"][:, 1:-1], g[tag + "_weights_inner"], atol=1.2e-7, rtol=1e-6)
    # edge rows (tools/make_golden.py g_composite): transparent, negative, opaque
    if tag == "a":                                            # noise_std == 0
        assert r["opacity"][0] == 0 and r["opacity"][1] == 0
        assert abs(r["opacity"][2] - 1) < 1e-6 and abs(r["opacity"][3] - 1) < 1e-6


def _z(rays, P):
    return O.sample_z(rays, P)


@pytest.mark.parametrize("case", ["det", "rnd", "tie"])
def test_sample_pdf_stage2_bit_exact(golden, case):
    """search+lerp on the reference's own (cdf,u): indices and samples bit-exact."""
    g = golden("g5_sample_pdf")
    inds, samples = O.search_lerp(g["bins"], g[case + "_cdf"], g[case + "_u"])
    assert np.array_equal(inds, g[case + "_inds"])
    assert np.array_equal(samples, g[case + "_samples"])
    assert inds.dtype == np.int64


def test_sample_pdf_stage1_cdf(golden):
    g = golden("g5_sample_pdf")
    cdf = O.build_cdf(g["weights"])
    # row-sum order differs from torch's vectorised sum by <=1 ulp of the total;
    # the cumsum itself is bit-equal given the same pdf.
    np.testing.assert_allclose(cdf, g["det_cdf"], atol=2.4e-7, rtol=0)
    frac = (cdf == g["det_cdf"]).mean()
    assert frac > 0.5
    s, aux = O.sample_pdf(g["bins"], g["weights"], 64, det=True)
    agree = (aux["inds"] == g["det_inds"]).mean()
    assert agree > 0.99, agree
    s2, aux2 = O.sample_pdf(g["bins"][:, :20], g["weights"][:, :19], 37, det=True)
    assert (aux2["inds"] == g["odd_inds"]).mean() > 0.99
    assert np.abs(s2 - g["odd_samples"]).max() < 0.15  # a flipped index moves one sample by <= one bin


def test_searchsorted_matrix(golden):
    g = golden("g5_searchsorted")
    a, v = g["a"], g["v"]
    for nm, (aa, vv) in dict(full=(a, v), bca=(a[:1], v), bcv=(a, v[:1])).items():
        for side in ("left", "right"):
            assert np.array_equal(O.searchsorted(aa, vv, side), g[f"{nm}_{side}"])


RENDER_CASES = ["blender_det", "blender_train", "ndc_train", "blender_test_time", "blender_disp",
                "coarse_only", "odd_sizes"]


def run_oracle_case(g, keep=False):
    F = int(g["F"])
    params = [synth.nerf_params(1, sigma_bias=-1.0), synth.nerf_params(2, sigma_bias=0.5)]
    rng = {k[4:]: g[k] for k in g if k.startswith("rng_")}
    return params, O.render_rays(params, g["rays"], int(g["S"]), bool(g["use_disp"]), float(g["perturb"]),
                                 float(g["noise_std"]), F, bool(g["white_back"]), bool(g["test_time"]),
                                 rng=rng, keep=keep)


@pytest.mark.parametrize("case", RENDER_CASES)
def test_render_rays_end_to_end(golden, case):
    g = golden("g7_" + case)
    _, res = run_oracle_case(g)
    keys = [k[4:] for k in g if k.startswith("out_")]
    assert set(keys) == set(k for k in res if not k.startswith("_"))
    span = float((g["rays"][:, 7] - g["rays"][:, 6]).max())
    N = g["rays"].shape[0]
    flipped = np.zeros(N, bool)
    if int(g["F"]) > 0:
        # One ulp of cdf (torch's row-sum order is ISA dependent) can flip a
        # searchsorted index and move that fine sample by up to one coarse bin
        # (SURVEY section 7 'bit-exact sample indices', section 8 a3): such rays
        # are listed and held to a looser bound; all others to 1e-4.
        mism = res["_aux"]["inds"] != g["mid_inds"]
        flipped = mism.any(1)
        assert mism.mean() < 0.03, mism.mean()
        assert np.all(np.diff(res["_aux"]["z_fine"], axis=-1) >= 0)
    for k in keys:
        tol = 1e-4 * (span if "depth" in k else 1.0)       # SURVEY section 7 contract (iii)
        err = np.abs(res[k] - g["out_" + k]).reshape(N, -1).max(-1)
        loose = flipped & np.array(["fine" in k] * N)
        assert np.all(err[~loose] <= tol), (k, err[~loose].max())
        assert np.all(err[loose] <= 100 * tol), (k, err[loose].max())


def _loss_grads(g, res):
    """d loss / d outputs for the loss of tools/make_golden.py g_render."""
    N = g["rays"].shape[0]
    t = g["target"]
    grads = {"rgb_coarse": 2 * (res["rgb_coarse"] - t) / (3 * N), "depth_coarse": np.full(N, 0.1 / N, np.float32),
             "opacity_coarse": np.full(N, 0.3 / N, np.float32)}
    if int(g["F"]) > 0:
        grads.update({"rgb_fine": 2 * (res["rgb_fine"] - t) / (3 * N),
                      "depth_fine": 0.2 * 2 * res["depth_fine"] / N,
                      "opacity_fine": np.full(N, -0.1 / N, np.float32)})
    return grads


def grad_rel_errors(g, gs):
    """max over tensors of ||mine - ref|| / ||ref|| per model (subsampled for big tensors)."""
    worst = []
    for mi, gm in enumerate(gs):
        if gm is None:
            continue
        w = 0.0
        for k, v in gm.items():
            ref = g.get(f"grad{mi}_{k}")
            mine = np.asarray(v)
            if ref is None:
                ref = g[f"grad{mi}_{k}_sub"]
                mine = mine.reshape(-1)[::37]
            else:
                nrm = float(g[f"grad{mi}_{k}_norm"])
                assert abs(np.linalg.norm(mine.astype(np.float64)) - nrm) <= 2e-2 * nrm + 1e-9
            den = np.linalg.norm(ref.astype(np.float64)) + 1e-12
            w = max(w, float(np.linalg.norm(mine.reshape(-1).astype(np.float64) - ref.reshape(-1)) / den))
        worst.append(w)
    return worst


GRAD_CASES = ["blender_train", "ndc_train", "blender_disp", "coarse_only", "odd_sizes", "blender_det"]


@pytest.mark.parametrize("case", GRAD_CASES)
def test_render_rays_gradients(golden, case):
    """Manual backward of the oracle == autograd of the reference, per tensor, to 1e-4 relative (typical 1e-6).

    Rounds 1-2 allowed 5e-3 / 2e-2 here because (a) a ReLU whose pre-activation is ~0 lands on either side of the kink
    under a different fp32 GEMM summation order (MKL vs OpenBLAS), which switches that unit's gradient path, and (b) the
    fine model inherits sample_pdf's ill-conditioning.  Round 3 removes both causes instead of absorbing them: (a) the
    fixture stores the sign pattern the REFERENCE used for every unit within 1e-4 of the kink (tests/kinks.py) and the
    oracle's backward is evaluated with exactly that pattern -- the units that took the other side are counted (0..3 per
    case) and nothing else is exempted; (b) the fine pass runs on the reference's own merged depths (rng['z_fine'])."""
    import kinks
    g = dict(golden("g7_" + case))
    if int(g["F"]) > 0:
        g["rng_z_fine"] = g["mid_sort_out"]
    params, res = run_oracle_case(g, keep=True)
    masks, flips = [], 0
    for mi, tag in enumerate(("coarse", "fine")[: 2 if int(g["F"]) > 0 else 1]):
        m, f = kinks.reference_masks(g, mi, res["_aux"]["_" + tag][0])
        masks.append(m)
        flips += f
    if len(masks) == 1:
        masks.append(None)
    gs = O.render_rays_backward(params, res, _loss_grads(g, res), bool(g["white_back"]), masks=masks)
    w = grad_rel_errors(g, gs)
    print(f"[{case}] units on the other side of the kink in the reference: {flips}; worst relative gradient error {max(w):.2e}")
    assert flips <= 8, flips
    assert max(w) < 1e-4, w
    if int(g["F"]) > 0:
        for k in ("rgb_fine", "depth_fine", "opacity_fine"):
            assert np.abs(res[k] - g["out_" + k]).max() < 3e-6, k
    # ... and with its OWN pattern the oracle differs from the reference by exactly the flipped units' paths: no flip -> 1e-4
    if flips == 0:
        w0 = grad_rel_errors(g, O.render_rays_backward(params, res, _loss_grads(g, res), bool(g["white_back"])))
        assert max(w0) < 1e-4, w0


def test_siren_oracle_vs_reference(golden):
    """FiLMLayer + SemanticNeRF.forward_with_frequencies_phase_shifts (nerf.py:142-216)."""
    g = golden("g8_siren")
    p = synth.siren_params(3)
    assert sum(v.size for v in p.values()) == int(g["n_params"]) == 529156
    out = O.siren_forward(p, g["inp"], g["freq"], g["phase"], g["dirs"])
    assert np.abs(out - g["out"]).max() < 2e-6
    film = O.film_layer(p["network.1.layer.weight"], p["network.1.layer.bias"], g["film_in"],
                        g["freq"][:, :256], g["phase"][:, :256])
    assert np.abs(film - g["film_out"]).max() < 5e-7
    sig = O.siren_forward(p, g["inp"], g["freq"], g["phase"], g["dirs"], sigma_only=True)
    assert np.array_equal(sig[..., 0], out[..., 3])


def test_siren_backward_oracle_vs_reference_autograd(golden):
    """Manual backward of the FiLM-SIREN field vs the reference's autograd (all 22 parameter gradients of
    loss = sum(out * G) through the imported SemanticNeRF, tools/make_golden.py:g_siren)."""
    g, gg = golden("g8_siren"), golden("g8b_siren_grad")
    p = synth.siren_params(3)
    out, cache = O.siren_forward(p, g["inp"], g["freq"], g["phase"], g["dirs"], keep=True)
    assert np.abs(out - gg["out"]).max() < 2e-6
    grads, d_f, d_p = O.siren_backward(p, cache, gg["G"], cond=True)
    assert set(grads) == set(p)
    for k, v in grads.items():
        ref = gg["grad_" + k]
        assert v.shape == ref.shape, k
        rel = np.linalg.norm((v - ref).astype(np.float64)) / max(np.linalg.norm(ref.astype(np.float64)), 1e-30)
        assert rel < 2e-5, (k, rel)
    # round 3: the conditioning rows (three rows x 9 x 256) against the reference's autograd of the same loss
    for name, v in (("frequencies", d_f), ("phase_shifts", d_p)):
        ref = gg["cond_grad_" + name]
        assert v.shape == ref.shape == (3, 2304), name
        rel = np.linalg.norm((v - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64))
        assert rel < 2e-5, (name, rel)


# --------------------------------------------------------------------------- f2: loss + Adam
@pytest.mark.parametrize("tag", ["c64", "c1000", "coarse_only"])
def test_mse_loss_oracle_vs_reference(golden, tag):
    """losses.MSELoss (imported reference class) + its autograd vs the restatement."""
    g = golden("g17_loss_" + tag)
    fine = g["rgb_fine"] if "rgb_fine" in g else None
    o = O.mse_loss(g["rgb_coarse"], fine, g["targets"])
    np.testing.assert_allclose(o["loss"], g["loss"], rtol=2e-7)
    assert np.array_equal(o["g_coarse"], g["g_coarse"])
    if fine is not None:
        assert np.array_equal(o["g_fine"], g["g_fine"])


@pytest.mark.parametrize("tag,wd", [("wd0", 0.0), ("wd1e-4", 1e-4)])
def test_adam_oracle_vs_torch(golden, tag, wd):
    """12 torch.optim.Adam steps (lr 5e-4 halved at steps 4 and 8, eps 1e-8) vs the restatement."""
    g = golden("g17_adam_" + tag)
    ps = [g[f"p0_{i}"].copy() for i in range(3)]
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    lr = 5e-4
    for step in range(12):
        for i in range(3):
            grad = synth.hash_normal(ps[i].shape, 1000 + 10 * step + i) * np.float32(0.1 if step % 3 else 3.0)
            ps[i], ms[i], vs[i] = O.adam_step(ps[i], grad, ms[i], vs[i], step + 1, lr, weight_decay=wd)
        if step + 1 in (4, 8):
            lr *= 0.5
    for i in range(3):
        np.testing.assert_allclose(ps[i], g[f"p12_{i}"], rtol=2e-6, atol=1e-8)


# --------------------------------------------------------------------------- f1: ray generation (parity unpinned)
def _torch_ray_formulas(c2w, H, W, focal, ndc):
    """datasets/ray_utils.py:5-93 re-typed with torch CPU ops (the reference module itself needs kornia, absent here):
    this is what the reference executes, given create_meshgrid(H, W, False) = integer pixel coordinates."""
    import torch
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    directions = torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)
    c2w = torch.from_numpy(c2w)
    rays_d = directions @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:, 3].expand(rays_d.shape)
    rays_d, rays_o = rays_d.reshape(-1, 3), rays_o.reshape(-1, 3)
    if ndc:
        near = 1.0
        t = -(near + rays_o[..., 2]) / rays_d[..., 2]
        rays_o = rays_o + t[..., None] * rays_d
        ox_oz = rays_o[..., 0] / rays_o[..., 2]
        oy_oz = rays_o[..., 1] / rays_o[..., 2]
        o0 = -1. / (W / (2. * focal)) * ox_oz
        o1 = -1. / (H / (2. * focal)) * oy_oz
        o2 = 1. + 2. * near / rays_o[..., 2]
        d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - ox_oz)
        d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - oy_oz)
        d2 = 1 - o2
        rays_o, rays_d = torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)
    return directions.numpy(), rays_o.numpy(), rays_d.numpy()


@pytest.mark.parametrize("H,W,focal,ndc", [(5, 7, 6.3, False), (40, 40, 55.5555, False), (27, 36, 0.809 * 36, True),
                                           (1, 1, 1.0, False)])
def test_raygen_oracle_vs_torch_formulas(H, W, focal, ndc):
    c2w = synth._look_at_c2w(0.4, 1.1, 4.0311).astype(np.float32) if not ndc else \
        np.concatenate([np.eye(3, dtype=np.float32), np.array([[0.1], [-0.05], [0.2]], np.float32)], 1)
    dirs, o_t, d_t = _torch_ray_formulas(c2w, H, W, focal, ndc)
    assert np.array_equal(O.ray_directions(H, W, focal), dirs)                       # exact: integer grid, one sub + one div
    rays = O.generate_rays(c2w[None], H, W, focal, ndc=ndc)
    # the (H*W,3)@(3,3) product's summation order / fma use inside torch's matmul is unspecified: a few ulp
    np.testing.assert_allclose(rays[:, 3:6], d_t, rtol=4e-6, atol=1e-7)
    np.testing.assert_allclose(rays[:, 0:3], o_t, rtol=4e-6, atol=1e-7)
    assert np.array_equal(rays[:, 6:], np.broadcast_to(np.float32([0, 1] if ndc else [2, 6]), (H * W, 2)))


def test_raygen_oracle_matches_synthetic_ray_pool():
    """The bench / golden inputs (synth.blender_rays) are the same arithmetic: picking those pixels through the oracle
    reproduces them."""
    W = H = 400
    focal = np.float32(0.5 * W / np.tan(0.5 * synth.LEGO_ANGLE_X))
    uv = synth.hash_uniform((100, 2), 3 * 7919 + 11)
    c2ws = np.stack([synth._look_at_c2w(e, a, synth.LEGO_RADIUS) for e, a in
                     zip(uv[:, 0].astype(np.float64) * np.deg2rad(60.0), uv[:, 1].astype(np.float64) * 2 * np.pi)])
    pick = synth.hash_uniform((64, 3), 3 * 7919 + 13)
    v = np.minimum((pick[:, 0] * 100).astype(np.int64), 99)
    i = np.minimum((pick[:, 1] * W).astype(np.int64), W - 1)
    j = np.minimum((pick[:, 2] * H).astype(np.int64), H - 1)
    rays = O.generate_rays(c2ws.astype(np.float32), H, W, float(focal), pixel_index=v * H * W + j * W + i)
    np.testing.assert_allclose(rays, synth.blender_rays(64, 3), rtol=2e-6, atol=1e-7)


# --------------------------------------------------------------------------- f4: PFM files
def test_pfm_writer_matches_reference_bytes(golden, tmp_path):
    """io_utils.save_pfm writes byte-identical files to datasets/depth_utils.save_pfm; read_pfm round-trips them."""
    from nerf_siren_amd.io_utils import read_pfm, save_pfm
    g = golden("g18_pfm")
    for tag, scale in (("gray", 1), ("gray1", 1), ("color", 2.5)):
        f = tmp_path / f"{tag}.pfm"
        save_pfm(str(f), g["img_" + tag], scale=scale)
        assert np.array_equal(np.frombuffer(f.read_bytes(), np.uint8), g["bytes_" + tag]), tag
        if tag != "gray1":
            back, sc = read_pfm(str(f))
            assert np.array_equal(back, g["back_" + tag]) and sc == float(g["scale_" + tag])
            assert np.array_equal(back, g["img_" + tag])
    with pytest.raises(Exception):
        save_pfm(str(tmp_path / "bad.pfm"), np.zeros((2, 2), np.float64))


# --------------------------------------------------------------------------- split-bf16 arithmetic (opt-in math)
def test_bf16_three_way_split_is_exact():
    """The claim behind csrc/bf16x3_core.h: an fp32 value is EXACTLY the sum of three bf16 terms (round-to-nearest at
    each step), and the six kept products a1b1+a1b2+a2b1+a2b2+a1b3+a3b1 miss the exact product by <= ~2^-23 of it."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.integers(-20, 20, 200000),
                        [0.0, -0.0, 1.0, -1.0, 3.0e38, 2.0 ** -100, 1 + 2.0 ** -23, 255.99998, 16777215.0]]).astype(np.float32)
    # (values whose residuals would be subnormal, |x| < 2^-102, are outside the claim: the kernels flush them)
    p0, p1, p2 = O.split3(x)
    assert np.array_equal((p0.astype(np.float64) + p1.astype(np.float64)) + p2.astype(np.float64), x.astype(np.float64))
    for p in (p0, p1, p2):                                   # each term really is a bf16 number
        assert np.all(p.view(np.uint32) & 0xFFFF == 0)
    a, b = x[:100000], x[100000:200000]
    a1, a2, a3 = (t.astype(np.float64) for t in O.split3(a))
    b1, b2, b3 = (t.astype(np.float64) for t in O.split3(b))
    kept = a1 * b1 + a1 * b2 + a2 * b1 + a2 * b2 + a1 * b3 + a3 * b1
    exact = a.astype(np.float64) * b.astype(np.float64)
    ok = np.isfinite(exact) & (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    assert np.abs(kept[ok] / exact[ok] - 1).max() < 2.0 ** -22


# --------------------------------------------------------------------------- cpu_baseline: the torch-CPU restatement
@pytest.mark.parametrize("case", RENDER_CASES)
def test_torch_cpu_ref_vs_reference(golden, case):
    """oracle/torch_cpu_ref.py (what bench.py times as `cpu_baseline`) runs the reference's torch op sequence: on the
    G7 fixtures (outputs and autograd gradients of the imported reference, captured draws injected) it must reproduce
    the reference to rounding."""
    import torch
    from oracle import torch_cpu_ref as TR
    g = golden("g7_" + case)
    F, test_time = int(g["F"]), bool(g["test_time"])
    ps = [synth.nerf_params(1, sigma_bias=-1.0), synth.nerf_params(2, sigma_bias=0.5)]
    params = [{k: torch.from_numpy(v.copy()).requires_grad_(not test_time) for k, v in p.items()} for p in ps]
    rng = {k[4:]: torch.from_numpy(g[k]) for k in g if k.startswith("rng_")}
    res = TR.render_rays(params, torch.from_numpy(g["rays"]), int(g["S"]), bool(g["use_disp"]), float(g["perturb"]),
                         float(g["noise_std"]), F, 1024 * 32, bool(g["white_back"]), test_time, rng=rng)
    keys = [k[4:] for k in g if k.startswith("out_")]
    assert list(res.keys()) == keys
    for k in keys:
        np.testing.assert_allclose(res[k].detach().numpy(), g["out_" + k], rtol=0, atol=2e-6, err_msg=k)
    if test_time:
        return
    t = torch.from_numpy(g["target"])
    loss = ((res["rgb_coarse"] - t) ** 2).mean() + 0.1 * res["depth_coarse"].mean() + 0.3 * res["opacity_coarse"].mean()
    if F > 0:
        loss = loss + ((res["rgb_fine"] - t) ** 2).mean() + 0.2 * (res["depth_fine"] ** 2).mean() \
            - 0.1 * res["opacity_fine"].mean()
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-6)
    gs = [{k: v.grad.numpy() for k, v in p.items()} if mi == 0 or F > 0 else None for mi, p in enumerate(params)]
    w = grad_rel_errors(g, gs)
    assert max(w) < 1e-4, w


def test_torch_cpu_ref_siren_field_vs_reference(golden):
    """The FiLM-SIREN leg of oracle/torch_cpu_ref.py (bench.py's `cpu_baseline` on the headline workload): each of G8's three
    conditioning rows through siren_mlp's field interface reproduces the imported SemanticNeRF's outputs (fixture g8) and,
    under torch autograd, its parameter gradients (fixture g8b) to rounding."""
    import torch
    from oracle import torch_cpu_ref as TR
    g, gg = golden("g8_siren"), golden("g8b_siren_grad")
    base = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.siren_params(3).items()}
    loss = 0
    for r in range(3):
        P = dict(base, frequencies=torch.from_numpy(g["freq"][r:r + 1]), phase_shifts=torch.from_numpy(g["phase"][r:r + 1]))
        x = torch.zeros(41, 90)
        x[:, :3] = torch.from_numpy(g["inp"][r])
        x[:, 63:66] = torch.from_numpy(g["dirs"][r])
        out = TR.field_mlp(P, x)
        np.testing.assert_allclose(out.detach().numpy(), g["out"][r], rtol=0, atol=2e-6)
        np.testing.assert_allclose(TR.field_mlp(P, x[:, :63], sigma_only=True).detach().numpy(), g["out"][r][:, 3:], rtol=0,
                                   atol=2e-6)
        loss = loss + (out * torch.from_numpy(gg["G"][r])).sum()
    loss.backward()
    for k, v in base.items():
        ref = gg["grad_" + k]
        rel = np.linalg.norm((v.grad.numpy() - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64))
        assert rel < 1e-5, (k, rel)


def test_torch_cpu_ref_host_info_and_sample():
    from oracle import torch_cpu_ref as TR
    info = TR.host_info()
    assert info["physical_cores"] >= 1 and info["cpu_model"] and info["allowed_cpus"] >= 1
    ps = [synth.nerf_params(1, False), synth.nerf_params(2, False)]
    out = TR.timed_sample("infer", ps, lambda i: synth.blender_rays(64, seed=i), None, budget_s=0.2, n_rays=64, max_steps=2)
    assert out["steps"] >= 1 and out["ray_samples_per_s"] > 0 and "parallel_info" in out
    assert 1 <= out["threads"] <= out["nproc"] and str(out["threads"]) in out["gemm_gflops"]
    # the headline workload: FiLM-SIREN fields (conditioning rows are inputs, not optimised)
    sp = [dict(synth.siren_params(s), frequencies=synth.hash_normal((1, 2304), 10 + s), phase_shifts=synth.hash_normal((1, 2304), 20 + s))
          for s in (1, 2)]
    out = TR.timed_sample("train", sp, lambda i: synth.blender_rays(16, seed=i), lambda i: synth.hash_uniform((16, 3), i),
                          budget_s=0.2, n_rays=16, max_steps=1)
    assert out["steps"] == 1 and out["ray_samples_per_s"] > 0


# --------------------------------------------------------------------------- f1 / f3 pinned by the reference's own code
def test_ray_utils_oracle_vs_reference(golden):
    """datasets/ray_utils.py:5-93 run from the reference file (tools/make_golden.py:g_ray_utils; kornia 0.2.0's
    create_meshgrid supplied from its published definition) vs the restatement."""
    g = golden("g20_ray_utils")
    for t in ("b", "l"):
        H, W, focal = int(g[t + "_H"]), int(g[t + "_W"]), float(g[t + "_focal"])
        d = O.ray_directions(H, W, focal)
        assert np.array_equal(d, g[t + "_directions"])
        o, rd = O.get_rays(d, g[t + "_c2w"])
        assert np.array_equal(o, g[t + "_rays_o"])
        np.testing.assert_allclose(rd, g[t + "_rays_d"], rtol=0, atol=1.2e-7)       # matmul + norm order: 1 ulp
    no, nd = O.ndc_rays(int(g["l_H"]), int(g["l_W"]), float(g["l_focal"]), 1.0, g["l_rays_o"], g["l_rays_d"])
    np.testing.assert_allclose(no, g["l_ndc_o"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(nd, g["l_ndc_d"], rtol=2e-6, atol=2e-7)


def test_grid_queries_oracle_vs_reference(golden):
    """extract_color_mesh.py:117-140 (grid order, sigma clamp), extract_color_mesh_eg3d.py:72-94 (create_samples) and
    extract_mesh.ipynb cell 7 (.vol records), all executed from the reference's own text (tools/make_golden.py:g_grid)."""
    g = golden("g21_grids")
    N = int(g["mesh_N"])
    pts = O.grid_points(N, *[tuple(r) for r in g["mesh_ranges"]])
    assert np.array_equal(pts, g["mesh_xyz"])
    assert np.array_equal(np.maximum(g["mesh_sigma_in"], 0).reshape(N, N, N), g["mesh_sigma_grid"])
    for n in (6, 32):
        smp, origin, vs = O.create_samples(n, [0, 0, 0], float(g[f"cs{n}_cube"]))
        assert np.array_equal(origin, g[f"cs{n}_origin"]) and vs == float(g[f"cs{n}_voxel_size"])
        assert smp.shape == g[f"cs{n}_samples"].shape
        np.testing.assert_allclose(smp, g[f"cs{n}_samples"], rtol=0, atol=2.4e-7)
    vol = O.pack_vol(g["vol_rgbsigma"], int(g["vol_N"]), float(g["vol_extent"]))
    assert vol.dtype == np.uint32 and np.array_equal(vol, g["vol_records"])


def test_philox_known_answers():
    """Random123's known-answer vectors for philox4x32-10 (kat_vectors): the generator behind the perf-mode draws."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
        assert [int(v) for v in got] == want
    a = O.render_draws(1234, 7, (1000, 1000, 999, 2001))
    assert [len(v) for v in a] == [1000, 1000, 999, 2001]
    assert all(0 <= v.min() and v.max() < 1 for v in (a[0], a[2]))
    assert abs(a[3].mean()) < 0.1 and abs(a[3].std() - 1) < 0.1


def test_png_gif_round_trip(tmp_path):
    """eval.py:185-194 output formats (imageio.imwrite png, imageio.mimsave gif at 30 fps), written with the standard
    library only: a decoder gets back exactly the pixels that were written (PNG) / their 3-3-2 palette images (GIF)."""
    from nerf_siren_amd import io_utils as IO
    rng = np.random.default_rng(0)
    pred = rng.random((23, 17, 3)).astype(np.float32)
    img = IO.to_uint8(pred)
    assert img.dtype == np.uint8 and np.array_equal(img, (pred * 255).astype(np.uint8))
    for a in (img, img[:, :, 0], rng.integers(0, 256, (9, 31, 4), dtype=np.uint8)):
        f = str(tmp_path / "a.png")
        IO.imwrite_png(f, a)
        assert np.array_equal(IO.imread_png(f), a)
        assert open(f, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    frames = [IO.to_uint8(rng.random((40, 56, 3))) for _ in range(3)]
    yy, xx = np.mgrid[0:40, 0:56]
    frames.append(np.stack([xx * 4 % 256, yy * 6 % 256, (xx + yy) * 2 % 256], -1).astype(np.uint8))   # compressible
    f = str(tmp_path / "scene.gif")
    IO.mimsave_gif(f, frames, fps=30)
    got, delay = IO.mimread_gif(f)
    assert delay == 3 and len(got) == len(frames)                 # round(100 / 30) hundredths of a second
    for a, b in zip(frames, got):
        assert np.array_equal(IO.PALETTE_332[IO._palette_332(a)], b)
    big = [rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)]    # forces LZW dictionary resets
    IO.mimsave_gif(f, big)
    assert np.array_equal(IO.PALETTE_332[IO._palette_332(big[0])], IO.mimread_gif(f)[0][0])
    with pytest.raises(TypeError):
        IO.imwrite_png(str(tmp_path / "b.png"), pred)
