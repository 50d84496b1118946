"""Pins oracle/flowdec_oracle.py against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_err
from oracle import flowdec_oracle as O


def test_window_and_stft():
    g = load_golden("g1_stft.npz")
    assert np.allclose(O.hann_sym(1534, np.float32), g["window"], atol=1e-7)
    S = O.stft(g["y"])
    assert S.shape == g["stft"].shape == (2, 1, 768, 13)
    assert rel_err(S, g["stft"]) < 2e-6
    assert O.num_frames(4800) == 13 and O.num_frames(96000) == 251 and O.num_frames(48000) == 126


def test_frame_indexing_exact():
    # frame j covers padded samples [384 j, 384 j + 1534) with reflect padding 767 (integer exact)
    y = np.arange(5000, dtype=np.float64)[None, :]
    fr = O.frame_signal(y)
    assert fr.shape == (1, 14, 1534)
    assert fr[0, 0, 767] == 0 and fr[0, 0, 0] == 767 and fr[0, 0, 766] == 1
    assert fr[0, 3, 0] == 3 * 384 - 767
    assert fr[0, 13, 1533] == 2 * 4999 - (13 * 384 + 1533 - 767)


def test_compress_and_invert():
    g = load_golden("g1_stft.npz")
    C = O.compress(g["stft"])
    assert rel_err(C, g["compressed"]) < 2e-6
    D = O.decompress(g["compressed"])
    assert rel_err(D, g["decompressed"]) < 5e-6


def test_istft():
    g = load_golden("g1_stft.npz")
    rt = O.istft(g["decompressed"], 4800)
    assert rel_err(rt, g["roundtrip"]) < 5e-6
    assert rel_err(rt, g["y"]) < 1e-4  # invertibility claim (feature_extractors.py:21-22)
    a = O.istft(O.decompress(g["Z"]), 3000)
    b = O.istft(O.decompress(g["Z"]), 3072)
    assert a.shape == (2, 1, 3000) and b.shape == (2, 1, 3072)
    assert rel_err(a, g["istft_len3000"]) < 5e-6
    assert rel_err(b, g["istft_len3072"]) < 5e-6


def test_pad_and_normalize():
    g = load_golden("g3_pad_norm.npz")
    yn, nf = O.normalize_noisy(g["y"])
    assert np.array_equal(nf, g["normfac"])       # silent clip -> normfac 1 (other.py:77)
    assert nf[1, 0, 0] == 1.0
    assert np.array_equal(yn, g["y_norm"])
    P, T = O.pad_spec(g["spec"])
    assert P.shape[-1] == int(g["padded_T"]) == 64 and T == int(g["orig_T"]) == 13
    assert np.array_equal(P[..., :T], g["spec"]) and not P[..., T:].any()
    assert [O.padded_frames(t) for t in (1, 64, 65, 126, 251, 501)] == [64, 64, 128, 128, 256, 512]


def test_upfirdn2d():
    g = load_golden("g4_upfirdn2d.npz")
    for nm in ("a", "b"):
        x = g["x" + nm]
        assert rel_err(O.upsample_2d(x), g["up_" + nm]) < 1e-6
        assert rel_err(O.downsample_2d(x), g["down_" + nm]) < 1e-6
        # polyphase closed forms (what the HIP kernels implement)
        assert rel_err(O.fir_up2_polyphase(x), g["up_" + nm]) < 1e-6
        assert rel_err(O.fir_down2_polyphase(x), g["down_" + nm]) < 1e-6
    # generic call: up (2,3) down (1,2) pad x(1,2) y(0,1): oracle's upfirdn2d is symmetric in x/y,
    # so emulate the asymmetric reference call with two separable passes is not possible ->
    # check the symmetric special case against a direct evaluation instead.
    k = O.setup_fir_kernel((1, 3, 3, 1))
    x = g["xa"]
    ref = O.upfirdn2d(x, k * 4, up=2, pad=(2, 1))
    assert rel_err(ref, g["up_a"]) < 1e-6


def test_groupnorm_silu():
    g = load_golden("g5_groupnorm_silu.npz")
    for C in (64, 256, 320, 384, 512):
        o = O.silu(O.group_norm(g[f"x{C}"], O.gn_groups(C), g[f"gamma{C}"], g[f"beta{C}"]))
        assert rel_err(o, g[f"out{C}"]) < 2e-6, C
    assert [O.gn_groups(c) for c in (64, 128, 256, 320, 384, 512)] == [16, 32, 32, 32, 32, 32]


@pytest.mark.parametrize("name", list(O.RESBLOCK_CASES))
def test_resblock(name):
    g = load_golden("g6_resblock.npz")
    seed, ci, co, up, down = O.RESBLOCK_CASES[name]
    p = O.random_resblock_params(seed, ci, co, has_conv2=(ci != co or up or down))
    net = O.NCSNppOracle({f"all_modules.0.{k}": v for k, v in p.items()}, prefix="")
    o = net.resblock(0, dict(cin=ci, cout=co, up=up, down=down), g[f"{name}_x"], g[f"{name}_temb"])
    assert o.shape == g[f"{name}_out"].shape
    assert rel_err(o, g[f"{name}_out"]) < 5e-6


def test_time_embedding_and_ncsnpp_nf8():
    g = load_golden("g8_ncsnpp_nf8.npz")
    net = O.NCSNppOracle(O.random_state_dict(seed=int(g["seed"]), nf=8), nf=8)
    assert rel_err(net.time_embedding(g["t"]), g["temb"]) < 2e-5
    o = net.forward(g["x"], g["y"], np.array([0.25], np.float32))
    assert o.shape == (2, 1, 768, 64)
    assert rel_err(o, g["out_t025"]) < 2e-5
    o2 = net.forward(g["x"], g["y"], np.array([0.1, 0.9], np.float32))
    assert rel_err(o2, g["out_t01_09"]) < 2e-5


def test_ncsnpp_full_width():
    g = load_golden("g10_ncsnpp_nf64.npz")
    net = O.NCSNppOracle(O.random_state_dict(seed=int(g["seed"]), nf=64), nf=64)
    o = net.forward(g["x"], g["y"], np.array([0.5], np.float32))
    assert rel_err(o, g["out"]) < 5e-5


def test_state_dict_manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        ref = json.load(f)
    ours = {k: list(v) for k, v in O.state_dict_manifest(nf=64).items()}
    bb = {k: v for k, v in ref.items() if k.startswith("backbone.")}
    assert ours == bb
    assert ref["feature_extractor.complex_stft.window"] == [1534]
    assert ref["sigma_y"] == [768, 1] and ref["sigma_x"] == []
    n_params = sum(int(np.prod(v)) for v in bb.values())
    assert n_params == 23703704


def test_sigma_y_curve():
    g = load_golden("g12_sigma_y.npz")
    for nm in ("75m", "25s"):
        c = O.sigma_y_curve(g[nm + "_raw"], 1.0, 3.0)
        assert c.shape == (768, 1) and c.dtype == np.float64
        assert np.allclose(c, g[nm], rtol=1e-12, atol=0)


def test_t_span_matches_torch_linspace():
    import torch
    for N in (1, 2, 3, 5, 6, 7, 25, 32, 50):
        assert np.array_equal(O.t_span_linspace(N), torch.linspace(0, 1, N + 1).numpy()), N


@pytest.mark.parametrize("solver,N", [("euler", 6), ("midpoint", 3), ("heun2", 3), ("heun2_eulerlast", 3)])
def test_enhance_nf8(solver, N):
    g = load_golden("g9_enhance_nf8.npz")
    net = O.NCSNppOracle(O.random_state_dict(seed=int(g["seed"]), nf=8), nf=8)
    x = O.enhance(net, g["y"], g["noise"], g["sigma_y"], N=N, solver=solver)
    ref = g[f"{solver}_N{N}"]
    assert x.shape == ref.shape == (2, 1, 24000)
    assert rel_err(x, ref) < 1e-4
    assert O.solver_nfe(solver, N) == {"euler": 6, "midpoint": 6, "heun2": 6, "heun2_eulerlast": 5}[solver]


def test_enhance_traj_and_1d():
    g = load_golden("g9_enhance_nf8.npz")
    net = O.NCSNppOracle(O.random_state_dict(seed=int(g["seed"]), nf=8), nf=8)
    traj, waves = O.enhance(net, g["y"], g["noise"], g["sigma_y"], N=2, solver="euler", return_traj=True)
    norms = np.array([np.sqrt((np.abs(X.astype(np.complex128)) ** 2).sum()) for X in traj])
    assert np.allclose(norms, g["traj_feat_norms"], rtol=1e-4)
    assert rel_err(waves[-1], g["traj_last_wave"]) < 1e-4
    x1 = O.enhance(net, g["y"][0:1], g["noise"][0:1], g["sigma_y"], N=2, solver="euler")
    assert rel_err(x1[0, 0], g["euler_N2_1d"]) < 1e-4


# ---- (f3) ScoreDec / regression baselines ---------------------------------------------------------------
SCORE_CASES = {
    "rd_ald_N3": dict(N=3, predictor="reverse_diffusion", corrector="ald", corrector_steps=1, snr=0.5),
    "rd_none_N4": dict(N=4, predictor="reverse_diffusion", corrector="none"),
    "em_ald2_N2": dict(N=2, predictor="euler_maruyama", corrector="ald", corrector_steps=2, snr=0.33),
    "rd_ald_N3_nodenoise": dict(N=3, predictor="reverse_diffusion", corrector="ald", corrector_steps=1, snr=0.5, denoise=False),
}


def score_net(g):
    sd = O.random_state_dict(seed=int(g["weight_seed"]), nf=8)
    sd["backbone.output_layer.weight"] = sd["backbone.output_layer.weight"] * np.float32(g["out_scale"])
    return O.NCSNppOracle(sd, nf=8), sd


def test_ouve_closed_forms_and_timesteps():
    g = load_golden("g13_score_nf8.npz")
    sde = O.OUVE(*[float(v) for v in g["sde"]])
    for t, s, d in zip(g["std_t"], g["std"], g["diffusion"]):
        assert abs(sde.std(t) - s) <= 2e-7 * abs(s) + 1e-9
        assert abs(sde.diffusion(t) - d) <= 2e-7 * abs(d)
    assert np.array_equal(O.linspace_f32(1.0, 3e-2, 30), g["timesteps_N30"])        # bit-exact like t_span
    assert np.array_equal(O.linspace_f32(0.0, 1.0, 7), O.t_span_linspace(6))


@pytest.mark.parametrize("case,item", [("rd_ald_N3", 1), ("rd_none_N4", 0), ("em_ald2_N2", 1)])
def test_score_sampler_matches_reference(case, item):
    """One batch item per case keeps the CPU suite short (items are independent end to end); the GPU tests check both."""
    g = load_golden("g13_score_nf8.npz")
    net, _ = score_net(g)
    y = g["y"][item:item + 1]
    Tp = O.padded_frames(O.num_frames(y.shape[-1]))
    sde = O.OUVE(*[float(v) for v in g["sde"]])
    noises = (z[item:item + 1] for z in O.seeded_noises(int(g["noise_seed"]), (2, 1, 768, Tp)))
    kw = dict(SCORE_CASES[case])
    Y, info = O.preprocess(y)
    (x_mean, x), nfe = O.score_pc_sample(net, Y, noises, sde, kw.pop("N"), t_eps=float(g["t_eps"]), denoise="both", **kw)
    assert nfe == O.score_noise_count(SCORE_CASES[case]["N"], **{k: v for k, v in kw.items() if k != "snr"}) - 1
    assert rel_err(O.postprocess(x_mean, info), g[case][item:item + 1]) < 2e-4
    if case + "_nodenoise" in g:
        assert rel_err(O.postprocess(x, info), g[case + "_nodenoise"][item:item + 1]) < 2e-4


def test_regression_matches_reference():
    g = load_golden("g13_score_nf8.npz")
    net, _ = score_net(g)
    assert rel_err(O.regression_enhance(net, g["y"][:1]), g["regression"][:1]) < 1e-4


def test_score_ode_sampler_matches_reference():
    """Black-box probability-flow sampler (scipy RK45 on the host) incl. the final denoising step and the NFE count."""
    g = load_golden("g13_score_nf8.npz")
    net, _ = score_net(g)
    y = g["y"][:1]
    Tp = O.padded_frames(O.num_frames(y.shape[-1]))
    sde = O.OUVE(*[float(v) for v in g["sde"]])
    z0 = next(O.seeded_noises(int(g["noise_seed"]), (1, 1, 768, Tp)))
    out, nfe = O.score_ode_enhance(net, y, z0, sde, N=30, t_eps=float(g["t_eps"]), rtol=1e-3, atol=1e-3)
    assert nfe == int(g["ode_rk45_nfe"])
    assert rel_err(out, g["ode_rk45"]) < 1e-3


def test_dopri5_driver_against_scipy():
    """The restated adaptive Dormand-Prince driver (unpinned w.r.t. torchdyn) is at least a correct DP5(4): it integrates a
    stiff-ish linear complex system and a nonlinear one to tolerance, agrees with scipy's RK45, hits the checkpoints of
    t_span exactly and reuses the last stage (6 evaluations per attempted step + 2)."""
    from scipy import integrate
    rng = np.random.default_rng(5)
    A = (rng.standard_normal((6, 6)) + 1j * rng.standard_normal((6, 6))).astype(np.complex64) - 3 * np.eye(6, dtype=np.complex64)
    x0 = (rng.standard_normal(6) + 1j * rng.standard_normal(6)).astype(np.complex64)
    calls = [0]

    def f(t, x):
        calls[0] += 1
        return (A @ x + np.float32(np.sin(3 * t)) * x * np.abs(x)).astype(np.complex64)
    ts = O.linspace_f32(0.0, 1.0, 9)
    traj, nfe = O.odeint_dopri5(f, x0, ts, atol=1e-6, rtol=1e-6, return_traj=True)
    assert nfe == calls[0] and len(traj) == 9 and (nfe - 2) % 6 == 0
    sol = integrate.solve_ivp(lambda t, x: A.astype(np.complex128) @ x + np.sin(3 * t) * x * np.abs(x), (0.0, 1.0), x0.astype(np.complex128),
                              method="RK45", rtol=1e-10, atol=1e-12, t_eval=[float(v) for v in ts])
    for i in range(9):
        assert rel_err(traj[i], sol.y[:, i]) < 2e-5
    # looser tolerance -> fewer evaluations, still inside that tolerance
    x1, nfe1 = O.odeint_dopri5(f, x0, O.linspace_f32(0.0, 1.0, 2), atol=1e-3, rtol=1e-3)
    assert nfe1 < nfe and rel_err(x1, sol.y[:, -1]) < 5e-3


def test_torch_oracle_matches_numpy_oracle_and_golden():
    """oracle/flowdec_oracle_torch.py (the oracle's graph + solver on torch CPU kernels: bench.py's same-box `cpu_baseline`) is
    pinned like the NumPy oracle: forward against the reference's NCSNpp (G8), enhance against the reference's
    FlowModel.enhance (G9, three solvers), the polyphase FIR pair against upfirdn2d (G4) -- at the fp32 tolerances."""
    import torch
    from oracle import flowdec_oracle_torch as OT
    g = load_golden("g8_ncsnpp_nf8.npz")
    net = OT.NCSNppTorchCPU(O.random_state_dict(seed=int(g["seed"]), nf=8), nf=8)
    out = net.forward(torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), torch.tensor([0.25])).numpy()
    assert rel_err(out, g["out_t025"]) < 2e-5
    out2 = net.forward(torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), torch.from_numpy(g["t"]).float().reshape(-1)).numpy() if g["t"].size == 2 else None
    if out2 is not None:
        assert rel_err(out2, g["out_t01_09"]) < 2e-5
    x = np.random.default_rng(0).standard_normal((2, 3, 12, 8)).astype(np.float32)
    assert rel_err(OT.upsample_2d(torch.from_numpy(x)).numpy(), O.upsample_2d(x)) < 1e-6
    assert rel_err(OT.downsample_2d(torch.from_numpy(x)).numpy(), O.downsample_2d(x)) < 1e-6
    g9 = load_golden("g9_enhance_nf8.npz")
    net = OT.NCSNppTorchCPU(O.random_state_dict(seed=int(g9["seed"]), nf=8), nf=8)
    for solver, N in (("euler", 6), ("midpoint", 3), ("heun2", 3)):
        got = OT.enhance(net, g9["y"], g9["noise"], g9["sigma_y"], N=N, solver=solver)
        assert rel_err(got, g9[f"{solver}_N{N}"]) < 1e-4, solver


def test_adaptive_tableaus_order_conditions_and_tsit5_against_scipy():
    """Both embedded pairs of the adaptive driver ('dopri5', and 'tsit5' = torchdyn's NeuralODE default) satisfy every order condition
    up to order 5 (the Tsitouras coefficients are typed from the publication: this is their check), their error weights those up to
    order 4 with weight sum 0, the solution converges with order >= 5; 'tsit5' integrates the nonlinear complex test system to
    tolerance, agrees with scipy's RK45, hits the checkpoints exactly and reuses its last stage (6 evaluations per step + 2)."""
    from scipy import integrate
    for name, (C_, A_, E_) in O.ADAPTIVE_TABLEAUS.items():
        c = np.array(C_, np.float64)
        A = np.zeros((7, 7))
        for i in range(7):
            A[i, :len(A_[i])] = A_[i]
        b, e = A[6].copy(), np.array(E_, np.float64)
        assert np.abs(A.sum(1) - c).max() < 1e-15, name                                     # row sums
        one = np.ones(7)
        conds = [(b @ one, 1), (b @ c, 1 / 2), (b @ c ** 2, 1 / 3), (b @ A @ c, 1 / 6), (b @ c ** 3, 1 / 4), ((b * c) @ A @ c, 1 / 8), (b @ A @ c ** 2, 1 / 12),
                 (b @ A @ A @ c, 1 / 24), (b @ c ** 4, 1 / 5), ((b * c ** 2) @ A @ c, 1 / 10), ((b * c) @ A @ c ** 2, 1 / 15), ((b * c) @ A @ A @ c, 1 / 30),
                 (b @ (A @ c) ** 2, 1 / 20), (b @ A @ c ** 3, 1 / 20), (b @ A @ (c * (A @ c)), 1 / 40), (b @ A @ A @ c ** 2, 1 / 60), (b @ A @ A @ A @ c, 1 / 120)]
        assert max(abs(v - w) for v, w in conds) < 1e-15, (name, [abs(v - w) for v, w in conds])
        econds = [e @ one, e @ c, e @ c ** 2, e @ A @ c, e @ c ** 3, (e * c) @ A @ c, e @ A @ c ** 2, e @ A @ A @ c]
        assert max(abs(v) for v in econds) < 1e-15, (name, econds)                        # the embedded 4th-order solution
        assert A[6, 6] == 0 and c[6] == 1.0                                                 # FSAL: stage 7 is f at the 5th-order solution

        def solve(n):
            x, h = np.array([1.0, 0.0]), 2.0 / n
            f = lambda t, x_: np.array([x_[1], -x_[0]]) * (1 + 0.3 * np.sin(t))
            for i in range(n):
                k = [f(i * h, x)]
                for s in range(1, 7):
                    k.append(f(i * h + c[s] * h, x + h * sum(A[s, j] * k[j] for j in range(s))))
                x = x + h * sum(b[j] * k[j] for j in range(7))
            return x
        ref = solve(2048)
        errs = [np.abs(solve(n) - ref).max() for n in (8, 16, 32)]
        assert all(np.log2(a / b_) > 4.7 for a, b_ in zip(errs, errs[1:])), (name, errs)
    rng = np.random.default_rng(5)
    A = (rng.standard_normal((6, 6)) + 1j * rng.standard_normal((6, 6))).astype(np.complex64) - 3 * np.eye(6, dtype=np.complex64)
    x0 = (rng.standard_normal(6) + 1j * rng.standard_normal(6)).astype(np.complex64)
    f = lambda t, x: (A @ x + np.float32(np.sin(3 * t)) * x * np.abs(x)).astype(np.complex64)
    ts = O.linspace_f32(0.0, 1.0, 9)
    traj, nfe = O.odeint_adaptive(f, x0, ts, "tsit5", atol=1e-6, rtol=1e-6, return_traj=True)
    assert len(traj) == 9 and (nfe - 2) % 6 == 0
    sol = integrate.solve_ivp(lambda t, x: A.astype(np.complex128) @ x + np.sin(3 * t) * x * np.abs(x), (0.0, 1.0), x0.astype(np.complex128),
                              method="RK45", rtol=1e-10, atol=1e-12, t_eval=[float(v) for v in ts])
    for i in range(9):
        assert rel_err(traj[i], sol.y[:, i]) < 2e-5
    xd, nd = O.odeint_adaptive(f, x0, ts, "dopri5", atol=1e-6, rtol=1e-6)
    assert rel_err(traj[-1], xd) < 2e-5
