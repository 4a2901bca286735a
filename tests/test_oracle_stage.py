"""Stage::NonZero (voices with GAMMA != 0, LSP spectra, MGLSA filter) in the oracle.
PARITY UNPINNED: no reference test sets a stage and no such voice exists in this container, so the
restatement (oracle/jbo_hot.c, after src/vocoder/{lsp.rs, generalized.rs, cepstrum.rs:69-103, mglsa.rs,
mod.rs:90-107,142-176}) is held by analytic identities only.  No GPU."""
import numpy as np
import pytest

from oracle import oracle as O


def _lsp(n, rng):
    """n strictly increasing line spectral frequencies in (0, pi): an even grid, each point moved by
    less than a third of the spacing (neighbours stay at least a third of it apart)"""
    h = np.pi / (n + 1)
    return h * (np.arange(1, n + 1) + rng.uniform(-0.33, 0.33, n))


@pytest.mark.parametrize("m", [2, 5, 8, 24, 35])
def test_lsp2lpc_roots_are_the_line_spectral_frequencies(m):
    """lsp2lpc (lsp.rs:27-94) builds A(z) of order m from m frequencies: A = (P + Q) / 2 with P, Q the
    symmetric / antisymmetric polynomials whose unit-circle roots are the even- / odd-indexed
    frequencies.  Recover P and Q from the returned A and evaluate them at the frequencies."""
    rng = np.random.default_rng(m)
    w = _lsp(m, rng)
    a = O.lsp2lpc(w)  # [1, a_1 .. a_m]
    assert a[0] == 1.0 and len(a) == m + 1
    A = np.concatenate([a, [0.0]])            # degree m + 1 holder
    Ar = A[::-1]                              # z^-(m+1) A(1/z)
    P, Q = A + Ar, A - Ar
    z = lambda om: np.exp(-1j * om * np.arange(m + 2))
    for i, om in enumerate(w):
        val = abs(np.dot(P if i % 2 == 0 else Q, z(om)))
        assert val < 1e-6 * np.sum(np.abs(P)), (m, i, val)  # a wrong root gives O(1)


@pytest.mark.parametrize("gamma", [-1.0, -0.5, -1.0 / 3.0, 0.0])
def test_gnorm_ignorm_are_inverse(gamma):
    """generalized.rs:6-38"""
    rng = np.random.default_rng(3)
    c = rng.normal(0, 0.3, 20)
    c[0] = 0.4
    back = O.ignorm(O.gnorm(c, gamma), gamma)
    np.testing.assert_allclose(back, c, rtol=1e-12, atol=1e-13)
    g = O.gnorm(c, gamma)
    if gamma != 0.0:
        k = 1.0 + gamma * c[0]
        assert abs(g[0] - k ** (1.0 / gamma)) < 1e-14 and np.allclose(g[1:], c[1:] / k)
    else:
        assert abs(g[0] - np.exp(c[0])) < 1e-15 and np.array_equal(g[1:], c[1:])


def test_gc2gc_identity_and_spectrum_preservation():
    """gc2gc (cepstrum.rs:69-92) re-expresses the same spectrum with another gamma.  Same gamma: the
    identity.  gamma -1 -> 0 of a gain-normalised all-pole model 1/A(z): the LPC-to-cepstrum recursion,
    checked against log|1/A| on the unit circle."""
    rng = np.random.default_rng(4)
    c = rng.normal(0, 0.1, 12)
    same = O.gc2gc(c, -0.5, 11, -0.5)
    np.testing.assert_allclose(same, c, rtol=1e-12, atol=1e-14)
    # stable all-pole model: roots inside the unit circle
    r = rng.uniform(0.3, 0.8, 4) * np.exp(1j * rng.uniform(0.3, 2.8, 4))
    a = np.real(np.poly(np.concatenate([r, r.conj()])))     # 1 + a1 z^-1 + ... (order 8)
    c1 = np.concatenate([[0.0], -a[1:]])                    # gamma = -1: (1 + gamma * C(z))^(1/gamma) = 1 / A(z)
    cep = O.gc2gc(c1, -1.0, 200, 0.0)
    om = np.linspace(0, np.pi, 64)
    A = np.array([np.dot(a, np.exp(-1j * w * np.arange(len(a)))) for w in om])
    C = np.array([np.real(np.dot(cep[1:], np.exp(-1j * w * np.arange(1, len(cep))))) for w in om])
    np.testing.assert_allclose(C, -np.log(np.abs(A)), atol=1e-8)


def test_mgc2mgc_same_alpha_same_gamma_is_identity_up_to_normalisation():
    """mgc2mgc (cepstrum.rs:94-102) with unchanged alpha and gamma: gnorm -> gc2gc(same) -> ignorm."""
    rng = np.random.default_rng(5)
    c = rng.normal(0, 0.1, 10)
    c[0] = 0.7
    out = O.mgc2mgc(c, 0.55, -0.5, 9, 0.55, -0.5)
    np.testing.assert_allclose(out, c, rtol=1e-11, atol=1e-13)


def test_stage1_mglsa_is_the_warped_all_pole_filter():
    """One MGLSA stage (mglsa.rs:23-41) with alpha = 0 is the all-pole recursion
    y[n] = x[n] - sum_k c[k+1] y'[n-k] ... with d[0] = y: compare with a direct difference equation."""
    rng = np.random.default_rng(6)
    n = 8
    c = np.concatenate([[1.0], rng.normal(0, 0.08, n - 1)])
    d = np.zeros((1, n))
    x = rng.normal(0, 1, 300)
    got = np.array([O.mglsa_df(d, float(v), 0.0, c) for v in x])
    # alpha = 0: d[i] holds the outputs delayed by i + 1 samples; y[n] = x[n] - sum_{i>=1} c[i] y[n-i]
    y = np.zeros(len(x))
    for t in range(len(x)):
        acc = x[t]
        for i in range(1, n):
            if t - i >= 0:
                acc -= c[i] * y[t - i]
        y[t] = acc
    np.testing.assert_allclose(got, y, rtol=1e-10, atol=1e-12)


def test_check_lsp_stability_and_postfilter_properties():
    """check_lsp_stability (lsp.rs:141-165) keeps an already well separated set; postfilter_lsp (lsp.rs:113-139)
    leaves the end entries, keeps the order and re-normalises the energy (lsp2en) through the gain."""
    rng = np.random.default_rng(7)
    n = 25
    lsp = np.concatenate([[0.5], _lsp(n - 1, rng)])
    assert np.array_equal(O.check_lsp_stability(lsp), lsp)
    squeezed = lsp.copy()
    squeezed[10] = squeezed[9] + 1e-4
    fixed = O.check_lsp_stability(squeezed)
    assert fixed[10] - fixed[9] > 1e-3
    for log_gain in (False, True):
        pf = O.postfilter_lsp(lsp, 0.55, log_gain, 2, 0.3)
        assert np.array_equal(pf[1:2], lsp[1:2]) and pf[-1] == lsp[-1]
        assert np.all(np.diff(pf[1:]) > 0)
        e1 = np.sum(O.lsp2mgc(lsp, 0.55, log_gain, 2) ** 2)
        e2 = np.sum(O.lsp2mgc(pf, 0.55, log_gain, 2) ** 2)
        assert np.isfinite(e1) and np.isfinite(e2)
        assert np.array_equal(O.postfilter_lsp(lsp, 0.55, log_gain, 2, 0.0), lsp)


def test_vocoder_stage_runs_and_is_linear_in_volume():
    """The whole NonZero loop: finite output, linear in volume, first frame un-filtered (beta changes
    every frame but the coefficients the first frame STARTS from, mod.rs:92-106)."""
    rng = np.random.default_rng(8)
    T, n = 12, 25
    base = np.concatenate([[0.02], _lsp(n - 1, rng)])
    mcp = base[None, :] + rng.normal(0, 0.003, (T, n))
    mcp[:, 1:] = np.sort(mcp[:, 1:], axis=1)
    lf0 = np.where(rng.random(T) < 0.7, np.log(120.0), -1e10)
    lpf = np.zeros((T, 31))
    lpf[:, 15] = 1.0
    a = O.vocoder(48000, 240, 0.55, 1.0, lf0, mcp, lpf, stage=2)
    b = O.vocoder(48000, 240, 0.55, 2.0, lf0, mcp, lpf, stage=2)
    assert np.all(np.isfinite(a)) and np.max(np.abs(a)) > 0
    np.testing.assert_allclose(b, 2.0 * a, rtol=1e-13)
    c0 = O.stage_coefficients(mcp[0], 0.55, 0.4, False, 2, filtered=False)
    c1 = O.stage_coefficients(mcp[0], 0.55, 0.4, False, 2, filtered=True)
    c2 = O.stage_coefficients(mcp[0], 0.55, 0.0, False, 2, filtered=False)
    assert np.array_equal(c0, c2) and not np.array_equal(c0, c1)


def test_lsp_conversion_is_ill_conditioned():
    """Why the GPU tests of this stage compare coefficients at 1e-6 and audio at 1e-4: the reference's
    lsp2lpc (lsp.rs:58-86) expands the product of its second-order sections as a cascade whose partial
    products (all low-frequency roots first) reach ~1e10 before the high-frequency sections cancel them,
    so ONE ulp of relative change in the input moves the LPC by ~1e-9 and the filter coefficients by
    ~1e-8 -- a property of the algorithm as written (two libm's cosines differ by that much)."""
    L = 35
    h = np.pi / L
    lsp = np.concatenate([[0.05], h * np.arange(1, L)])
    a = O.lsp2lpc(lsp)
    c = O.stage_coefficients(lsp, 0.55, 0.0, False, 2)
    rng = np.random.default_rng(0)
    worst_a = worst_c = 0.0
    for _ in range(20):
        l2 = lsp * (1.0 + 2.2e-16 * rng.choice([-1.0, 1.0], L))
        worst_a = max(worst_a, np.abs(O.lsp2lpc(l2) - a).max())
        worst_c = max(worst_c, np.abs(O.stage_coefficients(l2, 0.55, 0.0, False, 2) - c).max() / np.abs(c).max())
    print("one-ulp input change: LPC moves by", worst_a, "coefficients by", worst_c, "(relative)")
    assert 1e-12 < worst_a < 1e-6 and 1e-11 < worst_c < 1e-6
