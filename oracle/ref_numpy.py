"""Float64 brute-force definitions of the two hot paths (independent of bpmf_oracle.c).

TEST INFRASTRUCTURE ONLY.  These follow the *mathematical* definitions (SURVEY.md
section 8a MF-1 / BP-1 and Appendix C), in float64 and with no care for summation order,
so they check the C oracle's indexing and conventions, not its rounding.
Small sizes only (pure NumPy, O(T*S*C*n_corr*L)).
"""
import numpy as np


def matched_filter_f64(templates, moveouts, weights, data, step=1, network_sum=True):
    tp = np.asarray(templates, dtype=np.float64)
    T, S, C, L = tp.shape
    mv = np.broadcast_to(np.asarray(moveouts).reshape(T, S, -1), (T, S, C)).astype(np.int64)
    w = np.broadcast_to(np.asarray(weights).reshape(T, S, -1), (T, S, C)).astype(np.float64)
    d = np.asarray(data, dtype=np.float64)
    N = d.shape[-1]
    n_corr = (N - L) // step + 1
    out = np.zeros((T, n_corr, S, C), dtype=np.float64)
    for t in range(T):
        act = w[t] != 0
        if not act.any():
            continue
        mv_min, mv_max = mv[t][act].min(), mv[t][act].max()
        i_first = 0 if mv_min >= 0 else -(mv_min // step)  # ceil(-mv_min/step)
        if N - L - mv_max < 0:
            continue
        i_last = min((N - L - mv_max) // step, n_corr - 1)
        if i_first > i_last:
            continue
        lags = np.arange(i_first, i_last + 1)
        for s in range(S):
            for c in range(C):
                if not act[s, c]:
                    continue
                starts = lags * step + mv[t, s, c]
                win = np.lib.stride_tricks.sliding_window_view(d[s, c], L)[starts]
                num = win @ tp[t, s, c]
                den = (tp[t, s, c] ** 2).sum() * (win ** 2).sum(axis=1)
                cc = np.where(den > 1e-6, num / np.sqrt(np.maximum(den, 1e-300)), 0.0)
                out[t, lags, s, c] = cc
    if network_sum:
        return np.einsum("tisc,tsc->ti", out, w)
    return out


def prestack_f64(features, w_phases):
    # U[s,p,t] = sum_c alpha[s,c,p] * feat[s,c,t]
    return np.einsum("scp,scn->spn", np.asarray(w_phases, np.float64), np.asarray(features, np.float64))


def beamform_f64(features, moveouts, w_phases, w_sources, out_of_bounds="strict", reduce="max"):
    U = prestack_f64(features, w_phases)
    S, P, N = U.shape
    tau = np.asarray(moveouts, dtype=np.int64)
    beta = np.asarray(w_sources, dtype=np.float64)
    K = tau.shape[0]
    beam = np.zeros((K, N))
    computed = np.zeros((K, N), dtype=bool)
    t = np.arange(N)
    for k in range(K):
        act = beta[k] != 0
        if not act.any():
            continue
        if out_of_bounds == "strict":
            lo, hi = tau[k][act].min(), tau[k][act].max()
            ok = (t + lo >= 0) & (t + hi < N)
        else:
            ok = np.ones(N, dtype=bool)
        b = np.zeros(N)
        for s in np.where(act)[0]:
            for p in range(P):
                x = t + tau[k, s, p]
                inb = (x >= 0) & (x < N)
                b[inb] += beta[k, s] * U[s, p, x[inb]]
        beam[k, ok] = b[ok]
        computed[k] = ok
    if reduce == "none":
        return beam
    # max over computed beams, floor (0, k=0), lowest k on ties
    masked = np.where(computed, beam, -np.inf)
    best = masked.max(axis=0)
    arg = masked.argmax(axis=0)
    keep = best > 0
    return np.where(keep, best, 0.0), np.where(keep, arg, 0).astype(np.int32), beam
