"""Marker-annotation priors for single-trait BayesC / BayesR (host side of the marker path).

The reference regresses the inclusion indicators on marker annotations with a probit model and feeds the resulting
per-marker prior into the sweep (markers/readgenotypes.jl:56-160, markers/annotation_setup.jl, MCMC/annotation_updates.jl,
types.jl:167-215).  The sweep side is already per marker on the device (`pi_vec` for BayesC, the p x 4 `pi_matrix` for
BayesR, include/jwas_hip.h); what lives here is the O(p x annotations) host update between sweeps: latent liabilities
(truncated normals), a coordinate Gibbs pass over the annotation coefficients, their shrinkage variance, and the
rebuilt per-marker priors.  Annotated 2-trait BayesC uses the device's marker-specific joint-state prior
(`log_prior_states_matrix`) the same way."""
import numpy as np
from scipy.special import ndtr, ndtri

EPS = np.finfo(np.float64).eps


class MarkerAnnotations:
    """types.jl:167-215"""

    def __init__(self, design_matrix, *, variance=1.0, nsteps=1, nclasses=2, coefficients=None, snp_pi=None):
        X = np.asarray(design_matrix, dtype=np.float64)
        nrows, ncols = X.shape
        self.design_matrix = X
        self.nsteps, self.nclasses = nsteps, nclasses
        self.coefficients = (np.zeros(ncols) if nsteps == 1 else np.zeros((ncols, nsteps))) if coefficients is None else coefficients
        self.mean_coefficients = np.zeros_like(self.coefficients)
        self.mean_coefficients2 = np.zeros_like(self.coefficients)
        self.variance = float(variance) if nsteps == 1 else np.full(nsteps, float(variance))
        self.liability = np.zeros(nrows) if nsteps == 1 else np.zeros((nrows, nsteps))
        self.mu = np.zeros_like(self.liability)
        self.snp_pi = False if snp_pi is None else snp_pi


def validate_annotations_input(annotations, nmarkers, method):
    """readgenotypes.jl:56-70"""
    if annotations is False or annotations is None:
        return False
    if method not in ("BayesC", "BayesR"):
        raise ValueError('annotations are only supported with method="BayesC" or method="BayesR".')
    try:
        A = np.asarray(annotations, dtype=np.float64)
    except (TypeError, ValueError):
        raise ValueError("annotations must be a numeric matrix with one row per marker.")
    if A.ndim != 2:
        raise ValueError("annotations must be a numeric matrix with one row per marker.")
    if A.shape[0] != nmarkers:
        raise ValueError(f"annotations rows ({A.shape[0]}) must match the number of raw markers ({nmarkers}).")
    return A


def validate_annotation_design(A):
    """readgenotypes.jl:72-88"""
    if A.shape[1] == 0:
        return
    const = [j + 1 for j in range(A.shape[1]) if np.unique(A[:, j]).size == 1]
    if const:
        raise ValueError(f"annotations contain constant column(s) {const}. Remove constant columns because JWAS "
                         "automatically adds an intercept.")
    D = np.hstack([np.ones((A.shape[0], 1)), A])
    if np.linalg.matrix_rank(D) != D.shape[1]:
        raise ValueError("annotations are collinear after adding the intercept. Remove duplicate or perfectly collinear "
                         "annotation columns.")


def bayesr_annotation_probabilities(pi):
    """readgenotypes.jl:90-105"""
    pi = np.asarray(pi, dtype=np.float64)
    if pi.size != 4:
        raise ValueError("BayesR Pi must have length 4.")
    total_nonzero = pi[1] + pi[2] + pi[3]
    total_larger = pi[2] + pi[3]
    if not total_nonzero > 0:
        raise ValueError("Annotated BayesR requires positive nonzero prior mass.")
    if not total_larger > 0:
        raise ValueError("Annotated BayesR requires positive prior mass in classes 3 or 4.")
    p1, p2, p3 = total_nonzero, total_larger / total_nonzero, pi[3] / total_larger
    if not 0.0 < p1 < 1.0:
        raise ValueError("Annotated BayesR requires 0 < Pr(delta > 1) < 1. Adjust Pi so the zero-vs-nonzero split is nondegenerate.")
    if not 0.0 < p2 < 1.0:
        raise ValueError("Annotated BayesR requires 0 < Pr(delta > 2 | delta > 1) < 1. Adjust Pi so classes 2 versus 3/4 are both represented.")
    if not 0.0 < p3 < 1.0:
        raise ValueError("Annotated BayesR requires 0 < Pr(delta > 3 | delta > 2) < 1. Adjust Pi so classes 3 and 4 are both represented.")
    return p1, p2, p3


def build_marker_annotations(A, method, Pi):
    """readgenotypes.jl:127-150 (A: validated annotation matrix of the markers kept by QC)"""
    validate_annotation_design(A)
    D = np.hstack([np.ones((A.shape[0], 1)), A])
    if method == "BayesR":
        pi = np.array([0.95, 0.03, 0.015, 0.005]) if (np.isscalar(Pi) and Pi == 0.0) else np.asarray(Pi, dtype=np.float64)
        bayesr_annotation_probabilities(pi)
        return MarkerAnnotations(D, nsteps=3, nclasses=4, coefficients=np.zeros((D.shape[1], 3)),
                                 snp_pi=np.repeat(pi.reshape(1, 4), D.shape[0], axis=0))
    return MarkerAnnotations(D)


def initialize_bayesc_single_trait(ann, start_pi):
    """annotation_setup.jl:70-76: intercept = probit of the mean starting inclusion probability"""
    start_inclusion = float(np.clip(np.mean(1.0 - np.asarray(start_pi, dtype=np.float64)), EPS, 1 - EPS))
    ann.coefficients[:] = 0.0
    ann.coefficients[0] = ndtri(start_inclusion)
    ann.mu = ann.design_matrix @ ann.coefficients


def sample_binary_liabilities(mu, response, rng):
    """annotation_updates.jl:21-60: l_i ~ N(mu_i, 1) truncated to [0, inf) when z_i = 1 and to (-inf, 0] when z_i = 0.
    Inverse-CDF draw on the side where it is well conditioned (by symmetry the z = 0 case is the mirrored z = 1 case)."""
    sgn = np.where(np.asarray(response) != 0, 1.0, -1.0)
    m = sgn * mu                                     # draw x ~ N(m, 1) truncated to [0, inf), liability = sgn * x
    # survival function of the truncated law: Pr(X > x) = Phi(m - x) / Phi(m)  =>  x = m - Phi^-1(U * Phi(m));
    # U * Phi(m) stays representable down to m ~ -37, where the law is an Exp(|m|) overshoot of 0 to 1e-3 relative
    u = rng.random(m.shape)
    with np.errstate(divide="ignore"):
        x = m - ndtri(np.maximum(u, 1e-300) * ndtr(m))
    tail = m < -30.0
    if np.any(tail):
        x[tail] = rng.exponential(1.0 / (-m[tail]))
    return sgn * np.maximum(x, 0.0)


def gibbs_update_coefficients(coeffs, X, latent_residual, prior_var, rng):
    """annotation_updates.jl:96-123: flat prior on the intercept, N(0, prior_var) on the slopes; in place."""
    nobs = X.shape[0]
    old = coeffs[0]
    inv_lhs = 1.0 / nobs
    ahat = inv_lhs * (latent_residual.sum() + nobs * old)
    coeffs[0] = rng.standard_normal() * np.sqrt(inv_lhs) + ahat
    latent_residual += old - coeffs[0]
    for k in range(1, X.shape[1]):
        old = coeffs[k]
        xk = X[:, k]
        diag = float(xk @ xk)
        inv_lhs = 1.0 / (diag + 1.0 / prior_var)
        ahat = inv_lhs * (float(xk @ latent_residual) + diag * old)
        coeffs[k] = rng.standard_normal() * np.sqrt(inv_lhs) + ahat
        latent_residual += xk * (old - coeffs[k])


def sample_effect_variance(coeffs, rng):
    """annotation_updates.jl:125-137"""
    return (float(np.sum(coeffs[1:] ** 2)) + 2.0) / rng.chisquare(len(coeffs) + 1.0)


def update_bayesc_binary_priors(ann, delta, rng):
    """annotation_updates.jl:181-194: returns the per-marker Pr(effect = 0)."""
    ann.liability = sample_binary_liabilities(ann.mu, delta, rng)
    resid = ann.liability - ann.mu
    gibbs_update_coefficients(ann.coefficients, ann.design_matrix, resid, ann.variance, rng)
    ann.mu = ann.design_matrix @ ann.coefficients
    if len(ann.coefficients) > 1:
        ann.variance = sample_effect_variance(ann.coefficients, rng)
    return np.clip(1.0 - ndtr(ann.mu), EPS, 1 - EPS)


def _nested_probit_steps(ann, z, active, rng):
    """sample_nested_annotation_probit_step! for steps 1..3 (annotation_updates.jl:232-268)"""
    for step in range(3):
        coeffs = ann.coefficients[:, step]
        ann.mu[:, step] = ann.design_matrix @ coeffs
        act = active[step]
        if act.size == 0:
            continue
        X = ann.design_matrix[act, :]
        mu_a = ann.mu[act, step]
        liab = sample_binary_liabilities(mu_a, z[step][act], rng)
        ann.liability[act, step] = liab
        resid = liab - mu_a
        gibbs_update_coefficients(coeffs, X, resid, ann.variance[step], rng)
        if X.shape[1] > 1:
            ann.variance[step] = sample_effect_variance(coeffs, rng)
        ann.mu[:, step] = ann.design_matrix @ coeffs
    return np.clip(ndtr(ann.mu), EPS, 1 - EPS)


def update_bayesr_nested_priors(ann, delta, rng):
    """annotation_updates.jl:196-285,337-351: three nested step-up probit models (delta > 1; > 2 | > 1; > 3 | > 2);
    rebuilds ann.snp_pi (p x 4) and returns its column means."""
    delta = np.asarray(delta)
    z = (delta > 1, delta > 2, delta > 3)
    active = (np.arange(delta.size), np.nonzero(z[0])[0], np.nonzero(z[1])[0])
    probs = _nested_probit_steps(ann, z, active, rng)
    ann.snp_pi[:, 0] = 1.0 - probs[:, 0]
    ann.snp_pi[:, 1] = probs[:, 0] * (1.0 - probs[:, 1])
    ann.snp_pi[:, 2] = probs[:, 0] * probs[:, 1] * (1.0 - probs[:, 2])
    ann.snp_pi[:, 3] = probs[:, 0] * probs[:, 1] * probs[:, 2]
    return ann.snp_pi.mean(axis=0)


# ---- annotated 2-trait BayesC: a tree over the joint states 00 / 10 / 01 / 11 -------------------------------------
def bayesc_mt_start_row(pi):
    """annotation_setup.jl:101-121 (+ validate_bayesc_mt_start_row :57-68).  `pi`: the 4 joint prior probabilities in
    the order 00, 10, 01, 11 (= the device's state index sum_k delta_k << k), or 0.0 for the legacy all-active default."""
    if np.isscalar(pi) and pi == 0.0:
        row = np.array([0.0, 0.0, 0.0, 1.0])
    else:
        if np.isscalar(pi):
            raise ValueError("Annotated multi-trait BayesC requires Pi=0.0 or a joint Pi dictionary.")   # annotation_setup.jl:118
        row = np.asarray(pi, dtype=np.float64).reshape(-1)
        if row.size != 4:
            raise ValueError("Annotated multi-trait BayesC v1 expects four joint prior probabilities.")
    if not row[1] + row[3] > 0.0:
        raise ValueError("Annotated multi-trait BayesC requires positive startup prior mass in states {10,11} for trait 1.")
    if not row[2] + row[3] > 0.0:
        raise ValueError("Annotated multi-trait BayesC requires positive startup prior mass in states {01,11} for trait 2.")
    if not row[3] > 0.0:
        raise ValueError("Annotated multi-trait BayesC requires positive startup prior mass in shared state 11.")
    return row


def initialize_bayesc_mt(design_matrix, start_row):
    """annotation_setup.jl:123-133"""
    D = np.asarray(design_matrix, dtype=np.float64)
    return MarkerAnnotations(D, nsteps=3, nclasses=4, coefficients=np.zeros((D.shape[1], 3)),
                             snp_pi=np.repeat(np.asarray(start_row, dtype=np.float64).reshape(1, 4), D.shape[0], axis=0))


def update_bayesc_mt_tree_priors(ann, delta1, delta2, rng):
    """annotation_updates.jl:287-326,353-361: step 1 zero vs active, step 2 (among active) 11 vs a single trait,
    step 3 (among single-trait markers) 10 vs 01; rebuilds ann.snp_pi (p x 4, order 00,10,01,11), returns column means."""
    d1, d2 = np.asarray(delta1) != 0, np.asarray(delta2) != 0
    z = (d1 | d2, d1 & d2, d1 & ~d2)
    active = (np.arange(d1.size), np.nonzero(z[0])[0], np.nonzero(d1 ^ d2)[0])
    probs = _nested_probit_steps(ann, z, active, rng)
    p1, p2, p3 = probs[:, 0], probs[:, 1], probs[:, 2]
    ann.snp_pi[:, 0] = 1.0 - p1
    ann.snp_pi[:, 1] = p1 * (1.0 - p2) * p3
    ann.snp_pi[:, 2] = p1 * (1.0 - p2) * (1.0 - p3)
    ann.snp_pi[:, 3] = p1 * p2
    return ann.snp_pi.mean(axis=0)


def accumulate(ann, nsamples):
    """output.jl:597-601"""
    ann.mean_coefficients = ann.mean_coefficients + (ann.coefficients - ann.mean_coefficients) / nsamples
    ann.mean_coefficients2 = ann.mean_coefficients2 + (ann.coefficients ** 2 - ann.mean_coefficients2) / nsamples


def coefficients_table(ann, method):
    """output.jl:151-177"""
    import pandas as pd
    names = ["Intercept"] + [f"Annotation_{i}" for i in range(1, ann.design_matrix.shape[1])]
    sd = np.sqrt(np.abs(ann.mean_coefficients2 - ann.mean_coefficients ** 2))
    if ann.nsteps == 1:
        return pd.DataFrame({"Annotation": names, "Estimate": ann.mean_coefficients, "SD": sd})
    steps = (["step1_zero_vs_nonzero", "step2_small_vs_larger", "step3_medium_vs_large"] if method == "BayesR" else
             ["step1_zero_vs_active", "step2_11_vs_singleton", "step3_10_vs_01"])
    return pd.DataFrame({"Annotation": np.repeat(names, ann.nsteps), "Step": steps[:ann.nsteps] * len(names),
                         "Estimate": ann.mean_coefficients.reshape(-1), "SD": sd.reshape(-1)})
