"""impute_genotypes / A-inverse (single_step/SSBR.jl:65-142, PedModule.jl:167-219) against dense numpy algebra."""
import numpy as np
import pandas as pd
import pytest

import jwas_jl_amd as J  # noqa: F401
from jwas_jl_amd import api, single_step as SS
from conftest import make_dataset


def _random_pedigree(n_founders, n_offspring, seed, inbred=True):
    rng = np.random.default_rng(seed)
    rows = [(f"a{i}", "missing", "missing") for i in range(n_founders)]
    ids = [r[0] for r in rows]
    for k in range(n_offspring):
        pool = ids if inbred else ids[:n_founders]
        s, d = rng.choice(len(pool), 2, replace=False)
        u = rng.random()
        rows.append((f"o{k}", pool[s] if u > 0.1 else "missing", pool[d] if u < 0.95 else "0"))
        ids.append(f"o{k}")
    perm = rng.permutation(len(rows))                       # the file need not list parents first
    return pd.DataFrame([rows[i] for i in perm], columns=["ID", "sire", "dam"])


def _tabular_A(ped):
    n = len(ped.ids)
    A = np.zeros((n, n))
    for i in range(n):
        s, d = ped.sire[i], ped.dam[i]
        A[i, i] = 1.0 + (0.5 * A[s, d] if s >= 0 and d >= 0 else 0.0)
        for j in range(i):
            A[i, j] = A[j, i] = 0.5 * ((A[j, s] if s >= 0 else 0.0) + (A[j, d] if d >= 0 else 0.0))
    return A


def test_inbreeding_and_a_inverse_match_the_tabular_method():
    ped = SS.get_pedigree(_random_pedigree(12, 140, seed=3))
    A = _tabular_A(ped)
    np.testing.assert_allclose(ped.f, np.diag(A) - 1.0, atol=1e-12)
    assert ped.f.max() > 0.05                               # the pedigree is inbred
    Ai = SS.a_inverse(ped).toarray()
    np.testing.assert_allclose(Ai @ A, np.eye(len(A)), atol=1e-9)
    order = np.random.default_rng(0).permutation(len(A))    # any ordering, e.g. [non-genotyped; genotyped]
    np.testing.assert_allclose(SS.a_inverse(ped, order).toarray(), Ai[np.ix_(order, order)], atol=1e-12)
    with pytest.raises(ValueError, match="pedigree loop"):
        SS.get_pedigree(pd.DataFrame([("x", "y", "missing"), ("y", "x", "missing")]))


def test_impute_genotypes_matches_the_dense_solve():
    """M_n = -(A^nn)^-1 A^ng M_g (SSBR.jl:90-104), aligned to the phenotyped individuals; config-5 shape at reduced scale
    (a third of the phenotyped individuals genotyped, chunks of 64 markers)."""
    pdf = _random_pedigree(20, 380, seed=5)
    ped = SS.get_pedigree(pdf)
    rng = np.random.default_rng(1)
    genotyped = sorted(rng.choice(ped.ids, 130, replace=False))
    d = make_dataset(n=130, p=300, ncausal=5, seed=8, center=False)
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(300)]); gdf.insert(0, "ID", genotyped)
    geno = api.get_genotypes(gdf, 1.0, method="BayesC", Pi=0.9, quality_control=False)
    pheno_ids = list(rng.permutation(ped.ids)[:360])
    out = SS.impute_genotypes(geno, ped, pheno_ids, markers_per_chunk=64, return_host=True)
    assert out.genotypes.shape == (360, 300) and out.genotypes.dtype == np.float32 and out.obsID == pheno_ids
    # dense reference
    A = _tabular_A(ped)
    gset = set(genotyped)
    non = [i for i, v in enumerate(ped.ids) if v not in gset]
    gen = [i for i, v in enumerate(ped.ids) if v in gset]
    Ai = np.linalg.inv(A[np.ix_(non + gen, non + gen)])
    nn = len(non)
    gi = {g: k for k, g in enumerate(geno.obsID)}
    Mg = geno.genotypes[[gi[ped.ids[i]] for i in gen]].astype(np.float64)
    Mn = -np.linalg.solve(Ai[:nn, :nn], Ai[:nn, nn:] @ Mg)
    full = {ped.ids[i]: row for i, row in zip(non + gen, np.vstack([Mn, Mg]))}
    ref = np.stack([full[v] for v in pheno_ids])
    np.testing.assert_allclose(out.genotypes, ref, atol=2e-5)
    # genotyped individuals keep their genotypes; an ungenotyped offspring of two genotyped parents gets their average
    for v in pheno_ids[:50]:
        if v in gset:
            np.testing.assert_array_equal(out.genotypes[pheno_ids.index(v)], geno.genotypes[gi[v]])
    frac = np.abs(out.genotypes[[k for k, v in enumerate(pheno_ids) if v not in gset]] % 1.0)
    assert ((frac > 1e-3) & (frac < 1 - 1e-3)).mean() > 0.5     # real-valued rows
    with pytest.raises(ValueError, match="not in the pedigree"):
        SS.impute_genotypes(geno, ped, ["nobody"], return_host=True)
