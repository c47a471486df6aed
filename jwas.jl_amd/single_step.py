"""Single-step input: pedigree, A-inverse and impute_genotypes (row a21).

What the reference does before the marker sweep of a single-step analysis (src/1.JWAS/src/single_step/SSBR.jl:65-142,
src/2.PedModule/src/PedModule.jl): order the pedigree as [non-genotyped; genotyped], form the sparse A-inverse by
Henderson's rules with inbreeding (AInverse/HAi, PedModule.jl:167-219), and impute the genotypes of the non-genotyped
individuals marker chunk by marker chunk,

    A^nn M_n = -A^ng M_g        (SSBR.jl:83-135, 1000 markers per chunk),

then align [M_n; M_g] to the phenotyped individuals.  The result is the dense REAL-VALUED n_pheno x p matrix the sweep
runs on (SSBR.jl:137-138) -- 672 GB at config 5, which is why it is produced and uploaded chunk by chunk here and never
exists on the host as a whole: every chunk is solved on the host (one sparse LU of A^nn, reused) and written straight into
the device matrix (jwas_hip_set_columns).  The epsilon / J model terms of SSBRrun (SSBR.jl:23-53) are host-model terms
outside the marker path and stay on the reference.
"""
import numpy as np

MISSING = ("missing", "0", "", "nan", "NA", "None")


class Pedigree:
    """IDs in an order where parents precede offspring, parent indices (-1 = unknown) and inbreeding coefficients."""

    def __init__(self, ids, sire, dam):
        self.ids = list(ids)
        self.sire = np.asarray(sire, dtype=np.int64)
        self.dam = np.asarray(dam, dtype=np.int64)
        self.index = {v: i for i, v in enumerate(self.ids)}
        self.f = inbreeding(self.sire, self.dam)


def get_pedigree(source, separator=",", header=False):
    """PedModule.mkPed (PedModule.jl): rows (individual, sire, dam), unknown parents "missing" / "0"; parents that never
    appear as individuals are added as founders; individuals are ordered so that parents come first."""
    import pandas as pd
    if isinstance(source, str):
        tab = pd.read_csv(source, sep=separator, header=0 if header else None, dtype=str, keep_default_na=False)
    else:
        tab = pd.DataFrame(source).astype(str)
    rows = {}
    for ind, s, d in zip(tab.iloc[:, 0], tab.iloc[:, 1], tab.iloc[:, 2]):
        ind, s, d = ind.strip(), s.strip(), d.strip()
        rows[ind] = (None if s in MISSING else s, None if d in MISSING else d)
    for s, d in list(rows.values()):
        for par in (s, d):
            if par is not None and par not in rows:
                rows[par] = (None, None)
    order, state = [], {}
    for start in rows:                                        # iterative depth-first: parents before offspring
        if start in state:
            continue
        stack = [(start, 0)]
        while stack:
            node, stage = stack.pop()
            if stage == 1:
                order.append(node); state[node] = 2
                continue
            if node in state:
                if state[node] == 1:
                    raise ValueError(f"pedigree loop at individual {node}")
                continue
            state[node] = 1
            stack.append((node, 1))
            for par in rows[node]:
                if par is not None and state.get(par) != 2:
                    if state.get(par) == 1:
                        raise ValueError(f"pedigree loop at individual {par}")
                    stack.append((par, 0))
    idx = {v: i for i, v in enumerate(order)}
    sire = [idx[rows[v][0]] if rows[v][0] is not None else -1 for v in order]
    dam = [idx[rows[v][1]] if rows[v][1] is not None else -1 for v in order]
    return Pedigree(order, sire, dam)


def inbreeding(sire, dam):
    """Inbreeding coefficients (Meuwissen & Luo 1992): A_ii = sum_j l_ij^2 d_j over the ancestors of i, traced through a
    descending linked list; parents must have smaller indices than their offspring."""
    n = len(sire)
    F = np.zeros(n + 1)
    F[n] = -1.0                                               # "unknown parent"
    s_ = np.where(np.asarray(sire) < 0, n, sire)
    d_ = np.where(np.asarray(dam) < 0, n, dam)
    L = np.zeros(n + 1)
    D = np.zeros(n + 1)
    nxt = np.full(n + 1, -1, dtype=np.int64)                  # linked list of ancestors still to visit, descending
    for i in range(n):
        D[i] = 0.5 - 0.25 * (F[s_[i]] + F[d_[i]])
        if s_[i] == n or d_[i] == n:
            F[i] = 0.0
            continue
        fi = -1.0
        L[i] = 1.0
        head = i
        while head != -1:
            k = head
            head = nxt[k]
            nxt[k] = -1
            r = 0.5 * L[k]
            for par in (s_[k], d_[k]):
                if par == n:
                    continue
                if L[par] == 0.0:                             # insert into the descending list
                    if head == -1 or par > head:
                        nxt[par] = head
                        head = par
                    else:
                        q = head
                        while nxt[q] != -1 and nxt[q] > par:
                            q = nxt[q]
                        if nxt[q] != par and q != par:
                            nxt[par] = nxt[q]
                            nxt[q] = par
                L[par] += r
            fi += L[k] * L[k] * D[k]
            L[k] = 0.0
        F[i] = fi
    return F[:n]


def a_inverse(ped, order=None):
    """Sparse A-inverse by Henderson's rules with inbreeding (AInverse = hAi'hAi, PedModule.jl:167-219), rows / columns in
    `order` (a permutation of pedigree indices; default: pedigree order)."""
    import scipy.sparse as sp
    n = len(ped.ids)
    pos = np.arange(n) if order is None else np.argsort(np.asarray(order))
    ii, jj, vv = [], [], []
    for i in range(n):
        s, d = ped.sire[i], ped.dam[i]
        if s >= 0 and d >= 0:
            dd = np.sqrt(4.0 / (2.0 - ped.f[s] - ped.f[d]))
            ii += [pos[i], pos[i], pos[i]]; jj += [pos[s], pos[d], pos[i]]; vv += [-0.5 * dd, -0.5 * dd, dd]
        elif s >= 0 or d >= 0:
            par = s if s >= 0 else d
            dd = np.sqrt(4.0 / (3.0 - ped.f[par]))
            ii += [pos[i], pos[i]]; jj += [pos[par], pos[i]]; vv += [-0.5 * dd, dd]
        else:
            ii.append(pos[i]); jj.append(pos[i]); vv.append(1.0)
    h = sp.csr_matrix((vv, (ii, jj)), shape=(n, n))
    return (h.T @ h).tocsc()


def impute_genotypes(geno, ped, pheno_ids, engine=None, markers_per_chunk=1000, return_host=False):
    """impute_genotypes (SSBR.jl:83-142).  `geno`: Genotypes of the genotyped individuals (dense, processed by
    get_genotypes); `ped`: Pedigree; `pheno_ids`: the individuals of the phenotype file, in its order.  Returns a
    Genotypes object whose n_pheno x p matrix is resident on `engine` (a HipEngine; api.device_genotypes semantics) --
    or, with return_host=True (small problems, tests), whose `genotypes` is the host matrix."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from .api import Genotypes, Variance
    gset = set(geno.obsID)
    for g in geno.obsID:
        if g not in ped.index:
            raise ValueError(f"genotyped individual {g} is not in the pedigree")
    pheno_ids = [str(v) for v in pheno_ids]
    for v in pheno_ids:
        if v not in ped.index:
            raise ValueError(f"phenotyped individual {v} is not in the pedigree")
    # pedigree order [non-genotyped; genotyped]  (genoSet!, calc_Ai: SSBR.jl:72-80)
    non = [i for i, v in enumerate(ped.ids) if v not in gset]
    gen = [i for i, v in enumerate(ped.ids) if v in gset]
    order = np.array(non + gen, dtype=np.int64)
    nn = len(non)
    Ai = a_inverse(ped, order)
    Ai_nn, Ai_ng = Ai[:nn, :nn].tocsc(), Ai[:nn, nn:].tocsr()
    gidx = {g: i for i, g in enumerate(geno.obsID)}
    Mg_rows = np.array([gidx[ped.ids[i]] for i in gen], dtype=np.int64)            # Z * genotypes (:88-89)
    lu = spla.splu(Ai_nn) if nn else None
    where = {int(i): k for k, i in enumerate(order)}
    prow = np.array([where[ped.index[v]] for v in pheno_ids], dtype=np.int64)      # rows of [M_n; M_g] per phenotyped individual
    n_ph, p = len(pheno_ids), geno.nMarkers
    G = geno.genotypes
    host = np.empty((n_ph, p), dtype=np.float32, order="F") if return_host else None
    if not return_host:
        if engine is None:
            raise ValueError("impute_genotypes needs a HipEngine to hold the imputed matrix (or return_host=True)")
        engine.alloc_dense(n_ph, p)
    for j0 in range(0, p, markers_per_chunk):                                      # :112-131
        j1 = min(p, j0 + markers_per_chunk)
        Mg = np.asarray(G[Mg_rows, j0:j1], dtype=np.float64)
        Mn = lu.solve(-(Ai_ng @ Mg)) if nn else np.zeros((0, j1 - j0))
        chunk = np.vstack([Mn, Mg])[prow].astype(np.float32)                       # Z * Mped_chunk, data_type.(...) (:128,:137-138)
        if return_host:
            host[:, j0:j1] = chunk
        else:
            engine.set_columns(j0, chunk)
    out = Genotypes(pheno_ids, geno.markerID, n_ph, p, geno.alleleFreq, geno.sum2pq, geno.centered,
                    host if return_host else np.zeros((n_ph, 0), dtype=np.float32))
    if not return_host:
        out.storage_mode, out.device_backend = "device", engine
    for k in ("G", "genetic_variance", "method", "estimatePi", "pi", "multi_trait_sampler", "name"):
        setattr(out, k, getattr(geno, k))
    return out
