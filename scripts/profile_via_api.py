"""cProfile of the runMCMC() path on a device-resident config-2 matrix (development aid)."""
import cProfile, pstats, os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import jwas_jl_amd as J
n, p = 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 600000
e = J.HipEngine(0); e.alloc_dense(n, p); e.synth(2026, 0, True); e.setup_blocks(512, "mfma"); e.add_block_size(1024, "mfma")
e.init_state("BayesC")
rng = np.random.default_rng(1)
a = np.zeros(p, np.float32); idx = rng.choice(p, p // 1000, replace=False); a[idx] = rng.standard_normal(len(idx))
e.set_state(alpha=a); g = e.mul_alpha().astype(np.float64); g *= np.sqrt(0.5 / g.var())
y = (1 + g + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32)
import time
_orig = e.sweep
_log = []
def _sweep(**kw):
    t0 = time.perf_counter(); st = _orig(**kw); t1 = time.perf_counter()
    _log.append((kw["iteration"], e.block_size, st["n_events"], st["sweep_ms"], 1e3 * (t1 - t0), float(kw["pi"]), float(kw["vare"]), float(kw["var_effect"])))
    return st
e.sweep = _sweep
geno = J.device_genotypes(e, method="BayesC", Pi=0.95, estimatePi=True)
model = J.build_model("y = intercept + geno", genotypes={"geno": geno})
ph = pd.DataFrame({"ID": geno.obsID, "y": y})
folder = tempfile.mkdtemp()
pr = cProfile.Profile()
if os.environ.get("NOPROF") is None: pr.enable()
out = J.runMCMC(model, ph, chain_length=50, burnin=30, seed=1, outputEBV=False, output_samples_frequency=100, output_folder=os.path.join(folder, "r"), printout_model_info=False)
pr.disable()
ts = out["_timing"]["iteration_end_s"]
print("per-iteration ms:", np.round(np.diff(ts) * 1e3, 1)[-25:])
print("call_ms:", np.round([r[4] for r in _log[-26:]], 0))
for row in _log[-2:]:
    print("it %d bs %d events %.0f sweep_ms %.1f call_ms %.1f pi %.5f vare %.4f G %.3e" % row)
print("device sweep ms total", out["_timing"]["device_sweep_ms_total"], "block", out["_timing"]["block_size"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(6)
shutil.rmtree(folder, ignore_errors=True)
