"""The reference's streaming backend on disk (.jgb2 + sidecars): writer/reader round trip, codec tables of
test/unit/test_streaming_codec.jl:21-51 and test_streaming_prepare_lowmem.jl:22-67, stream-mode runMCMC."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
import oracle as O
import jwas_jl_amd.api as api
from jwas_jl_amd import streaming as S


def test_pack_layout_is_the_reference_bit_layout():
    # individual i (0-based): byte i>>2, shift (i&3)<<1 (streaming_genotypes.jl:364-367,622-627)
    codes = np.array([[0], [1], [2], [3], [2], [1]], dtype=np.uint8)            # one marker, 6 individuals
    payload = S.pack_2bit(codes)
    assert payload.shape == (1, 2)
    assert payload[0, 0] == (0 | 1 << 2 | 2 << 4 | 3 << 6) and payload[0, 1] == (2 | 1 << 2)
    assert np.array_equal(S.unpack_2bit(payload, 6), codes)


def test_prepare_load_decode_round_trip(tmp_path):
    """6 x 4 table of test_streaming_codec.jl:21-51: missing -> marker mean, centred; x'x sidecar = sum v^2 - mu sum v."""
    raw = np.array([[0, 1, 2, 0], [1, 1, 0, 2], [2, 9, 1, 1], [0, 2, 2, 0], [1, 0, 9, 2], [2, 1, 1, 1]], dtype=np.float64)
    ids = [f"a{i}" for i in range(6)]
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "g6", obs_ids=ids, marker_ids=["m1", "m2", "m3", "m4"],
                                           quality_control=False)
    for ext in (".jgb2", ".meta", ".obsid.txt", ".markerid.txt", ".selected.i32", ".mean.f32", ".xpRinvx.f32", ".afreq.f32"):
        assert os.path.isfile(prefix + ext), ext
    meta = dict(line.rstrip("\n").split("\t", 1) for line in open(prefix + ".meta"))
    assert meta["version"] == "1" and meta["nObs"] == "6" and meta["nMarkers"] == "4" and meta["stride_bytes"] == "2"
    assert meta["centered"] == "1" and os.path.getsize(prefix + ".jgb2") == 8
    b = S.load_streaming_backend(prefix + ".meta")
    assert b["obsID"] == ids and b["markerID"] == ["m1", "m2", "m3", "m4"]
    X = S.decode_markers(b)
    dense = raw.astype(np.float32).copy()
    for j in range(4):
        col = dense[:, j]
        miss = col == 9
        col[miss] = col[~miss].mean(dtype=np.float32)
    mu = np.array([dense[:, j][raw[:, j] != 9].mean(dtype=np.float32) for j in range(4)], dtype=np.float32)
    np.testing.assert_allclose(b["marker_means"], mu, rtol=1e-6)
    np.testing.assert_allclose(X, dense - b["marker_means"][None, :], atol=1e-6)
    np.testing.assert_allclose(b["xpRinvx"], (X.astype(np.float64) ** 2).sum(axis=0), rtol=1e-5)
    np.testing.assert_allclose(b["allele_freq"], b["marker_means"] / 2)
    # the C oracle's decode agrees element for element
    payload = np.fromfile(b["data_path"], dtype=np.uint8).reshape(4, 2)
    for j in range(4):
        assert np.array_equal(O.decode_marker_2bit(payload, 6, j, b["marker_means"][j], True), X[:, j])


def test_prepare_applies_the_dense_path_qc(tmp_path):
    d = make_dataset(n=120, p=60, ncausal=3, seed=2, center=False)
    raw = d["raw"].astype(np.float64)
    raw[:, 7] = 0                                   # fixed locus -> removed (readgenotypes.jl:388-399)
    raw[5, 11] = 9
    ids = [str(i) for i in range(120)]
    gdf = pd.DataFrame(raw, columns=[f"s{j}" for j in range(60)])
    gdf.insert(0, "ID", ids)
    dense = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "qc", obs_ids=ids, marker_ids=[f"s{j}" for j in range(60)])
    g = api.get_genotypes(prefix, method="BayesC", Pi=0.9, storage="stream")
    assert g.storage_mode == "stream" and g.genotypes.shape == (120, 0)
    assert g.markerID == dense.markerID and g.nMarkers == dense.nMarkers == 59
    assert g.sum2pq == pytest.approx(dense.sum2pq, rel=1e-5)
    np.testing.assert_allclose(S.decode_markers(g.stream_backend), dense.genotypes, atol=2e-6)
    with pytest.raises(ValueError, match="requires a file path or prefix"):
        api.get_genotypes(gdf, storage="stream")
    with pytest.raises(FileNotFoundError, match="Streaming manifest is not found"):
        api.get_genotypes(str(tmp_path / "nope"), storage="stream")
    os.truncate(prefix + ".jgb2", 10)
    with pytest.raises(ValueError, match="file size does not match metadata"):
        api.get_genotypes(prefix, storage="stream")


def test_stream_mode_runmcmc_equals_dense_mode(tmp_path):
    """test_streaming_codec.jl:100-104: stream vs dense posterior means within 1e-4 -- here the same engine sees
    the same decoded matrix, so the chains are identical."""
    d = make_dataset(n=160, p=130, ncausal=4, seed=8, center=False)
    ids = [str(i) for i in range(160)]
    raw = d["raw"].astype(np.float64)
    gdf = pd.DataFrame(raw, columns=[f"s{j}" for j in range(130)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "st", obs_ids=ids, marker_ids=list(gdf.columns[1:]))
    outs = []
    for mode in ("dense", "stream"):
        geno = api.get_genotypes(gdf if mode == "dense" else prefix, method="BayesC", Pi=0.9, storage=mode)
        model = api.build_model("y1 = intercept + geno")
        outs.append(api.runMCMC(model, ph, chain_length=40, burnin=5, seed=4, outputEBV=False,
                                output_folder=str(tmp_path / mode), _engine=OracleEngine("block"), block_size=64))
    a, b = (o["marker effects geno"] for o in outs)
    assert list(a["Marker_ID"]) == list(b["Marker_ID"])
    np.testing.assert_allclose(a["Estimate"], b["Estimate"], atol=1e-4)
    np.testing.assert_allclose(a["Model_Frequency"], b["Model_Frequency"], atol=1e-4)
    geno = api.get_genotypes(prefix, method="BayesC", Pi=0.9, storage="stream")
    model = api.build_model("y1 = intercept + geno")
    with pytest.raises(ValueError, match="requires exact genotype/phenotype ID match and order"):
        api.runMCMC(model, ph.iloc[::-1], chain_length=2, output_folder=str(tmp_path / "bad"), _engine=OracleEngine("block"))


def _write_geno_file(path, n=37, p=23, seed=4, missing=True):
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.02, 0.5, p)
    G = (rng.random((n, p)) < f).astype(int) + (rng.random((n, p)) < f).astype(int)
    G[:, 5] = 1                                           # fixed marker: removed by QC
    if missing:
        G[rng.random((n, p)) < 0.03] = 9
    ids = [f"a{i}" for i in range(n)]
    tab = pd.DataFrame(G, columns=[f"m{j + 1}" for j in range(p)])
    tab.insert(0, "ID", ids)
    tab.to_csv(path, index=False)
    return G, ids


@pytest.mark.parametrize("center", [True, False])
def test_lowmem_file_conversion_equals_dense_conversion(tmp_path, center):
    """test_streaming_prepare_lowmem.jl:22-67,114-160: the chunked two-pass converter writes the same backend as the
    load-everything converter (payload, sidecars, QC), and both decode to the dense get_genotypes matrix."""
    import pandas as pd  # noqa: F811
    path = str(tmp_path / "geno.csv")
    G, ids = _write_geno_file(path)
    pre_low = S.prepare_streaming_genotypes(path, str(tmp_path / "low"), conversion_mode="lowmem", chunk_rows=8, center=center)
    pre_den = S.prepare_streaming_genotypes(path, str(tmp_path / "den"), conversion_mode="dense", center=center)
    for ext in (".jgb2", ".obsid.txt", ".markerid.txt", ".selected.i32", ".mean.f32", ".xpRinvx.f32", ".afreq.f32"):
        assert open(pre_low + ext, "rb").read() == open(pre_den + ext, "rb").read(), ext
    b = S.load_streaming_backend(pre_low)
    dense = api.get_genotypes(path, separator=",", header=True, method="BayesC", center=center)
    assert b["markerID"] == dense.markerID and b["nObs"] == dense.nObs == 37 and b["nMarkers"] == dense.nMarkers < 23
    Xs = S.decode_markers(b)
    np.testing.assert_allclose(Xs, np.asarray(dense.genotypes), atol=1e-5)
    np.testing.assert_allclose((Xs.astype(np.float64) ** 2).sum(0), b["xpRinvx"], rtol=2e-5)


def test_auto_mode_and_guards(tmp_path, capsys):
    """test_streaming_prepare_lowmem.jl:97-145"""
    path = str(tmp_path / "geno.csv")
    _write_geno_file(path, missing=False)
    S.prepare_streaming_genotypes(path, str(tmp_path / "auto_d"), conversion_mode="auto", auto_dense_max_bytes=2 ** 30)
    assert "Auto conversion mode selected :dense" in capsys.readouterr().out
    S.prepare_streaming_genotypes(path, str(tmp_path / "auto_l"), conversion_mode="auto", auto_dense_max_bytes=10)
    assert "Auto conversion mode selected :lowmem" in capsys.readouterr().out
    assert open(str(tmp_path / "auto_d") + ".jgb2", "rb").read() == open(str(tmp_path / "auto_l") + ".jgb2", "rb").read()
    with pytest.raises(OSError, match="Insufficient disk"):
        S.prepare_streaming_genotypes(path, str(tmp_path / "guard"), conversion_mode="lowmem", disk_guard_ratio=0.0)
    with pytest.raises(ValueError, match="conversion_mode"):
        S.prepare_streaming_genotypes(path, str(tmp_path / "bad"), conversion_mode="fast")
    assert S.prepare_streaming_genotypes(path).endswith("geno_stream")              # default prefix: <file>_stream


def test_streaming_qc_matches_the_reference_rules(tmp_path):
    """prepare_streaming_genotypes QC (streaming_genotypes.jl:274-296): a marker with only missing values is an error;
    the MAF filter is strict in Float32 (a marker AT the threshold is dropped); fixed markers are dropped."""
    from jwas_jl_amd.streaming import prepare_streaming_genotypes
    rng = np.random.default_rng(0)
    G = rng.integers(0, 3, size=(40, 6)).astype(np.float32)
    G[:, 2] = 9.0                                            # only missing values
    with pytest.raises(ValueError, match="Marker m3 has only missing values"):
        prepare_streaming_genotypes(G, str(tmp_path / "a"), marker_ids=[f"m{j + 1}" for j in range(6)])
    G[:, 2] = 1.0                                            # fixed: allele frequency 0.5 but no variance
    G[:, 3] = 0.0; G[:4, 3] = 1.0                            # mean 0.1 -> allele frequency exactly 0.05 = MAF: dropped (strict <)
    prefix = prepare_streaming_genotypes(G, str(tmp_path / "b"), marker_ids=[f"m{j + 1}" for j in range(6)], MAF=0.05)
    kept = open(prefix + ".markerid.txt").read().split()
    assert kept == ["m1", "m2", "m5", "m6"]
    G[:] = 0.0
    with pytest.raises(ValueError, match="No markers remain"):
        prepare_streaming_genotypes(G, str(tmp_path / "c"))
