"""jwas.jl_amd -- MI355X-native marker-effect Gibbs sweep behind JWAS.jl's
get_genotypes() / build_model() / runMCMC() surface.

The directory name carries a dot, so import it as `jwas_jl_amd` (the shim at the repo root
registers this package under that name).
"""
from ._lib import JwasHipError, LIB_PATH  # noqa: F401
from .engine import HipEngine, BAYESR_GAMMA  # noqa: F401

__all__ = ["HipEngine", "JwasHipError", "BAYESR_GAMMA", "LIB_PATH"]


def __getattr__(name):
    # host-side API (imports pandas etc.) is loaded lazily
    if name in ("get_genotypes", "build_model", "runMCMC", "Genotypes", "Model", "set_covariate", "outputEBV", "outputMCMCsamples", "device_genotypes"):
        from . import api
        return getattr(api, name)
    if name == "GWAS":
        from . import gwas
        return gwas.GWAS
    if name in ("prepare_streaming_genotypes", "load_streaming_backend"):
        from . import streaming
        return getattr(streaming, name)
    if name in ("samples", "single_step"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
