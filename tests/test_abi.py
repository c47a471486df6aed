"""The C-ABI library loads and exports every symbol include/jwas_hip.h declares (no compute calls:
there is no GPU on the CPU test box), and the shipped package never routes through the oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "jwas_hip.h")).read()
    return sorted(set(re.findall(r"\b(jwas_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from jwas_jl_amd import _lib
    L = _lib.load()
    declared = _declared()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/jwas_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (jwas_hip_\w+)", out))
    assert exported == set(declared)


def test_library_contains_gfx950_code_object():
    from jwas_jl_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_block_step" in blob and b"k_prepare" in blob


def test_memory_estimate_formula():
    """HBM analogue of estimate_marker_memory (tools4genotypes.jl:99-235; test_memory_guardrails.jl:9-60):
    dense X (rows padded to 256) + block Grams and cross-Grams (2*p*b) + x'x + state + residuals + partials;
    stream storage: 2-bit payload + means instead of the fp32 matrix (the reference's stream formula has the same
    ceil(n/4) term, test_memory_guardrails.jl:62-75)."""
    from jwas_jl_amd import HipEngine
    n, p, t, b = 1000, 5000, 1, 256
    ld = 1024
    rest = 2 * 4 * b * p + 4 * p + t * p * 4 * 6 + 4 * ld * 4 + b * (ld // 256) * t * 8
    assert HipEngine.estimate_bytes(n, p, t, b) == 4 * ld * p + rest
    assert HipEngine.estimate_bytes(n, p, t, b, storage="stream") == (ld // 4) * p + 4 * p + rest
    assert HipEngine.estimate_bytes(50_000, 600_000, 1, 256) < 288e9        # config 2 fits one MI355X
    # multi-trait sampler I on 256-marker blocks: + the per-sweep section inverses of Rule T, (64 t)^2 floats per 64-marker section
    t3 = HipEngine.estimate_bytes(n, p, 3, 256) - HipEngine.estimate_bytes(n, p, 3, 128)
    assert t3 == 2 * 4 * 128 * p + 128 * (ld // 256) * 3 * 8 + (p // 256) * 4 * (192 * 192) * 4


def test_no_cpu_fallback_without_gpu():
    import jwas_jl_amd as J
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(J.JwasHipError, match="no CPU fallback"):
        J.HipEngine(0)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "jwas.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                # comments may NAME the oracle (the arithmetic contract is shared); code must not use it
                for pat in (r"#\s*include[^\n]*oracle", r"^\s*(import|from)\s+oracle", r"libjwas_oracle", r"oracle_engine",
                            r"dlopen[^\n]*oracle", r"CDLL[^\n]*oracle"):
                    assert not re.search(pat, src, flags=re.M), (f, pat)


def test_julia_struct_layout(tmp_path):
    """julia/JWASHip.jl mirrors jwas_sweep_params / jwas_sweep_stats as isbits structs.  Julia is absent here, so the
    struct definitions are parsed and their C layout (natural alignment -- what an isbits Julia struct has) is compared
    with gcc's offsetof / sizeof of the header's structs: a field-order or length slip would corrupt silently."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _check_julia_struct_layout(open(os.path.join(root, "julia", "JWASHip.jl")).read(), tmp_path)


def test_integration_md_struct_excerpt_matches_the_header(tmp_path):
    """INTEGRATION.md section 1 prints the struct mirrors a maintainer would paste: the excerpt must be the text of
    julia/JWASHip.jl (round 3's copy had drifted four fields short) and must itself match the header's layout."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    jl = open(os.path.join(root, "julia", "JWASHip.jl")).read()
    blocks = "\n".join(re.findall(r"```julia\n(.*?)```", md, re.S))
    for name in ("HipSweepParams", "HipSweepStats"):
        pat = r"struct %s\n.*?\nend" % name
        assert re.search(pat, blocks, re.S).group(0) == re.search(pat, jl, re.S).group(0), name
    # the positional constructor call of the excerpt has one argument per field
    nfields = len(re.search(r"struct HipSweepParams\n(.*?)\nend", blocks, re.S).group(1).strip().splitlines())
    ctor = re.search(r"function HipSweepParams\(.*?\n    (HipSweepParams\(.*?)\nend", blocks, re.S).group(1)
    ctor = re.sub(r"#.*", "", ctor)
    depth, nargs = 0, 1
    for ch in ctor[ctor.index("(") + 1:ctor.rindex(")")]:
        depth += ch in "({[" 
        depth -= ch in ")}]"
        nargs += (ch == "," and depth == 0)
    assert nargs == nfields, (nargs, nfields)
    _check_julia_struct_layout(blocks, tmp_path)


def _check_julia_struct_layout(src, tmp_path):
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    size = {"Int32": 4, "UInt32": 4, "UInt64": 8, "Int64": 8, "Float32": 4, "Float64": 8}

    def layout(name):
        body = re.search(r"struct %s\n(.*?)\nend" % name, src, re.S).group(1)
        off, fields, maxal = 0, [], 1
        for line in body.strip().splitlines():
            fname, ftype = [v.strip() for v in line.strip().split("::")]
            m = re.match(r"NTuple\{(\d+),(\w+)\}", ftype)
            if m:
                cnt, el = int(m.group(1)), size[m.group(2)]
            elif ftype.startswith("Ptr{"):
                cnt, el = 1, 8
            else:
                cnt, el = 1, size[ftype]
            off = (off + el - 1) // el * el
            fields.append((fname, off))
            off += cnt * el
            maxal = max(maxal, el)
        return fields, (off + maxal - 1) // maxal * maxal

    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "jwas_hip.h"', "int main(void){"]
    expect = []
    for jl, cname in (("HipSweepParams", "jwas_sweep_params"), ("HipSweepStats", "jwas_sweep_stats")):
        fields, total = layout(jl)
        for fname, off in fields:
            prog.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
            expect.append(off)
        prog.append(f'printf("%zu\\n", sizeof({cname}));')
        expect.append(total)
    prog.append("return 0;}")
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(cfile), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == expect
