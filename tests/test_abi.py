"""The C-ABI library loads on a CPU-only box and exports exactly what include/webdgs.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

from webdgs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "webdgs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wdgs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) > 70
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/webdgs.h but not exported by libwebdgs_hip.so"
    assert sorted(_lib.SIGNATURES) == names, "webdgs_amd/_lib.py SIGNATURES and include/webdgs.h disagree"


def test_no_undeclared_exports():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T wdgs_" in l})
    assert exported == declared_functions()


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.wdgs_abi_version() == 1
    assert isinstance(lib.wdgs_last_error(), bytes)


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "webdgs.h"\nint main(void){ wdgs_tiled_forward_config c; (void)c; return sizeof(wdgs_kernel_time) == 56 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_struct_layouts_match_ctypes():
    # sizes the C compiler sees == sizes ctypes uses (guards against silent ABI drift)
    code = r'''
#include <stdio.h>
#include "webdgs.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(wdgs_tiled_forward_config), sizeof(wdgs_tiled_forward_resources),
 sizeof(wdgs_training_config), sizeof(wdgs_tiled_backward_config), sizeof(wdgs_tiled_backward_resources), sizeof(wdgs_adam_hyperparameters),
 sizeof(wdgs_optimizer_state), sizeof(wdgs_densify_config), sizeof(wdgs_densify_prepared)); return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")], text=True).split()]
    expect = [ctypes.sizeof(t) for t in (_lib.TiledForwardConfig, _lib.TiledForwardResources, _lib.TrainingConfig, _lib.TiledBackwardConfig,
                                         _lib.TiledBackwardResources, _lib.AdamHyperparameters, _lib.OptimizerState, _lib.DensifyConfig, _lib.DensifyPrepared)]
    assert sizes == expect


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle (only tests/, smoke() and bench.py's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "webdgs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("parity oracle", "").replace("the oracle's", "").lower() or f in ("dmath.h", "backward.hip", "raster.hip"), \
                    f"{f} mentions the oracle"
                assert "from oracle" not in text and "import oracle" not in text and "liboracle" not in text, f"{f} uses the oracle"
