"""Audit by the dynamic section: what the product's binaries load. libsmilehip.so / libsmilehip_comm.so /
libsmilehip_host.so / smilextract_hip need the HIP runtime (RCCL for the gather library), each other and the C / C++ run time --
nothing under oracle/, no torch, no rocFFT / rocBLAS. The openSMILE-side plugin additionally needs libopensmile.so: that is its HOST
(the process it is loaded into), found through the run path the Makefile's SMILE_HOST names -- in this tree the one
reference build there is (oracle/_ref, where the task statement puts every output of the reference's sources); the plugin imports
the component framework from it (cDataMemory, cVectorProcessor, ...) and no checker code. No product source mentions oracle/."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AMD = os.path.join(ROOT, "opensmile_amd")
ALLOWED = {"libamdhip64", "librccl", "libstdc++", "libm", "libgcc_s", "libc", "ld-linux-x86-64", "libsmilehip", "libpthread", "libdl", "librt"}


def dyn(path):
    out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    needed = [re.sub(r"\.so.*", "", m) for m in re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", out)]
    runpath = ":".join(re.findall(r"\((?:RUNPATH|RPATH)\)\s+Library (?:runpath|rpath): \[([^\]]*)\]", out))
    return needed, runpath


@pytest.mark.parametrize("name", ["libsmilehip.so", "libsmilehip_comm.so", "libsmilehip_host.so", "smilextract_hip"])
def test_product_binaries_load_nothing_of_the_checker(name):
    path = os.path.join(AMD, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not built")
    needed, runpath = dyn(path)
    assert set(needed) <= ALLOWED, needed
    assert "oracle" not in runpath and "reference" not in runpath, runpath


def test_plugin_loads_its_host_and_the_library_only():
    path = os.path.join(AMD, "plugin", "plugins", "libsmilehip_plugin.so")
    if not os.path.exists(path):
        pytest.skip("plugin not built (needs the reference's headers)")
    needed, runpath = dyn(path)
    assert set(needed) <= ALLOWED | {"libopensmile"}, needed
    assert "libopensmile" in needed and "libsmilehip" in needed
    und = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
    assert "lld_oracle" not in und and "lldo" not in und               # nothing of the restatement is imported


def test_no_product_source_mentions_the_oracle():
    bad = []
    for d, _dirs, files in os.walk(AMD):
        if "__pycache__" in d:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".inc")):
                continue
            p = os.path.join(d, f)
            for i, line in enumerate(open(p, errors="replace"), 1):
                code = line.split("//")[0] if not f.endswith(".py") else line.split("#")[0]       # comments may cite the oracle's files
                if re.search(r"#\s*include[^\n]*oracle|import oracle|from oracle|liblld_oracle|\"[^\"]*oracle[^\"]*\"", code):
                    bad.append(f"{os.path.relpath(p, ROOT)}:{i}: {line.strip()[:100]}")
    assert not bad, "\n".join(bad)
