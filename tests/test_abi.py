"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/*.h declares, and the MAF structs have the reference's layout.
No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "multiz_amd", "libmzamd.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "multiz_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    return C.CDLL(LIB)


def _declared_functions():
    names = set()
    for h in ("mz_amd.h", "mz_yama.h", "mz_preyama.h", "mz_scores.h", "maf.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?[\s\*](\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            names.add(m.group(1))
    return names


def test_every_declared_entry_point_is_exported(lib):
    names = _declared_functions()
    assert {"yama", "pre_yama", "pre_yama2", "smooth", "mafBuild", "rmColDash", "mapping", "init_scores70", "init_scores85",
            "mafScoreRange", "mz_yama_batch", "mz_dev_run", "mz_init"} <= names
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_score_globals_exported(lib):
    for g in ("ss", "gop", "gap_open", "gap_extend"):
        C.c_void_p.in_dll(lib, g)
    lib.init_scores70()
    assert C.c_int.in_dll(lib, "gap_open").value == 400 and C.c_int.in_dll(lib, "gap_extend").value == 30
    ss = C.POINTER(C.POINTER(C.c_int)).in_dll(lib, "ss")
    assert ss[ord("A")][ord("A")] == 91 and ss[ord("a")][ord("T")] == -123 and ss[ord("-")][ord("C")] == -30
    assert ss[ord("N")][ord("A")] == -100 and ss[ord("-")][ord("-")] == 0
    lib.init_scores85()
    assert C.c_int.in_dll(lib, "gap_open").value == 600
    lib.init_scores70()


def test_maf_struct_layout_matches_reference(tmp_path):
    # offsets probed from the reference's own maf.h in SURVEY.md section 8b
    src = tmp_path / "abi.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "maf.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(struct mafAli),offsetof(struct mafAli,score),offsetof(struct mafAli,components),offsetof(struct mafAli,textSize),offsetof(struct mafAli,chain_len),'
                   'sizeof(struct mafComp),offsetof(struct mafComp,name),offsetof(struct mafComp,src),offsetof(struct mafComp,text),offsetof(struct mafComp,contig),'
                   'offsetof(struct mafComp,mafPosMap),offsetof(struct mafComp,srcSize),offsetof(struct mafComp,start),offsetof(struct mafComp,size),'
                   'offsetof(struct mafComp,nameID),offsetof(struct mafComp,strand),offsetof(struct mafComp,paralog),sizeof(struct mafFile));return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [32, 8, 16, 24, 28, 64, 8, 16, 24, 32, 40, 48, 52, 56, 60, 62, 63, 56]


def test_no_cpu_fallback_without_device(lib):
    # on a box without a HIP device every compute entry point must fail loudly, never fall back
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib.mz_last_error.restype = C.c_char_p
    assert lib.mz_init(0) == -1
    assert b"no HIP device" in lib.mz_last_error()
    assert lib.mz_yama_batch(1, None, None) == -1


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under multiz_amd/ may import, link or call it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "multiz_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "mzoracle" not in text and "liboracle" not in text and "mzo_" not in text, os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", LIB]).decode()
    assert "oracle" not in out


def test_abi_revision_and_struct_sizes_agree(lib, tmp_path):
    # the header's MZ_AMD_ABI, the library's mz_abi_version() and the ctypes mirrors in multiz_amd/api.py describe ONE layout
    from multiz_amd import api
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mz_amd.h"\nint main(void){printf("%d %zu %zu %zu %zu\\n", MZ_AMD_ABI,'
                   'sizeof(mz_job),sizeof(mz_out),sizeof(mz_prejob),sizeof(mz_preout));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[0] == lib.mz_abi_version() == api.MZ_AMD_ABI
    assert got[1:] == [C.sizeof(api.Job), C.sizeof(api.Out), C.sizeof(api.PreJob), C.sizeof(api.PreOut)]
