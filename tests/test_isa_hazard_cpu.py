"""Static guard (round 4): the shipped library must not contain the packed-fp32 instruction form that MI355X evaluates wrongly next to
f16 / bf16 MFMA kernels (tools/check_isa.py, tools/ubench/pk_opsel_mfma.hip).  Runs on the CPU: it disassembles the gfx950
code objects inside libcermvs.so."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def _objdump():
    llvm = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    return os.path.exists(os.path.join(llvm, "llvm-objdump")) or shutil.which("llvm-objdump")


@pytest.mark.skipif(not _objdump(), reason="llvm-objdump of the ROCm toolchain not found")
def test_shipped_library_has_no_packed_fp32_src1_high_select():
    import check_isa
    lib = os.environ.get("CER_MVS_LIB") or os.path.join(REPO, "cer-mvs_amd", "csrc", "libcermvs.so")
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    ncos, n, hits = check_isa.scan(lib)
    assert ncos >= 10 and n > 1000, (ncos, n)              # (the scan saw the kernels: every .hip of the library is one code object)
    assert not hits, "packed-fp32 instructions taking the low result from src1's high half:\n" + "\n".join(f"{f}: {l}" for f, l in hits[:20])


def test_checker_recognises_the_form():
    import check_isa
    m = check_isa.PK_F32.search("	v_pk_add_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]")
    assert m and check_isa.OPSEL.search(m.group(2)).group(2) == "1"
    for ok in ("v_pk_mul_f32 v[0:1], v[28:29], v[0:1] op_sel_hi:[1,0]", "v_pk_fma_f32 v[2:3], v[26:27], s[6:7], v[2:3] op_sel:[1,0,0]",
               "v_pk_add_f32 v[0:1], v[2:3], v[4:5]"):
        m = check_isa.PK_F32.search(ok)
        o = check_isa.OPSEL.search(m.group(2))
        assert m and not (o and o.group(2) == "1"), ok
    m = check_isa.PK_F32.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert check_isa.OPSEL.search(m.group(2)).group(2) == "1"
