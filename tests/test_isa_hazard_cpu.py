"""Static guard (round 4): the shipped library must not contain the packed-fp32 instruction form that MI355X evaluates wrongly next to
f16 / bf16 MFMA kernels (tools/check_isa.py, tools/ubench/pk_opsel_mfma.hip).  Runs on the CPU: it disassembles the gfx950
code objects inside libcermvs.so."""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def _lib_path():
    return os.environ.get("CER_MVS_LIB") or os.path.join(REPO, "cer-mvs_amd", "csrc", "libcermvs.so")


def test_shipped_library_has_no_packed_fp32_src1_high_select():
    """Fails CLOSED (round 5): with llvm-objdump the library is scanned (and the scan record refreshed); without it the
    library must carry the record of a PASSED scan of exactly these bytes (tools/check_isa.py: <lib>.isa_scan.json, written by build());
    a box that has hipcc but no llvm-objdump, or a library nobody scanned, is an error - never a skip."""
    import check_isa
    lib = _lib_path()
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    if check_isa.have_objdump():
        ncos, n, hits = check_isa.scan(lib)
        assert ncos >= 10 and n > 1000, (ncos, n)          # (the scan saw the kernels: every .hip of the library is one code object)
        assert not hits, "packed-fp32 instructions taking the low result from src1's high half:\n" + "\n".join(f"{f}: {l}" for f, l in hits[:20])
        check_isa.write_sidecar(lib, ncos, n, hits)
        assert check_isa.verify_sidecar(lib)
    else:
        assert not check_isa.have_hipcc(), f"hipcc is here but llvm-objdump is not under {check_isa.LLVM}: set LLVM_BIN - the guard does not skip"
        assert check_isa.verify_sidecar(lib), f"{lib} was never scanned (no matching {os.path.basename(check_isa.sidecar_path(lib))}): run build() where the toolchain is"


def test_scan_record_is_bound_to_the_library_bytes(tmp_path):
    import check_isa
    lib = tmp_path / "libx.so"
    lib.write_bytes(b"\x7fELF" + b"\0" * 100)
    check_isa.write_sidecar(str(lib), 1, 0, [])
    assert check_isa.verify_sidecar(str(lib))
    lib.write_bytes(b"\x7fELF" + b"\1" * 100)              # other bytes: the record no longer vouches for them
    assert not check_isa.verify_sidecar(str(lib))
    check_isa.write_sidecar(str(lib), 1, 5, [("f", "v_pk_mul_f32 ...")])
    assert not check_isa.verify_sidecar(str(lib))           # a scan that FOUND the form does not vouch either


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc to build the planted object")
def test_a_planted_hazardous_instruction_turns_the_guard_red(tmp_path):
    """The guard is exercised, not only trusted: a kernel with the hazardous form PLANTED (inline assembly: v_pk_mul_f32 with the src1 bit of
    op_sel set - the form hipcc emitted on its own for u = X * r, w = Y * r in round 5's reciprocal-projection experiment,
    profiles/r05_cost_lines_fastdiv_ab.txt) is compiled for gfx950; the scan must list it, a harmless twin must pass."""
    import subprocess
    import check_isa
    assert check_isa.have_objdump(), "hipcc without llvm-objdump: the guard cannot work on this box"
    src = tmp_path / "planted.hip"
    src.write_text("""
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void planted(const f2* a, const f2* b, f2* o) {
    f2 x = a[threadIdx.x], y = b[threadIdx.x], r;
#ifdef PLANT
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
#else
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
#endif
    o[threadIdx.x] = r;
}
""")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for flag, expect in (("-DPLANT", 1), ("-DCLEAN", 0)):
        obj = tmp_path / f"planted{expect}.o"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-c", str(src), flag, "-o", str(obj)])
        ncos, n, hits = check_isa.scan(str(obj))
        assert ncos == 1 and n >= 1 and len(hits) == expect, (flag, ncos, n, hits)


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
