#!/usr/bin/env python3
"""Static guard against a silent-corruption hazard found in round 4 (tools/ubench/pk_opsel_mfma.hip):

    On MI355X (gfx950) a packed-fp32 VOP3P instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose LOW result takes the HIGH half
    of src1 (the src1 bit of op_sel set, e.g. `op_sel:[0,1] op_sel_hi:[1,0]`) returns a wrong low half about once per 10^4 executions
    while waves of an f16 / bf16 MFMA kernel are resident on the same CU.  hipcc emits the form when it vectorises two fp32 operations
    whose operands sit crosswise in two register pairs.

This script disassembles the gfx950 code objects inside objects / shared libraries and lists every instruction of that form.
usage: check_isa.py <file.o | file.so> ...     exit status 1 if any is found."""
import os, re, subprocess, sys, tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PK_F32 = re.compile(r"\b(v_pk_(?:add|mul|fma|max|min)_f32)\b(.*)")
OPSEL = re.compile(r"op_sel:\[([01])(?:,([01]))?(?:,([01]))?\]")


def code_objects(path, tmp):
    """The gfx950 code objects bundled in `path` -> list of files (a plain AMDGPU ELF is returned as is)."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] == b"\x7fELF" and data[18:20] == (224).to_bytes(2, "little"):        # EM_AMDGPU
        return [path]
    # offload bundles: "__CLANG_OFFLOAD_BUNDLE__" header, entries (offset, size, triple)
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        p = i + len(magic)
        n = int.from_bytes(data[p:p + 8], "little"); p += 8
        for _ in range(n):
            off = int.from_bytes(data[p:p + 8], "little"); size = int.from_bytes(data[p + 8:p + 16], "little")
            tl = int.from_bytes(data[p + 16:p + 24], "little"); triple = data[p + 24:p + 24 + tl].decode(); p += 24 + tl
            if "gfx950" in triple and size > 0:
                fn = os.path.join(tmp, f"co_{len(out)}_{os.path.basename(path)}.elf")
                with open(fn, "wb") as g:
                    g.write(data[i + off:i + off + size])
                out.append(fn)
        pos = i + len(magic)
    return out


def scan(path):
    hits, ninstr = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(path, tmp)
        for co in cos:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
            if dis.returncode != 0:
                raise RuntimeError(f"llvm-objdump failed on {co}: {dis.stderr[:300]}")
            func = "?"
            for line in dis.stdout.splitlines():
                if line.endswith(">:"):
                    func = line.split("<")[-1][:-2]
                    continue
                m = PK_F32.search(line)
                if not m:
                    continue
                ninstr += 1
                o = OPSEL.search(m.group(2))
                if o and o.group(2) == "1":                     # src1's low-result selector = high half
                    hits.append((func, line.strip()))
        return len(cos), ninstr, hits


SCANNER_VERSION = 2


def sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def sidecar_path(lib):
    return lib + ".isa_scan.json"


def write_sidecar(lib, ncos, n, hits):
    """Record of a scan next to the library it scanned: a library that travels to a box without the toolchain (or is simply loaded
    later) can be matched to a PASSED scan by its sha256 (verify_sidecar) - round 5, VERDICT r4 item 7(ii)."""
    import json
    rec = {"library": os.path.basename(lib), "sha256": sha256(lib), "code_objects": ncos, "packed_fp32_instructions": n,
           "hazardous_forms": len(hits), "scanner_version": SCANNER_VERSION,
           "form": "v_pk_{add,mul,fma,max,min}_f32 with the src1 bit of op_sel set (low result from src1's high half)"}
    with open(sidecar_path(lib), "w") as f:
        json.dump(rec, f, indent=1)
    return rec


def verify_sidecar(lib):
    """True iff a scan record exists for exactly these bytes and it found nothing."""
    import json
    try:
        with open(sidecar_path(lib)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return False
    return rec.get("sha256") == sha256(lib) and rec.get("hazardous_forms") == 0 and rec.get("scanner_version") == SCANNER_VERSION


def have_objdump():
    return os.path.exists(os.path.join(LLVM, "llvm-objdump"))


def have_hipcc():
    import shutil
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def main(paths):
    bad = 0
    for p in paths:
        ncos, n, hits = scan(p)
        print(f"{p}: {ncos} gfx950 code object(s), {n} packed-fp32 instructions, {len(hits)} of the hazardous form")
        for func, line in hits[:40]:
            print(f"   {func}: {line}")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
