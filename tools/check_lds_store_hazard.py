#!/usr/bin/env python3
"""Scan gfx950 device assembly for the hazard that produced intermittently wrong tile rows in round 3: a DS store of more
than 64 bits whose DATA registers are overwritten by one of the next instructions.  hipcc (ROCm 7.2) places such a VALU write directly
behind `ds_write_b128`; on the MI355X the store's last lanes then pick up the new value (measured: tools/exp_f8_stat.py, 74-99 % of the
launches wrong with one generator variant, 0 % with `s_nop` behind the stores).
usage: check_lds_store_hazard.py file.s [...]   (hipcc -S --cuda-device-only);  exit status 1 if a site is found"""
import re
import sys

WIDE = {"ds_write_b128": 4, "ds_write_b96": 3, "ds_write_b64": 2, "ds_write2_b64": 2, "ds_write2st64_b64": 2}
WINDOW = 2          # instructions after the store that are checked (gfx940+: 2 wait states for the VMEM form of this hazard)


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    total = 0
    for path in sys.argv[1:]:
        kernel = "?"
        lines = open(path).read().split("\n")
        ins = []
        for ln in lines:
            t = ln.split(";")[0].strip()
            m = re.match(r"^(_Z\w+|\w+):\s*$", t)
            if m and not t.startswith(".L"):
                kernel = m.group(1)
            if not t or t.startswith(".") or t.endswith(":"):
                continue
            ins.append((kernel, t))
        for i, (k, t) in enumerate(ins):
            op = t.split()[0]
            if op not in WIDE:
                continue
            ops = [x.strip() for x in t[len(op):].split(",")]
            data = set()
            for tok in ops[1:3]:                           # data0 (, data1 for the write2 forms)
                data |= regs(tok.split()[0])
            for j in range(1, WINDOW + 1):
                if i + j >= len(ins):
                    break
                t2 = ins[i + j][1]
                op2 = t2.split()[0]
                if op2.startswith("s_nop"):
                    break
                if not op2.startswith("v_") or op2.startswith("v_cmp") or op2.startswith("v_cmpx"):
                    continue
                dst = regs(t2[len(op2):].split(",")[0].strip().split()[0])
                if dst & data:
                    total += 1
                    print(f"{path}: {k[:60]}: `{t}`  then (+{j}) `{t2}`")
    print(f"{total} site(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
