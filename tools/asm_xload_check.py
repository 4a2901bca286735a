#!/usr/bin/env python3
"""ISA check for k_vocoder_lt's excitation loads.

The kernel issues its excitation loads from inline asm (`global_load_dwordx2 vD, ...`) and waits for them with
hand-written `s_waitcnt vmcnt(N)`; the compiler does not know the destination registers are in flight.  That
is correct only while NOTHING touches vD between the load and the next hand-written wait -- a phi copy, a spill
or a reuse of those VGPRs after a toolchain change would silently read stale data (the hand-off check cannot
see it for the first chunk of an utterance).  This script walks the ISA (`hipcc -S --cuda-device-only`) of every
k_vocoder_lt instantiation: for each asm-issued load it follows the program order (through the loop's back
edge) to the next asm `s_waitcnt vmcnt` and fails if any instruction in between reads or writes vD.

usage: asm_xload_check.py file.s        (exit code 1 on a violation)
"""
import re
import sys


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def kernels(asm):
    cur, body = None, None
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                yield cur, body
                cur = None


def check_kernel(body):
    """Returns (number of asm loads checked, list of violations)."""
    ins, in_asm, labels = [], False, {}
    for ln in body:
        t = ln.strip()
        if t.startswith(";;#ASMSTART") or t.startswith(";#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND") or t.startswith(";#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ins.append((t.split(";")[0].strip(), in_asm))
    loads = [i for i, (t, a) in enumerate(ins) if a and t.startswith("global_load_dwordx2")]
    bad = []
    for i in loads:
        dst = _regs(ins[i][0].split(",")[0])
        j, steps, found = i + 1, 0, False
        while steps < 4 * len(ins):
            if j >= len(ins):
                break
            t, a = ins[j]
            if a and t.startswith("s_waitcnt") and "vmcnt" in t:
                found = True
                break
            if _regs(t) & dst:
                bad.append(f"'{ins[i][0]}' (#{i}): '{t}' (#{j}) touches the destination before the hand-written wait")
                break
            m = re.match(r"s_(c?branch\w*)\s+(\.LBB\w+)", t)
            if m and m.group(2) in labels and labels[m.group(2)] <= j and not m.group(1).startswith("cbranch_execz"):
                # follow a backward branch (the sample loop's back edge): program order continues at the label
                if m.group(1) == "branch" or labels[m.group(2)] <= i:
                    j = labels[m.group(2)]
                    steps += 1
                    continue
            j += 1
            steps += 1
        if not found and not bad:
            bad.append(f"'{ins[i][0]}' (#{i}): no hand-written s_waitcnt vmcnt found behind it")
    return len(loads), bad


def check(asm, name_part="k_vocoder_lt"):
    res = {}
    for k, body in kernels(asm):
        if name_part in k:
            res[k] = check_kernel(body)
    return res


if __name__ == "__main__":
    r = check(open(sys.argv[1]).read())
    rc = 0
    if not r:
        print("no k_vocoder_lt kernel in the file")
        rc = 1
    for k, (n, bad) in r.items():
        print(f"{k}: {n} asm loads, {len(bad)} violation(s)")
        for b in bad:
            print("   ", b)
        rc |= 1 if (bad or n == 0) else 0
    sys.exit(rc)
