"""Checks the one invariant of k_pyramid_stream that the compiler does not know about (csrc/pyramid.hip, stream_load):
the registers the uncounted prefetch loads write must not be read, copied or overwritten by any instruction between
the load and the kernel's own s_waitcnt (stream_rows_arrived).  Compiles pyramid.hip to gfx950 assembly (hipcc -S,
seconds, no GPU) and scans every instantiation of the kernel.  Exit code 0 = holds; tests/test_abi.py runs it.

usage: python tools/check_pyramid_isa.py [assembly.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tadataka_amd", "csrc", "pyramid.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assembly():
    out = os.path.join(tempfile.mkdtemp(prefix="tdk_isa_"), "pyramid.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                    "--cuda-device-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    return out


def regs_of(token):
    """'v[76:77]' -> {76, 77}; 'v5' -> {5}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", token)
    return {int(m.group(1))} if m else set()


def vregs_in(line):
    found = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        found |= regs_of(tok)
    return found


def check(path):
    text = open(path).read().splitlines()
    problems, kernels = [], 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN\S*k_pyramid_stream[^:\s]*):", text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i
        while j < len(text) and "s_endpgm" not in text[j]:
            j += 1
        body = text[i:j]
        kernels += 1
        in_asm, pending, waits, loads = False, {}, 0, 0
        for n, line in enumerate(body):
            s = line.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            if in_asm and s.startswith("global_load_dwordx2"):
                dest = regs_of(s.split()[1].rstrip(","))
                for r in dest:
                    pending[r] = n
                loads += 1
                # the address operand may not be a pending register either
                addr = vregs_in(s.split(",", 1)[1])
                if addr & (set(pending) - dest):
                    problems.append(f"{name}: line {n}: address of a prefetch reads a pending register: {s}")
                continue
            if in_asm and s.startswith("s_waitcnt vmcnt"):
                waits += 1
                if s.startswith("s_waitcnt vmcnt(0)"):
                    continue                      # (the first of the two waits of the statement; the label follows)
                pending = {}
                continue
            if in_asm and (s.startswith("s_cmp") or s.startswith("s_cbranch") or s.startswith("s_branch") or
                           re.match(r"^\d+:$", s)):
                continue
            touched = vregs_in(s) & set(pending)
            if touched:
                problems.append(f"{name}: line {n}: '{s}' touches v{sorted(touched)} between its prefetch "
                                f"(line {pending[min(touched)]}) and the wait")
        if loads == 0 or waits == 0:
            problems.append(f"{name}: no uncounted prefetch / wait found (loads {loads}, waits {waits}): the check is stale")
        if pending:
            # loads after the last wait in layout order are the next trip's: the loop closes on the wait
            pass
        i = j
    if kernels == 0:
        problems.append("no k_pyramid_stream instantiation in the assembly")
    return kernels, problems


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else assembly()
    kernels, problems = check(path)
    for p in problems:
        print("FAIL", p)
    print(f"{kernels} instantiation(s) of k_pyramid_stream checked: " + ("ok" if not problems else f"{len(problems)} problem(s)"))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
