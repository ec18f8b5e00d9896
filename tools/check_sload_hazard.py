"""Static check of the hand-issued scalar loads of subflat.hip: the Sturm table is fetched by inline-asm
`s_load_dwordx8` into registers the hardware writes ASYNCHRONOUSLY; the only ordering the compiler sees is a later asm
`s_waitcnt lgkmcnt(0)` tied to the same registers.  Nothing stops a future compiler from copying or spilling those registers
in between (k_subscore_tri has SGPR spills), which would silently read stale data.  This tool compiles the file to assembly
and verifies, for every s_load_dwordx8 of the kernel, that no instruction between the load and the next
`s_waitcnt lgkmcnt(0)` (or any s_waitcnt with lgkmcnt(0)) touches the destination registers.
    python tools/check_sload_hazard.py [file.hip]      exit status 0 = clean; prints the offending lines otherwise."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--offload-device-only", "-S"]


def sregs(text):
    """scalar registers an operand string mentions: s12, s[4:11] -> set of ints"""
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    return out


def check(hip):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [os.path.abspath(hip), "-o", out], check=True,
                       capture_output=True, cwd=os.path.dirname(os.path.abspath(hip)))
        lines = open(out).read().split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.?LBB[\w]+):", ln)
        if m:
            labels[m.group(1)] = i
    loads, bad = 0, []
    for i, ln in enumerate(lines):
        m = re.match(r"\s*s_load_dwordx8\s+s\[(\d+):(\d+)\]", ln)
        if not m:
            continue
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
        loads += 1
        # every path from the load to its first s_waitcnt lgkmcnt(0): conditional branches fork, plain branches jump
        work, seen, steps = [i + 1], set(), 0
        while work:
            j = work.pop()
            while j < len(lines) and steps < 20000:
                steps += 1
                if j in seen:
                    break
                seen.add(j)
                body = lines[j].split(";")[0].strip()
                if not body or body.endswith(":") or body.startswith("."):
                    j += 1
                    continue
                if body.startswith("s_waitcnt") and "lgkmcnt(0)" in body:
                    break
                if body.startswith("s_endpgm"):
                    break   # (the registers die with the wave)
                mb = re.match(r"s_(c?)branch\w*\s+(\.?LBB\w+)", body)
                if mb:
                    tgt = labels.get(mb.group(2))
                    if tgt is None:
                        bad.append((i + 1, j + 1, "branch to an unknown label", lines[j].strip()))
                        break
                    if mb.group(1) == "c":
                        work.append(tgt)
                        j += 1
                        continue
                    j = tgt
                    continue
                ops = body.split(None, 1)[1] if " " in body else ""
                if sregs(ops) & dst:
                    bad.append((i + 1, j + 1, "touches the destination of an outstanding s_load", lines[j].strip()))
                j += 1
        if steps >= 20000:
            bad.append((i + 1, i + 1, "no s_waitcnt lgkmcnt(0) found on some path", ln.strip()))
    return loads, bad


if __name__ == "__main__":
    f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "splitp_amd", "csrc", "subflat.hip")
    n, bad = check(f)
    print(f"{os.path.basename(f)}: {n} s_load_dwordx8, {len(bad)} hazards")
    for b in bad[:20]:
        print("  load at line %d, line %d: %s: %s" % b)
    sys.exit(1 if bad or n == 0 else 0)
