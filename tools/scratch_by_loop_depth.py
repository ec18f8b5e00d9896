"""Where a translation unit's scratch (private-segment) and SGPR-spill instructions sit, by loop depth - from the
device assembly (no GPU needed):  python tools/scratch_by_loop_depth.py splitp_amd/csrc/sparse.hip [more.hip ...]
A spill at depth 0 runs once per call of the function; only spills inside loops (depth >= 1, and really only the inner
product / list loops at depth >= 2) cost time.  Complements tools/kernel_resources.py, which gives the totals."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--offload-device-only", "-S"]


def analyse(hip):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [os.path.abspath(hip), "-o", out],
                       check=True, capture_output=True, cwd=os.path.dirname(os.path.abspath(hip)))
        lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[\w]+:", l)]
    filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "/usr/bin/c++filt"
    for idx, (i, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        depth, scratch, lanes = 0, {}, {}
        size = 0
        for ln in lines[i:end]:
            if re.match(r"^(\.LBB|; %bb)", ln):
                m = re.search(r"Depth[ =](\d+)", ln)
                depth = int(m.group(1)) if m else 0
            if "scratch_" in ln:
                scratch[depth] = scratch.get(depth, 0) + 1
            if "v_readlane_b32" in ln or "v_writelane_b32" in ln:
                lanes[depth] = lanes.get(depth, 0) + 1
            m = re.match(r"; ScratchSize: (\d+)", ln)
            if m:
                size = int(m.group(1))
        pretty = name
        if os.path.exists(filt):
            pretty = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
        pretty = re.sub(r"\(.*$", "", pretty)
        if scratch or lanes or size:
            print(f"{os.path.basename(hip)}: {pretty[:80]}\n    scratch bytes/lane {size}; scratch instructions by loop depth {dict(sorted(scratch.items()))}; "
                  f"SGPR-spill lane moves by loop depth {dict(sorted(lanes.items()))}")


for f in sys.argv[1:]:
    analyse(f)
