"""Per-kernel resource table of the library's device code, from the compiler's own remarks (no GPU needed):
    python tools/kernel_resources.py [out.json] [file.hip ...]
Compiles every csrc/*.hip (or the given ones) device-only with the Makefile's flags plus
-Rpass-analysis=kernel-resource-usage and records, per kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane, LDS bytes,
occupancy, SGPR / VGPR spills.  `ScratchSize` > 0 means a kernel keeps data in global memory behind the compiler's back
(runtime-indexed local arrays, spills, callee-saved registers of __noinline__ functions): DESIGN 9 "check ScratchSize of
every kernel".  tests/test_abi_and_host.py pins the entries that must stay 0."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "splitp_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--offload-device-only",
         "-Rpass-analysis=kernel-resource-usage", "-c"]
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "LDS Size [bytes/block]": "lds_bytes", "Occupancy [waves/SIMD]": "occupancy_waves_per_simd",
          "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills", "Dynamic Stack": "dynamic_stack"}


def demangle(names):
    filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "/usr/bin/c++filt"
    if not names or not os.path.exists(filt):
        return {n: n for n in names}
    out = subprocess.run([filt] + list(names), capture_output=True, text=True).stdout.split("\n")
    return {n: (out[i].strip() or n) for i, n in enumerate(names)}


def resources(hip_file, extra=()):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + list(extra) + [hip_file, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    table, cur = {}, None
    for line in err.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = table.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            v = m.group(2)
            cur[FIELDS[m.group(1).strip()]] = int(v) if v.isdigit() else v
    names = demangle(list(table))
    return {re.sub(r"\(.*$", "", names[k]): v for k, v in table.items()}


def main():
    args = sys.argv[1:]
    out = args[0] if args and args[0].endswith(".json") else None
    files = [a for a in args if a.endswith(".hip")] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    result = {}
    for f in files:
        result[os.path.basename(f)] = resources(f)
    text = json.dumps(result, indent=1, sort_keys=True)
    if out:
        with open(out, "w") as fh:
            fh.write(text + "\n")
    for fn, ks in result.items():
        for k, v in sorted(ks.items()):
            print(f"{fn:16s} {k[:70]:70s} vgpr {v.get('vgprs', '?'):>4} sgpr-spill {v.get('sgpr_spills', '?'):>4} "
                  f"vgpr-spill {v.get('vgpr_spills', '?'):>4} scratch {v.get('scratch_bytes_per_lane', '?'):>5} occ {v.get('occupancy_waves_per_simd', '?')}")


if __name__ == "__main__":
    main()
