#!/usr/bin/env python3
"""Static audit of the HIP kernels (no GPU needed): tools/isa_audit.py [file.hip ...]   (default: every csrc/kernels_*.hip)
Per kernel: VGPRs, scratch bytes per lane, waves per SIMD, LDS per block (hipcc -Rpass-analysis=kernel-resource-usage) and, from the
gfx950 ISA (-save-temps), the number of global loads, of `s_waitcnt vmcnt(N)` with N == 0 (a FULL wait: behind every load it means the
loads are serial round trips) and of scalar branches.  Flags: SPILL (scratch > 0), LOWOCC (<= 2 waves per SIMD), SERIAL (>= 8 loads
and at least every second one followed by a full wait).  This is how round 4 found the 96 serial loads of the window-sum kernel and
the spill in the rescale's column pass; a software-pipelined loop legitimately waits in full once per trip."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def short(name):
    name = re.sub(r"_ZN6fhelin12_GLOBAL__N_1\d+", "", name)
    return name[:64]


def audit(path):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", path, "-o", os.path.join(tmp, "k.o"), "-save-temps",
               "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "include")]
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            raise SystemExit("hipcc failed on " + path)
        res = {}
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: .*Function Name: (\S+)", line)
            if m:
                cur = res.setdefault(m.group(1), {})
                continue
            for key, pat in (("vgprs", r"VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None and " Spill" not in line:
                    cur[key] = int(m.group(1))
        asm = [f for f in glob.glob(os.path.join(tmp, "*gfx950*.s"))]
        text = open(asm[0]).read() if asm else ""
        names = [(m.start(), m.group(1)) for m in re.finditer(r"\n(_ZN6fhelin\S*):", text)]
        for k, (pos, name) in enumerate(names):
            end = names[k + 1][0] if k + 1 < len(names) else len(text)
            body = text[pos:end]
            d = res.setdefault(name, {})
            waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
            d["loads"] = len(re.findall(r"global_load", body))
            d["full_waits"] = sum(1 for w in waits if w == "0")
            d["branches"] = len(re.findall(r"s_cbranch", body))
        return res


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "fhe-linformer_amd", "csrc", "kernels_*.hip")))
    flagged = 0
    for f in files:
        print("==", os.path.relpath(f, ROOT))
        for name, d in sorted(audit(f).items(), key=lambda kv: -kv[1].get("vgprs", 0)):
            if "vgprs" not in d:
                continue
            flags = []
            if d.get("scratch", 0) > 0:
                flags.append("SPILL")
            if d.get("occ", 8) <= 2:
                flags.append("LOWOCC")
            if d.get("loads", 0) >= 8 and 2 * d.get("full_waits", 0) >= d.get("loads", 0):
                flags.append("SERIAL")
            flagged += bool(flags)
            print(f"  {short(name):66s} vgpr {d.get('vgprs', 0):4d} scratch {d.get('scratch', 0):4d} waves/SIMD {d.get('occ', 0)} lds {d.get('lds', 0):6d} "
                  f"loads {d.get('loads', 0):4d} full waits {d.get('full_waits', 0):3d} branches {d.get('branches', 0):3d}  {' '.join(flags)}")
    print(f"{flagged} kernel instantiation(s) flagged")


if __name__ == "__main__":
    main()
