"""Register / scratch / LDS / occupancy of every kernel the product library ships: hipcc -Rpass-analysis=kernel-resource-usage on the shipped translation units with their
.flags (build container, no GPU).  usage: python tools/diag/code_objects.py > profiles/rNN_code_objects.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(ROOT)
units = sys.argv[1:] or ["jh_engine_v5", "jh_engine_v5_cap64", "jh_engine_v6", "jh_engine_v4", "jh_policy", "jh_simple", "jh_update", "jh_reward"]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
print(f"# hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage, shipped flags, tree at {head}")
print(f"# {'kernel':78s} VGPR AGPR  SGPR  spilled VGPR / SGPR  scratch B/lane  LDS B/workgroup  waves/SIMD")
for u in units:
    src = f"judo_amd/csrc/{u}.hip"; fl = f"judo_amd/csrc/{u}.flags"
    flags = open(fl).read().split() if os.path.exists(fl) else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Iinclude", "-Ijudo_amd/csrc", *flags, "-c", src, "-o", "/tmp/co_tmp.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    print(f"## {src}  {' '.join(flags)}")
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:[^:]*:)?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?)(?: \[-Rpass|$)", line)
        if not m: continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            n = re.sub(r"\(anonymous namespace\)::", "", cur["name"]); n = re.sub(r"\(.*$", "", n).replace("void ", "")
            print(f"  {n:78s} {cur.get('VGPRs', '?'):>4s} {cur.get('AGPRs', '?'):>4s} {cur.get('TotalSGPRs', '?'):>5s}  {cur.get('VGPRs Spill', '?'):>8s} / {cur.get('SGPRs Spill', '?'):<6s} "
                  f"{cur.get('ScratchSize [bytes/lane]', '?'):>12s} {cur.get('LDS Size [bytes/block]', '?'):>16s} {cur.get('Occupancy [waves/SIMD]', '?'):>10s}")
