#!/usr/bin/env python3
"""Copy what tools/profile_round2.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>_* (tracked) and derive profiles/<tag>_traffic.json
(HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes) which bench.py reports as roofline.traffic.
usage: tools/collect_profiles.py <tag>"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
traffic = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --task <task> --steps 5 --warmup 3 "
                     "--no-cpu-baseline --no-cube-only` (tools/profile_round2.sh; summaries in profiles/%s_<case>_rocprofv3_summary.txt).  Counter unit = KB "
                     "(calibrated in round 1 on torch.randn's 16 MiB write: WRITE_SIZE = 16384.0); the guide's x2 FETCH_SIZE correction applies to 16 B/lane "
                     "streams and is NOT applied (4 B/lane accesses).  Almost all of it is register spill traffic (scratch), not rollout data." % tag}
for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if name.endswith("_summary.txt"):
        case = name[: -len("_summary.txt")]
        shutil.copy(p, os.path.join(dst, f"{tag}_{case}_rocprofv3_summary.txt"))
        txt = open(p).read()
        vals, kern = {}, None
        # the dominant kernel = the one rocpd_summary.py lists dispatch by dispatch (largest total duration); its counter rows, not the first row of the
        # table (sorted by value: with the cube's contacts only, torch.randn's 16 MiB write outranks the rollout kernel's -- round 2 picked that row)
        top = re.search(r"^dispatches of (.+?) \[us\]:", txt, re.M)
        prefix = re.escape(top.group(1)[:40]) if top else r"\S"
        for cn in ("FETCH_SIZE", "WRITE_SIZE"):
            m = re.search(r"^(" + prefix + r".*?)\s+" + cn + r"\s+(\d+)\s+([\d.]+)\s", txt, re.M)
            if m:
                vals[cn], kern = float(m.group(3)), m.group(1)
        issue = {}
        for cn in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE"):  # the issue side of the same launches (per launch, summed over the GPU)
            m = re.search(r"^" + prefix + r".*?\s+" + cn + r"\s+(\d+)\s+([\d.]+)\s", txt, re.M)
            if m:
                issue[cn] = float(m.group(2))
        ms = re.findall(r"mean of the last 20: ([\d.]+) us", txt)
        if vals:
            traffic[case] = {"kernel": re.sub(r"\(float const\*.*", "", kern).strip(), "kernel_ms_default_bench_last20": float(ms[-1]) / 1e3 if ms else None,
                             "FETCH_SIZE_KB_per_launch": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": vals.get("WRITE_SIZE"),
                             "hbm_bytes_per_launch": int(1024 * (vals.get("FETCH_SIZE", 0) + vals.get("WRITE_SIZE", 0)))}
            if "SQ_INSTS_VALU" in issue and "SQ_BUSY_CYCLES" in issue:
                # SQ_BUSY_CYCLES is per shader engine (32 of them), in clock cycles: the kernel's own duration; 1 024 SIMDs; one wave64 VALU instruction per 2.74 cycles is
                # the best rate tools/ubench/valu_rate.hip measures on this part (profiles/r03_valu_rate.txt), 2 cycles the nominal one
                cyc = issue["SQ_BUSY_CYCLES"] / 32.0
                traffic[case]["issue"] = {"valu_wave_instructions_per_launch": issue["SQ_INSTS_VALU"], "kernel_cycles": cyc,
                                          "cycles_per_valu_instruction_per_simd": cyc * 1024 / issue["SQ_INSTS_VALU"], "measured_best_cycles_per_valu_instruction": 2.74,
                                          "frac_of_measured_valu_issue_rate": 2.74 * issue["SQ_INSTS_VALU"] / (cyc * 1024),
                                          "frac_of_nominal_valu_issue_rate": 2.0 * issue["SQ_INSTS_VALU"] / (cyc * 1024),
                                          "active_inst_any_over_simd_cycles": (4.0 * issue["SQ_ACTIVE_INST_ANY"] / (cyc * 1024)) if "SQ_ACTIVE_INST_ANY" in issue else None,
                                          "lds_array_busy_frac": (issue["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)) if "SQ_LDS_IDX_ACTIVE" in issue else None}
    elif name.endswith("_bench_under_rocprof.json") or ((name.startswith("bench_") or name.startswith("materialize_")) and name.endswith(".json")):
        lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_{name}"), "w").write(lines[-1] + "\n")
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in traffic.items() if k != "method"}, indent=1))
for name, out in (("spot_summary.txt", "spot_policy_rollout_rocprofv3_summary.txt"), ("spot_under_rocprof.txt", "spot_policy_rollout_timing.txt"), ("benchmark_sweep.txt", "benchmark_sweep.txt")):
    if os.path.exists(os.path.join(src, name)):  # (tools/profile_spot.sh, python -m judo_amd.benchmark)
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{out}"))
        stale = os.path.join(dst, f"{tag}_{name[:-len('_summary.txt')]}_rocprofv3_summary.txt") if name.endswith("_summary.txt") else None
        if stale and stale != os.path.join(dst, f"{tag}_{out}") and os.path.exists(stale):
            os.remove(stale)
