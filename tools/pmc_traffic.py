"""profiles/traffic_latest.json from a tools/pmc_pass.sh summary: HBM bytes per launch of every heavy
kernel = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide streaming reads: MI355X_MICROARCH.md,
HBM section) + WRITE_SIZE, both in KiB per dispatch, collected in separate --pmc passes.
Usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag>/summary.txt profiles/<name>_pmc_summary.txt"""
import json
import re
import sys

src, note_path = sys.argv[1], sys.argv[2]
per, cur = {}, None
for line in open(src):
    if not line.startswith(" "):
        cur = line.strip()
        per[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
        if m and m.group(1) in ("FETCH_SIZE", "WRITE_SIZE"):
            per[cur][m.group(1)] = float(m.group(2))
out = {"_note": "HBM bytes per launch at 3 Gbase k=31 from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 wide-stream "
                "correction per MI355X_MICROARCH.md) + WRITE_SIZE, separate passes; " + note_path,
       "per_kernel": {}}
# (the "sk_count" phase of bench.py is two launches: sk_count_clean, then sk_count over the buckets it left)
phase_of = {"sk_count_kernel": "sk_count", "sk_count_clean_kernel": "sk_count", "sk_scatter0_kernel": "sk_scatter0", "sk_hist0_kernel": "sk_hist0",
            "sk_scatter1_kernel": "sk_scatter1", "sk_hist1_kernel": "sk_hist1", "sk_regroup_kernel": "sk_regroup",
            "leaves_kernel": "leaves", "level_hist_kernel<false>": "level1_hist", "level_hist_kernel<true>": "level0_hist",
            "level_scatter_wc_kernel<false": "level1_scatter", "level_scatter_wc_kernel<true": "level0_scatter"}
for kern, c in per.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    f, w = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    out["per_kernel"][kern] = {"fetch_bytes_corrected_x2": f, "write_bytes": w}
    for pat, phase in phase_of.items():
        if kern.startswith(pat):
            out[phase] = out.get(phase, 0) + int(f + w)
if any(k_.startswith("sk_") for k_ in out):
    out["sk_step_total"] = int(sum(v for k_, v in out.items() if k_.startswith("sk_") and isinstance(v, (int, float))))
print(json.dumps(out, indent=1))
