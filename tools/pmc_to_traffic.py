"""Fold the per-kernel counter averages of tools/probes/profile_pmc.sh (gpurun_out/pmc_<tag>_<WL>/summary.txt) into
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB): the gfx950 read-side correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 bytes).

    python tools/pmc_to_traffic.py C2 gpurun_out/pmc_r1_C2/summary.txt [profiles/r01_pmc_c2_summary.txt]
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    wl, summary = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(out)) if os.path.exists(out) else {}
    entry = {}
    for line in open(summary):
        tok = line.split()
        first = next((k for k, x in enumerate(tok) if "=" in x), len(tok))
        name, kv = " ".join(tok[:first]), tok[first:]
        # pass A variants: 0 sweep, 1 sweep + superset list, 2 walk (k_density_sweep_w: the large-channel form of 0 / 1)
        mode = re.search(r"k_density(?:_sweep_w)?<\d+,\s*(\d)[,>]", name)
        name = re.sub(r"<.*", "", name)
        if mode:
            name += {"0": "", "1": "_build", "2": "_walk"}[mode.group(1)]
        # the large-channel forms of the passes are launched under the names of the passes (bench.py's kernels_ms)
        name = name.replace("k_density_sweep_w", "k_density")
        name = {"k_density_w": "k_density_walk", "k_kgc_w": "k_kgc", "k_forces_w": "k_forces",
                "k_continuity_density_w": "k_continuity_density"}.get(name, name)
        v = {}
        for item in kv:
            if "=" in item:
                k, x = item.split("=")
                try:
                    v[k] = float(x)
                except ValueError:
                    pass
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        e = dict(fetch_size_bytes=v["FETCH_SIZE"] * 1024, write_size_bytes=v["WRITE_SIZE"] * 1024)
        e["traffic_bytes"] = 2 * e["fetch_size_bytes"] + e["write_size_bytes"]
        if v.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the TA / SQ sums over 256 CUs: busy fraction of the average
            # CU = sum / (256/8 * GRBM)
            e["ta_busy_frac"] = v.get("TA_TA_BUSY_sum", 0.0) / (32.0 * v["GRBM_GUI_ACTIVE"])
            e["valu_active_frac"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / (32.0 * v["GRBM_GUI_ACTIVE"])
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        e["launches_profiled"] = v.get("launches")
        entry[name] = e
    table[wl] = entry
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        shutil.copyfile(summary, os.path.join(ROOT, sys.argv[3]))
    print(wl, {k: round(e["traffic_bytes"]) for k, e in entry.items()})


if __name__ == "__main__":
    main()
